#!/usr/bin/env python3
"""Where the per-record time of a many-record work list goes (host wall clock per call)."""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import dense_model, random_genome
from insilicoseq_amd.engine import ReadEngine
n_rec = 2000
dense = dense_model("novaseq")
big = random_genome(5, 20000 * 50).encode()
seqs = [big[(k % 50) * 20000:(k % 50 + 1) * 20000] for k in range(n_rec)]
eng = ReadEngine(0); eng.load_model(dense); eng.reserve(1 << 18)
d = tempfile.mkdtemp(dir="/dev/shm")
f1 = open(os.path.join(d, "a"), "wb"); f2 = open(os.path.join(d, "b"), "wb")
t = {"add": 0.0, "gen": 0.0, "emit": 0.0}
t0 = time.perf_counter()
for k in range(n_rec):
    a = time.perf_counter(); gid = eng.add_genome(seqs[k]); b = time.perf_counter()
    eng.generate(gid, 50, first_ordinal=k * 50, seed=1, out_first_pair=0); c = time.perf_counter()
    eng.fastq_emit(f1.fileno(), f2.fileno(), "contig_%d" % k, 0, 0, 0, 50); e = time.perf_counter()
    t["add"] += b - a; t["gen"] += c - b; t["emit"] += e - c
eng.fastq_flush()
tot = time.perf_counter() - t0
print("per record: total %.1f us; add_genome %.1f, generate %.1f, fastq_emit %.1f" % (tot / n_rec * 1e6, t["add"] / n_rec * 1e6, t["gen"] / n_rec * 1e6, t["emit"] / n_rec * 1e6))
shutil.rmtree(d, ignore_errors=True)
