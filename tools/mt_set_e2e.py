#!/usr/bin/env python3
"""The byte-identical mode end to end with W workers on one GPU: `python -m insilicoseq_amd generate --rng mt --cpus W --devices 1`
(the files `iss generate --cpus W` writes) on one random 5 Mbp record, NovaSeq, FASTQ on /dev/shm -- wall time of the whole
command (start-up, generation + text + the workers' temp files, the parent's concatenation as the reference does it) and of a
tiny run of the same command (start-up alone).
    python tools/mt_set_e2e.py [W] [million read pairs]"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_genome  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pairs = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 16_000_000
tmp = "/dev/shm/mt_set_e2e"
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp)
fasta = os.path.join(tmp, "g.fasta")
with open(fasta, "wb") as fh:
    fh.write(b">rec0\n" + random_genome(11, 5_000_000).encode() + b"\n")


def run(n_reads, tag):
    out = os.path.join(tmp, tag)
    t0 = time.perf_counter()
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes", fasta, "--model", "novaseq", "-n", str(n_reads),
                           "--seed", "7", "--cpus", str(W), "--devices", "1", "--rng", "mt", "-o", out], cwd=ROOT)
    dt = time.perf_counter() - t0
    size = sum(os.path.getsize(out + s) for s in ("_R1.fastq", "_R2.fastq"))
    for s in ("_R1.fastq", "_R2.fastq", "_abundance.txt"):
        os.remove(out + s)
    return dt, size


t_small, _ = run(2 * 64 * W, "tiny")
t_big, size = run(2 * pairs, "big")
print("W = %d: %d pairs -> %.2f GB of FASTQ in %.2f s (start-up alone %.2f s): %.3g pairs/s for the command, %.3g beyond start-up" % (
    W, pairs, size / 1e9, t_big, t_small, pairs / t_big, pairs / max(t_big - t_small, 1e-9)))
shutil.rmtree(tmp, ignore_errors=True)
