#!/usr/bin/env python3
"""Golden outputs of `iss generate --cpus 8` (the reference's own parallelism: eight workers, seeds seed + cpu_number,
iss/app.py:81-106, iss/generator.py:234-236), captured by running the reference like make_golden_cli.py does (same
stand-in Bio package).  The build's W-workers-per-launch MT mode (iss_generate_mt_workers) must reproduce the files byte
for byte.

Outputs: tests/golden/generate/genomes_hiseq_n1600_seed42_cpus8.npz  (data/genomes.fasta: short, low-complexity records --
                                                                       the sequential walker's cases)
         tests/golden/generate/syn3_novaseq_n3000_seed7_cpus8.npz     (three random A/C/G/T records of 20 kbp, carried in
                                                                       the fixture: the resolver's fast path, 8 workers)

Usage:  python tests/golden/tooling/make_golden_cpus8.py   (from the repo root, build container only)
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "bio_shim")
REF = "/root/reference"

work = tempfile.mkdtemp(prefix="iss_ref_")
REFCOPY = os.path.join(work, "refcopy")
shutil.copytree(REF, REFCOPY)
subprocess.check_call(["chmod", "-R", "u+w", REFCOPY])
env = dict(os.environ, PYTHONPATH=SHIM + ":" + REFCOPY)
os.makedirs(os.path.join(GOLDEN, "generate"), exist_ok=True)


def run(name, fasta, model, n, seed, extra=()):
    outp = os.path.join(work, name)
    subprocess.check_call([sys.executable, "-m", "iss", "generate", "--genomes", fasta, "--model", model, "-n", str(n),
                           "--seed", str(seed), "--cpus", "8", "-o", outp, "--quiet"] + list(extra), env=env, cwd=REFCOPY)
    blob = lambda suffix: np.frombuffer(open(outp + suffix, "rb").read(), dtype=np.uint8)  # noqa: E731
    return blob("_R1.fastq"), blob("_R2.fastq"), blob("_abundance.txt")


r1, r2, ab = run("g8", "data/genomes.fasta", "hiseq", 1600, 42)
np.savez_compressed(os.path.join(GOLDEN, "generate", "genomes_hiseq_n1600_seed42_cpus8.npz"), r1=r1, r2=r2, abundance=ab)
print("genomes_hiseq cpus8", len(r1), len(r2))

rng = np.random.RandomState(2024)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
fasta = os.path.join(work, "syn3.fasta")
text = b""
for k in range(3):
    seq = letters[rng.randint(0, 4, size=20000)].tobytes()
    text += b">syn_%d some description\n" % k + b"\n".join(seq[i:i + 70] for i in range(0, len(seq), 70)) + b"\n"
with open(fasta, "wb") as fh:
    fh.write(text)
r1, r2, ab = run("s8", fasta, "novaseq", 3000, 7)
np.savez_compressed(os.path.join(GOLDEN, "generate", "syn3_novaseq_n3000_seed7_cpus8.npz"), r1=r1, r2=r2, abundance=ab,
                    fasta=np.frombuffer(text, dtype=np.uint8))
print("syn3_novaseq cpus8", len(r1), len(r2))
shutil.rmtree(work, ignore_errors=True)
