#!/usr/bin/env python3
"""BASELINE configs[0] AS WRITTEN: `iss generate --genomes data/ecoli.fasta --mode basic -n 10000 --cpus 1` (seed 42), captured by
running the reference like make_golden.py does (same stand-in Bio package): /root/reference/iss/app.py:333-341,
/root/reference/iss/error_models/basic.py:40-63.

Output: tests/golden/generate/ecoli_basic_n10000_seed42_cpus1.npz -- the two FASTQ files and `_abundance.txt` whole, their SHA-256,
and the input `data/ecoli.fasta` (a data file of the reference's own tests) so that the test needs nothing outside the repo.

Usage:  python tests/golden/tooling/make_golden_configs0.py   (from the repo root, build container only)
"""
import hashlib
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
outp = os.path.join(work, "configs0")
subprocess.check_call([sys.executable, "-m", "iss", "generate", "--genomes", "data/ecoli.fasta", "--mode", "basic", "-n", "10000",
                       "--seed", "42", "--cpus", "1", "-o", outp, "--quiet"], env=env, cwd=REFCOPY)
blobs = {s: np.frombuffer(open(outp + s, "rb").read(), dtype=np.uint8) for s in ("_R1.fastq", "_R2.fastq", "_abundance.txt")}
fasta = np.frombuffer(open(os.path.join(REF, "data", "ecoli.fasta"), "rb").read(), dtype=np.uint8)
np.savez_compressed(os.path.join(GOLDEN, "generate", "ecoli_basic_n10000_seed42_cpus1.npz"), r1=blobs["_R1.fastq"],
                    r2=blobs["_R2.fastq"], abundance=blobs["_abundance.txt"], fasta=fasta,
                    sha_r1=np.array(hashlib.sha256(blobs["_R1.fastq"].tobytes()).hexdigest()),
                    sha_r2=np.array(hashlib.sha256(blobs["_R2.fastq"].tobytes()).hexdigest()))
print("configs[0]", len(blobs["_R1.fastq"]), len(blobs["_R2.fastq"]), blobs["_R1.fastq"].tobytes().count(b"\n") // 4, "records per file")
shutil.rmtree(work, ignore_errors=True)
