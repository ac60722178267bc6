#!/usr/bin/env python3
"""Golden vector of a reference worker with `--fragment-length` whose work list holds records NOT longer than the read
length: simulate_read draws its fragment length (np.random.normal, numpy's legacy polar Box-Muller with a cached second
value, iss/generator.py:121-123) BEFORE the assertion of iss/generator.py:130 fails, so a skipped record moves the numpy
stream and flips the gaussian cache for everything after it.  Captured by importing the reference like make_golden.py
does (same stand-in Bio package, same file format); kept separate so that the existing fixtures are not rewritten.

Output: tests/golden/worker/syn_novaseq_frag_short.npz

Usage:  python tests/golden/tooling/make_golden_frag_short.py   (from the repo root, build container only)
"""
import json
import os
import random
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
sys.path.insert(0, REFCOPY)
sys.path.insert(0, SHIM)
os.chdir(REFCOPY)

from Bio.Seq import Seq  # noqa: E402
from Bio.SeqRecord import SeqRecord  # noqa: E402

from iss import generator  # noqa: E402
from iss.error_models import kde  # noqa: E402


def random_genome(seed, n):
    rnd = random.Random(seed)
    return "".join(rnd.choice("ACGT") for _ in range(n))


case, model, frag, sd, cpu_number, seed = "syn_novaseq_frag_short", "novaseq", 400, 30, 2, 13
em = kde.KDErrorModel(os.path.join(REFCOPY, "iss/profiles/NovaSeq"), frag, sd, False)
# (length, pairs): the second, fourth and fifth records are skipped (length <= read length 151) -- each after ONE draw of
# its fragment length, whatever its pair count; odd pair counts in between so that the cache is met in both states
spec = [(4000, 7), (140, 3), (4500, 6), (151, 2), (90, 1), (5000, 5)]
records = [SeqRecord(Seq(random_genome(90 + i, L)), id="fs%d" % i, description="") for i, (L, _n) in enumerate(spec)]
counts = [n for _L, n in spec]
prefix = os.path.join(work, case)
generator.worker_iterator([(r, n, "default") for r, n in zip(records, counts)], em, cpu_number, prefix, seed, "metagenomics", False)
blobs = {}
for suffix in ("_R1.fastq", "_R2.fastq", ".vcf"):
    with open(prefix + suffix, "rb") as fh:
        blobs[suffix] = np.frombuffer(fh.read(), dtype=np.uint8)
assert blobs["_R1.fastq"].tobytes().count(b"\n") == 4 * (7 + 6 + 5)
meta = dict(case=case, model=model, ids=[r.id for r in records], counts=counts, cpu_number=cpu_number, seed=seed,
            sequence_type="metagenomics", gc_bias=False, store_mutations=False, fragment_length=frag, fragment_sd=sd)
os.makedirs(os.path.join(GOLDEN, "worker"), exist_ok=True)
np.savez_compressed(os.path.join(GOLDEN, "worker", case + ".npz"), r1=blobs["_R1.fastq"], r2=blobs["_R2.fastq"],
                    vcf=blobs[".vcf"], meta=np.array(json.dumps(meta)),
                    **{"genome_%d" % i: np.frombuffer(str(r.seq).encode("ascii"), dtype=np.uint8) for i, r in enumerate(records)})
print("worker", case, len(blobs["_R1.fastq"]))
shutil.rmtree(work)
