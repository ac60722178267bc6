#!/usr/bin/env python3
"""BASELINE configs[4] as written: "a custom .npz from data/ecoli.bam via iss model".  The reference's own `iss model`
(iss/app.py:147-170 -> iss/bam.py:103-227 -> iss/modeller.py) builds the model from the reference's own BAM file on top of a
stand-in for pysam (tests/golden/tooling/pysam_shim: pysam is absent from the build container) -- after the reference's own
bam / modeller tests have passed on that stand-in (iss/test/test_bam.py, iss/test/test_modeller.py: read_1_2's mismatch,
read_4_1's insertion).  Then, like make_golden.py: the model's dense tables and a pair set of the reference's reads_generator
under random.seed(s); np.random.seed(s) with the next doubles of both streams.

Outputs: tests/golden/models/ecoli-bam.dense.npz, tests/golden/pairs/ecoli_bam_acgt.npz, tests/golden/pairs/ecoli_bam_gc.npz

Usage:  python tests/golden/tooling/make_golden_bam_model.py   (from the repo root, build container only)
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
REPO = os.path.dirname(os.path.dirname(GOLDEN))
REF = "/root/reference"

work = tempfile.mkdtemp(prefix="iss_ref_")
REFCOPY = os.path.join(work, "refcopy")
shutil.copytree(REF, REFCOPY)
subprocess.check_call(["chmod", "-R", "u+w", REFCOPY])
paths = [os.path.join(HERE, "pysam_shim"), os.path.join(HERE, "bio_shim"), REFCOPY]
env = dict(os.environ, PYTHONPATH=":".join(paths))
# the reference's own tests of the BAM reader and the modeller must pass on the stand-in
subprocess.check_call([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "iss/test/test_bam.py", "iss/test/test_modeller.py"],
                      env=env, cwd=REFCOPY)
out_prefix = os.path.join(work, "ecoli_bam")
subprocess.check_call([sys.executable, "-m", "iss", "model", "-b", "data/ecoli.bam", "-o", out_prefix], env=env, cwd=REFCOPY)
for p in reversed(paths):
    sys.path.insert(0, p)
sys.path.insert(0, REPO)
os.chdir(REFCOPY)
from Bio.Seq import Seq  # noqa: E402
from Bio.SeqRecord import SeqRecord  # noqa: E402

from iss import generator  # noqa: E402
from iss.error_models import kde  # noqa: E402

from insilicoseq_amd.model import DenseModel  # noqa: E402

os.makedirs(os.path.join(GOLDEN, "models"), exist_ok=True)
dense = DenseModel.from_reference_npz(out_prefix + ".npz")
dense.save(os.path.join(GOLDEN, "models", "ecoli-bam.dense.npz"))
z = np.load(out_prefix + ".npz", allow_pickle=True)
print("iss model: read_length", int(z["read_length"]), "mean counts", z["mean_count_forward"], z["mean_count_reverse"],
      "non-zero / NaN indel entries", int(np.count_nonzero(np.nan_to_num(dense.ins)) + np.count_nonzero(np.nan_to_num(dense.dele))),
      int(np.isnan(dense.ins).sum() + np.isnan(dense.dele).sum()))


def genome(seed, n):
    rnd = random.Random(seed)
    return "".join(rnd.choice("ACGT") for _ in range(n))


def to_u8(s):
    return np.frombuffer(s.encode("ascii"), dtype=np.uint8)


for case, gseed, seed, n_pairs, gc_bias in (("ecoli_bam_acgt", 11, 71, 160, False), ("ecoli_bam_gc", 12, 72, 160, True)):
    em = kde.KDErrorModel(out_prefix + ".npz")
    g = genome(gseed, 3000)
    random.seed(seed)
    np.random.seed(seed)
    rec = SeqRecord(Seq(g), id="g", description="")
    RL = int(em.read_length)
    out = [np.zeros((n_pairs, RL), dtype=np.uint8) for _ in range(4)]
    n = 0
    for fwd, rev, _ in generator.reads_generator(n_pairs, rec, em, 0, gc_bias, "metagenomics"):
        out[0][n] = to_u8(str(fwd.seq))
        out[1][n] = fwd.letter_annotations["phred_quality"]
        out[2][n] = to_u8(str(rev.seq))
        out[3][n] = rev.letter_annotations["phred_quality"]
        n += 1
    tail_py = np.array([random.random() for _ in range(4)])
    tail_np = np.array([np.random.random_sample() for _ in range(4)])
    meta = dict(case=case, model="ecoli-bam", genome="acgt3k_%d" % gseed, seed=seed, n_pairs=n_pairs, n_done=n, sequence_type="metagenomics",
                fragment_length=None, fragment_sd=None, gc_bias=gc_bias, indel=None)
    np.savez_compressed(os.path.join(GOLDEN, "pairs", case + ".npz"), r1_base=out[0], r1_qual=out[1], r2_base=out[2], r2_qual=out[3],
                        tail_py=tail_py, tail_np=tail_np, genome=to_u8(g), meta=np.array(json.dumps(meta)))
    print("pairs", case, n)
shutil.rmtree(work, ignore_errors=True)
