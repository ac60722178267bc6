#!/usr/bin/env python3
"""Which allocation's placement moves the step time?  One engine; re-allocate the outputs (reserve a little more), then the
genomes (clear + upload), then the model tables (load_model), timing 100 steps after each."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402
from insilicoseq_amd.model import DenseModel  # noqa: E402

dense = DenseModel.load(os.path.join(ROOT, "insilicoseq_amd", "profiles", "novaseq.dense.npz"))
genomes = bench.synthetic_genomes(5, bench.GENOME_LEN, 123)
rng = np.random.RandomState(123)
w = rng.lognormal(size=5)
pairs = [int(5_000_000 * x / w.sum()) for x in w]


def timed(eng, gids, tag):
    for k in range(10):
        eng.generate_batch(gids, pairs, first_ordinal=k * sum(pairs), seed=1, out_first_pair=0)
    eng.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        eng.generate_batch(gids, pairs, first_ordinal=k * sum(pairs), seed=1, out_first_pair=0)
    eng.synchronize()
    print("%-28s %.4f ms/step  out=%s" % (tag, (time.perf_counter() - t0) / 100 * 1e3, hex(eng.device_ptrs()[0])), flush=True)


with ReadEngine(0) as eng:
    eng.load_model(dense)
    gids = [eng.add_genome(g) for g in genomes]
    eng.reserve(sum(pairs))
    timed(eng, gids, "fresh")
    timed(eng, gids, "same again")
    for i in range(1, 7):
        eng.reserve(sum(pairs) + 8192 * i)
        timed(eng, gids, "outputs re-allocated %d" % i)
    for i in range(1, 5):
        eng.clear_genomes()
        gids = [eng.add_genome(g) for g in genomes]
        timed(eng, gids, "genomes re-uploaded %d" % i)
    for i in range(1, 7):
        eng.load_model(dense)
        eng.clear_genomes()
        gids = [eng.add_genome(g) for g in genomes]
        eng.reserve(sum(pairs) + 8192 * 6)
        timed(eng, gids, "model re-loaded %d" % i)
for i in range(1, 5):
    with ReadEngine(0) as eng2:
        eng2.load_model(dense)
        gids = [eng2.add_genome(g) for g in genomes]
        eng2.reserve(sum(pairs))
        timed(eng2, gids, "new engine %d" % i)
