#!/usr/bin/env python3
"""How long do the GPU's clocks take to settle under the bench's step?  The default bench's step (5 M NovaSeq pairs over 5
records, one iss_generate_batch) repeated from a cold start; wall time per group of steps (one synchronize per group).
Usage: python tools/ramp_probe.py [groups] [steps_per_group] [idle_seconds_before]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402
from insilicoseq_amd.model import DenseModel  # noqa: E402

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 25
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
dense = DenseModel.load(os.path.join(ROOT, "insilicoseq_amd", "profiles", "novaseq.dense.npz"))
genomes = bench.synthetic_genomes(5, bench.GENOME_LEN, 123)
rng = np.random.RandomState(123)
w = rng.lognormal(size=5)
pairs = [int(5_000_000 * x / w.sum()) for x in w]
with ReadEngine(0) as eng:
    eng.load_model(dense)
    gids = [eng.add_genome(g) for g in genomes]
    eng.reserve(sum(pairs))
    eng.generate_batch(gids, pairs, first_ordinal=0, seed=1, out_first_pair=0)
    eng.synchronize()
    time.sleep(idle)
    t_all = time.perf_counter()
    for g in range(groups):
        t0 = time.perf_counter()
        for k in range(per):
            eng.generate_batch(gids, pairs, first_ordinal=(g * per + k) * sum(pairs), seed=1, out_first_pair=0)
        eng.synchronize()
        t1 = time.perf_counter()
        print("t=%.3f s  steps %4d..%4d  %.4f ms/step" % (t1 - t_all, g * per, g * per + per - 1, (t1 - t0) / per * 1e3), flush=True)
