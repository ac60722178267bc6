#!/usr/bin/env python3
"""Does the step time depend on where the engine's buffers land?  Several engines one after the other in ONE process (each
allocates its own outputs / tables), 100 steps of the default bench's work list each.
Usage: python tools/placement_probe.py [engines] [keep_alive 0/1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402
from insilicoseq_amd.model import DenseModel  # noqa: E402

n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 6
keep = len(sys.argv) > 2 and sys.argv[2] == "1"
dense = DenseModel.load(os.path.join(ROOT, "insilicoseq_amd", "profiles", "novaseq.dense.npz"))
genomes = bench.synthetic_genomes(5, bench.GENOME_LEN, 123)
rng = np.random.RandomState(123)
w = rng.lognormal(size=5)
pairs = [int(5_000_000 * x / w.sum()) for x in w]
alive = []
for e in range(n_eng):
    eng = ReadEngine(0)
    eng.load_model(dense)
    gids = [eng.add_genome(g) for g in genomes]
    eng.reserve(sum(pairs))
    ptrs = eng.device_ptrs() if hasattr(eng, "device_ptrs") else None
    for rep in range(3):
        for k in range(10):
            eng.generate_batch(gids, pairs, first_ordinal=k * sum(pairs), seed=1, out_first_pair=0)
        eng.synchronize()
        t0 = time.perf_counter()
        for k in range(100):
            eng.generate_batch(gids, pairs, first_ordinal=k * sum(pairs), seed=1, out_first_pair=0)
        eng.synchronize()
        t1 = time.perf_counter()
        print("engine %d rep %d: %.4f ms/step  out=%s" % (e, rep, (t1 - t0) / 100 * 1e3, hex(ptrs[0]) if ptrs else "?"), flush=True)
    if keep:
        alive.append(eng)
    else:
        eng.close()
