#!/usr/bin/env python3
"""MT mode (rng="mt"): one worker through iss_generate_mt -- on the 2 Mbp genome / one 200 k-pair call of tools/mt_mode_speed.py
and on bench.py's 5 Mbp genome / 2^18-pair calls -- and W workers side by side (iss_generate_mt_workers).
    python tools/mt_workers_speed.py [model] [W ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import dense_model, random_genome  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "novaseq"
sets = tuple(int(x) for x in sys.argv[2:]) or (1, 2, 8, 32, 64, 128, 256)
dense = dense_model(model)
with ReadEngine(0) as eng:  # the shape of tools/mt_mode_speed.py
    eng.load_model(dense)
    gid = eng.add_genome(random_genome(1, 2000000))
    eng.seed_mt(42)
    eng.generate_mt(gid, 1000)
    for n in (200000, 200000, 1 << 18):
        t0 = time.perf_counter()
        assert eng.generate_mt(gid, n) == n
        dt = time.perf_counter() - t0
        print("one worker, 2 Mbp genome, one call of %d pairs: %.0f pairs/s" % (n, n / dt), flush=True)
genome = bench.synthetic_genomes(1, bench.GENOME_LEN, 123)[0]
print(json.dumps(bench.mt_mode_leg(0, dense, genome, worker_sets=sets), indent=1))
