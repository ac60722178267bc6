#!/usr/bin/env python3
"""Throughput of the reference-compatible MT mode (one sequential wavefront per worker)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from helpers import dense_model, random_genome  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402

for model, n in (("novaseq", 200000), ("hiseq", 200000), ("miseq", 100000)):
    dense = dense_model(model)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(random_genome(1, 2000000))
        eng.seed_mt(42)
        eng.generate_mt(gid, 1000)
        t0 = time.perf_counter()
        assert eng.generate_mt(gid, n) == n
        dt = time.perf_counter() - t0
        print("%s: %d pairs in %.3f s = %.0f pairs/s (MT-compatible mode)" % (model, n, dt, n / dt))
