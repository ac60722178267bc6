#!/usr/bin/env python3
"""Throughput of the sequential walker (k_mt_walk): BasicErrorModel, custom fragment length, --store_mutations."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import dense_model, random_genome  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for name, model, frag, mut in (("basic", "basic", None, False), ("novaseq + fragment length", "novaseq", (400, 30), False),
                               ("novaseq + store_mutations", "novaseq", None, True), ("miseq-legacy (indel-heavy)", "miseq-legacy", None, False)):
    with ReadEngine(0) as eng:
        eng.load_model(dense_model(model))
        gid = eng.add_genome(random_genome(1, 2000000))
        eng.seed_mt(42)
        if frag:
            eng.mt_set_fragment(*frag)
        if mut:
            eng.mt_mutations_reserve(8 * n)
        eng.generate_mt(gid, 1000)
        t0 = time.perf_counter()
        assert eng.generate_mt(gid, n) == n
        dt = time.perf_counter() - t0
        print("%s: %d pairs in %.3f s = %.0f pairs/s, paths (resolved, walked) %s" % (name, n, dt, n / dt, eng.mt_path_counts()),
              flush=True)
