#!/usr/bin/env python3
"""One MT worker's rate with / without torch's CUDA context and after other engines have come and gone (bench.py's situation)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if "--torch" in sys.argv:  # (as bench.py does: torch first -- once the library has initialised HIP, torch finds no GPU)
    import torch

    torch.cuda.set_device(0)
    _x = torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
import bench  # noqa: E402
from helpers import dense_model  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402

dense = dense_model("novaseq")
genome = bench.synthetic_genomes(1, bench.GENOME_LEN, 123)[0]


def rate(tag):
    with ReadEngine(0) as mt:
        mt.load_model(dense)
        gm = mt.add_genome(genome)
        mt.seed_mt(42)
        mt.generate_mt(gm, 2000)
        n = 1 << 18
        t0 = time.perf_counter()
        assert mt.generate_mt(gm, n) == n
        mt.synchronize()
        print("%-60s %8.0f pairs/s" % (tag, n / (time.perf_counter() - t0)), flush=True)


rate("fresh process" + (" (torch's CUDA context alive)" if "--torch" in sys.argv else ""))
for k in range(3):
    with ReadEngine(0) as e:
        e.load_model(dense)
        g = e.add_genome(genome)
        e.generate(g, 100000, first_ordinal=0, seed=1)
        e.synchronize()
rate("after three engines came and went")
if "--torch" in sys.argv:
    rate("with torch's CUDA context alive")
    keep = ReadEngine(0)
    keep.load_model(dense)
    gk = keep.add_genome(genome)
    keep.timing_enable(2)
    keep.generate(gk, 1000000, first_ordinal=0, seed=1)
    keep.synchronize()
    rate("... and a Philox engine with kernel events alive")
    keep.close()
