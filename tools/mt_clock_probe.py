#!/usr/bin/env python3
"""Does MT-mode throughput (3 active workgroups) depend on how busy the rest of the GPU is (clock ramp)?"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import dense_model, random_genome  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402


def run(tag):
    dense = dense_model("novaseq")
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(random_genome(1, 2000000))
        eng.seed_mt(42)
        eng.generate_mt(gid, 1000)
        t0 = time.perf_counter()
        n = 200000
        assert eng.generate_mt(gid, n) == n
        dt = time.perf_counter() - t0
        print("%s: %.0f pairs/s" % (tag, n / dt), flush=True)


run("idle GPU")
stop = False


def burn():
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    while not stop:
        for _ in range(20):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()


th = threading.Thread(target=burn)
th.start()
time.sleep(1.0)
run("busy GPU (bf16 matmuls on another stream)")
stop = True
th.join()
os.system("rocm-smi --showclocks 2>/dev/null | head -20")
