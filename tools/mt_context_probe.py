#!/usr/bin/env python3
"""Why does one MT-mode worker run at 3.7e5 pairs/s by itself and at 2.3e5 inside bench.py?  The single-worker rate cold, right
behind 20 s of the Philox path's k_main at full load (bench.py's legs in front of the MT leg), and again after pauses; the SMI's
clocks beside each."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import dense_model  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402

dense = dense_model("novaseq")
genome = bench.synthetic_genomes(1, bench.GENOME_LEN, 123)[0]


def smi(tag):
    out = subprocess.run("rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i 'sclk\\|mclk\\|power\\|junction' | head -6", shell=True,
                         capture_output=True, text=True).stdout
    print("   [%s] %s" % (tag, " | ".join(x.strip() for x in out.splitlines())), flush=True)


def mt_rate(eng, gid, tag, n=1 << 18):
    t0 = time.perf_counter()
    assert eng.generate_mt(gid, n) == n
    eng.synchronize()
    dt = time.perf_counter() - t0
    print("%-46s %8.0f pairs/s" % (tag, n / dt), flush=True)


with ReadEngine(0) as mt, ReadEngine(0) as ph:
    mt.load_model(dense)
    gm = mt.add_genome(genome)
    mt.seed_mt(42)
    mt.generate_mt(gm, 2000)
    ph.load_model(dense)
    gp = ph.add_genome(genome)
    ph.reserve(5_000_000)
    smi("idle")
    mt_rate(mt, gm, "one MT worker, cold")
    mt_rate(mt, gm, "one MT worker, again")
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 20.0:
        for _ in range(50):
            ph.generate(gp, 5_000_000, first_ordinal=steps * 5_000_000, seed=1)
            steps += 1
        ph.synchronize()
    print("%d Philox steps of 5 M pairs in %.1f s" % (steps, time.perf_counter() - t0), flush=True)
    smi("right behind the load")
    mt_rate(mt, gm, "one MT worker, right behind 20 s of k_main")
    smi("after it")
    mt_rate(mt, gm, "one MT worker, again")
    time.sleep(5)
    mt_rate(mt, gm, "one MT worker, 5 s later")
    time.sleep(15)
    smi("20 s later")
    mt_rate(mt, gm, "one MT worker, 20 s later")
with ReadEngine(0) as mt:  # the other Philox engine closed (its 3.2 GB of rows back with the allocator)
    mt.load_model(dense)
    gm = mt.add_genome(genome)
    mt.seed_mt(42)
    mt.generate_mt(gm, 2000)
    mt_rate(mt, gm, "one MT worker, fresh engine, no other engine")
