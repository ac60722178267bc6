#!/usr/bin/env python3
"""Per-record overhead of one worker: many small records (a draft genome's contigs), few pairs each."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import dense_model, random_genome  # noqa: E402
from insilicoseq_amd.generator import Record, worker_iterator  # noqa: E402

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dense = dense_model("novaseq")
big = random_genome(5, 20000 * 50)
recs = [Record(big[(k % 50) * 20000:(k % 50 + 1) * 20000], id="contig_%d" % k) for k in range(n_rec)]
work = [(r, 50, "default") for r in recs]
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    for rng in ("philox", "mt"):
        prefix = os.path.join(d, "w_" + rng)
        worker_iterator(work[:20], dense, 0, prefix, 42, "metagenomics", False, device=0, rng=rng)
        t0 = time.perf_counter()
        worker_iterator(work, dense, 0, prefix, 42, "metagenomics", False, device=0, rng=rng)
        dt = time.perf_counter() - t0
        print("rng=%s: %d records x 50 pairs in %.2f s = %.0f us per record (%.0f records/s)" % (
            rng, n_rec, dt, dt / n_rec * 1e6, n_rec / dt), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
