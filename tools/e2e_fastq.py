#!/usr/bin/env python3
"""End-to-end rate of one worker: genomes -> GPU -> FASTQ files on tmpfs (SURVEY.md section 8 d: "report both")."""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from helpers import dense_model, random_genome  # noqa: E402
from insilicoseq_amd.generator import Record, lognormal_abundance, worker_iterator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=5_000_000)
ap.add_argument("--mt-pairs", type=int, default=500_000)
ap.add_argument("--model", default="novaseq")
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--compress", action="store_true", help="gzip members built on the device (iss_fastq_compress)")
args = ap.parse_args()

dense = dense_model(args.model)
recs = [Record(random_genome(123 + k, 5_000_000), id="genome_%d" % k) for k in range(5)]
ab = lognormal_abundance([r.id for r in recs], np.random.RandomState(123))
for rng, total in (("philox", args.pairs), ("mt", args.mt_pairs)):
    counts = [int(total * ab[r.id]) for r in recs]
    work = [(r, n, "default") for r, n in zip(recs, counts)]
    d = tempfile.mkdtemp(dir=args.dir)
    try:
        prefix = os.path.join(d, "w")
        worker_iterator([(recs[0], 1000, "default")], dense, 0, prefix, 42, "metagenomics", False, device=0, rng=rng,
                        compress=args.compress)  # warm-up
        t0 = time.perf_counter()
        worker_iterator(work, dense, 0, prefix, 42, "metagenomics", False, device=0, rng=rng, compress=args.compress)
        dt = time.perf_counter() - t0
        size = os.path.getsize(prefix + "_R1.fastq") + os.path.getsize(prefix + "_R2.fastq")
        print("%s rng=%s%s: %d pairs -> %.2f GB of %s in %.2f s = %.3g pairs/s end to end (%.2f GB/s written)" % (
            args.model, rng, " compress" if args.compress else "", sum(counts), size / 1e9,
            "gzip members" if args.compress else "FASTQ", dt, sum(counts) / dt, size / 1e9 / dt), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
