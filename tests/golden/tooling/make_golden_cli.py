#!/usr/bin/env python3
"""Golden outputs of `iss generate` for the abundance / coverage / read-count inputs of the reference's CLI
(iss/generator.py:497-594, iss/abundance.py), captured by running the reference like make_golden.py does (same
stand-in Bio package).  Kept separate so that the existing fixtures are not rewritten.

Outputs: tests/golden/generate/cli_<case>.npz (FASTQ files, the abundance / coverage file the run wrote, the
input file it was given).

Usage:  python tests/golden/tooling/make_golden_cli.py   (from the repo root, build container only)
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "bio_shim")
REF = "/root/reference"

work = tempfile.mkdtemp(prefix="iss_ref_")
REFCOPY = os.path.join(work, "refcopy")
shutil.copytree(REF, REFCOPY)
subprocess.check_call(["chmod", "-R", "u+w", REFCOPY])
env = dict(os.environ, PYTHONPATH=SHIM + ":" + REFCOPY)

ids = ["genome_A", "genome_T", "genome_GC", "genome_ATCG", "genome_TA"]
inputs = {
    "abundance_file": "".join("%s\t%s\n" % (r, a) for r, a in zip(ids, [0.3, 0.05, 0.25, 0.35, 0.05])),
    "coverage_file": "".join("%s\t%s\n" % (r, a) for r, a in zip(ids, [27.0, 4.5, 20.25, 36.0, 9.0])),
    "readcount_file": "".join("%s\t%s\n" % (r, a) for r, a in zip(ids, [120, 31, 90, 200, 14])),
}
CASES = {
    "halfnormal": ["--abundance", "halfnormal", "-n", "600"],
    "zero_inflated_lognormal": ["--abundance", "zero_inflated_lognormal", "-n", "600"],
    "exponential": ["--abundance", "exponential", "-n", "600"],
    "uniform": ["--abundance", "uniform", "-n", "600"],
    "coverage_lognormal": ["--coverage", "lognormal", "-n", "600"],
    "coverage_halfnormal": ["--coverage", "halfnormal", "-n", "500"],
    "abundance_file": ["--abundance_file", "@abundance_file", "-n", "600"],
    "coverage_file": ["--coverage_file", "@coverage_file", "-n", "600"],
    "readcount_file": ["--readcount_file", "@readcount_file"],
}
os.makedirs(os.path.join(GOLDEN, "generate"), exist_ok=True)
for case, flags in CASES.items():
    outp = os.path.join(work, "cli_" + case)
    argv, given = [], b""
    for f in flags:
        if f.startswith("@"):
            path = os.path.join(work, f[1:] + ".txt")
            with open(path, "w") as fh:
                fh.write(inputs[f[1:]])
            given = inputs[f[1:]].encode()
            argv.append(path)
        else:
            argv.append(f)
    subprocess.check_call([sys.executable, "-m", "iss", "generate", "--genomes", "data/genomes.fasta", "--model", "hiseq",
                           "--seed", "42", "--cpus", "2", "-o", outp, "--quiet"] + argv, env=env, cwd=REFCOPY)
    blobs = {}
    for suffix in ("_R1.fastq", "_R2.fastq", "_abundance.txt", "_coverage.txt"):
        blobs[suffix] = np.frombuffer(open(outp + suffix, "rb").read(), dtype=np.uint8) if os.path.exists(outp + suffix) else None
    np.savez_compressed(os.path.join(GOLDEN, "generate", "cli_%s.npz" % case), r1=blobs["_R1.fastq"], r2=blobs["_R2.fastq"],
                        abundance=blobs["_abundance.txt"] if blobs["_abundance.txt"] is not None else np.zeros(0, np.uint8),
                        coverage=blobs["_coverage.txt"] if blobs["_coverage.txt"] is not None else np.zeros(0, np.uint8),
                        has_abundance=np.array(blobs["_abundance.txt"] is not None),
                        has_coverage=np.array(blobs["_coverage.txt"] is not None),
                        given=np.frombuffer(given, dtype=np.uint8), flags=np.array(" ".join(flags)))
    print(case, len(blobs["_R1.fastq"]), "abundance" if blobs["_abundance.txt"] is not None else "",
          "coverage" if blobs["_coverage.txt"] is not None else "")
shutil.rmtree(work, ignore_errors=True)
