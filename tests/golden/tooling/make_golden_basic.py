#!/usr/bin/env python3
"""Golden vectors of the reference's BasicErrorModel (`iss generate --mode basic`), captured by importing the
reference like make_golden.py does (same stand-in Bio package, same file formats).  Kept separate so that the
existing fixtures are not rewritten.

Outputs: tests/golden/pairs/basic_*.npz, tests/golden/worker/genomes_basic_cpu2.npz,
tests/golden/generate/genomes_basic_n400_seed42_cpus2.npz.

Usage:  python tests/golden/tooling/make_golden_basic.py   (from the repo root, build container only)
"""
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
REPO = os.path.dirname(os.path.dirname(GOLDEN))
SHIM = os.path.join(HERE, "bio_shim")
REF = "/root/reference"

work = tempfile.mkdtemp(prefix="iss_ref_")
REFCOPY = os.path.join(work, "refcopy")
shutil.copytree(REF, REFCOPY)
subprocess.check_call(["chmod", "-R", "u+w", REFCOPY])
sys.path.insert(0, REFCOPY)
sys.path.insert(0, SHIM)
os.chdir(REFCOPY)
env = dict(os.environ, PYTHONPATH=SHIM + ":" + REFCOPY)

from Bio.Seq import Seq  # noqa: E402
from Bio.SeqRecord import SeqRecord  # noqa: E402

from iss import generator  # noqa: E402
from iss.error_models import basic  # noqa: E402


def random_genome(seed, n, alphabet="ACGT"):
    rnd = random.Random(seed)
    return "".join(rnd.choice(alphabet) for _ in range(n))


def mixed_genome(seed, n):
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        x = rnd.random()
        if x < 0.80:
            out.append(rnd.choice("ACGT"))
        elif x < 0.92:
            out.append(rnd.choice("acgt"))
        elif x < 0.97:
            out.append(rnd.choice("NRYWSMKHBVD"))
        else:
            out.append(rnd.choice("nrywsmkhbvd"))
    return "".join(out)


GENOMES = {  # same generators / seeds as make_golden.py
    "acgt20k": random_genome(1, 20000), "mixed5k": mixed_genome(2, 5000), "short420": random_genome(3, 420),
    "amplicon700": random_genome(6, 700), "acgt3k": random_genome(7, 3000),
}

# (case, genome, seed, n_pairs, seq_type, frag, sd, gc_bias)
CASES = [
    ("basic_acgt", "acgt20k", 63, 96, "metagenomics", None, None, False),
    ("basic_mixed", "mixed5k", 64, 64, "metagenomics", None, None, False),
    ("basic_frag300", "acgt3k", 65, 96, "metagenomics", 300, 30, False),
    ("basic_gcbias", "acgt20k", 66, 96, "metagenomics", None, None, True),
    ("basic_amplicon", "amplicon700", 67, 48, "amplicon", None, None, False),
    ("basic_short420", "short420", 68, 64, "metagenomics", None, None, False),
]


def to_u8(s):
    return np.frombuffer(s.encode("ascii"), dtype=np.uint8)


for (case, gkey, seed, n_pairs, seq_type, frag, sd, gc_bias) in CASES:
    em = basic.BasicErrorModel(frag, sd)
    random.seed(seed)
    np.random.seed(seed)
    rec = SeqRecord(Seq(GENOMES[gkey]), id="g", description="")
    RL = int(em.read_length)
    out = [np.zeros((n_pairs, RL), dtype=np.uint8) for _ in range(4)]
    n = 0
    for fwd, rev, _ in generator.reads_generator(n_pairs, rec, em, 0, gc_bias, seq_type):
        out[0][n] = to_u8(str(fwd.seq))
        out[1][n] = fwd.letter_annotations["phred_quality"]
        out[2][n] = to_u8(str(rev.seq))
        out[3][n] = rev.letter_annotations["phred_quality"]
        n += 1
    tail_py = np.array([random.random() for _ in range(4)])
    tail_np = np.array([np.random.random_sample() for _ in range(4)])
    meta = dict(case=case, model="basic", genome=gkey, seed=seed, n_pairs=n_pairs, n_done=n, sequence_type=seq_type,
                fragment_length=frag, fragment_sd=sd, gc_bias=gc_bias, indel=None)
    np.savez_compressed(os.path.join(GOLDEN, "pairs", case + ".npz"), r1_base=out[0], r1_qual=out[1], r2_base=out[2],
                        r2_qual=out[3], tail_py=tail_py, tail_np=tail_np, genome=to_u8(GENOMES[gkey]),
                        meta=np.array(json.dumps(meta)))
    print("pairs", case, n)

# worker_iterator with the basic model (store_mutations on: VCF rows too)
from Bio import SeqIO  # noqa: E402

records = list(SeqIO.parse("data/genomes.fasta", "fasta"))
counts = [40, 11, 23, 17, 9]
em = basic.BasicErrorModel(None, None, True)
prefix = os.path.join(work, "wbasic")
generator.worker_iterator([(r, n, "default") for r, n in zip(records, counts)], em, 2, prefix, 42, "metagenomics", False)
blobs = {}
for suffix in ("_R1.fastq", "_R2.fastq", ".vcf"):
    with open(prefix + suffix, "rb") as fh:
        blobs[suffix] = np.frombuffer(fh.read(), dtype=np.uint8)
meta = dict(case="genomes_basic_cpu2", model="basic", ids=[r.id for r in records], counts=counts, cpu_number=2, seed=42,
            sequence_type="metagenomics", gc_bias=False, store_mutations=True, fragment_length=None, fragment_sd=None)
np.savez_compressed(os.path.join(GOLDEN, "worker", "genomes_basic_cpu2.npz"), r1=blobs["_R1.fastq"], r2=blobs["_R2.fastq"],
                    vcf=blobs[".vcf"], meta=np.array(json.dumps(meta)),
                    **{"genome_%d" % i: to_u8(str(r.seq)) for i, r in enumerate(records)})
print("worker genomes_basic_cpu2")

# `iss generate --mode basic` end to end
outp = os.path.join(work, "gen_basic")
subprocess.check_call([sys.executable, "-m", "iss", "generate", "--genomes", "data/genomes.fasta", "--mode", "basic", "-n",
                       "400", "--seed", "42", "--cpus", "2", "-o", outp, "--quiet"], env=env, cwd=REFCOPY)
blobs = {}
for suffix in ("_R1.fastq", "_R2.fastq", "_abundance.txt"):
    with open(outp + suffix, "rb") as fh:
        blobs[suffix] = np.frombuffer(fh.read(), dtype=np.uint8)
np.savez_compressed(os.path.join(GOLDEN, "generate", "genomes_basic_n400_seed42_cpus2.npz"), r1=blobs["_R1.fastq"],
                    r2=blobs["_R2.fastq"], abundance=blobs["_abundance.txt"],
                    sha_r1=np.array(hashlib.sha256(blobs["_R1.fastq"].tobytes()).hexdigest()))
print("generate basic", len(blobs["_R1.fastq"]))
shutil.rmtree(work, ignore_errors=True)
