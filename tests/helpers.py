"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from insilicoseq_amd.model import DenseModel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROFILES = os.path.join(os.path.dirname(GOLDEN), "..", "insilicoseq_amd", "profiles")


def dense_model(name, indel=None):
    d = DenseModel.load(os.path.join(PROFILES, name + ".dense.npz"))
    if indel is not None:
        d.ins[:] = indel[0]
        d.dele[:] = indel[1]
    return d


def load_pairs_case(case):
    z = np.load(os.path.join(GOLDEN, "pairs", case + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def pairs_cases():
    return sorted(f[:-4] for f in os.listdir(os.path.join(GOLDEN, "pairs")) if f.endswith(".npz"))


def random_genome(seed, n, alphabet="ACGT"):
    rng = np.random.RandomState(seed)
    idx = rng.randint(0, len(alphabet), size=n)
    return np.frombuffer(alphabet.encode(), dtype=np.uint8)[idx].tobytes().decode()


def mixed_genome(seed, n):
    rng = np.random.RandomState(seed)
    x = rng.random_sample(n)
    plain = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, n)]
    lower = np.frombuffer(b"acgt", dtype=np.uint8)[rng.randint(0, 4, n)]
    amb = np.frombuffer(b"NRYWSMKHBVD", dtype=np.uint8)[rng.randint(0, 11, n)]
    ambl = np.frombuffer(b"nrywsmkhbvd", dtype=np.uint8)[rng.randint(0, 11, n)]
    out = np.where(x < 0.8, plain, np.where(x < 0.92, lower, np.where(x < 0.97, amb, ambl)))
    return out.astype(np.uint8).tobytes().decode()
