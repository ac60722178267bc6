"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from insilicoseq_amd.model import DenseModel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROFILES = os.path.join(os.path.dirname(GOLDEN), "..", "insilicoseq_amd", "profiles")


def dense_model(name, indel=None):
    path = os.path.join(PROFILES, name + ".dense.npz")
    if not os.path.exists(path):  # models minted for the tests only (tests/golden/models: the reference's `iss model` on data/ecoli.bam)
        path = os.path.join(GOLDEN, "models", name + ".dense.npz")
    d = DenseModel.basic() if name == "basic" else DenseModel.load(path)
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


def synthetic_model(read_length, n_q, n_isize, seed, indel=(1e-3, 2e-3), nonempty=((1, 1, 1, 1), (1, 0, 1, 1))):
    """A random but valid dense model at arbitrary sizes (limits of the engine: read_length 1024, 60 phred entries,
    8000 insert sizes).  phred_thr follows util.phred_to_prob like the real profiles."""
    rng = np.random.RandomState(seed)
    RL = read_length

    def cdf(shape):
        w = rng.gamma(0.3, size=shape) + 1e-12
        c = np.cumsum(w, axis=-1)
        c /= c[..., -1:]
        c[..., -1] = 1.0
        return c

    isize = cdf((n_isize,))
    bin_w = rng.random_sample((2, 4)) * np.asarray(nonempty, dtype=np.float64) + 1e-9 * np.asarray(nonempty)
    bin_cdf = np.cumsum(bin_w, axis=1)
    bin_cdf /= bin_cdf[:, -1:]
    qcdf = cdf((2, 4, RL, n_q))
    subst_cdf = cdf((2, RL, 4, 3))
    alts = {0: b"TCG", 1: b"ACG", 2: b"ATG", 3: b"ATC"}  # base order A, T, C, G
    subst_alt = np.zeros((2, RL, 4, 3), dtype=np.uint8)
    for b in range(4):
        subst_alt[:, :, b, :] = np.frombuffer(alts[b], dtype=np.uint8)
    ins = np.full((2, RL, 4), indel[0]) * rng.random_sample((2, RL, 4))
    dele = np.full((2, RL, 4), indel[1]) * rng.random_sample((2, RL, 4))
    ins_letter = np.zeros((2, RL, 4), dtype=np.uint8)
    ins_letter[:] = np.frombuffer(b"ATCG", dtype=np.uint8)
    phred_thr = 1.0 - 10.0 ** (-np.arange(n_q + 1) / 10.0)
    return DenseModel(RL, isize, bin_cdf, np.asarray(nonempty, dtype=np.uint8), qcdf, subst_cdf, subst_alt, ins, ins_letter,
                      dele, phred_thr)
