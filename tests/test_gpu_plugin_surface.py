"""The inner plugin surface on the GPU: the batched C-ABI entries behind the reference's ErrorModel
methods -- gen_phred_scores, mut_sequence, introduce_indels (+ adjust_seq_length), random_insert_size -- and their mirrors
on insilicoseq_amd.model.KDErrorModel, against the CPU oracle's function-level entry points (the functions
tests/test_oracle_golden.py pins to the reference's own unit goldens, iss/test/test_error_model.py:30-105), same Philox
address per read.  Bit-exact."""
import os

import numpy as np
import pytest

from helpers import dense_model, mixed_genome, random_genome

pytestmark = pytest.mark.gpu

SEED = 2024


@pytest.fixture(scope="module")
def engine():
    from insilicoseq_amd.engine import ReadEngine

    eng = ReadEngine(0)
    yield eng
    eng.close()


@pytest.mark.parametrize("model", ["novaseq", "hiseq", "miseq", "ecoli"])
def test_gen_phred_scores_and_insert_size(engine, model):
    from oracle import oracle as O

    dense = dense_model(model)
    engine.load_model(dense)
    orc = O.Oracle(dense)
    n, first = 300, 2**33 + 11
    isz = engine.random_insert_size(n, first, SEED)
    for o in (0, 1):
        got = engine.gen_phred_scores(o, n, first, SEED)
        for i in range(n):
            rng = O.Rng().seed_philox(SEED)
            rng.set_address(first + i)
            assert np.array_equal(got[i], orc.gen_phred_scores(rng, o)), (o, i)
            if o == 0:
                assert isz[i] == orc.random_insert_size(rng)


@pytest.mark.parametrize("model,indel", [("novaseq", None), ("hiseq", None), ("novaseq", (0.02, 0.05)), ("miseq-legacy", None)])
def test_mut_sequence(engine, model, indel):
    from oracle import oracle as O

    dense = dense_model(model, indel)
    engine.load_model(dense)
    orc = O.Oracle(dense)
    RL, n = dense.read_length, 200
    rng0 = np.random.RandomState(5)
    seqs = np.frombuffer(b"ACGTacgtNRYK", dtype=np.uint8)[rng0.choice(12, size=(n, RL), p=[.22] * 4 + [.02] * 4 + [.01] * 4)]
    quals = rng0.randint(0, 41, size=(n, RL)).astype(np.uint8)
    quals[::3] = rng0.randint(0, 6, size=quals[::3].shape)  # plenty of substitutions
    for o in (0, 1):
        got, st = engine.mut_sequence(o, seqs, quals, 77, SEED)
        for i in range(n):
            rng = O.Rng().seed_philox(SEED)
            rng.set_address(77 + i)
            rc, exp = orc.mut_sequence(rng, seqs[i].tobytes().decode(), quals[i], o)
            assert st[i] == rc, (o, i)
            if rc == 0:
                assert got[i].tobytes().decode() == exp, (o, i)


@pytest.mark.parametrize("indel", [(0.0, 0.0), (0.01, 0.03), (0.2, 0.35), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0)])
def test_introduce_indels(engine, indel):
    from oracle import oracle as O

    dense = dense_model("novaseq", indel)
    engine.load_model(dense)
    orc = O.Oracle(dense)
    RL, n = dense.read_length, 120
    genome = mixed_genome(9, 4000) if indel == (0.01, 0.03) else random_genome(9, 4000)
    g = np.frombuffer(genome.encode(), dtype=np.uint8)
    rng0 = np.random.RandomState(3)
    for o in (0, 1):
        starts = rng0.randint(0, len(genome) - RL, size=n)
        starts[:4] = [0, 1, len(genome) - RL, len(genome) - RL - 2]  # padding runs off either end of the reference
        lens = np.where(rng0.rand(n) < 0.15, rng0.randint(1, RL, size=n), RL)
        seqs = np.zeros((n, RL), dtype=np.uint8)
        bounds = np.zeros((n, 2), dtype=np.int64)
        perfect = []
        for i in range(n):
            piece = genome[starts[i]:starts[i] + lens[i]]
            if o == 1:
                piece = O.rev_comp(piece)
            perfect.append(piece)
            seqs[i, :lens[i]] = np.frombuffer(piece.encode(), dtype=np.uint8)
            bounds[i] = (starts[i], starts[i] + RL)
        got, st = engine.introduce_indels(o, seqs, lens, genome, bounds, 5000, SEED)
        for i in range(n):
            rng = O.Rng().seed_philox(SEED)
            rng.set_address(5000 + i)
            rc, exp = orc.introduce_indels(rng, perfect[i], o, genome, (int(bounds[i, 0]), int(bounds[i, 1])))
            assert st[i] == rc, (o, i, st[i], rc)
            if rc == 0:
                assert got[i].tobytes().decode() == exp, (o, i)
    assert g.size == len(genome)


def test_kd_error_model_methods_match_the_generated_pair(engine):
    """The mirror of the reference's plugin surface on KDErrorModel, read by read: indels -> phred scores -> substitutions at
    one Philox address give the very mate iss_generate produces for that pair (same seed, same ordinal)."""
    from insilicoseq_amd.generator import Record
    from insilicoseq_amd.model import KDErrorModel
    from oracle import oracle as O

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    em = KDErrorModel(os.path.join(root, "insilicoseq_amd", "profiles", "hiseq.dense.npz"))
    genome = random_genome(21, 30000)
    em.bind(engine, seed=SEED, ordinal=0)
    gid = engine.add_genome(genome)
    n = 40
    engine.generate(gid, n, first_ordinal=0, seed=SEED)
    engine.synchronize()
    rows = engine.download(0, n)
    coords = engine.coords(0, n)
    RL = int(em.read_length)
    for i in range(n):
        em.ordinal = i
        fs, rs, re, isz = (int(x) for x in coords[i])
        assert em.random_insert_size() == isz
        for orientation, key_b, key_q, start in (("forward", "r1_base", "r1_qual", fs), ("reverse", "r2_base", "r2_qual", rs)):
            piece = genome[start:start + RL]
            rec = Record(piece if orientation == "forward" else O.rev_comp(piece), id="x")
            rec = em.introduce_indels(rec, orientation, genome, (start, start + RL))
            rec = em.introduce_error_scores(rec, orientation)
            seq = em.mut_sequence(rec, orientation)
            assert seq == rows[key_b][i].tobytes().decode(), (i, orientation)
            assert rec.letter_annotations["phred_quality"] == rows[key_q][i].tolist(), (i, orientation)


@pytest.mark.parametrize("case", ["novaseq", "miseq-legacy", "configs4", "extremes"])
def test_event_process_step_equals_oracle(engine, case):
    """iss_ev_step: the device function the indel kernels loop over (one draw of the event process: state + uniform ->
    next firing test + event mask) against the oracle's ev_step, whose interval structure
    tests/test_oracle_golden.py::test_event_process_exact_* pins to the reference's per-test probabilities
    (iss/error_models/__init__.py:193-196, :209).  Random states x numerators of every magnitude, both mates."""
    from oracle import oracle as O

    if case in ("novaseq", "miseq-legacy"):
        dense = dense_model(case)
    elif case == "configs4":
        dense = dense_model("novaseq", (0.001, 0.003))
    else:
        dense = dense_model("hiseq")
        rs = np.random.RandomState(11)
        dense.ins[:] = rs.choice([0.0, 1e-4, 3e-3, 0.05], size=dense.ins.shape, p=[0.4, 0.3, 0.2, 0.1])
        dense.dele[:] = rs.choice([0.0, 2e-4, 1e-2, 0.2], size=dense.dele.shape, p=[0.3, 0.3, 0.3, 0.1])
        dense.ins[0, 7, 2] = 1.0
        dense.dele[0, 40, :] = [0.5, 1.0, 0.0, 0.25]
        dense.ins[0, 60:64, :] = 0.9
        dense.dele[1, 3, :] = float("nan")
        dense.ins[1, 5, 1] = 5e-324
    engine.load_model(dense)
    orc = O.Oracle(dense)
    ns = 5 * (dense.read_length - 1)
    rs = np.random.RandomState(3)
    n = 400000
    cur = rs.randint(-1, ns - 1, size=n).astype(np.int32)

    def numerators():
        full = rs.randint(0, 1 << 30, size=n).astype(np.uint64) << np.uint64(23) | rs.randint(0, 1 << 23, size=n).astype(np.uint64)
        return full >> rs.randint(0, 53, size=n).astype(np.uint64)

    m53, v53 = numerators(), numerators()
    m53[:1000] = 0                      # fires at the first test with p > 0
    m53[1000:2000] = (1 << 53) - 1      # the largest numerator
    for o in (0, 1):
        got = engine.ev_step(o, cur, m53, v53)
        exp = orc.ev_step(o, cur, m53, v53)
        for g, e, what in zip(got, exp, ("next state", "slot", "mask")):
            assert np.array_equal(g, e), (case, o, what)
        if case != "novaseq":
            assert (exp[1] >= 0).mean() > 0.01  # (the sample does reach firing tests)
    with pytest.raises(Exception):
        engine.ev_step(0, [ns - 1], [0])
    with pytest.raises(Exception):
        engine.ev_step(0, [0], [1 << 53])
