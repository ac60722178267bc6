"""k_main_g (round 6): the rows of a group of passes wait in registers until the group's deferred bases are settled, the exact
path's byte patches follow the rows out.  Every instantiation the library holds, forced and as the host picks it, against the CPU
oracle -- with FEW workgroups, so that a workgroup makes many passes (groups that end early, entries carried into the next
group, the descriptors two groups ahead) -- and the model that defers every base (the ring fills inside a group: rows leave
early, rounds run at once)."""
import numpy as np
import pytest

from helpers import dense_model, mixed_genome, random_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from insilicoseq_amd.engine import ReadEngine

    eng = ReadEngine(0)
    yield eng
    eng.close()


def _compare(engine, dense, genome, n_pairs, seed, first_ordinal=0):
    from oracle import oracle as O

    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.generate(gid, n_pairs, first_ordinal=first_ordinal, seed=seed)
    engine.synchronize()
    got = engine.download(0, n_pairs)
    exp = O.Oracle(dense).simulate(O.Rng().seed_philox(seed), genome, n_pairs, first_ordinal=first_ordinal)
    assert exp["status"] == 0 and exp["n_done"] == n_pairs
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        bad = np.argwhere(got[k] != exp[k])
        assert bad.size == 0, "%s differs at (pair, pos) %s ... (%d cells)" % (k, bad[:5].tolist(), len(bad))
    return engine.main_kernel()


# (model, passes per group to force, the kernel that must have run): read lengths 151 / 126 / 301 / 301 / 20 / 125 -> 5 / 2 / 2 / 4 / 1 / 4
# iterations per pass of a position tile
FORCED = [("novaseq", 1, "k_main_g<5, 1>"), ("hiseq", 1, "k_main_g<2, 1>"), ("hiseq", 2, "k_main_g<2, 2>"), ("miseq", 1, "k_main_g<2, 1>"),
          ("miseq", 2, "k_main_g<2, 2>"), ("nextseq", 1, "k_main_g<4, 1>"), ("ecoli", 2, "k_main_g<1, 2>"), ("basic", 1, "k_main_g<4, 1>")]


@pytest.mark.parametrize("wgs", ["3", "40"])
@pytest.mark.parametrize("model,np_,kernel", FORCED)
def test_every_instantiation_matches_the_oracle(engine, model, np_, kernel, wgs, monkeypatch):
    monkeypatch.setenv("ISS_MAIN_GROUP", str(np_))
    monkeypatch.setenv("ISS_MAIN_WGS", wgs)
    n = 9000 if model in ("miseq", "nextseq") else 14000  # (3 workgroups: ~12-18 passes each; partial last blocks)
    assert _compare(engine, dense_model(model), random_genome(71, 150000), n + 37, 4242, first_ordinal=2**33 + 11) == kernel


@pytest.mark.parametrize("model,expect", [("novaseq", "k_main<false, true, false>"), ("hiseq", "k_main_g<2, 2>"), ("miseq", "k_main_g<2, 1>"),
                                          ("nextseq", "k_main_g<4, 1>")])
def test_the_host_picks_the_kernel_by_the_model(engine, model, expect, monkeypatch):
    """Groups of about one round's worth of deferred lane-items; a model that defers little keeps k_main (iss_mi355x.hip:
    main_group_passes).  ISS_MAIN_GROUP=0 is k_main for everybody; records with IUPAC letters take k_main as well."""
    monkeypatch.delenv("ISS_MAIN_GROUP", raising=False)
    assert _compare(engine, dense_model(model), random_genome(72, 100000), 6000, 99) == expect
    monkeypatch.setenv("ISS_MAIN_GROUP", "0")
    assert _compare(engine, dense_model(model), random_genome(72, 100000), 6000, 99) == "k_main<false, true, false>"
    monkeypatch.delenv("ISS_MAIN_GROUP", raising=False)
    assert _compare(engine, dense_model(model), mixed_genome(73, 40000), 3000, 98) == "k_main<false, false, false>"


@pytest.mark.parametrize("np_", [1, 2])
@pytest.mark.parametrize("min_round", ["1", "64"])
def test_a_model_that_defers_every_base(engine, np_, min_round, monkeypatch):
    """Every phred's substitution test fires (threshold 0: `u > 0`): 16 deferred bases per lane-item, 64 entries per wavefront
    and iteration -- the ring cannot take a group's pushes, the rows computed so far leave early and rounds run inside the
    group; the closing round's patches wait for the rows; entries come back sixteen times.  (The quality rows, and with them the
    position tiles, stay HiSeq's.)"""
    d = dense_model("hiseq")
    d.phred_thr[:] = 0.0
    monkeypatch.setenv("ISS_MAIN_GROUP", str(np_))
    monkeypatch.setenv("ISS_MAIN_GROUP_MIN", min_round)
    monkeypatch.setenv("ISS_MAIN_WGS", "2")
    assert _compare(engine, d, random_genome(74, 50000), 2600, 5).startswith("k_main_g<2, %d>" % np_)


def test_groups_of_a_launch_of_many_chunks(engine, monkeypatch):
    """Chunks of a call (ISS_CHUNK_PAIRS) each start their groups anew; the last chunk is short."""
    monkeypatch.setenv("ISS_CHUNK_PAIRS", "2000")
    monkeypatch.setenv("ISS_MAIN_WGS", "2")
    assert _compare(engine, dense_model("hiseq"), random_genome(75, 80000), 7001, 6) == "k_main_g<2, 2>"
