"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle in Philox mode, bit-exact.

The oracle itself is pinned to the reference in MT mode (tests/test_oracle_golden.py); both
providers feed the same semantic function, so HIP == oracle(Philox) closes the chain."""
import os

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


def _compare(engine, dense, genome, n_pairs, seed, first_ordinal=0, sequence_type="metagenomics", gc_bias=False):
    from oracle import oracle as O

    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.generate(gid, n_pairs, first_ordinal=first_ordinal, seed=seed, sequence_type=sequence_type,
                    gc_bias=gc_bias)
    engine.synchronize()
    got = engine.download(0, n_pairs)
    coords = engine.coords(0, n_pairs)
    rng = O.Rng().seed_philox(seed)
    exp = O.Oracle(dense).simulate(rng, genome, n_pairs, first_ordinal=first_ordinal, sequence_type=sequence_type,
                                   gc_bias=gc_bias, want_coords=True)
    assert exp["status"] == 0 and exp["n_done"] == n_pairs
    assert (coords == exp["coords"]).all(), "pair coordinates differ"
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        bad = np.argwhere(got[k] != exp[k])
        assert bad.size == 0, "%s differs at (pair, pos) %s ... (%d cells)" % (k, bad[:5].tolist(), len(bad))
    return got, engine.stats_read()


CASES = [
    # (model, indel override, genome builder, n_pairs, seed, first_ordinal, seq_type, gc_bias)
    ("novaseq", None, lambda: random_genome(11, 200000), 6000, 42, 0, "metagenomics", False),
    ("novaseq", None, lambda: mixed_genome(12, 50000), 3000, 2**40 + 17, 2**33 + 5, "metagenomics", False),
    ("novaseq", None, lambda: random_genome(13, 200), 1500, 7, 100, "metagenomics", False),
    ("novaseq", None, lambda: random_genome(14, 700), 1500, 8, 0, "amplicon", False),
    ("novaseq", None, lambda: random_genome(15, 100000), 3000, 9, 12345, "metagenomics", True),
    ("hiseq", None, lambda: random_genome(16, 300000), 5000, 10, 0, "metagenomics", False),
    ("hiseq", None, lambda: mixed_genome(17, 20000), 3000, 11, 0, "metagenomics", True),
    ("miseq", None, lambda: random_genome(18, 100000), 3000, 12, 0, "metagenomics", False),
    ("miseq", None, lambda: random_genome(19, 420), 1000, 13, 0, "metagenomics", False),
    ("miseq-legacy", None, lambda: random_genome(20, 50000), 2500, 14, 0, "metagenomics", False),
    ("miseq-legacy", None, lambda: mixed_genome(21, 30000), 1500, 15, 0, "metagenomics", False),
    ("nextseq", None, lambda: random_genome(22, 50000), 1500, 16, 0, "metagenomics", False),
    ("miseq-36", None, lambda: random_genome(23, 50000), 1000, 17, 0, "metagenomics", False),
    ("ecoli", None, lambda: random_genome(24, 3000), 4000, 18, 0, "metagenomics", False),
    ("novaseq", (0.001, 0.003), lambda: random_genome(25, 100000), 3000, 19, 0, "metagenomics", False),
    ("novaseq", (0.01, 0.03), lambda: mixed_genome(26, 30000), 2000, 20, 0, "metagenomics", False),
    ("novaseq", (0.2, 0.35), lambda: random_genome(27, 5000), 600, 21, 0, "metagenomics", False),
    ("novaseq", (1.0, 0.0), lambda: random_genome(28, 5000), 200, 22, 0, "metagenomics", False),
    ("novaseq", (0.0, 1.0), lambda: random_genome(29, 5000), 200, 23, 0, "metagenomics", False),
    ("novaseq", (1.0, 1.0), lambda: mixed_genome(30, 5000), 200, 24, 0, "amplicon", False),
    # BasicErrorModel on the Philox path (basic.py:40-63): its score distribution as the quality rows, constant insert size
    ("basic", None, lambda: random_genome(32, 150000), 5000, 25, 0, "metagenomics", False),
    ("basic", None, lambda: mixed_genome(33, 20000), 3000, 26, 2**35 + 3, "metagenomics", True),
    ("basic", None, lambda: random_genome(34, 460), 1500, 27, 0, "amplicon", False),
    ("basic", None, lambda: random_genome(35, 300), 1000, 28, 0, "metagenomics", False),  # record < fragment: generator.py:144
    # BASELINE configs[4] as written: the model the reference's own `iss model` builds from its data/ecoli.bam (minted on a pysam
    # stand-in by tests/golden/tooling/make_golden_bam_model.py; read_length 20, a 2 000-point insert-size CDF, no indel detected)
    ("ecoli-bam", None, lambda: random_genome(36, 3000), 4000, 29, 0, "metagenomics", False),
    ("ecoli-bam", None, lambda: mixed_genome(37, 2500), 3000, 30, 5, "metagenomics", True),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_hip_matches_oracle(engine, case):
    model, indel, mk_genome, n_pairs, seed, first, seq_type, gc_bias = CASES[case]
    _, stats = _compare(engine, dense_model(model, indel), mk_genome(), n_pairs, seed, first, seq_type, gc_bias)
    if indel is not None and indel[0] + indel[1] >= 0.01:
        assert stats["fixup_reads"] > 0  # the indel fix-up path really ran


@pytest.mark.parametrize("mode", ["0", "2"])
@pytest.mark.parametrize("case", [0, 5, 7, 9, 10, 11, 14, 15, 16])
def test_both_indel_paths_match_oracle(engine, case, mode, monkeypatch):
    """A model picks its indel path by how often a read has an event (DevModel::p_read_event against ISS_LIGHT_INDELS, 2e-3 by
    default): rare -- every such read to the one-wavefront-per-read kernel, k_main as it is; else k_indel_scan lists the
    events, k_indel_script turns them into edit scripts and k_main builds the reads from shifted windows / explicit letters.
    Both paths for shipped and indel-heavy models alike ("0": never the light path, "2": always)."""
    monkeypatch.setenv("ISS_LIGHT_INDELS", mode)
    model, indel, mk_genome, n_pairs, seed, first, seq_type, gc_bias = CASES[case]
    _compare(engine, dense_model(model, indel), mk_genome(), n_pairs, seed, first, seq_type, gc_bias)


@pytest.mark.parametrize("model,indel,min_share", [("novaseq", (0.001, 0.003), 0.95), ("miseq-legacy", None, 0.9), ("novaseq", (0.01, 0.03), 0.0)])
def test_scripted_reads_are_built_by_k_main(engine, model, indel, min_share, monkeypatch):
    """Round 4: the reads with (few) indel events of a heavy model are built by k_main itself from their edit scripts -- no
    second pass over their letters.  BASELINE configs[4]'s rates (0.001 / 0.003) and MiSeq-legacy (five position tiles):
    nearly every read with an event is scripted, the one-wavefront-per-read kernel takes the rest (more than EV_K events,
    windows leaving the record).  Ten times configs[4]'s rates: most reads have more events than a script holds, more explicit
    letters or more explicit pieces than a row does -- the overflow exits, exact as well."""
    monkeypatch.setenv("ISS_LIGHT_INDELS", "0")
    engine.stats_read()
    n_pairs = 4000
    _, stats = _compare(engine, dense_model(model, indel), random_genome(61, 120000), n_pairs, 77)
    assert stats["scripted_reads"] > 0
    assert stats["scripted_reads"] >= min_share * (stats["scripted_reads"] + stats["fixup_reads"])
    if min_share == 0.0:
        assert stats["fixup_reads"] > n_pairs // 4


@pytest.mark.parametrize("entries", ["nan", "inf", "mixed"])
@pytest.mark.parametrize("mode", ["0", "2"])
def test_bam_built_indel_tables_nan_inf_zero_one(engine, entries, mode, monkeypatch):
    """A model built by `iss model` from a BAM file divides indel counts by match counts (iss/modeller.py:338-349): positions
    nobody covered give 0/0 = NaN and n/0 = inf, others 0 or values >= 1.  `random() < NaN` never fires, `random() < inf`
    and >= 1 always do (__init__.py:194, :209); both indel paths (scripts / one wavefront per read) against the oracle, which
    compares IEEE doubles on the raw tables."""
    monkeypatch.setenv("ISS_LIGHT_INDELS", mode)
    dense = dense_model("novaseq", (0.0005, 0.002))
    r = np.random.RandomState(5)
    for tab in (dense.ins, dense.dele):
        for o in range(2):
            pos = r.choice(dense.read_length - 1, size=6, replace=False)
            vals = {"nan": [np.nan] * 6, "inf": [np.inf, np.nan, np.inf, 0.0, np.nan, 0.0],
                    "mixed": [np.nan, np.inf, 0.0, 1.0, 1.5, 0.9999999]}[entries]
            for p_, v in zip(pos, vals):
                tab[o, p_, r.randint(0, 4)] = v
    _compare(engine, dense, random_genome(91, 80000), 4000, 31)


@pytest.mark.parametrize("ahead", ["1", "0"])
@pytest.mark.parametrize("model,indel", [("novaseq", None), ("novaseq", (0.001, 0.003)), ("hiseq", None)])
def test_back_to_back_calls_pipeline(model, indel, ahead, monkeypatch):
    """k_setup of a call runs on its own stream beside the kernels of the call before (double-buffered descriptors, flags and
    fix-up lists; ISS_SETUP_AHEAD=0: one stream).  Forty calls into the SAME rows without a host synchronisation in
    between, alternating between two work lists and two genomes: the rows, coordinates (the copy k_main leaves for the
    host) of the last call and of a call in the middle (synchronised there) equal the oracle's."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    monkeypatch.setenv("ISS_SETUP_AHEAD", ahead)
    dense = dense_model(model, indel)
    genomes = [random_genome(41, 60000), random_genome(42, 45000)]
    orc = O.Oracle(dense)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gids = [eng.add_genome(g) for g in genomes]
        lists = [([0, 1], [9000, 7000]), ([1, 0, 1], [4000, 8000, 3000])]
        eng.reserve(16000)

        def check(call):
            ids, counts = lists[call % 2]
            eng.synchronize()
            row, ordinal = 0, 1000 * call
            for g, n in zip(ids, counts):
                exp = orc.simulate(O.Rng().seed_philox(5 + call), genomes[g], n, first_ordinal=ordinal, want_coords=True)
                got, cg = eng.download(row, n), eng.coords(row, n)
                for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                    assert np.array_equal(got[key], exp[key]), (call, g, key)
                assert np.array_equal(np.asarray(cg), exp["coords"]), (call, g)
                row += n
                ordinal += n

        for call in range(40):
            ids, counts = lists[call % 2]
            eng.generate_batch([gids[g] for g in ids], counts, first_ordinal=1000 * call, seed=5 + call, out_first_pair=0)
            if call in (17, 39):
                check(call)


@pytest.mark.parametrize("switches", [{"ISS_TILES": "3"}, {"ISS_TILES": "2", "ISS_GUIDE_BITS": "8"}, {"ISS_GUIDE_BITS": "6"},
                                      {"ISS_SETUP_AHEAD": "0"}])
def test_tuning_switches_do_not_change_results(switches, monkeypatch):
    """The switches the library still reads (k_main's position tiles and guide bits -- an indel-heavy model
    cut into several tiles also exercises the edit scripts' rows per tile --, everything on one stream) leave the reads alone,
    also across a re-allocation of the rows."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    dense = dense_model("novaseq", (0.001, 0.003))
    genome = random_genome(77, 50000)
    orc = O.Oracle(dense)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        for n in (70000, 150000):  # (the second reserve frees the first rows)
            eng.reserve(n)
            eng.generate(gid, n, first_ordinal=n, seed=9)
            got = eng.download(n - 3000, 3000)
            exp = orc.simulate(O.Rng().seed_philox(9), genome, 3000, first_ordinal=2 * n - 3000)
            for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                assert np.array_equal(got[key], exp[key]), (n, key)


def test_calls_with_and_without_setup_stream_share_the_counter_rings(monkeypatch):
    """Regression (round 3): calls whose k_setup runs on the setup stream and calls that keep everything on one stream
    (custom fragment lengths: the host reads k_setup's results back) take their fix-up / read-list / substitution-list
    counters from the same rings -- every chunk clears its own.  Twenty calls of the first kind, then one of the second
    on a slot in the middle of the ring, compared with the oracle incl. its --store_mutations rows."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    dense = dense_model("novaseq", (0.001, 0.003))
    genome = random_genome(55, 2000)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        for call in range(21):
            eng.generate(gid, 3000, first_ordinal=call, seed=7)
        eng.mutations_reserve(1_000_000)
        eng.set_fragment(200, 60)
        eng.generate(gid, 3000, first_ordinal=11, seed=7)
        eng.synchronize()
        rows, got = eng.mutations(), eng.download(0, 3000)
    exp = O.Oracle(dense).simulate(O.Rng().seed_philox(7), genome, 3000, first_ordinal=11, store_mutations=True,
                                   fragment_length=200, fragment_sd=60)
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        assert np.array_equal(got[k], exp[k]), k
    assert len(rows) == len(exp["mutations"]) and all(np.array_equal(rows[f], exp["mutations"][f]) for f in ("pair", "mate", "type", "position", "ref", "alt"))


def test_rows_and_ordinals_compose(engine):
    """Two calls writing adjacent rows with consecutive ordinals == one call (work items of a worker)."""
    dense = dense_model("hiseq")
    genome = random_genome(31, 80000)
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.generate(gid, 3000, first_ordinal=50, seed=99)
    engine.synchronize()
    whole = engine.download(0, 3000)
    whole = {k: v.copy() for k, v in whole.items() if not k.startswith("_")}
    engine.generate(gid, 1234, first_ordinal=50, seed=99, out_first_pair=0)
    engine.generate(gid, 3000 - 1234, first_ordinal=50 + 1234, seed=99, out_first_pair=1234)
    engine.synchronize()
    parts = engine.download(0, 3000)
    for k in whole:
        assert (whole[k] == parts[k]).all()


def test_short_record_is_reported(engine):
    from insilicoseq_amd import _native

    engine.load_model(dense_model("novaseq"))
    engine.clear_genomes()
    gid = engine.add_genome("ACGT" * 30)  # 120 <= read_length 151
    with pytest.raises(_native.EngineError) as e:
        engine.generate(gid, 10)
    assert e.value.code == _native.E_SHORT_RECORD


def test_invalid_letter_is_rejected(engine):
    from insilicoseq_amd import _native

    engine.load_model(dense_model("ecoli"))
    with pytest.raises(_native.EngineError) as e:
        engine.add_genome("ACGTXACGT" * 10)
    assert e.value.code == _native.E_INVALID


def test_worker_iterator_fastq_matches_oracle(tmp_path):
    """The drop-in boundary end to end on the GPU: worker_iterator -> FASTQ files == oracle (Philox,
    worker seed = seed + cpu_number, ordinals running across work items) + the same formatter."""
    from insilicoseq_amd.engine import fastq_write
    from insilicoseq_amd.generator import Record, worker_iterator
    from oracle import oracle as O

    dense = dense_model("hiseq")
    recs = [Record(random_genome(40 + i, 3000 + 700 * i), id="rec%d" % i) for i in range(3)]
    recs.insert(1, Record("ACGT" * 20, id="tiny"))  # shorter than read_length: skipped with a warning
    counts = [700, 50, 1, 1333]
    work = [(r, n, "default") for r, n in zip(recs, counts)]
    seed, cpu = 1234, 0
    prefix = str(tmp_path / "w")
    worker_iterator(work, dense, cpu, prefix, seed, "metagenomics", True, device=0)
    rng = O.Rng().seed_philox(seed + cpu)
    orc = O.Oracle(dense)
    p1, p2 = tmp_path / "e1", tmp_path / "e2"
    ordinal = 0
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for r, n in zip(recs, counts):
            if len(r.seq) <= dense.read_length:
                continue
            res = orc.simulate(rng, r.seq, n, first_ordinal=ordinal, gc_bias=True)
            assert res["status"] == 0
            fastq_write(f1.fileno(), f2.fileno(), r.id, 0, cpu, n, dense.read_length, dense.read_length,
                        res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], 1)
            ordinal += n
    assert open(prefix + "_R1.fastq", "rb").read() == p1.read_bytes()
    assert open(prefix + "_R2.fastq", "rb").read() == p2.read_bytes()
    assert os.path.exists(prefix + ".vcf")


def test_large_batch_properties(engine):
    """Size-independent properties at a scale the oracle cannot check (2 M pairs): every phred is a
    possible one for its (position), bases are in the alphabet, and un-mutated positions equal the
    template (spot-checked through the pair coordinates); a second run is bit-identical."""
    dense = dense_model("novaseq")
    genome = random_genome(77, 1000000)
    n = 2_000_000
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.generate(gid, n, first_ordinal=0, seed=5)
    engine.synchronize()
    a = engine.download(0, n)
    coords = engine.coords(0, n)
    assert int(a["r1_qual"].max()) <= 40 and int(a["r2_qual"].max()) <= 40
    for k in ("r1_base", "r2_base"):
        assert np.isin(a[k], np.frombuffer(b"ACGT", dtype=np.uint8)).all()
    g = np.frombuffer(genome.encode(), dtype=np.uint8)
    RL = dense.read_length
    idx = np.random.RandomState(0).randint(0, n, size=20000)
    fs = coords[idx, 0]
    tmpl = g[fs[:, None] + np.arange(RL)[None, :]]
    same = (a["r1_base"][idx] == tmpl).mean()
    assert same > 0.99  # substitutions are rare; indels rarer
    assert (coords[:, 0] >= 0).all() and (coords[:, 2] <= len(genome)).all()
    assert (coords[:, 1] == coords[:, 0] + RL + coords[:, 3]).mean() > 0.999  # reverse_start = fwd_end + insert
    checksum = [int(a[k].astype(np.uint64).sum()) for k in ("r1_base", "r1_qual", "r2_base", "r2_qual")]
    engine.generate(gid, n, first_ordinal=0, seed=5)
    engine.synchronize()
    b = engine.download(0, n)
    assert checksum == [int(b[k].astype(np.uint64).sum()) for k in ("r1_base", "r1_qual", "r2_base", "r2_qual")]


@pytest.mark.parametrize("workers", [1, 2])
def test_generate_cli_end_to_end(tmp_path, workers):
    """`python -m insilicoseq_amd generate` (the reference's `iss generate` flow): abundance file equals the
    reference's for the same seed; FASTQ equals the oracle (Philox, seed + worker) over the same chunks."""
    import subprocess
    import sys

    from helpers import GOLDEN
    from insilicoseq_amd.distributed import rank_work
    from insilicoseq_amd.engine import fastq_write
    from insilicoseq_amd.generator import parse_fasta
    from oracle import oracle as O

    root = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
    out = str(tmp_path / "run")
    fasta = os.path.join(GOLDEN, "genomes.fasta")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes", fasta, "--model", "hiseq",
                           "-n", "600", "--seed", "42", "--gpus", str(workers), "--devices", "1", "-o", out,
                           "--quiet"], cwd=root)
    z = np.load(os.path.join(GOLDEN, "generate", "genomes_hiseq_n600_seed42_cpus%d.npz" % workers))
    assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()  # same numpy stream as the reference
    dense = dense_model("hiseq")
    records = list(parse_fasta(fasta))
    abundance = {}
    for line in z["abundance"].tobytes().decode().splitlines():
        k, v = line.split("\t")
        abundance[k] = float(v)
    p1, p2 = tmp_path / "e1", tmp_path / "e2"
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for rank in range(workers):
            work, _, _ = rank_work(records, None, abundance, 600, None, None, dense, out, workers, rank)
            rng = O.Rng().seed_philox(42 + rank)
            ordinal = 0
            for rec, n, _ in work:
                if len(rec.seq) <= dense.read_length:
                    continue
                res = O.Oracle(dense).simulate(rng, rec.seq, n, first_ordinal=ordinal)
                assert res["status"] == 0
                fastq_write(f1.fileno(), f2.fileno(), rec.id, 0, rank, n, dense.read_length, dense.read_length,
                            res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], 1)
                ordinal += n
    assert open(out + "_R1.fastq", "rb").read() == p1.read_bytes()
    assert open(out + "_R2.fastq", "rb").read() == p2.read_bytes()
    assert not os.path.exists(out + ".iss.tmp.genomes.fasta") and not os.path.exists(out + ".iss.tmp.0_R1.fastq")


def test_device_genome_packing_roundtrip(engine):
    """k_pack_genome: every letter class survives (plain, lower case, IUPAC), at word boundaries too."""
    from helpers import mixed_genome as mg

    dense = dense_model("ecoli")  # read_length 20
    engine.load_model(dense)
    for L in (21, 31, 32, 33, 63, 64, 65, 1000, 4099):
        genome = mg(100 + L, L)
        engine.clear_genomes()
        gid = engine.add_genome(genome)
        n = 64
        engine.generate(gid, n, seed=3, sequence_type="amplicon")
        engine.synchronize()
        got = engine.download(0, n)
        from oracle import oracle as O

        exp = O.Oracle(dense).simulate(O.Rng().seed_philox(3), genome, n, sequence_type="amplicon")
        assert exp["status"] == 0
        for k in ("r1_base", "r2_base", "r1_qual", "r2_qual"):
            assert np.array_equal(got[k], exp[k]), (L, k)


@pytest.mark.parametrize("indel", [None, (0.001, 0.003)])
def test_chunked_launches_compose(engine, indel):
    """One call larger than the library's per-launch chunk (row byte offsets are 32 bits wide) == two smaller calls with
    consecutive ordinals; spot-checked around the chunk boundary and at both ends.  (The indel-heavy variant: event
    lists, read lists and their counters are per chunk.)"""
    dense = dense_model("hiseq", indel)  # rows of 512 bytes -> chunk = (2^32 - 1) / 512 = 8,388,607 pairs
    chunk = ((1 << 32) - 1) // 512
    n = chunk + 300_000
    genome = random_genome(123, 2_000_000)
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.generate(gid, n, first_ordinal=7, seed=31)
    engine.synchronize()
    windows = [(0, 2000), (chunk - 1500, 3000), (n - 2000, 2000)]
    whole = [{k: v.copy() for k, v in engine.download(a, m).items() if not k.startswith("_")} for a, m in windows]
    coords_whole = [engine.coords(a, m).copy() for a, m in windows]
    half = n // 2 + 17
    engine.generate(gid, half, first_ordinal=7, seed=31, out_first_pair=0)
    engine.generate(gid, n - half, first_ordinal=7 + half, seed=31, out_first_pair=half)
    engine.synchronize()
    for (a, m), w, cw in zip(windows, whole, coords_whole):
        got = engine.download(a, m)
        for k in w:
            assert np.array_equal(got[k], w[k]), (a, k)
        assert np.array_equal(engine.coords(a, m), cw)
    # and the boundary rows agree with the oracle
    from oracle import oracle as O

    a, m = windows[1]
    exp = O.Oracle(dense).simulate(O.Rng().seed_philox(31), genome, m, first_ordinal=7 + a)
    for k in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
        assert np.array_equal(whole[1][k], exp[k]), k


def test_api_misuse_is_reported_not_crashed():
    """Error conventions of the C ABI: negative code + message, never exit()/crash."""
    import ctypes as C

    from insilicoseq_amd import _native
    from insilicoseq_amd.engine import ReadEngine
    from insilicoseq_amd.model import DenseModel

    with ReadEngine(0) as eng:
        with pytest.raises(_native.EngineError) as e:  # no model yet
            eng._check(eng._lib.iss_output_reserve(eng._ctx, 10))
        assert e.value.code == _native.E_INVALID
        with pytest.raises(_native.EngineError):
            eng._check(eng._lib.iss_generate(eng._ctx, 0, 10, 0, 0, 0, 0, 0))
        d = dense_model("ecoli")
        eng.load_model(d)
        with pytest.raises(_native.EngineError):  # unknown genome id
            eng.generate(3, 10)
        gid = eng.add_genome("ACGT" * 100)
        eng.reserve(100)
        with pytest.raises(_native.EngineError):  # rows beyond the reservation
            eng._check(eng._lib.iss_generate(eng._ctx, gid, 200, 0, 0, 0, 0, 0))
        with pytest.raises(_native.EngineError):  # MT mode needs a seed first
            eng.generate_mt(gid, 10)
        with pytest.raises(ValueError):
            eng.generate(gid, 10, sequence_type="shotgun")
        with pytest.raises(_native.EngineError):  # numpy's legacy seeding range
            eng.seed_mt(2**32)
        # a model the engine cannot hold: more than 60 entries per quality CDF
        big = DenseModel(d.read_length, d.isize_cdf, d.bin_cdf, d.bin_nonempty,
                         np.concatenate([d.qcdf, np.ones(d.qcdf.shape[:3] + (30,))], axis=3), d.subst_cdf, d.subst_alt,
                         d.ins, d.ins_letter, d.dele, np.concatenate([d.phred_thr, np.ones(30)]))
        with pytest.raises(_native.EngineError):
            eng.load_model(big)
        # the context is still usable afterwards
        eng.load_model(d)
        gid = eng.add_genome("ACGT" * 100)
        eng.generate(gid, 50, seed=1)
        eng.synchronize()
        assert eng.download(0, 50)["r1_base"].shape == (50, 20)


def test_philox_mode_statistics_match_the_model(engine):
    """Size-independent property at scale (2 M pairs): the empirical distributions produced by the Philox
    address map match the model -- phred PMF per position (mixture over the mean-quality bins), insert sizes,
    uniform forward starts, substitution rate = E[10^(-Q/10)] -- within 6 sigma."""
    dense = dense_model("hiseq")
    L = 400_000
    genome = random_genome(99, L)
    n = 2_000_000
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.generate(gid, n, first_ordinal=0, seed=2024)
    engine.synchronize()
    got = engine.download(0, n)
    coords = engine.coords(0, n)
    RL, nq = dense.read_length, dense.n_q
    # phred PMF at a few positions, forward and reverse
    for o, key in ((0, "r1_qual"), (1, "r2_qual")):
        w = np.diff(np.concatenate(([0.0], dense.bin_cdf[o])))
        for p in (0, 37, RL - 1):
            pmf = np.zeros(nq + 1)
            for b in range(4):
                if dense.bin_nonempty[o, b]:
                    c = dense.qcdf[o, b, p]
                    pmf[:nq] += w[b] * np.diff(np.concatenate(([0.0], c)))
            emp = np.bincount(got[key][:, p], minlength=nq + 1)[: nq + 1] / n
            sigma = np.sqrt(np.maximum(pmf * (1 - pmf), 1e-12) / n)
            assert (np.abs(emp - pmf) < 6 * sigma + 1e-9).all(), (key, p)
    # insert sizes
    pmf = np.diff(np.concatenate(([0.0], dense.isize_cdf)))
    emp = np.bincount(coords[:, 3], minlength=len(pmf))[: len(pmf)] / n
    assert (np.abs(emp - pmf) < 6 * np.sqrt(np.maximum(pmf * (1 - pmf), 1e-12) / n) + 1e-9).all()
    # forward start uniform on [0, L - fragment): compare the mean of fs / width with 1/2
    width = L - (coords[:, 3] + 2 * RL)
    u = coords[:, 0] / width
    assert abs(u.mean() - 0.5) < 6 * np.sqrt(1 / 12 / n) and (coords[:, 0] < width).all()
    # substitution rate of the forward mate at one position: P(base != template) vs E[P(err)] * P(alt != base) = E[P(err)]
    g = np.frombuffer(genome.encode(), dtype=np.uint8)
    for p in (5, 100):
        exp_err = (10.0 ** (-got["r1_qual"][:, p].astype(np.float64) / 10)).mean()
        emp_err = (got["r1_base"][:, p] != g[coords[:, 0] + p]).mean()
        assert abs(emp_err - exp_err) < 6 * np.sqrt(exp_err / n) + 2e-5, (p, emp_err, exp_err)  # indels: ~1e-5


FRAG_CASES = [
    # (model, indel, genome, n_pairs, seed, seq_type, gc_bias, fragment_length, fragment_sd)
    ("novaseq", None, lambda: random_genome(41, 3000), 3000, 5, "metagenomics", False, 160, 40),     # negative inserts,
    ("novaseq", None, lambda: random_genome(42, 400), 1500, 6, "metagenomics", False, 160, 40),      # cut templates
    ("novaseq", None, lambda: random_genome(43, 50000), 3000, 7, "metagenomics", False, 450, 30),
    ("ecoli", None, lambda: mixed_genome(44, 3000), 3000, 8, "metagenomics", True, 300, 25),
    ("miseq-legacy", None, lambda: random_genome(45, 700), 600, 9, "amplicon", False, 1000, 10),
    ("novaseq", (0.01, 0.03), lambda: mixed_genome(46, 2000), 1500, 10, "metagenomics", False, 200, 60),
    ("hiseq", None, lambda: random_genome(47, 100000), 2000, 11, "metagenomics", False, 0, 0),        # degenerate sd
]


@pytest.mark.parametrize("case", range(len(FRAG_CASES)))
@pytest.mark.parametrize("guard", ["1e-6", "0.6"])
def test_custom_fragment_length_matches_oracle(case, guard, monkeypatch):
    """--fragment-length on the Philox path vs the oracle (Philox): per-pair polar Box-Muller, Python slice
    semantics for the odd geometries; guard 0.6 forces every pair through the host-evaluated override pass."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    model, indel, mk, n, seed, seq_type, gc_bias, mu, sd = FRAG_CASES[case]
    monkeypatch.setenv("ISS_MT_GUARD", guard)
    dense = dense_model(model, indel)
    genome = mk()
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        eng.set_fragment(mu, sd)
        eng.generate(gid, n, first_ordinal=3, seed=seed, sequence_type=seq_type, gc_bias=gc_bias)
        eng.synchronize()
        got = eng.download(0, n)
        coords = eng.coords(0, n)
    exp = O.Oracle(dense).simulate(O.Rng().seed_philox(seed), genome, n, first_ordinal=3, sequence_type=seq_type,
                                   gc_bias=gc_bias, fragment_length=mu, fragment_sd=sd, want_coords=True)
    assert exp["status"] == 0 and exp["n_done"] == n
    assert np.array_equal(coords[:, [0, 2, 3]], exp["coords"][:, [0, 2, 3]])  # forward start, reverse end, insert size
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        bad = np.argwhere(got[k] != exp[k])
        assert bad.size == 0, "%s differs at %s (%d cells)" % (k, bad[:5].tolist(), len(bad))


MUT_CASES = [
    # (model, indel, genome, n_pairs, seed, seq_type, gc_bias, fragment)
    ("novaseq", None, lambda: random_genome(51, 100000), 20000, 3, "metagenomics", False, None),
    ("hiseq", None, lambda: mixed_genome(52, 30000), 8000, 4, "metagenomics", True, None),
    ("novaseq", (0.01, 0.03), lambda: mixed_genome(53, 20000), 3000, 5, "metagenomics", False, None),
    ("novaseq", (0.2, 0.35), lambda: random_genome(54, 5000), 500, 6, "metagenomics", False, None),
    ("novaseq", (0.001, 0.003), lambda: random_genome(55, 2000), 3000, 7, "metagenomics", False, (200, 60)),
    ("ecoli", None, lambda: random_genome(56, 3000), 5000, 8, "amplicon", False, None),
]


@pytest.mark.parametrize("case", range(len(MUT_CASES)))
def test_store_mutations_rows_match_oracle(engine, case):
    """--store_mutations on the Philox path: rows (after the host's stale-row filter and sort) equal the oracle's,
    and the reads themselves are unchanged by the bookkeeping kernel variant."""
    from oracle import oracle as O

    model, indel, mk, n, seed, seq_type, gc_bias, frag = MUT_CASES[case]
    dense = dense_model(model, indel)
    genome = mk()
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.mutations_reserve(4_000_000)
    engine.set_fragment(*(frag or (None, None)))
    try:
        engine.generate(gid, n, first_ordinal=11, seed=seed, sequence_type=seq_type, gc_bias=gc_bias)
        engine.synchronize()
        got_rows = engine.mutations()
        got = engine.download(0, n)
    finally:
        engine.mutations_reserve(0)
        engine.set_fragment(None, None)
    exp = O.Oracle(dense).simulate(O.Rng().seed_philox(seed), genome, n, first_ordinal=11, sequence_type=seq_type,
                                   gc_bias=gc_bias, store_mutations=True,
                                   fragment_length=frag[0] if frag else None, fragment_sd=frag[1] if frag else None)
    assert exp["status"] == 0
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        assert np.array_equal(got[k], exp[k]), k
    rows = exp["mutations"]
    assert len(got_rows) == len(rows) and len(rows) > 0
    for f in ("pair", "mate", "type", "position", "ref", "alt", "quality"):
        bad = np.flatnonzero(got_rows[f] != rows[f])
        assert bad.size == 0, (f, bad[:5], got_rows[bad[:3]], rows[bad[:3]])


@pytest.mark.parametrize("kind", ["del", "ins"])
def test_store_mutations_script_row_overflow(engine, kind, monkeypatch):
    """A read whose edit script needs more than four explicit pieces in ONE 16-byte row (pieces 0, 4, 8, 12, 16 of a
    one-tile model: an event every 32 positions) leaves k_indel_script for k_indel_fixup.  With --store_mutations the
    script kernel must notice that BEFORE it emits the read's indel rows (round-4 advice: it noticed afterwards, and the
    fix-up kernel wrote the rows a second time).  Every read of this model takes that exit."""
    from oracle import oracle as O

    monkeypatch.setenv("ISS_LIGHT_INDELS", "0")
    dense = dense_model("novaseq", (0.0, 0.0))
    for pos in (4, 36, 68, 100, 132):  # five steps with an event, 32 positions apart: one residue class of pieces
        if kind == "del":
            dense.dele[:, pos, :] = 1.0
        else:
            dense.ins[:, pos, 2] = 1.0
    genome = random_genome(57, 60000)
    n, seed = 1500, 9
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.mutations_reserve(1_000_000)
    engine.stats_read()
    try:
        engine.generate(gid, n, first_ordinal=5, seed=seed)
        engine.synchronize()
        got_rows = engine.mutations()
        got = engine.download(0, n)
        stats = engine.stats_read()
    finally:
        engine.mutations_reserve(0)
    exp = O.Oracle(dense).simulate(O.Rng().seed_philox(seed), genome, n, first_ordinal=5, store_mutations=True)
    assert exp["status"] == 0
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        assert np.array_equal(got[k], exp[k]), k
    rows = exp["mutations"]
    assert len(rows) >= 10 * n  # five indel rows per read at least
    assert len(got_rows) == len(rows), "duplicate or missing --store_mutations rows: %d against %d" % (len(got_rows), len(rows))
    for f in ("pair", "mate", "type", "position", "ref", "alt", "quality"):
        assert np.array_equal(got_rows[f], rows[f]), f
    assert stats["fixup_reads"] >= 2 * n and stats["scripted_reads"] == 0  # the overflow exit, for every read


def test_worker_vcf_philox(tmp_path):
    """worker_iterator with store_mutations on the Philox path writes the rows in VCF form."""
    from insilicoseq_amd.generator import Record, worker_iterator
    from insilicoseq_amd.model import KDErrorModel
    from oracle import oracle as O

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    em = KDErrorModel(os.path.join(root, "insilicoseq_amd", "profiles", "hiseq.dense.npz"), None, None, True)
    recs = [Record(random_genome(60 + i, 5000), id="g%d" % i) for i in range(2)]
    prefix = str(tmp_path / "w")
    worker_iterator([(recs[0], 900, "default"), (recs[1], 400, "default")], em, 2, prefix, 5, "metagenomics", False, device=0)
    dense = dense_model("hiseq")
    rng = O.Rng().seed_philox(5 + 2)
    lines, ordinal = [], 0
    for r, n in zip(recs, (900, 400)):
        res = O.Oracle(dense).simulate(rng, r.seq, n, first_ordinal=ordinal, store_mutations=True)
        ordinal += n
        for m in res["mutations"]:
            ref, alt = chr(m["ref"]), chr(m["alt"])
            alt = ref + alt if m["type"] == 1 else alt
            qual = str(int(m["quality"])) if m["type"] == 0 else "."
            lines.append("\t".join(["%s_%d_2/%d" % (r.id, m["pair"], 1 + int(m["mate"])), str(int(m["position"]) + 1), ".",
                                    ref, alt, qual, "", ""]) + "\n")
    assert open(prefix + ".vcf").read() == "".join(lines) and len(lines) > 50


@pytest.mark.parametrize("model,rid,cpu,first_i,counts", [
    ("novaseq", "genome_A", 0, 0, [1205, 1, 7000]),
    ("ecoli", "x", 12, 95, [10, 900, 120000]),           # crosses 99|100, 999|1000, 9999|10000, 99999|100000
    ("miseq", "NZ_" + "k" * 300 + ".1", 3, 999990, [25, 3000]),
])
def test_device_fastq_equals_host_formatter(model, rid, cpu, first_i, counts, tmp_path):
    """iss_fastq_emit (text built on the device, closed-form record offsets, asynchronous copy + pwrite) against
    iss_fastq_write (host formatter) on the same rows: several appends to the same files, ids that cross digit
    boundaries, short / long reads and ids."""
    from helpers import random_genome
    from insilicoseq_amd.engine import ReadEngine, fastq_write

    dense = dense_model(model)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(random_genome(3, 50000))
        eng.reserve(max(counts))
        paths = [str(tmp_path / n) for n in ("d1.fq", "d2.fq", "h1.fq", "h2.fq")]
        fh = [open(p, "wb") for p in paths]
        for f in fh[:2]:
            f.write(b"# existing content\n")  # appended after whatever is in the files
            f.flush()
        for f in fh[2:]:
            f.write(b"# existing content\n")
            f.flush()
        i0 = first_i
        for k, n in enumerate(counts):
            eng.generate(gid, n, first_ordinal=1000 * k, seed=5)
            eng.fastq_emit(fh[0].fileno(), fh[1].fileno(), rid, i0, cpu, 0, n, n_threads=3)
            eng.synchronize()
            rows = eng.download(0, n)["_pitched"]
            fastq_write(fh[2].fileno(), fh[3].fileno(), rid, i0, cpu, n, eng.read_length, eng.pitch, rows[0], rows[1],
                        rows[2], rows[3], n_threads=2)
            i0 += n
        eng.fastq_flush()
        assert os.lseek(fh[0].fileno(), 0, os.SEEK_CUR) == os.path.getsize(paths[0])
        for f in fh:
            f.close()
    assert open(paths[0], "rb").read() == open(paths[2], "rb").read()
    assert open(paths[1], "rb").read() == open(paths[3], "rb").read()
    assert os.path.getsize(paths[0]) > 100 * sum(counts) // 4


@pytest.mark.parametrize("compress", [False, True])
def test_fastq_pipeline_many_tiny_jobs(compress, tmp_path):
    """Regression (round 3): the file offsets of the FASTQ pipeline were advanced by the caller outside the pipe's
    mutex while the writer thread added its own (zero, in text mode) byte count under it -- a lost update let a job
    overwrite its predecessor's bytes, once in ~1e4 runs of the randomized worker tests.  Thousands of tiny jobs back to
    back keep the writer finishing a job while the caller queues the next; the files must be the concatenation of the
    jobs' text, and iss_fastq_flush checks offset == attach offset + queued bytes."""
    import gzip
    from helpers import random_genome
    from insilicoseq_amd.engine import ReadEngine, fastq_write

    n, jobs = 512, int(os.environ.get("ISS_TINY_JOBS", "6000"))
    dense = dense_model("ecoli")  # read length 20: a job is a few hundred bytes
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(random_genome(3, 50000))
        eng.reserve(n)
        eng.generate(gid, n, first_ordinal=0, seed=5)
        eng.synchronize()
        rows = eng.download(0, n)["_pitched"]
        paths = [str(tmp_path / p) for p in ("d1.fq", "d2.fq", "h1.fq", "h2.fq")]
        fh = [open(p, "wb") for p in paths]
        if compress:
            eng.fastq_compress(True)
        r = np.random.RandomState(7)
        first = r.randint(0, n - 3, size=jobs)
        count = r.randint(1, 4, size=jobs)
        for j in range(jobs):
            eng.fastq_emit(fh[0].fileno(), fh[1].fileno(), "g", j, 1, int(first[j]), int(count[j]), n_threads=1)
        eng.fastq_flush()
        assert os.lseek(fh[0].fileno(), 0, os.SEEK_CUR) == os.path.getsize(paths[0])
        for j in range(jobs):  # the same text from the host formatter, job by job
            a, c = int(first[j]), int(count[j])
            fastq_write(fh[2].fileno(), fh[3].fileno(), "g", j, 1, c, eng.read_length, eng.pitch, rows[0][a:a + c],
                        rows[1][a:a + c], rows[2][a:a + c], rows[3][a:a + c], n_threads=1)
        for f in fh:
            f.close()
    rd = (lambda p: gzip.open(p, "rb").read()) if compress else (lambda p: open(p, "rb").read())
    assert rd(paths[0]) == open(paths[2], "rb").read()
    assert rd(paths[1]) == open(paths[3], "rb").read()


@pytest.mark.parametrize("model,indel,n_genomes,pairs_total,batch", [("novaseq", None, 5, 5_000_000, True), ("hiseq", None, 50, 6_250_000, True),
                                                                     ("novaseq", None, 5, 5_000_000, False),
                                                                     ("novaseq", (0.001, 0.003), 1, 5_000_000, True),
                                                                     # 5 M MiSeq pairs: 6.4 GB of rows in ONE k_main launch (64-bit row offsets, round 5)
                                                                     ("miseq", None, 5, 5_000_000, True)])
def test_baseline_sizes_sampled_against_oracle(model, indel, n_genomes, pairs_total, batch):
    """BASELINE.json's full sizes (configs[2]: 10 M NovaSeq reads over 5 x 5 Mbp; one rank's 6.25 M-pair share of
    configs[3]: HiSeq, 50 x 5 Mbp genomes; configs[4]: 10 M reads of an indel-heavy NovaSeq-shaped model over ONE 5 Mbp
    genome -- two reads in three carry an indel and are built from edit scripts): the whole work list is generated on the GPU exactly as bench.py times it
    (ONE iss_generate_batch call; `batch` False: one iss_generate call per record), and -- every pair being a pure
    function of (seed, ordinal, genome) -- windows of consecutive ordinals are recomputed by the CPU oracle and compared
    byte for byte, coordinates included: the first and last pairs of every work item (the item boundaries of the batch),
    random windows inside, and the pairs either side of the engine's launch-chunk edges."""
    from insilicoseq_amd.engine import ReadEngine
    from insilicoseq_amd.generator import lognormal_abundance
    from oracle import oracle as O

    dense = dense_model(model, indel)
    rng = np.random.RandomState(123)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = [letters[rng.randint(0, 4, size=5_000_000)].tobytes().decode() for _ in range(n_genomes)]
    ab = lognormal_abundance(list(range(n_genomes)), np.random.RandomState(123))
    counts = [max(1, int(pairs_total * ab[k])) for k in range(n_genomes)]
    orc = O.Oracle(dense)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gids = [eng.add_genome(g) for g in genomes]
        eng.reserve(sum(counts))
        row, ordinal, items = 0, 0, []
        for gid, n in zip(gids, counts):
            if not batch:
                eng.generate(gid, n, first_ordinal=ordinal, seed=42, out_first_pair=row)
            items.append((gid, n, row, ordinal))
            row += n
            ordinal += n
        if batch:
            eng.generate_batch(gids, counts, first_ordinal=0, seed=42, out_first_pair=0)
        eng.synchronize()
        pick = np.random.RandomState(7)
        total = sum(counts)
        # pairs around multiples of 2^20 (any launch-chunk edge of the engine is one) fall into some item's window list
        edges = [e for e in range(1 << 20, total, 1 << 20)][:: max(1, (total >> 20) // 6)]
        for k, (gid, n, row0, ord0) in enumerate(items):
            inner = (24 if n_genomes == 1 else 6) if n_genomes <= 8 else 1
            starts = set([0, max(n - 64, 0)] + list(pick.randint(0, max(n - 64, 1), size=inner)))
            starts.update(min(max(e - row0 - 32, 0), max(n - 64, 0)) for e in edges if row0 <= e < row0 + n)
            for start in sorted(starts):
                w = min(64, n - start)
                got = eng.download(row0 + start, w)
                cg = eng.coords(row0 + start, w)
                exp = orc.simulate(O.Rng().seed_philox(42), genomes[k], w, first_ordinal=ord0 + start, want_coords=True)
                for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                    assert np.array_equal(got[key], exp[key]), (k, start, key)
                assert np.array_equal(np.asarray(cg), exp["coords"]), (k, start)


@pytest.mark.parametrize("RL,n_q,n_isize,n", [(1024, 60, 8000, 300), (997, 41, 1000, 300), (2, 1, 1, 500), (5, 3, 7, 500)])
def test_engine_limits_match_oracle(RL, n_q, n_isize, n):
    """The engine's size limits (read_length 1024, 60 phred entries, 8000 insert sizes) and its smallest shapes,
    random valid tables with indels: Philox path and MT walker against the oracle."""
    from helpers import synthetic_model
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    dense = synthetic_model(RL, n_q, n_isize, seed=RL + n_q)
    genome = mixed_genome(RL, max(6 * RL + n_isize, 64))
    orc = O.Oracle(dense)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        eng.generate(gid, n, first_ordinal=3, seed=9)
        eng.synchronize()
        got = eng.download(0, n)
        exp = orc.simulate(O.Rng().seed_philox(9), genome, n, first_ordinal=3)
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[key], exp[key]), ("philox", key)
        eng.seed_mt(1234)
        assert eng.generate_mt(gid, n) == n
        got = eng.download(0, n)
        exp = orc.simulate(O.Rng().seed_mt(1234), genome, n)
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[key], exp[key]), ("mt", key)


def test_genome_positions_beyond_2_30():
    """A 1.2 Gbp record (MT mode takes genomes below 2^31 bp): position arithmetic near that limit, on both RNG
    paths, against the oracle at the pairs with the largest / smallest coordinates and a random sample."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    L = 1_200_000_000
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.RandomState(1).randint(0, 4, size=L)]
    g[L - 5000:L - 4000] = ord("N")
    gs = g.tobytes()
    dense = dense_model("novaseq")
    orc = O.Oracle(dense)
    n = 200000
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(g)
        eng.generate(gid, n, first_ordinal=0, seed=3)
        eng.synchronize()
        coords = eng.coords(0, n)
        assert int(coords[:, 2].max()) > (1 << 30) + 100_000_000
        order = np.argsort(coords[:, 2])
        picks = list(order[-30:]) + list(order[:10]) + list(np.random.RandomState(2).randint(0, n, 30))
        for i in picks:
            got = eng.download(int(i), 1)
            exp = orc.simulate(O.Rng().seed_philox(3), gs, 1, first_ordinal=int(i), want_coords=True)
            for k in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                assert np.array_equal(got[k], exp[k]), (int(i), k)
            assert np.array_equal(coords[i], exp["coords"][0])
        eng.seed_mt(9)
        assert eng.generate_mt(gid, 1500) == 1500
        got = eng.download(0, 1500)
        exp = orc.simulate(O.Rng().seed_mt(9), gs, 1500)
        for k in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[k], exp[k]), ("mt", k)


@pytest.mark.parametrize("model,indel,mixed,L", [("novaseq", (0.001, 0.003), False, 2**31 + 3_000_001), ("hiseq", None, True, 2**32 + 70_000_003),
                                                 ("novaseq", None, False, 2**32 + 90_000_001)])  # (plain: MT mode's resolver + emitter up there)
def test_records_of_2_31_bases_and_more(model, indel, mixed, L):
    """Round 4: records of 2^31 - 1 bases and more (the reference spills them to a memmap and carries on:
    iss/generator.py:313-331, util.py:271-304).  Coordinates are 36-bit in the pair descriptors, `random.randrange` draws a
    second word once the record passes 2^32 (CPython's getrandbits), k_main addresses the packed genome by 32-bit WORD numbers.
    A 2.15 Gbp record on the indel-heavy path (scripts at coordinates beyond 2^31) and a 4.36 Gbp record with IUPAC / lower-case
    stretches (mask and ASCII indexing up there; two-word randrange): the pairs with the largest and smallest coordinates and a
    random sample against the oracle, coordinates included.  The content has a prime period, so a coordinate off by a power of
    two reads other letters.  Round 5: the same records in MT mode (the reference's streams) against the oracle's MT mode."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    period = 67_108_859  # prime
    if mixed:
        block = np.frombuffer(mixed_genome(5, period).encode(), dtype=np.uint8)
    else:
        block = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.RandomState(5).randint(0, 4, size=period)]
    g = np.ascontiguousarray(np.tile(block, L // period + 1)[:L])
    dense = dense_model(model, indel)
    orc = O.Oracle(dense)
    n = 150_000
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(g)
        eng.generate(gid, n, first_ordinal=7, seed=13)
        eng.synchronize()
        coords = eng.coords(0, n)
        assert int(coords[:, 0].max()) >= 2**31 and int(coords[:, 2].max()) <= L and int(coords[:, 0].min()) >= 0
        if L > 2**32:
            assert int((coords[:, 0] >= 2**32).sum()) > n // 100
        order = np.argsort(coords[:, 2])
        picks = list(order[-24:]) + list(order[:8]) + list(np.random.RandomState(2).randint(0, n, 32))
        picks += [int(i) for i in np.flatnonzero((coords[:, 0] < 2**31) & (coords[:, 2] > 2**31))[:8]]  # pairs across 2^31
        for i in picks:
            got = eng.download(int(i), 1)
            exp = orc.simulate(O.Rng().seed_philox(13), g, 1, first_ordinal=7 + int(i), want_coords=True)
            assert exp["status"] == 0
            assert np.array_equal(coords[i], exp["coords"][0]), int(i)
            for k in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                assert np.array_equal(got[k], exp[k]), (int(i), k)
        if indel is not None:
            assert eng.stats_read()["scripted_reads"] > n // 2
        # round 5: MT mode takes such records too (36-bit coordinates in its kernels, two stream words per `randrange` candidate
        # once the bound passes 2^32 -- CPython's getrandbits, pinned on CPython itself in tests/test_oracle_golden.py): the
        # resolver + emitter for the plain record, the walker for the one with IUPAC stretches and for the indel-heavy model
        n_mt = 1200
        eng.seed_mt(9)
        assert eng.generate_mt(gid, n_mt) == n_mt
        got = eng.download(0, n_mt)
        cg = eng.coords(0, n_mt)
        rng = O.Rng().seed_mt(9)
        exp = orc.simulate(rng, g, n_mt, want_coords=True)
        assert exp["status"] == 0 and exp["n_done"] == n_mt
        assert np.array_equal(np.asarray(cg), exp["coords"]) and int(cg[:, 2].max()) >= 2**31
        for k in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[k], exp[k]), ("mt", k)
        py, npw = eng.mt_peek(8)  # both streams stand where the oracle's stand
        assert [int(x) for x in py] == [rng.py_word() for _ in range(8)] and [int(x) for x in npw] == [rng.np_word() for _ in range(8)]
        if indel is None and not mixed:
            assert eng.mt_path_counts()[0] >= n_mt - 8  # the offset resolver + parallel emitter did it (two-word randrange candidates)


def test_batch_arena_beyond_2_31_bases():
    """Round 5: the records of ONE iss_generate_batch call stand side by side in an arena of up to 2^34 - 4096 bases (until then:
    below 2^31, the worker fell back to one call per record).  Three records of 0.9 / 0.9 / 0.6 Gbp -- the third one's arena
    coordinates lie beyond 2^31 -- of an indel-heavy model (edit scripts, fix-up, scan at those coordinates): the pairs of every
    item with the largest and smallest coordinates and a random sample against the oracle, record coordinates included.  The
    content has a prime period: an arena coordinate off by a power of two reads other letters."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    period = 33_554_393  # prime
    block = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.RandomState(8).randint(0, 4, size=period)]
    lengths = [900_000_011, 900_000_017, 600_000_007]
    genomes = [np.ascontiguousarray(np.tile(np.roll(block, 1000 * k), L // period + 1)[:L]) for k, L in enumerate(lengths)]
    dense = dense_model("novaseq", (0.001, 0.003))
    orc = O.Oracle(dense)
    counts = [60_000, 50_000, 70_000]
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gids = [eng.add_genome(g) for g in genomes]
        eng.generate_batch(gids, counts, first_ordinal=11, seed=17, out_first_pair=0)
        eng.synchronize()
        row0, ord0 = 0, 11
        for k, n in enumerate(counts):
            coords = eng.coords(row0, n)
            assert int(coords[:, 0].min()) >= 0 and int(coords[:, 2].max()) <= lengths[k]
            order = np.argsort(coords[:, 2])
            picks = list(order[-12:]) + list(order[:6]) + list(np.random.RandomState(k).randint(0, n, 24))
            for i in picks:
                got = eng.download(row0 + int(i), 1)
                exp = orc.simulate(O.Rng().seed_philox(17), genomes[k], 1, first_ordinal=ord0 + int(i), want_coords=True)
                assert exp["status"] == 0
                assert np.array_equal(coords[i], exp["coords"][0]), (k, int(i))
                for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                    assert np.array_equal(got[key], exp[key]), (k, int(i), key)
            row0 += n
            ord0 += n
        assert eng.stats_read()["scripted_reads"] > sum(counts) // 2


def _fuzz_config(k):
    """Random but reproducible (model shape, tables, genome, options) for the differential test below."""
    from helpers import synthetic_model

    k += int(os.environ.get("ISS_FUZZ_OFFSET", "0"))  # (a soak run: other configurations than the 200 of the suite)
    r = np.random.RandomState(1000 + k)
    RL = int(r.choice([2, 3, 7, 20, 36, 75, 100, 126, 151, 200, 251, 301]))
    n_q = int(r.choice([1, 2, 8, 41, 41, 41, 60]))
    n_isize = int(r.choice([1, 5, 300, 1000, 2000]))
    scale = float(r.choice([0.0, 0.0, 1e-5, 1e-3, 3e-2, 0.3]))
    bins = [tuple(int(x) for x in r.randint(0, 2, size=4)) for _ in range(2)]
    bins = [b if any(b) else (0, 0, 1, 0) for b in bins]
    dense = synthetic_model(RL, n_q, n_isize, seed=2000 + k, indel=(scale, 2 * scale), nonempty=tuple(bins))
    kind = r.randint(0, 4)
    L = int(r.choice([RL + 1, RL + 5, 3 * RL + 10, 10 * RL + n_isize, 20000]))
    L = max(L, RL + 1)
    genome = (random_genome if kind else mixed_genome)(3000 + k, L)
    frag = None
    if r.rand() < 0.35:
        frag = (float(r.choice([RL // 2 + 1, 2 * RL, 2 * RL + 150, 3 * RL])), float(r.choice([1, 10, 60])))
    return dict(dense=dense, genome=genome, n=int(r.choice([1, 17, 200, 700])), seed=int(r.randint(0, 2**31)),
                first=int(r.choice([0, 5, 2**33])), seq_type="amplicon" if r.rand() < 0.2 else "metagenomics",
                gc_bias=bool(r.rand() < 0.3), frag=frag, mut=bool(r.rand() < 0.5))


# (18045, 20075: found by the round-3 soak -- a reverse mate with negative bounds, a full-length wrapped template and a deletion:
#  the letter behind the template is 'A', not the genome's)
@pytest.mark.parametrize("k", list(range(200)) + [18045, 20075])
def test_randomized_differential_both_paths(k):
    """Random model shapes (read length 2..301, 1..60 phred entries, arbitrary non-empty bins, indel rates from 0 to
    30 %), genomes (plain, mixed case + IUPAC, barely longer than a read), options (amplicon, gc_bias, custom fragment
    lengths incl. shorter than a read, --store_mutations): the Philox path and the MT path against the oracle --
    bases, phreds, coordinates, mutation rows."""
    from insilicoseq_amd import _native
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    c = _fuzz_config(k)
    dense, genome, n = c["dense"], c["genome"], c["n"]
    orc = O.Oracle(dense)
    fl, fsd = c["frag"] if c["frag"] else (None, None)
    kw = dict(sequence_type=c["seq_type"], gc_bias=c["gc_bias"])
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        # ---- Philox
        exp = orc.simulate(O.Rng().seed_philox(c["seed"]), genome, n, first_ordinal=c["first"], fragment_length=fl,
                           fragment_sd=fsd, store_mutations=c["mut"], want_coords=True, **kw)
        eng.set_fragment(fl, fsd)
        eng.mutations_reserve(max(64 * n * dense.read_length, 1 << 21) if c["mut"] else 0)  # (rows are reserved in 256-slot chunks per wavefront)
        eng.generate(gid, n, first_ordinal=c["first"], seed=c["seed"], **kw)
        eng.synchronize()
        got = eng.download(0, n)
        assert exp["status"] == 0 and exp["n_done"] == n
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[key], exp[key]), ("philox", key)
        assert np.array_equal(np.asarray(eng.coords(0, n)), exp["coords"])
        if c["mut"]:
            rows = eng.mutations()
            assert len(rows) == len(exp["mutations"])
            for f in ("pair", "mate", "type", "position", "ref", "alt", "quality"):
                assert np.array_equal(rows[f], exp["mutations"][f]), ("philox rows", f)
        eng.mutations_reserve(0)
        eng.set_fragment(None, None)
        # ---- MT (resolver + walker, whatever the model allows)
        s32 = c["seed"] & 0x7FFFFFFF
        exp = orc.simulate(O.Rng().seed_mt(s32), genome, n, fragment_length=fl, fragment_sd=fsd, store_mutations=c["mut"], **kw)
        eng.seed_mt(s32)
        eng.mt_set_fragment(fl, fsd)
        eng.mt_mutations_reserve(64 * n * dense.read_length if c["mut"] else 0)
        try:
            assert eng.generate_mt(gid, n, **kw) == n
        except _native.EngineError:
            assert c["frag"] is not None and exp["status"] != 0  # (a record the reference skips mid-stream)
            return
        got = eng.download(0, n)
        assert exp["status"] == 0
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[key], exp[key]), ("mt", key)
        if c["mut"]:
            rows = eng.mt_mutations()
            assert len(rows) == len(exp["mutations"])
            for f in ("pair", "mate", "type", "position", "ref", "alt", "quality"):
                assert np.array_equal(rows[f], exp["mutations"][f]), ("mt rows", f)


@pytest.mark.parametrize("model", ["miseq-36", "novaseq", "hiseq"])
def test_batch_with_more_work_items_than_the_indel_kernels_cache(model):
    """iss_generate_batch over 700 small records with an indel-heavy model: k_indel_apply keeps the table of up to 512 work
    items in LDS and searches it in global memory beyond that; --store_mutations rows included.  (Read lengths 301, 151,
    126: 32 and 8 lanes per read.)"""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    dense = dense_model(model, (0.01, 0.02))
    r = np.random.RandomState(8)
    RL = dense.read_length
    genomes = [random_genome(5000 + i, RL + int(r.randint(1, 300))) for i in range(700)]
    pairs = [int(r.choice([0, 1, 3, 9])) for _ in genomes]
    orc = O.Oracle(dense)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        eng.mutations_reserve(1 << 22)
        gids = [eng.add_genome(g) for g in genomes]
        total = sum(pairs)
        eng.reserve(total)
        eng.generate_batch(gids, pairs, first_ordinal=77, seed=5)
        eng.synchronize()
        got = eng.download(0, total)
        rows = eng.mutations()
        assert eng.stats_read()["fixup_reads"] > 0
    at, exp_rows = 0, []
    for g, n in zip(genomes, pairs):
        if not n:
            continue
        exp = orc.simulate(O.Rng().seed_philox(5), g, n, first_ordinal=77 + at, store_mutations=True)
        assert exp["status"] == 0
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[key][at:at + n], exp[key]), (at, key)
        exp_rows += [(int(m["pair"]) + at,) + tuple(int(m[k]) for k in ("mate", "type", "position", "ref", "alt", "quality"))
                     for m in exp["mutations"]]
        at += n
    assert [tuple(int(m[k]) for k in ("pair", "mate", "type", "position", "ref", "alt", "quality")) for m in rows] == exp_rows


def test_mt_single_pairs_with_gc_bias_rejections():
    """MT mode, one pair per call, gc_bias: a candidate pair that is rejected (generator.py:82-92, one time in ten) costs a
    whole pair's draws, and two or three rejections in a row need more stream words than the two pairs' worth a turn of
    one pair starts with (found by a soak run of the randomized test: "MT stream buffers too small for one read pair")."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    dense = dense_model("miseq")  # (long reads: many words per pair)
    genome = random_genome(4242, 900)
    orc = O.Oracle(dense)
    most = 0
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        for seed in range(400):
            rng = O.Rng().seed_mt(seed)
            exp = orc.simulate(rng, genome, 1, gc_bias=True)
            most = max(most, rng.words_used()[1])
            eng.seed_mt(seed)
            assert eng.generate_mt(gid, 1, gc_bias=True) == 1
            got = eng.download(0, 1)
            for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                assert np.array_equal(got[key], exp[key]), (seed, key)
    assert most >= 3 * 1212  # (a MiSeq pair draws >= 1212 numpy words: some seed met at least three candidate pairs)


def _expected_worker_files(dense, recs, counts, seed, cpu, rng_mode, seq_type, gc_bias, frag, tmp_path):
    """worker_iterator's three files as the oracle + the host formatter + the reference's VCF line format give them."""
    from insilicoseq_amd.engine import fastq_write
    from oracle import oracle as O

    orc = O.Oracle(dense)
    rng = O.Rng().seed_philox(seed + cpu) if rng_mode == "philox" else O.Rng().seed_mt(seed + cpu)
    fl, fsd = frag if frag else (None, None)
    p1, p2 = tmp_path / "e1", tmp_path / "e2"
    lines, ordinal = [], 0
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for r, n in zip(recs, counts):
            kw = dict(first_ordinal=ordinal) if rng_mode == "philox" else {}
            res = orc.simulate(rng, r.seq, n, sequence_type=seq_type, gc_bias=gc_bias, fragment_length=fl, fragment_sd=fsd,
                               store_mutations=True, **kw)
            if res["status"] == O.SKIP_RECORD:
                continue
            assert res["status"] == 0
            k = res["n_done"]
            if k:
                fastq_write(f1.fileno(), f2.fileno(), r.id, 0, cpu, k, dense.read_length, dense.read_length, res["r1_base"],
                            res["r1_qual"], res["r2_base"], res["r2_qual"], 1)
            ordinal += n
            for m in res["mutations"]:
                ref, alt = chr(m["ref"]), chr(m["alt"])
                alt = ref + alt if m["type"] == 1 else alt
                qual = str(int(m["quality"])) if m["type"] == 0 else "."
                lines.append("\t".join(["%s_%d_%d/%d" % (r.id, m["pair"], cpu, 1 + int(m["mate"])), str(int(m["position"]) + 1),
                                        ".", ref, alt, qual, "", ""]) + "\n")
    return p1.read_bytes(), p2.read_bytes(), "".join(lines)


@pytest.mark.parametrize("k", range(24))
def test_randomized_worker_lists(k, tmp_path):
    """Random work lists through worker_iterator (records of assorted lengths incl. some the reference skips, zero /
    one / many pairs, stream and ordinal carry-over between items, device FASTQ text, VCF) on both RNG paths."""
    from insilicoseq_amd.generator import Record, worker_iterator
    from insilicoseq_amd.model import BasicErrorModel

    r = np.random.RandomState(500 + k + int(os.environ.get("ISS_FUZZ_OFFSET", "0")))  # (soak runs: other lists)
    rng_mode = "philox" if k % 2 == 0 else "mt"
    model = str(r.choice(["novaseq", "hiseq", "ecoli", "miseq-36"] + (["basic"] if rng_mode == "mt" else [])))
    dense = dense_model(model)
    RL = dense.read_length
    recs, counts = [], []
    for i in range(int(r.randint(2, 7))):
        L = int(r.choice([RL - 3, RL, RL + 1, 2 * RL, 5 * RL, 4000, 30000]))
        g = (mixed_genome if r.rand() < 0.3 else random_genome)(700 + 10 * k + i, max(L, 4))
        recs.append(Record(g, id="rec%d.%d" % (k, i)))
        counts.append(int(r.choice([0, 1, 2, 63, 64, 65, 300, 1500])))
    frag = (float(r.choice([2 * RL + 80, 3 * RL])), float(r.choice([5, 40]))) if r.rand() < 0.3 else None
    if frag and rng_mode == "mt" and any(len(x.seq) <= RL for x in recs):
        frag = None  # (skipping a record under a custom fragment length is not supported in MT mode)
    seq_type = "amplicon" if r.rand() < 0.15 else "metagenomics"
    gc_bias = bool(r.rand() < 0.3)
    seed, cpu = int(r.randint(0, 10**6)), int(r.randint(0, 40))
    em = BasicErrorModel(*(frag or (None, None)), True) if model == "basic" else dense
    if model != "basic":
        em.store_mutations = True
        em.fragment_length, em.fragment_sd = frag if frag else (None, None)
    prefix = str(tmp_path / "w")
    worker_iterator([(x, n, "default") for x, n in zip(recs, counts)], em, cpu, prefix, seed, seq_type, gc_bias, device=0,
                    rng=rng_mode)
    e1, e2, evcf = _expected_worker_files(dense, recs, counts, seed, cpu, rng_mode, seq_type, gc_bias, frag, tmp_path)
    assert open(prefix + "_R1.fastq", "rb").read() == e1
    assert open(prefix + "_R2.fastq", "rb").read() == e2
    assert open(prefix + ".vcf").read() == evcf


@pytest.mark.parametrize("frag", [None, (420.0, 25.0)])
def test_basic_model_worker_philox(frag, tmp_path):
    """`--mode basic` through the drop-in boundary on the parallel (Philox) path: FASTQ text and VCF == oracle."""
    from insilicoseq_amd.generator import Record, worker_iterator
    from insilicoseq_amd.model import BasicErrorModel

    dense = dense_model("basic")
    recs = [Record(random_genome(1200 + i, L), id="b%d" % i) for i, L in enumerate([5000, 126, 90, 40000, 700])]
    counts = [800, 300, 5, 2500, 64]
    em = BasicErrorModel(*(frag or (None, None)), True)
    seed, cpu = 77, 3
    prefix = str(tmp_path / "w")
    worker_iterator([(x, n, "default") for x, n in zip(recs, counts)], em, cpu, prefix, seed, "metagenomics", False, device=0,
                    rng="philox")
    e1, e2, evcf = _expected_worker_files(dense, recs, counts, seed, cpu, "philox", "metagenomics", False, frag, tmp_path)
    assert open(prefix + "_R1.fastq", "rb").read() == e1
    assert open(prefix + "_R2.fastq", "rb").read() == e2
    assert open(prefix + ".vcf").read() == evcf
    assert e1.count(b"\n") == 4 * (800 + 300 + 2500 + 64)  # (the 90-base record is skipped: generator.py:130)


def test_worker_drops_resident_genomes_over_budget(tmp_path, monkeypatch):
    """A work list larger than the HBM budget for genomes: uploaded records are dropped and the files do not change."""
    from insilicoseq_amd import generator as G

    dense = dense_model("novaseq")
    recs = [G.Record(random_genome(900 + i, 3000 + 500 * i), id="r%d" % i) for i in range(5)]
    work = [(recs[0], 40, "default"), (recs[1], 70, "default"), (recs[1], 5, "default"), (recs[2], 0, "default"),
            (recs[3], 64, "default"), (recs[0], 9, "default"), (recs[4], 33, "default")]
    outs = []
    for budget in (None, 7000):
        if budget:
            monkeypatch.setattr(G.Worker, "GENOME_BUDGET", budget)
        prefix = str(tmp_path / ("w%s" % budget))
        G.worker_iterator(work, dense, 2, prefix, 11, "metagenomics", False, device=0)
        outs.append((open(prefix + "_R1.fastq", "rb").read(), open(prefix + "_R2.fastq", "rb").read()))
    assert outs[0] == outs[1] and outs[0][0].count(b"\n") == 4 * (40 + 70 + 5 + 64 + 9 + 33)


@pytest.mark.parametrize("rng_mode", ["philox", "mt"])
def test_device_gzip_members_hold_the_same_text(rng_mode, tmp_path, monkeypatch):
    """`--compress` on the device (iss_fastq_compress, iss_deflate.hip.h): the gunzipped files are the text files, for
    single and multiple batches, several records (one member per batch), blocks of every size class and ids of
    changing width."""
    import gzip

    from insilicoseq_amd import generator as G

    dense = dense_model("novaseq" if rng_mode == "philox" else "hiseq")
    recs = [G.Record(random_genome(40 + i, 30000), id="contig.%d" % i) for i in range(3)] + [G.Record(mixed_genome(44, 9000), id="m")]
    work = [(recs[0], 1, "default"), (recs[1], 3000, "default"), (recs[2], 0, "default"), (recs[3], 77, "default"),
            (recs[0], 1234, "default"), (recs[1], 99, "default")]
    if rng_mode == "mt":
        work = [(r, min(n, 300), m) for r, n, m in work]
    monkeypatch.setattr(G.Worker, "BATCH_PAIRS", 1000)  # 3000 pairs: three batches = three members
    out = {}
    for compress in (False, True):
        prefix = str(tmp_path / ("c%d" % compress))
        G.worker_iterator(work, dense, 3, prefix, 5, "metagenomics", False, device=0, rng=rng_mode, compress=compress)
        out[compress] = [open(prefix + s, "rb").read() for s in ("_R1.fastq", "_R2.fastq")]
    for plain, packed in zip(out[False], out[True]):
        assert packed[:4] == b"\x1f\x8b\x08\x00" and gzip.decompress(packed) == plain
        assert len(packed) < 0.4 * len(plain)  # (tiny members: the block headers weigh in)


def test_device_gzip_full_batch(tmp_path):
    """Full-size batches (2 600 DEFLATE blocks per file each) and a tail batch through the compressed path."""
    import gzip
    import hashlib

    from insilicoseq_amd import generator as G

    dense = dense_model("novaseq")
    rec = G.Record(random_genome(8, 200000), id="big_genome")
    work = [(rec, (1 << 20) + 12345, "default")]
    sums = {}
    for compress in (False, True):
        prefix = str(tmp_path / ("f%d" % compress))
        G.worker_iterator(work, dense, 0, prefix, 9, "metagenomics", False, device=0, compress=compress)
        for s in ("_R1.fastq", "_R2.fastq"):
            h = hashlib.sha256()
            with (gzip.open if compress else open)(prefix + s, "rb") as fh:
                for chunk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(chunk)
            sums[(compress, s)] = (h.hexdigest(), os.path.getsize(prefix + s))
            os.remove(prefix + s)
    for s in ("_R1.fastq", "_R2.fastq"):
        assert sums[(True, s)][0] == sums[(False, s)][0]
        assert sums[(True, s)][1] < 0.33 * sums[(False, s)][1]


@pytest.mark.parametrize("model,id_len", [("miseq", 3), ("miseq", 700), ("miseq-36", 4000), ("novaseq", 64)])
def test_device_gzip_record_distances(model, id_len, tmp_path):
    """Record lengths from 100 bytes to 4.7 KB (distance codes 12 .. 24 of the previous-record matches), pair numbers
    that change their digit count inside a batch."""
    import gzip

    from insilicoseq_amd import generator as G

    dense = dense_model(model)
    rec = G.Record(random_genome(77, 20000), id=("x" * id_len)[:id_len - 1] + "7")
    out = {}
    for compress in (False, True):
        prefix = str(tmp_path / ("d%d" % compress))
        G.worker_iterator([(rec, 1205, "default")], dense, 12, prefix, 3, "metagenomics", False, device=0, compress=compress)
        out[compress] = [open(prefix + s, "rb").read() for s in ("_R1.fastq", "_R2.fastq")]
    for plain, packed in zip(out[False], out[True]):
        assert gzip.decompress(packed) == plain
        assert len(packed) < 0.45 * len(plain)


@pytest.mark.parametrize("k", range(10))
def test_device_gzip_randomized(k, tmp_path, monkeypatch):
    """Random work lists, batch sizes, record ids and models through the compressed path: gunzip == the text path."""
    import gzip

    from insilicoseq_amd import generator as G

    r = np.random.RandomState(900 + k)
    dense = dense_model(str(r.choice(["novaseq", "hiseq", "miseq-36", "ecoli"])))
    monkeypatch.setattr(G.Worker, "BATCH_PAIRS", int(r.choice([7, 100, 999, 5000])))
    recs, work = [], []
    for i in range(int(r.randint(1, 5))):
        rid = "".join(chr(int(c)) for c in r.choice(list(range(48, 58)) + list(range(65, 91)) + [46, 95, 124], size=int(r.randint(1, 40))))
        gen = (mixed_genome if r.rand() < 0.3 else random_genome)(60 + 7 * k + i, int(r.choice([400, 5000, 40000])))
        recs.append(G.Record(gen, id=rid))
        work.append((recs[-1], int(r.choice([0, 1, 8, 9, 10, 99, 100, 1001, 12000])), "default"))
    out = {}
    for compress in (False, True):
        prefix = str(tmp_path / ("r%d" % compress))
        G.worker_iterator(work, dense, 5, prefix, 17 + k, "metagenomics", False, device=0,
                          compress=compress)
        out[compress] = [open(prefix + s, "rb").read() for s in ("_R1.fastq", "_R2.fastq")]
    for plain, packed in zip(out[False], out[True]):
        assert gzip.decompress(packed) == plain


@pytest.mark.parametrize("length", [3000, (1 << 20) + 4096])
def test_letters_outside_the_rev_comp_alphabet_are_rejected(length):
    """util.rev_comp raises KeyError on such letters (iss/util.py:90); the upload refuses the record -- small records
    are checked on the host, large ones by the pack kernel -- and names the first offender."""
    from insilicoseq_amd import _native
    from insilicoseq_amd.engine import ReadEngine

    eng = ReadEngine(0)
    try:
        eng.load_model(dense_model("ecoli"))
        g = bytearray(random_genome(3, length).encode())
        ok = eng.add_genome(bytes(g))
        assert eng.genome_length(ok) == length
        g[length - 17] = ord("x")
        g[length // 2] = ord("-")
        with pytest.raises(_native.EngineError) as exc:
            eng.add_genome(bytes(g))
        assert "0x2d at offset %d" % (length // 2) in str(exc.value) and "2 such letters" in str(exc.value)
        g[length - 17], g[length // 2] = ord("n"), ord("y")  # IUPAC / lower case are fine
        eng.add_genome(bytes(g))
    finally:
        eng.close()


@pytest.mark.parametrize("case", ["novaseq", "hiseq_gc", "miseq_mixed", "ecoli_amplicon", "indel", "novaseq_vcf",
                                  "novaseq_frag_vcf", "basic_frag_guard", "hiseq_mixed_frag_gc"])
def test_generate_batch_equals_consecutive_calls(case, monkeypatch):
    """iss_generate_batch (one set of launches for a work list) == consecutive iss_generate calls: bases, phreds,
    record coordinates, mutation rows; records repeated in the list, records with IUPAC / lower-case letters, zero-pair
    items, gc_bias, amplicon mode, an indel-heavy model (every other read through the fix-up kernel), custom fragment
    lengths (negative inserts, templates cut by the record's end; "guard": four pairs in ten get their fragment length
    from the host and are set up again)."""
    from insilicoseq_amd.engine import ReadEngine
    from helpers import synthetic_model

    model = {"novaseq": "novaseq", "hiseq_gc": "hiseq", "miseq_mixed": "miseq", "ecoli_amplicon": "ecoli", "novaseq_vcf": "novaseq",
             "novaseq_frag_vcf": "novaseq", "basic_frag_guard": "basic", "hiseq_mixed_frag_gc": "hiseq"}.get(case)
    frag = {"novaseq_frag_vcf": (330.0, 40.0), "basic_frag_guard": (420.0, 90.0), "hiseq_mixed_frag_gc": (500.0, 5.0)}.get(case)
    if "guard" in case:
        monkeypatch.setenv("ISS_MT_GUARD", "0.2")
    dense = synthetic_model(151, 41, 1000, 4, indel=(1e-3, 3e-3)) if case == "indel" else dense_model(model)
    genomes = [random_genome(300, 30000), mixed_genome(301, 9000) if "mixed" in case else random_genome(301, 9000),
               random_genome(302, 700), mixed_genome(303, 52000) if "mixed" in case else random_genome(303, 52000)]
    items = [(0, 1500), (1, 0), (1, 777), (2, 64), (3, 4000), (0, 9), (2, 1)]
    kw = dict(gc_bias="gc" in case, sequence_type="amplicon" if "amplicon" in case else "metagenomics", seed=91)
    n_total = sum(n for _, n in items)
    out = []
    for batch in (False, True):
        eng = ReadEngine(0)
        try:
            eng.load_model(dense)
            if frag:
                eng.set_fragment(*frag)
            if "vcf" in case or case == "indel":
                eng.mutations_reserve(1 << 23)  # (slots are handed out in chunks of 256 per wavefront)
            gids = [eng.add_genome(g) for g in genomes]
            eng.reserve(n_total)  # (growing the buffers between calls would drop the rows already there)
            rows, coords, muts = [], [], []
            if batch:
                eng.generate_batch([gids[g] for g, _ in items], [n for _, n in items], first_ordinal=1000, **kw)
                rows.append(eng.download(0, n_total))
                coords.append(eng.coords(0, n_total))
                if "vcf" in case or case == "indel":
                    muts = [tuple(int(m[k]) for k in ("pair", "mate", "type", "position", "ref", "alt", "quality")) for m in eng.mutations()]
            else:
                ordinal, row = 1000, 0
                for g, n in items:
                    eng.generate(gids[g], n, first_ordinal=ordinal, out_first_pair=row, **kw)
                    coords.append(eng.coords(row, n))
                    if n and ("vcf" in case or case == "indel"):  # (a call without pairs leaves the previous call's rows)
                        muts += [(int(m["pair"]) + row,) + tuple(int(m[k]) for k in ("mate", "type", "position", "ref", "alt", "quality"))
                                 for m in eng.mutations()]
                    ordinal += n
                    row += n
                rows.append(eng.download(0, n_total))
            out.append((rows, np.concatenate([np.asarray(c) for c in coords]), muts))
        finally:
            eng.close()
    (ra, ca, ma), (rb, cb, mb) = out
    for k in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
        assert np.array_equal(ra[0][k], rb[0][k]), k
    assert np.array_equal(ca, cb)
    assert ma == mb


def test_exact_path_store_order_stress():
    """k_main's exact path patches single bytes of rows the same wavefront stored a few iterations earlier and relies on a
    wavefront's vector memory operations reaching an address in issue order (iss_kernels.hip.h, drain_round).  Stress: a
    NovaSeq-shaped model whose rows put half of the mass on Q2, so that more than 30 % of all bases take the exact path (the
    substitution test of a Q2 base fires with probability 0.63) at full occupancy, 10^9 bases in one call.  EVERY base and
    phred is then cross-checked on the device by the one-lane-per-read unit kernels (iss_gen_phred_scores / iss_mut_sequence:
    other kernels, no deferred stores, exact 53-bit compares) on templates rebuilt from the coordinates, and windows of pairs
    are recomputed by the CPU oracle."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    dense = dense_model("novaseq", (0.0, 0.0))
    q = dense.qcdf.copy()
    q[..., :2] *= 0.5
    q[..., 2:] = 0.5 + 0.5 * q[..., 2:]
    q[..., -1] = 1.0
    dense.qcdf[:] = q
    dense.validate()
    RL = dense.read_length
    genome = random_genome(99, 2_000_000)
    g = np.frombuffer(genome.encode(), dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    n = (10**9 + 2 * RL - 1) // (2 * RL)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        eng.generate(gid, n, first_ordinal=5, seed=77)
        eng.synchronize()
        substituted = 0
        step = 250_000
        for at in range(0, n, step):
            m = min(step, n - at)
            got, co = eng.download(at, m), eng.coords(at, m)
            fwd = g[co[:, 0:1] + np.arange(RL)[None, :]]
            rev = comp[g[co[:, 2:3] - 1 - np.arange(RL)[None, :]]]
            for o, tmpl, kb, kq in ((0, fwd, "r1_base", "r1_qual"), (1, rev, "r2_base", "r2_qual")):
                ph = eng.gen_phred_scores(o, m, first_ordinal=5 + at, seed=77)
                assert np.array_equal(ph, got[kq]), (at, kq)
                mut, st = eng.mut_sequence(o, tmpl, ph, first_ordinal=5 + at, seed=77)
                assert not st.any()
                assert np.array_equal(mut, got[kb]), (at, kb)
                substituted += int((got[kb] != tmpl).sum())
        assert substituted >= 0.3 * 2 * RL * n  # (every one of them a late byte patch of the exact path)
        orc = O.Oracle(dense)
        for start in np.random.RandomState(3).randint(0, n - 64, size=24):
            got = eng.download(int(start), 64)
            exp = orc.simulate(O.Rng().seed_philox(77), genome, 64, first_ordinal=5 + int(start))
            for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                assert np.array_equal(got[key], exp[key]), (int(start), key)


@pytest.mark.parametrize("ahead", ["1", "0"])
@pytest.mark.parametrize("model,indel", [("novaseq", (0.001, 0.003)), ("novaseq", None), ("miseq-legacy", None)])
def test_calls_of_many_chunks_reuse_the_counter_rings(model, indel, ahead, monkeypatch):
    """Regression (round-3 advice): the fix-up / read-list counters live in rings of 16 slots indexed by the chunk number, and
    the kernels in front of k_main run on the setup stream, ahead of the main stream.  A call of more than 16 chunks (here 40
    and 25: ISS_CHUNK_PAIRS) comes back to a slot while the chunk that used it last may still be waiting on the main stream:
    the setup stream now waits for that chunk's last kernel before it clears the slot.  Two calls back to back into the same
    rows, then the oracle: every byte, and the coordinates."""
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    monkeypatch.setenv("ISS_SETUP_AHEAD", ahead)
    monkeypatch.setenv("ISS_CHUNK_PAIRS", "1500")
    dense = dense_model(model, indel)
    genome = random_genome(88, 90000)
    orc = O.Oracle(dense)
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        eng.generate(gid, 60000, first_ordinal=0, seed=21)
        eng.generate(gid, 37000, first_ordinal=1000, seed=22)  # (no synchronisation in between)
        eng.synchronize()
        got, cg = eng.download(0, 37000), eng.coords(0, 37000)
        exp = orc.simulate(O.Rng().seed_philox(22), genome, 37000, first_ordinal=1000, want_coords=True)
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(got[key], exp[key]), key
        assert np.array_equal(np.asarray(cg), exp["coords"])
        tail = eng.download(37000, 23000)  # rows 37000.. still hold the FIRST call's pairs
        exp1 = orc.simulate(O.Rng().seed_philox(21), genome, 23000, first_ordinal=37000)
        for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
            assert np.array_equal(tail[key], exp1[key]), ("first call", key)
