"""Reference-compatible MT mode on the GPU: the device consumes the reference's two MT19937 streams in
the reference's order, so its output must equal the REFERENCE's golden vectors directly (no oracle in
between): pair sets, stream positions afterwards, worker FASTQ files, and whole `iss generate` runs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, dense_model, load_pairs_case, pairs_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from insilicoseq_amd.engine import ReadEngine

    eng = ReadEngine(0)
    yield eng
    eng.close()


def _res53(words):
    w = words.astype(np.uint64)
    return ((w[0::2] >> np.uint64(5)) * np.uint64(67108864) + (w[1::2] >> np.uint64(6))).astype(np.float64) / 9007199254740992.0


def test_mt_streams_on_device(engine):
    z = np.load(os.path.join(GOLDEN, "mt_taps.npz"))
    engine.load_model(dense_model("ecoli"))
    for s in (0, 1, 42, 43, 2**31, 2**32 - 1):
        engine.seed_mt(s)
        py, npw = engine.mt_peek(624)
        assert (py == z["py_%d" % s][:624]).all()
        assert (_res53(npw) == z["npd_%d" % s][:312]).all()


MT_CASES = pairs_cases()  # every golden pair set, custom fragment lengths (negative inserts) included


@pytest.mark.parametrize("case", MT_CASES)
def test_pairs_equal_reference(engine, case):
    z, meta = load_pairs_case(case)
    dense = dense_model(meta["model"], meta["indel"])
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(z["genome"].tobytes())
    engine.seed_mt(meta["seed"])
    engine.mt_set_fragment(meta["fragment_length"], meta["fragment_sd"])
    n = meta["n_pairs"]
    if meta["n_done"] == 0:
        pytest.skip("record skipped by the reference")
    try:
        assert engine.generate_mt(gid, n, sequence_type=meta["sequence_type"], gc_bias=meta["gc_bias"]) == n
    finally:
        engine.mt_set_fragment(None, None)
    got = engine.download(0, n)
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        bad = np.argwhere(got[k] != z[k])
        assert bad.size == 0, "%s differs from the reference at %s (%d cells)" % (k, bad[:5].tolist(), len(bad))
    py, npw = engine.mt_peek(8)  # both streams stand where the reference's stand
    assert list(_res53(py)) == list(z["tail_py"]) and list(_res53(npw)) == list(z["tail_np"])


@pytest.mark.parametrize("case", ["genomes_hiseq_cpu0", "genomes_miseq_cpu1", "syn_novaseq_cpu3_gc", "genomes_basic_cpu2",
                                  "syn_novaseq_frag_short"])
def test_worker_files_equal_reference(case, tmp_path):
    """(syn_novaseq_frag_short: --fragment-length with records the worker skips after their fragment-length draw,
    iss/generator.py:121-130 -- the gaussian is replayed on the host and the streams stay the reference's.)"""
    from insilicoseq_amd.generator import Record, worker_iterator

    z = np.load(os.path.join(GOLDEN, "worker", case + ".npz"))
    meta = json.loads(str(z["meta"]))
    recs = [Record(z["genome_%d" % i].tobytes().decode(), id=rid) for i, rid in enumerate(meta["ids"])]
    work = [(r, n, "default") for r, n in zip(recs, meta["counts"])]
    prefix = str(tmp_path / "w")
    em = dense_model(meta["model"])
    if meta.get("fragment_length") is not None:
        em.fragment_length, em.fragment_sd = meta["fragment_length"], meta["fragment_sd"]
    worker_iterator(work, em, meta["cpu_number"], prefix, meta["seed"], meta["sequence_type"],
                    meta["gc_bias"], device=0, rng="mt")
    assert open(prefix + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(prefix + "_R2.fastq", "rb").read() == z["r2"].tobytes()


def test_generate_cli_basic_mode_equals_reference(tmp_path):
    """`--mode basic --rng mt` (BasicErrorModel on the device) == `iss generate --mode basic --cpus 2`."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "run")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes",
                           os.path.join(GOLDEN, "genomes.fasta"), "--mode", "basic", "--rng", "mt", "-n", "400", "--seed", "42",
                           "--cpus", "2", "--devices", "1", "-o", out, "--quiet"], cwd=root)
    z = np.load(os.path.join(GOLDEN, "generate", "genomes_basic_n400_seed42_cpus2.npz"))
    assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()
    assert open(out + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(out + "_R2.fastq", "rb").read() == z["r2"].tobytes()


def test_generate_cli_baseline_configs0_as_written(tmp_path):
    """BASELINE configs[0] as written -- `iss generate --genomes data/ecoli.fasta --mode basic -n 10000 --cpus 1` (seed 42;
    /root/reference/iss/app.py:333-341, /root/reference/iss/error_models/basic.py:40-63) -- through the device in the
    byte-identical mode: the reference's own files (tests/golden/tooling/make_golden_configs0.py), byte for byte."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(GOLDEN, "generate", "ecoli_basic_n10000_seed42_cpus1.npz"))
    fasta = str(tmp_path / "ecoli.fasta")
    with open(fasta, "wb") as fh:
        fh.write(z["fasta"].tobytes())
    out = str(tmp_path / "run")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes", fasta, "--mode", "basic", "--rng", "mt",
                           "-n", "10000", "--seed", "42", "--cpus", "1", "--devices", "1", "-o", out, "--quiet"], cwd=root)
    assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()
    for name, key, sha in (("_R1.fastq", "r1", "sha_r1"), ("_R2.fastq", "r2", "sha_r2")):
        got = open(out + name, "rb").read()
        assert got.count(b"\n") == 4 * 5000
        assert got == z[key].tobytes(), name
        import hashlib
        assert hashlib.sha256(got).hexdigest() == str(z[sha])


@pytest.mark.parametrize("cpus", [1, 2, 3])
def test_generate_cli_equals_reference(cpus, tmp_path):
    """python -m insilicoseq_amd generate --rng mt == the reference's `iss generate --cpus N` output files."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "run")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes",
                           os.path.join(GOLDEN, "genomes.fasta"), "--model", "hiseq", "-n", "600", "--seed", "42",
                           "--cpus", str(cpus), "--devices", "1", "--rng", "mt", "-o", out, "--quiet"], cwd=root)
    z = np.load(os.path.join(GOLDEN, "generate", "genomes_hiseq_n600_seed42_cpus%d.npz" % cpus))
    assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()
    assert open(out + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(out + "_R2.fastq", "rb").read() == z["r2"].tobytes()


@pytest.mark.parametrize("case", ["genomes_hiseq_n1600_seed42", "syn3_novaseq_n3000_seed7"])
def test_generate_cli_cpus8_equals_reference(case, tmp_path):
    """`iss generate --cpus 8` -- the reference's own parallelism, eight workers seeded seed + cpu_number (iss/app.py:81-106,
    iss/generator.py:234-236) -- as eight chains side by side on ONE GPU (iss_generate_mt_workers: every kernel of the MT
    path launched once for all workers, one workgroup each): the assembled files equal the reference's byte for byte.
    genomes_hiseq: data/genomes.fasta (short low-complexity records: genome-end fallbacks, a skipped record);
    syn3_novaseq: three random 20 kbp records carried in the fixture (the resolver's fast path)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(GOLDEN, "generate", case + "_cpus8.npz"))
    fasta = os.path.join(GOLDEN, "genomes.fasta")
    if "fasta" in z.files:
        fasta = str(tmp_path / "in.fasta")
        with open(fasta, "wb") as fh:
            fh.write(z["fasta"].tobytes())
    model, n, seed = case.split("_")[1], case.split("_")[2][1:], case.split("_")[3][4:]
    out = str(tmp_path / "run")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes", fasta, "--model", model, "-n", n,
                           "--seed", seed, "--cpus", "8", "--devices", "1", "--rng", "mt", "-o", out, "--quiet"], cwd=root)
    assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()
    assert open(out + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(out + "_R2.fastq", "rb").read() == z["r2"].tobytes()


@pytest.mark.parametrize("turn", [None, "37", "37/3"])
@pytest.mark.parametrize("case", ["novaseq", "hiseq_gc", "miseq", "indel_heavy", "basic", "novaseq_frag", "novaseq_gzip"])
def test_worker_set_equals_separate_workers(case, turn, tmp_path, monkeypatch):
    """W workers side by side in one context (worker_set_iterator -> iss_generate_mt_workers) write the files of W
    separate worker_iterator(rng="mt") runs -- which are the reference's (tests above) -- for work lists of different
    lengths over a plain record, a record with IUPAC / lower-case letters (the resolver hands such pairs to the walker) and
    a record shorter than a read (skipped after its draw).  turn = 37: many short turns per call (stream words produced
    ahead -- appended to a stream's buffer, or, at the buffer's end, into the other one with the unconsumed words moved in
    front: "/3" = buffers of three turns, a move every second turn -- and consumed across turn boundaries); indel_heavy: the walker only; basic / novaseq_frag: the workers take
    the single-worker path one after the other (draws the host's libm settles)."""
    import gzip

    from helpers import mixed_genome, random_genome
    from insilicoseq_amd.generator import Record, worker_iterator, worker_set_iterator

    if turn:
        monkeypatch.setenv("ISS_MT_SET_TURN", turn.split("/")[0])
        if "/" in turn:
            monkeypatch.setenv("ISS_MT_SET_BUF_TURNS", turn.split("/")[1])
    model = {"hiseq_gc": "hiseq", "indel_heavy": "novaseq", "novaseq_frag": "novaseq", "novaseq_gzip": "novaseq"}.get(case, case)
    em = dense_model(model, (0.01, 0.03) if case == "indel_heavy" else None)
    if case == "novaseq_frag":
        em.fragment_length, em.fragment_sd = 420.0, 35.0
    gc = case == "hiseq_gc"
    compress = case == "novaseq_gzip"
    recs = [Record(random_genome(201, 30000), id="plain"), Record(mixed_genome(202, 9000), id="mixed"),
            Record(random_genome(203, 120), id="short"), Record(random_genome(204, 2500), id="small")]
    r = np.random.RandomState(5)
    scale = 1 if case in ("indel_heavy", "basic", "miseq") else 4
    works = []
    for w in range(5):
        items = [(recs[int(k)], int(r.randint(1, 260 * scale)), "default") for k in r.randint(0, 4, size=int(r.randint(1, 5)))]
        works.append(items)
    works[3] = [(recs[0], 700 * scale, "default")]  # one long item: its worker is still busy when the others are done
    cpus = [0, 1, 2, 5, 9]
    seed = 77
    set_prefix = [str(tmp_path / ("set%d" % c)) for c in cpus]
    worker_set_iterator(works, em, cpus, set_prefix, seed, "metagenomics", gc, device=0, compress=compress, batch_pairs=300)
    rd = (lambda p: gzip.open(p, "rb").read()) if compress else (lambda p: open(p, "rb").read())
    for work, c, sp in zip(works, cpus, set_prefix):
        one = str(tmp_path / ("one%d" % c))
        worker_iterator(work, em, c, one, seed, "metagenomics", gc, device=0, rng="mt", compress=compress)
        for suffix in ("_R1.fastq", "_R2.fastq"):
            a, b = rd(sp + suffix), rd(one + suffix)
            assert a == b, (case, c, suffix, len(a), len(b))
    assert sum(len(rd(sp + "_R1.fastq")) for sp in set_prefix) > 100_000  # (a worker whose items are all short records writes nothing)


@pytest.mark.parametrize("cpus", [1, 2])
def test_generate_cli_compress_equals_reference(cpus, tmp_path):
    """`--compress` (gzip members built on the device, one per worker batch, concatenated in worker order; the .vcf
    through the host path): the gunzipped files are the reference's `iss generate --cpus N` files and no text file is left."""
    import gzip

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "run")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes",
                           os.path.join(GOLDEN, "genomes.fasta"), "--model", "hiseq", "-n", "600", "--seed", "42",
                           "--cpus", str(cpus), "--devices", "1", "--rng", "mt", "--compress", "--store_mutations",
                           "-o", out, "--quiet"], cwd=root)
    z = np.load(os.path.join(GOLDEN, "generate", "genomes_hiseq_n600_seed42_cpus%d.npz" % cpus))
    assert gzip.open(out + "_R1.fastq.gz", "rb").read() == z["r1"].tobytes()
    assert gzip.open(out + "_R2.fastq.gz", "rb").read() == z["r2"].tobytes()
    assert gzip.open(out + ".vcf.gz", "rb").read().startswith(b"##fileformat=VCFv4.1")
    left = sorted(os.listdir(str(tmp_path)))
    assert left == ["run.vcf.gz", "run_R1.fastq.gz", "run_R2.fastq.gz", "run_abundance.txt"], left


@pytest.mark.parametrize("case", ["halfnormal", "zero_inflated_lognormal", "exponential", "uniform", "coverage_lognormal",
                                  "coverage_halfnormal", "abundance_file", "coverage_file", "readcount_file"])
def test_generate_cli_abundance_inputs_equal_reference(case, tmp_path):
    """The abundance / coverage / read-count inputs of `iss generate` (iss/generator.py:497-594): the distribution file
    the run writes and the FASTQ files equal the reference's (`--cpus 2`, goldens of make_golden_cli.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(GOLDEN, "generate", "cli_%s.npz" % case))
    flags = str(z["flags"]).split()
    given = str(tmp_path / "given.txt")
    with open(given, "wb") as fh:
        fh.write(z["given"].tobytes())
    flags = [given if f.startswith("@") else f for f in flags]
    out = str(tmp_path / "run")
    subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes",
                           os.path.join(GOLDEN, "genomes.fasta"), "--model", "hiseq", "--seed", "42", "--cpus", "2",
                           "--devices", "1", "--rng", "mt", "-o", out, "--quiet"] + flags, cwd=root)
    assert open(out + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(out + "_R2.fastq", "rb").read() == z["r2"].tobytes()
    assert os.path.exists(out + "_abundance.txt") == bool(z["has_abundance"])
    assert os.path.exists(out + "_coverage.txt") == bool(z["has_coverage"])
    if z["has_abundance"]:
        assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()
    if z["has_coverage"]:
        assert open(out + "_coverage.txt", "rb").read() == z["coverage"].tobytes()


def test_mt_mode_large_equals_oracle(engine):
    """20k pairs (several stream refills) against the CPU oracle in MT mode, plus the stream positions."""
    from helpers import random_genome
    from oracle import oracle as O

    dense = dense_model("novaseq")
    genome = random_genome(91, 300000)
    n = 20000
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.seed_mt(4242)
    assert engine.generate_mt(gid, n) == n
    got = engine.download(0, n)
    rng = O.Rng().seed_mt(4242)
    exp = O.Oracle(dense).simulate(rng, genome, n)
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        assert np.array_equal(got[k], exp[k]), k
    py, npw = engine.mt_peek(8)
    assert list(_res53(py)) == [rng.py_random() for _ in range(4)]
    assert list(_res53(npw)) == [rng.np_random() for _ in range(4)]


def _sparse_exception_genome(seed, n):
    """Mostly plain ACGT with a few N runs and lower-case stretches: most pairs resolve, some are walked."""
    from helpers import random_genome

    g = bytearray(random_genome(seed, n).encode())
    rng = np.random.RandomState(seed + 1)
    for start in rng.randint(0, n - 200, size=12):
        g[start:start + 40] = b"N" * 40
    for start in rng.randint(0, n - 200, size=12):
        g[start:start + 60] = bytes(g[start:start + 60]).lower()
    return g.decode()


RESOLVER_CASES = {
    "novaseq": dict(model="novaseq", L=300000, n=30000),
    "hiseq": dict(model="hiseq", L=250000, n=12000),
    "miseq": dict(model="miseq", L=250000, n=9000),          # digit rows too large for the LDS: read from memory
    "miseq-36": dict(model="miseq-36", L=250000, n=9000),    # 2000-entry insert-size CDF, a single quality bin
    "novaseq_gc": dict(model="novaseq", L=300000, n=12000, gc_bias=True),
    "novaseq_amplicon": dict(model="novaseq", L=5000, n=6000, sequence_type="amplicon"),
    "novaseq_short": dict(model="novaseq", L=400, n=6000),   # width <= 0 and reverse-end fallback randrange
    "novaseq_exceptions": dict(model="novaseq", L=200000, n=20000, exceptions=True),
    "miseq_legacy_indels": dict(model="miseq-legacy", L=200000, n=3000),  # indel candidates everywhere: walker only
    "novaseq_vcf": dict(model="novaseq", L=300000, n=20000, mut=True),              # --store_mutations rows
    "hiseq_vcf_exceptions": dict(model="hiseq", L=150000, n=9000, mut=True, exceptions=True),
    "novaseq_frag400": dict(model="novaseq", L=300000, n=12000, frag=(400, 30)),
    "novaseq_frag160": dict(model="novaseq", L=3000, n=6000, frag=(160, 40)),   # negative inserts, templates cut by the ends
    "hiseq_frag_gc": dict(model="hiseq", L=100000, n=8000, frag=(350, 60), gc_bias=True),
}


@pytest.mark.parametrize("case", sorted(RESOLVER_CASES))
def test_resolver_equals_walker(engine, monkeypatch, case):
    """iss_generate_mt has two device paths (offset resolver + parallel emitter; sequential walker).  The walker
    is pinned to the reference by the goldens above; here both paths must agree byte for byte, stream
    positions included, and the resolver must really have been the one running."""
    from helpers import random_genome

    c = RESOLVER_CASES[case]
    dense = dense_model(c["model"])
    genome = _sparse_exception_genome(7, c["L"]) if c.get("exceptions") else random_genome(7, c["L"])
    kw = dict(sequence_type=c.get("sequence_type", "metagenomics"), gc_bias=c.get("gc_bias", False))
    n = c["n"]
    outs = {}
    for path in ("walk", "resolve"):
        if path == "walk":
            monkeypatch.setenv("ISS_MT_PATH", "walk")
        else:
            monkeypatch.delenv("ISS_MT_PATH", raising=False)
        engine.load_model(dense)
        engine.clear_genomes()
        gid = engine.add_genome(genome)
        engine.seed_mt(77)
        engine.mt_set_fragment(*c.get("frag", (None, None)))
        engine.reserve(n + 500)
        engine.mt_mutations_reserve(16 * n if c.get("mut") else 0)
        r0, w0 = engine.mt_path_counts()
        assert engine.generate_mt(gid, n, **kw) == n
        rows = engine.mt_mutations().copy() if c.get("mut") else None
        # a second work item on the same streams (offsets that do not start at a buffer boundary)
        assert engine.generate_mt(gid, 500, out_first_pair=n, **kw) == 500
        r1, w1 = engine.mt_path_counts()
        got = engine.download(0, n + 500)
        coords = engine.coords(0, n + 500)
        outs[path] = (got, coords, engine.mt_peek(16), (r1 - r0, w1 - w0), rows)
        engine.mt_set_fragment(None, None)
        engine.mt_mutations_reserve(0)
    (ga, ca, pa, cnt_walk, rows_a), (gb, cb, pb, cnt_res, rows_b) = outs["walk"], outs["resolve"]
    if c.get("mut"):
        assert len(rows_a) > 0.2 * n and rows_a.tobytes() == rows_b.tobytes()
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        bad = np.argwhere(ga[k] != gb[k])
        assert bad.size == 0, "%s differs at %s (%d cells)" % (k, bad[:5].tolist(), len(bad))
    assert np.array_equal(np.asarray(ca), np.asarray(cb))
    assert (pa[0] == pb[0]).all() and (pa[1] == pb[1]).all()
    assert cnt_walk == (0, n + 500)
    if case == "miseq_legacy_indels":
        assert cnt_res == (0, n + 500)
    elif case in ("novaseq_exceptions", "novaseq_frag160", "hiseq_vcf_exceptions"):
        assert cnt_res[0] > 0.5 * n and cnt_res[1] > 50 and sum(cnt_res) == n + 500
    else:
        assert cnt_res[0] > 0.95 * n and sum(cnt_res) == n + 500


def test_config1_scale_short_genomes_miseq(engine):
    """BASELINE configs[1] at its full size: data/genomes.fasta (records shorter than the MiSeq fragment: fallback
    branches everywhere), --model miseq, 1 M reads, seed-fixed, GPU (MT mode) vs CPU (oracle with the reference's MT
    streams) bit-exact -- 500 k pairs spread over the records like a worker's work list."""
    from insilicoseq_amd.generator import parse_fasta
    from oracle import oracle as O

    dense = dense_model("miseq")
    records = list(parse_fasta(os.path.join(GOLDEN, "genomes.fasta")))
    counts = [200000, 75000, 100000, 124900, 100]
    engine.load_model(dense)
    engine.clear_genomes()
    engine.seed_mt(42)
    rng = O.Rng().seed_mt(42)
    orc = O.Oracle(dense)
    from insilicoseq_amd import _native

    for rec, n in zip(records, counts):
        exp = orc.simulate(rng, rec.seq, n)
        if exp["status"] == O.SKIP_RECORD:
            with pytest.raises(_native.EngineError):
                engine.generate_mt(engine.add_genome(rec.seq), n)
            continue
        gid = engine.add_genome(rec.seq)
        assert engine.generate_mt(gid, n) == n
        got = engine.download(0, n)
        for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
            assert np.array_equal(got[k], exp[k]), (rec.id, k)
    py, npw = engine.mt_peek(8)
    assert list(_res53(py)) == [rng.py_random() for _ in range(4)]
    assert list(_res53(npw)) == [rng.np_random() for _ in range(4)]


def test_config1_full_size_philox_windows(engine):
    """The same work list (BASELINE configs[1]: 1 M reads, data/genomes.fasta, --model miseq) on the parallel path: ONE
    iss_generate_batch call, windows of every work item (its first and last 1500 pairs: ordinals run across the items)
    recomputed by the oracle's Philox provider."""
    from insilicoseq_amd.generator import parse_fasta
    from oracle import oracle as O

    dense = dense_model("miseq")
    records = [r for r in parse_fasta(os.path.join(GOLDEN, "genomes.fasta")) if len(r.seq) > dense.read_length]
    counts = [200000, 75000, 100000, 124900, 100][:len(records)]
    counts[-1] += 500000 - sum(counts)
    engine.load_model(dense)
    engine.clear_genomes()
    gids = [engine.add_genome(r.seq) for r in records]
    engine.reserve(sum(counts))
    engine.generate_batch(gids, counts, first_ordinal=7, seed=42, out_first_pair=0)
    engine.synchronize()
    orc = O.Oracle(dense)
    first = 0
    for rec, n in zip(records, counts):
        for w0 in sorted({0, max(0, n - 1500)}):
            m = min(1500, n - w0)
            got = engine.download(first + w0, m)
            exp = orc.simulate(O.Rng().seed_philox(42), rec.seq, m, first_ordinal=7 + first + w0)
            assert exp["status"] == 0
            for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
                assert np.array_equal(got[k], exp[k]), (rec.id, w0, k)
        first += n


@pytest.mark.parametrize("case", ["syn_novaseq_vcf", "genomes_basic_cpu2"])
def test_store_mutations_vcf_equals_reference(case, tmp_path):
    """--store_mutations in MT mode: the worker's .vcf (and FASTQ) equal the reference's files."""
    from insilicoseq_amd.generator import Record, worker_iterator
    from insilicoseq_amd.model import BasicErrorModel, KDErrorModel

    z = np.load(os.path.join(GOLDEN, "worker", case + ".npz"))
    meta = json.loads(str(z["meta"]))
    recs = [Record(z["genome_%d" % i].tobytes().decode(), id=rid) for i, rid in enumerate(meta["ids"])]
    work = [(r, n, "default") for r, n in zip(recs, meta["counts"])]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if meta["model"] == "basic":
        em = BasicErrorModel(None, None, True)
    else:
        em = KDErrorModel(os.path.join(root, "insilicoseq_amd", "profiles", "novaseq.dense.npz"), None, None, True)
    prefix = str(tmp_path / "w")
    worker_iterator(work, em, meta["cpu_number"], prefix, meta["seed"], meta["sequence_type"], meta["gc_bias"],
                    device=0, rng="mt")
    assert open(prefix + ".vcf", "rb").read() == z["vcf"].tobytes()
    assert open(prefix + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(prefix + "_R2.fastq", "rb").read() == z["r2"].tobytes()


@pytest.mark.parametrize("gc_bias", [False, True])
def test_mutation_records_with_indels_equal_oracle(engine, gc_bias):
    """Insertion / deletion / substitution rows of an indel-heavy model on a mixed-case + IUPAC genome."""
    from helpers import mixed_genome
    from oracle import oracle as O

    dense = dense_model("novaseq", (0.01, 0.03))
    genome = mixed_genome(5, 20000)
    n = 1500
    engine.load_model(dense)
    engine.clear_genomes()
    gid = engine.add_genome(genome)
    engine.mt_mutations_reserve(200000)
    engine.seed_mt(77)
    assert engine.generate_mt(gid, n, gc_bias=gc_bias) == n
    got = engine.mt_mutations()
    engine.mt_mutations_reserve(0)
    exp = O.Oracle(dense).simulate(O.Rng().seed_mt(77), genome, n, gc_bias=gc_bias, store_mutations=True)["mutations"]
    assert len(got) == len(exp) and len(got) > 1000
    for f in ("pair", "mate", "type", "position", "ref", "alt", "quality"):
        assert np.array_equal(got[f], exp[f]), f
    assert set(np.unique(got["type"])) == {0, 1, 2}


def test_fragment_length_host_guard_path(monkeypatch):
    """Widen the 'too close to an integer' guard so EVERY normal draw is re-evaluated on the host: the
    stop / override / resume protocol must still reproduce the reference (incl. the cached gaussian)."""
    from insilicoseq_amd.engine import ReadEngine

    monkeypatch.setenv("ISS_MT_GUARD", "0.6")
    # (BasicErrorModel: the same guard sends nine phred scores out of ten to the host's libm as well)
    for case in ("novaseq_frag160", "ecoli_gcbias_frag", "basic_frag300", "basic_mixed", "basic_gcbias"):
        z, meta = load_pairs_case(case)
        with ReadEngine(0) as eng:
            eng.load_model(dense_model(meta["model"], meta["indel"]))
            gid = eng.add_genome(z["genome"].tobytes())
            eng.seed_mt(meta["seed"])
            eng.mt_set_fragment(meta["fragment_length"], meta["fragment_sd"])
            n = meta["n_pairs"]
            assert eng.generate_mt(gid, n, sequence_type=meta["sequence_type"], gc_bias=meta["gc_bias"]) == n
            got = eng.download(0, n)
            for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
                assert np.array_equal(got[k], z[k]), (case, k)
            py, npw = eng.mt_peek(8)
            assert list(_res53(py)) == list(z["tail_py"]) and list(_res53(npw)) == list(z["tail_np"])


@pytest.mark.parametrize("n_short", [1, 3, 40])
def test_short_record_first_after_seeding_keeps_the_streams(n_short):
    """Regression (round-3 advice): with --fragment-length a record shorter than a read still costs the reference one
    np.random.normal draw before its assertion fails (iss/generator.py:121-130).  The host replays that gaussian from the
    stream words -- which must be read BEHIND the refill mt_ensure queues on the engine's (non-blocking) streams.  Short
    records as the very first calls after seeding (nothing filled yet; n_short > 1: cached second values in between, 40: past
    a refill), then a normal record: its reads and the positions of both streams equal the oracle's in MT mode."""
    from helpers import random_genome
    from insilicoseq_amd._native import EngineError
    from insilicoseq_amd.engine import ReadEngine
    from oracle import oracle as O

    dense = dense_model("novaseq")
    short, genome = random_genome(71, dense.read_length - 3), random_genome(72, 40000)
    n = 300
    with ReadEngine(0) as eng:
        eng.load_model(dense)
        g_short, g_long = eng.add_genome(short), eng.add_genome(genome)
        eng.seed_mt(991)
        eng.mt_set_fragment(400.0, 35.0)
        for _ in range(n_short):
            with pytest.raises(EngineError):
                eng.generate_mt(g_short, 5)
        assert eng.generate_mt(g_long, n) == n
        got = eng.download(0, n)
        py, npw = eng.mt_peek(8)
    orc, rng = O.Oracle(dense), O.Rng().seed_mt(991)
    for _ in range(n_short):
        assert orc.simulate(rng, short, 5, fragment_length=400.0, fragment_sd=35.0)["status"] != 0
    exp = orc.simulate(rng, genome, n, fragment_length=400.0, fragment_sd=35.0)
    assert exp["status"] == 0
    for k in ("r1_qual", "r2_qual", "r1_base", "r2_base"):
        assert np.array_equal(got[k], exp[k]), k
    assert [int(x) for x in py] == [rng.py_word() for _ in range(8)]  # both streams stand where the oracle's stand
    assert [int(x) for x in npw] == [rng.np_word() for _ in range(8)]
