"""CPU-side tests: C-ABI library loads and exports every declared symbol; host logic (FASTA parse,
work divider, FASTQ formatter) against reference goldens.  No GPU compute here."""
import io
import json
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN, dense_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as ge

    ge.build()
    from insilicoseq_amd import _native

    return _native


def test_library_exports_every_declared_symbol(native):
    lib = native.lib()
    header = open(os.path.join(ROOT, "include", "iss_mi355x.h")).read()
    declared = set(re.findall(r"\b(iss_[a-z0-9_]+)\s*\(", header))
    declared -= {"iss_ctx"}
    assert declared == set(native.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.iss_abi_version() == 8


def test_no_gpu_means_loud_failure(native):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from insilicoseq_amd.engine import ReadEngine

    with pytest.raises(native.EngineError):
        ReadEngine(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "insilicoseq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("the cpu oracle", "").replace("cpu oracle", ""), f


def _worker_case(name):
    z = np.load(os.path.join(GOLDEN, "worker", name + ".npz"))
    meta = json.loads(str(z["meta"]))
    genomes = [z["genome_%d" % i].tobytes() for i in range(len(meta["ids"]))]
    return z, meta, genomes


@pytest.mark.parametrize("case", ["genomes_hiseq_cpu0", "genomes_miseq_cpu1", "syn_novaseq_cpu3_gc", "genomes_basic_cpu2",
                                  "syn_novaseq_frag_short"])
def test_worker_fastq_matches_reference(native, case, tmp_path):
    """Oracle (MT streams, seeded like worker_iterator: seed + cpu_number) + the product's FASTQ
    formatter reproduce the reference worker's R1/R2 files byte for byte (ids, order, skipped
    records, gc_bias rejections, stream carry-over between work items; syn_novaseq_frag_short: --fragment-length with
    records the worker skips AFTER their fragment-length draw, iss/generator.py:121-130)."""
    from insilicoseq_amd.engine import fastq_write
    from oracle import oracle as O

    z, meta, genomes = _worker_case(case)
    d = dense_model(meta["model"])
    orc = O.Oracle(d)
    rng = O.Rng().seed_mt(meta["seed"] + meta["cpu_number"])
    p1, p2 = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for rid, n, g in zip(meta["ids"], meta["counts"], genomes):
            res = orc.simulate(rng, g, n, sequence_type=meta["sequence_type"], gc_bias=meta["gc_bias"],
                               fragment_length=meta.get("fragment_length"), fragment_sd=meta.get("fragment_sd"))
            if res["status"] == O.SKIP_RECORD:
                continue
            assert res["status"] == 0
            k = res["n_done"]
            fastq_write(f1.fileno(), f2.fileno(), rid, 0, meta["cpu_number"], k, d.read_length, d.read_length,
                        res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], n_threads=3)
    assert p1.read_bytes() == z["r1"].tobytes()
    assert p2.read_bytes() == z["r2"].tobytes()


def test_fastq_writer_large_and_threaded(native, tmp_path):
    from insilicoseq_amd.engine import fastq_write

    rng = np.random.RandomState(0)
    n, RL, pitch = 40000, 37, 40
    arrs = [rng.randint(65, 85, size=(n, pitch)).astype(np.uint8) if k % 2 == 0 else
            rng.randint(0, 41, size=(n, pitch)).astype(np.uint8) for k in range(4)]
    outs = []
    for nt in (1, 5):
        p1, p2 = tmp_path / ("a%d" % nt), tmp_path / ("b%d" % nt)
        with open(p1, "wb") as f1, open(p2, "wb") as f2:
            fastq_write(f1.fileno(), f2.fileno(), "rec.1", 7, 12, n, RL, pitch, *arrs, n_threads=nt)
        outs.append((p1.read_bytes(), p2.read_bytes()))
    assert outs[0] == outs[1]
    lines = outs[0][0].split(b"\n")
    assert lines[0] == b"@rec.1_7_12/1" and lines[2] == b"+"
    assert lines[1] == arrs[0][0, :RL].tobytes()
    assert lines[3] == bytes(33 + int(x) for x in arrs[1][0, :RL])
    assert lines[4 * (n - 1)] == b"@rec.1_%d_12/1" % (7 + n - 1)
    assert outs[0][1].split(b"\n")[0] == b"@rec.1_7_12/2"


def test_parse_fasta_and_work_divider_match_reference_generate():
    """`iss generate --genomes data/genomes.fasta --model hiseq -n 600 --seed 42 --cpus {1,2,3}`:
    the per-record pair counts and chunk boundaries implied by the reference's FASTQ ids are
    reproduced by generate_work_divider fed with the reference's own abundance file."""
    from insilicoseq_amd.generator import generate_work_divider, parse_fasta

    records = list(parse_fasta(os.path.join(GOLDEN, "genomes.fasta")))
    assert [r.id for r in records] == ["genome_A", "genome_T", "genome_GC", "genome_ATCG", "genome_TA"]
    dense = dense_model("hiseq")
    for cpus in (1, 2, 3):
        z = np.load(os.path.join(GOLDEN, "generate", "genomes_hiseq_n600_seed42_cpus%d.npz" % cpus))
        abundance = {}
        for line in z["abundance"].tobytes().decode().splitlines():
            k, v = line.split("\t")
            abundance[k] = float(v)
        n_reads = 600
        chunk_size = -((n_reads // 2) // -cpus)
        chunks = list(generate_work_divider(records, None, abundance, n_reads, None, None, dense, "x", chunk_size))
        # counts per (record, cpu) observed in the reference output
        seen = {}
        for line in z["r1"].tobytes().decode().splitlines()[0::4]:
            m = re.match(r"@(.+)_(\d+)_(\d+)/1$", line)
            key = (m.group(1), int(m.group(3)))
            seen[key] = max(seen.get(key, 0), int(m.group(2)) + 1)
        expect = {}
        for cpu, chunk in enumerate(chunks[:cpus]):  # zip(work_chunks, temp_file_list) drops a surplus chunk
            for rec, n, _ in chunk:
                if len(rec.seq) > dense.read_length:  # shorter records are skipped by the worker
                    expect[(rec.id, cpu)] = expect.get((rec.id, cpu), 0) + n
        # a record split inside one worker restarts its ids; compare totals per key via max id only when single
        for key, n in seen.items():
            assert key in expect and expect[key] >= n, (key, n, expect.get(key))
        assert set(seen) == set(expect)


@pytest.mark.parametrize("case", ["syn_novaseq_vcf", "genomes_basic_cpu2"])
def test_oracle_vcf_rows_match_reference(case):
    """--store_mutations: the oracle's mutation records, formatted like write_mutations
    (iss/generator.py:598-620), reproduce the reference worker's .vcf byte for byte."""
    from oracle import oracle as O

    z, meta, genomes = _worker_case(case)
    assert meta["store_mutations"]
    d = dense_model(meta["model"])
    orc = O.Oracle(d)
    rng = O.Rng().seed_mt(meta["seed"] + meta["cpu_number"])
    lines = []
    for rid, n, g in zip(meta["ids"], meta["counts"], genomes):
        res = orc.simulate(rng, g, n, store_mutations=True)
        if res["status"] == O.SKIP_RECORD:
            continue
        assert res["status"] == 0
        for m in res["mutations"]:
            read_id = "%s_%d_%d/%d" % (rid, m["pair"], meta["cpu_number"], 1 + int(m["mate"]))
            ref, alt = chr(m["ref"]), chr(m["alt"])
            if m["type"] == 1:  # insertion: alt = ref + inserted letter (__init__.py:203)
                alt = ref + alt
            qual = str(int(m["quality"])) if m["type"] == 0 else "."
            lines.append("\t".join([read_id, str(int(m["position"]) + 1), ".", ref, alt, qual, "", ""]) + "\n")
    assert "".join(lines).encode() == z["vcf"].tobytes()


def test_compress_file_is_one_gzip_stream(tmp_path):
    """--compress (iss/util.py:255-268): block-parallel gzip members must read back as the original bytes."""
    import gzip

    from insilicoseq_amd.app import compress_file

    rng = np.random.RandomState(3)
    for n in (0, 17, 3_000_000):
        data = rng.randint(33, 74, size=n).astype(np.uint8).tobytes()
        path = str(tmp_path / ("reads_%d.fastq" % n))
        with open(path, "wb") as fh:
            fh.write(data)
        gz = compress_file(path, block_bytes=1 << 20, threads=3)
        assert gz == path + ".gz" and not os.path.exists(path)
        with gzip.open(gz, "rb") as fh:
            assert fh.read() == data


def _length_code(n):
    """RFC 1951 3.2.5 for match lengths 3..32: (symbol, extra bits, their value)"""
    if n <= 10:
        return 254 + n, 0, 0
    k = n - 11
    if k < 8:
        return 265 + (k >> 1), 1, k & 1
    return 269 + ((k - 8) >> 2), 2, (k - 8) & 3


def _tokens(data, dist=0):
    """The device's tokens (iss_deflate.hip.h, deflate_tokens): 32-byte chunks; at every position the run (the byte
    repeats its predecessor) and the previous record (the same bytes `dist` earlier) are tried, the longer one wins
    with >= 3 (run) / >= 4 (previous record) bytes inside the chunk, else a literal.
    -> (symbol, kind 0 literal / 1 run / 2 previous record, extra bits of the length code, their value)"""
    out = []
    for at in range(0, len(data), 32):
        chunk = data[at:at + 32]
        has_src = bool(dist) and at >= dist
        i = 0
        while i < len(chunk):
            c = chunk[i]
            r1 = rd = 0
            while i + r1 < len(chunk) and at + i + r1 > 0 and data[at + i + r1] == data[at + i + r1 - 1]:
                r1 += 1
            while has_src and i + rd < len(chunk) and chunk[i + rd] == data[at + i + rd - dist]:
                rd += 1
            if r1 >= 3 and r1 >= rd:
                out.append((*_length_code(r1)[:1], 1, *_length_code(r1)[1:]))
                i += r1
            elif rd >= 4:
                out.append((*_length_code(rd)[:1], 2, *_length_code(rd)[1:]))
                i += rd
            else:
                out.append((c, 0, 0, 0))
                i += 1
    return out


def _deflate_block(native, data, hist=None, dist=0):
    """One DEFLATE block of `data` built on the CPU with the code tables of iss_deflate_code_build (what the device
    kernels pack): header bits, the tokens' codes, end of block, then an empty stored block and a final empty block."""
    import ctypes as C

    toks = _tokens(data, dist)
    if hist is None:
        hist = np.bincount(np.array([t[0] for t in toks] + [256], dtype=np.int64), minlength=273).astype(np.uint32)
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    assert hist.size == 273
    entry = np.zeros(273, dtype=np.uint32)
    hdr = np.zeros(64, dtype=np.uint32)
    dcode = np.zeros(3, dtype=np.uint32)
    nbits = C.c_uint32(0)
    assert native.lib().iss_deflate_code_build(hist.ctypes.data, dist, entry.ctypes.data, C.byref(nbits), hdr.ctypes.data,
                                               dcode.ctypes.data) == 0
    lens = (entry >> 16).astype(np.int64)
    assert lens.min() >= 1 and lens.max() <= 15
    assert sum(2.0 ** -int(x) for x in lens) == 1.0  # complete code (inflate rejects anything else)
    acc = 0
    for w in range((nbits.value + 31) // 32):
        acc |= int(hdr[w]) << (32 * w)
    acc &= (1 << nbits.value) - 1
    n = nbits.value
    for sym, kind, xbits, xval in toks:
        acc |= (int(entry[sym]) & 0xffff) << n
        n += int(lens[sym])
        acc |= xval << n           # extra bits of the length code
        n += xbits
        if kind == 1:              # distance 1: the bit 0
            n += 1
        elif kind == 2:            # the record distance: the bit 1, then its extra bits
            acc |= (1 | (int(dcode[2]) << 1)) << n
            n += 1 + int(dcode[1])
    acc |= (int(entry[256]) & 0xffff) << n
    n += int(lens[256])
    n += 3                      # empty stored block: BFINAL 0, BTYPE 00
    n = (n + 7) // 8 * 8
    acc |= 0xffff0000 << n      # LEN 0, NLEN 0xffff
    n += 32
    return acc.to_bytes(n // 8, "little") + b"\x03\x00", lens


def test_deflate_code_builder_makes_valid_streams(native):
    """The code builder shared by the host and the device (iss_deflate.hip.h): length-limited complete Huffman codes
    and the dynamic-block header, checked by inflating CPU-packed blocks with zlib."""
    import zlib

    rng = np.random.RandomState(5)
    acgt, quals = np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(b"#-8F", dtype=np.uint8)
    fastq = b"".join(b"@genome_%d_%d_0/1\n" % (7, 1000 + i) + rng.choice(acgt, 151).tobytes() + b"\n+\n" +
                     rng.choice(quals, 151, p=[0.005, 0.015, 0.03, 0.95]).tobytes() + b"\n" for i in range(300))
    rec = len(fastq) // 300
    cases = [(fastq, 0), (fastq, rec), (fastq, rec + 1), (b"A", 0), (b"", 0), (bytes(range(256)) * 3, 256), (bytes(range(256)) * 3, 5),
             (bytes(rng.randint(0, 256, 5000).astype(np.uint8)), 1000), (b"\x00" * 4000 + b"\x01", 8),
             (b"FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF", 0), (b"ab" + b"c" * 9 + b"dd" + b"e" * 3, 4), (fastq[:5000] * 8, 5000),
             (fastq[:4097] * 9, 4097), (fastq[:1000] * 40, 32768)]
    for data, dist in cases:
        block, lens = _deflate_block(native, data, dist=dist)
        assert zlib.decompressobj(-15).decompress(block) == data, dist
    runs_only, _ = _deflate_block(native, fastq)
    block, lens = _deflate_block(native, fastq, dist=rec)
    assert len(block) < 0.9 * len(runs_only) < 0.31 * len(fastq)
    # a code built from one histogram still encodes symbols the histogram never saw
    hist = np.bincount(np.array([t[0] for t in _tokens(fastq)] + [256]), minlength=273).astype(np.uint32)
    other = b"nnnnNNNN@@@\xff\x00" * 50
    block, _ = _deflate_block(native, other, hist=hist, dist=13)
    assert zlib.decompressobj(-15).decompress(block) == other
    # extreme counts (a long batch) keep the 15-bit limit
    hist = np.ones(273, dtype=np.uint32)
    hist[65], hist[67], hist[10] = 4_000_000_000, 200_000_000, 3
    block, lens = _deflate_block(native, b"ACCA\n", hist=hist)
    assert zlib.decompressobj(-15).decompress(block) == b"ACCA\n" and lens[65] == 1


def test_deflate_code_builder_randomized(native):
    """Random token statistics (flat, geometric, one-symbol, zero-heavy, huge counts) through the code builder: every
    block inflates, the code is complete and within 15 bits."""
    import zlib

    rng = np.random.RandomState(11)
    for case in range(120):
        kind = case % 6
        n_sym = int(rng.choice([1, 2, 3, 5, 17, 64, 200, 256]))
        alphabet = rng.choice(256, size=n_sym, replace=False).astype(np.uint8)
        if kind == 0:
            p = np.ones(n_sym)
        elif kind == 1:
            p = 0.5 ** np.arange(n_sym)
        elif kind == 2:
            p = rng.rand(n_sym) ** 8 + 1e-12
        elif kind == 3:
            p = 1.0 / (1 + np.arange(n_sym)) ** 3
        elif kind == 4:
            p = np.where(np.arange(n_sym) == 0, 1.0, 1e-4)
        else:
            p = rng.rand(n_sym)
        p = p / p.sum()
        size = int(rng.choice([0, 1, 7, 8, 9, 100, 3000]))
        data = alphabet[rng.choice(n_sym, size=size, p=p)].tobytes()
        if case % 4 == 1 and size >= 100:  # periodic data: previous-record matches
            data = (data[:37] * (size // 37 + 1))[:size]
        dist = int(rng.choice([0, 0, 5, 37, 300, 4000, 32768]))
        scale = int(rng.choice([1, 1, 1000, 1_000_000]))  # the same statistics at a long batch's counts
        toks = _tokens(data, dist)
        hist = np.bincount(np.array([t[0] for t in toks] + [256], dtype=np.int64), minlength=273) * scale
        hist = np.minimum(hist, 2**32 - 2).astype(np.uint32)
        block, lens = _deflate_block(native, data, hist=hist, dist=dist)
        assert zlib.decompressobj(-15).decompress(block) == data, case


@pytest.mark.parametrize("case", ["halfnormal", "zero_inflated_lognormal", "exponential", "uniform", "coverage_lognormal",
                                  "coverage_halfnormal"])
def test_abundance_distributions_match_reference(case, tmp_path):
    """The abundance / coverage distributions of the CLI (iss/abundance.py:80-175, 196-228) drawn after the reference's
    seeding (generator.py:397-400): the file the reference's `iss generate --seed 42` wrote, byte for byte."""
    import random

    from insilicoseq_amd import app
    from insilicoseq_amd.generator import parse_fasta

    z = np.load(os.path.join(GOLDEN, "generate", "cli_%s.npz" % case))
    flags = str(z["flags"]).split()
    records = list(parse_fasta(os.path.join(GOLDEN, "genomes.fasta")))
    ids = [r.id for r in records]
    random.seed(42)
    np.random.seed(42)
    out = str(tmp_path / "o")
    if flags[0] == "--abundance":
        dic = app.ABUNDANCE[flags[1]](ids)
        app._write_distribution(dic, out, "abundance")
        assert open(out + "_abundance.txt", "rb").read() == z["abundance"].tobytes()
    else:
        dic = app.coverage_scaling(int(flags[3]), app.ABUNDANCE[flags[1]](ids), records, dense_model("hiseq").read_length)
        app._write_distribution(dic, out, "coverage")
        assert open(out + "_coverage.txt", "rb").read() == z["coverage"].tobytes()


def test_batched_worker_loop_cuts_batches_across_items(monkeypatch):
    """_simulate_work_batched (the worker's loop on the parallel path) against a recording stand-in for the engine:
    batches of BATCH_PAIRS pairs cut across work items, pair ids continuing inside an item, running ordinals, records
    not longer than the read length skipped with the reference's two warnings, one emit job per batch."""
    import io

    from insilicoseq_amd import generator as G

    class FakeEngine:
        read_length = 100

        def __init__(self):
            self.calls = []

        def generate_batch(self, gids, counts, first_ordinal, seed, sequence_type, gc_bias, out_first_pair):
            self.calls.append(("gen", list(gids), list(counts), first_ordinal))

        def fastq_emit_batch(self, fd1, fd2, items, cpu):
            self.calls.append(("emit", list(items), cpu))

    class FakeWorker:
        BATCH_PAIRS = 100
        GENOME_BUDGET = 10**9

        def __init__(self):
            self.engine = FakeEngine()
            self.ordinal, self.seed, self.cpu_number, self.store_mutations = 7, 5, 3, False
            self.ids = {}

        def needs_room_for(self, record):
            return False

        def genome_id(self, record):
            return self.ids.setdefault(record.id, len(self.ids))

    recs = [G.Record("A" * 500, id="r0"), G.Record("C" * 100, id="short"), G.Record("G" * 300, id="r2"), G.Record("T" * 999, id="r3")]
    work = [(recs[0], 30, "default"), (recs[1], 50, "default"), (recs[2], 0, "default"), (recs[2], 170, "default"),
            (recs[3], 1, "default"), (recs[0], 99, "default")]
    w = FakeWorker()
    f1, f2 = open(os.devnull, "wb"), open(os.devnull, "wb")
    G._simulate_work_batched(w, work, f1, f2, io.StringIO(), "metagenomics", False)
    gens = [c for c in w.engine.calls if c[0] == "gen"]
    emits = [c for c in w.engine.calls if c[0] == "emit"]
    assert [c[2] for c in gens] == [[30, 70], [100], [1, 99]] and [c[3] for c in gens] == [7, 107, 207]
    assert [c[1] for c in gens] == [[0, 1], [1], [2, 0]]
    assert emits[0][1] == [("r0", 0, 0, 30), ("r2", 0, 30, 70)]
    assert emits[1][1] == [("r2", 70, 0, 100)]          # pair ids continue inside the work item
    assert emits[2][1] == [("r3", 0, 0, 1), ("r0", 0, 1, 99)]  # ... and restart for the next item of the same record
    assert w.ordinal == 7 + 300 and all(e[2] == 3 for e in emits)


@pytest.mark.parametrize("case", ["halfnormal", "zero_inflated_lognormal", "coverage_lognormal", "coverage_halfnormal",
                                  "abundance_file", "coverage_file", "readcount_file"])
def test_work_divider_matches_reference_cli_inputs(case):
    """The pair counts per (record, worker) that the reference's FASTQ ids imply, for every kind of abundance / coverage /
    read-count input (`--cpus 2`), against generate_work_divider fed with what the reference run wrote or was given."""
    from insilicoseq_amd.generator import generate_work_divider, parse_fasta

    z = np.load(os.path.join(GOLDEN, "generate", "cli_%s.npz" % case))
    flags = str(z["flags"]).split()
    records = list(parse_fasta(os.path.join(GOLDEN, "genomes.fasta")))
    dense = dense_model("hiseq")
    text = (z["given"] if z["given"].size else (z["coverage"] if z["has_coverage"] else z["abundance"])).tobytes().decode()
    table = {line.split()[0]: float(line.split()[1]) for line in text.splitlines() if line.strip()}
    coverage = flags[1] if flags[0] == "--coverage" else None
    coverage_file = "given" if flags[0] == "--coverage_file" else None
    readcount = {k: int(v) for k, v in table.items()} if flags[0] == "--readcount_file" else None
    n_reads = sum(readcount.values()) if readcount else int(flags[flags.index("-n") + 1])
    chunk_size = -((n_reads // 2) // -2)
    chunks = list(generate_work_divider(records, readcount, None if readcount else table, n_reads, coverage, coverage_file, dense,
                                        "x", chunk_size))
    expect = {}
    for cpu, chunk in enumerate(chunks[:2]):
        for rec, n, _ in chunk:
            if len(rec.seq) > dense.read_length:
                expect[(rec.id, cpu)] = expect.get((rec.id, cpu), 0) + n
    seen = {}
    for line in z["r1"].tobytes().decode().splitlines()[0::4]:
        m = re.match(r"@(.+)_(\d+)_(\d+)/1$", line)
        key = (m.group(1), int(m.group(3)))
        seen[key] = seen.get(key, 0) + 1
    assert seen == expect


def test_concatenate_rank_files(tmp_path):
    """util.concatenate over the workers' temp files (iss/app.py:123-133, iss/util.py:213-234): rank order, the VCF
    header line, renamed outputs for workers that wrote gzip members, temp files removed, a missing file is an error."""
    from insilicoseq_amd.distributed import VCF_HEADER, concatenate_rank_files, temp_prefix

    out = str(tmp_path / "run")

    def make(world, suffixes):
        for r in range(world):
            for s in suffixes:
                with open(temp_prefix(out, r) + s, "wb") as fh:
                    fh.write(("%s of rank %d\n" % (s, r)).encode() * (r + 1))

    make(3, ("_R1.fastq", "_R2.fastq", ".vcf"))
    concatenate_rank_files(out, 3, suffixes=("_R1.fastq", "_R2.fastq", ".vcf"), headers={".vcf": VCF_HEADER})
    assert open(out + "_R1.fastq").read() == "".join("_R1.fastq of rank %d\n" % r * (r + 1) for r in range(3))
    assert open(out + ".vcf").read() == VCF_HEADER + "\n" + "".join(".vcf of rank %d\n" % r * (r + 1) for r in range(3))
    assert sorted(os.listdir(str(tmp_path))) == ["run.vcf", "run_R1.fastq", "run_R2.fastq"]
    make(2, ("_R1.fastq", "_R2.fastq"))
    concatenate_rank_files(out, 2, out_suffixes={"_R1.fastq": "_R1.fastq.gz", "_R2.fastq": "_R2.fastq.gz"})
    assert open(out + "_R2.fastq.gz").read() == "_R2.fastq of rank 0\n" + "_R2.fastq of rank 1\n" * 2
    make(1, ("_R1.fastq", "_R2.fastq"))
    with pytest.raises(FileNotFoundError):  # fewer chunks than workers
        concatenate_rank_files(out, 2)


def test_device_code_has_no_stale_scc_select():
    """tools/scan_isa.py: hipcc 7.2 lowered a uniform 64-bit `min` in the deflate kernels to a v_cmp followed by an
    s_cselect on a stale SCC (wrong block sizes); the source avoids the construct now and the build is scanned for it."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("scan_isa", os.path.join(ROOT, "tools", "scan_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_write_mutations_format():
    """The VCF rows of write_mutations (iss/generator.py:598-620, __init__.py:98-108, 197-221) against a row-by-row
    restatement: substitutions carry their quality, insertions alt = ref + letter, deletions alt '.'."""
    import io

    from insilicoseq_amd.engine import MUT_DTYPE
    from insilicoseq_amd.generator import write_mutations

    r = np.random.RandomState(1)
    n = 5000
    rows = np.zeros(n, dtype=MUT_DTYPE)
    rows["pair"] = np.sort(r.randint(0, 3000, n))
    rows["mate"] = r.randint(0, 2, n)
    rows["type"] = r.choice([0, 0, 0, 1, 2], n)
    rows["position"] = r.randint(0, 301, n)
    rows["ref"] = r.choice(list(b"ACGTNacgt"), n)
    rows["alt"] = r.choice(list(b"ACGT."), n)
    rows["quality"] = r.randint(-1, 41, n)
    expect = []
    for m in rows:
        ref, alt = chr(m["ref"]), chr(m["alt"])
        alt = ref + alt if m["type"] == 1 else alt
        qual = str(int(m["quality"])) if m["type"] == 0 else "."
        expect.append("\t".join(["rec|1_%d_%d/%d" % (77 + int(m["pair"]), 12, 1 + int(m["mate"])), str(int(m["position"]) + 1), ".",
                                 ref, alt, qual, "", ""]) + "\n")
    got = io.StringIO()
    write_mutations(rows, got, "rec|1", 77, 12)
    assert got.getvalue() == "".join(expect)
    got = io.StringIO()
    write_mutations(rows[:0], got, "x", 0, 0)
    assert got.getvalue() == ""


def test_parse_fasta_fast_path_equals_line_parser(tmp_path):
    """parse_fasta on a path (records cut at header lines, line ends removed in one pass) == the line-by-line parser:
    CRLF, blank lines, text before the first header, empty ids, white space inside sequence lines, no final newline."""
    import io

    from insilicoseq_amd.generator import _parse_fasta_lines, parse_fasta

    cases = [b"", b"\n\n", b">a\nACGT\n", b">a desc here\r\nAC\r\nGT\r\n>b\r\n\r\nTT", b"junk\nmore\n>x\nAA\n\n>y z\nCC\n>\nGG\n>empty\n",
             b">s\nA C\n G\tT \n>t\nAAA", b">only", b"ACGT\n", b">a\nAC>GT\n>b\nT", b"\n>a\nAC\n", b">a\n" + b"ACGTacgtNN\n" * 5000 + b">b\nT"]
    for k, data in enumerate(cases):
        path = tmp_path / ("c%d.fasta" % k)
        path.write_bytes(data)
        fast = [(r.id, r.description, r.seq) for r in parse_fasta(str(path))]
        slow = [(r.id, r.description, r.seq) for r in _parse_fasta_lines(io.StringIO(data.decode()))]
        assert fast == slow, k


def test_batched_worker_loop_falls_back_to_single_calls():
    """Records too long for one arena (iss_generate_batch: ISS_E_INVALID, "... must stay below 2^34 - 4096 bases"): the batch is generated item
    by item into the same rows (running ordinals), mutation rows renumbered to the batch, one emit job as before."""
    import io

    from insilicoseq_amd import _native
    from insilicoseq_amd import generator as G
    from insilicoseq_amd.engine import MUT_DTYPE

    class FakeEngine:
        read_length = 100

        def __init__(self):
            self.calls, self.last = [], None

        def generate_batch(self, *a, **k):
            raise _native.EngineError(_native.E_INVALID, "iss_generate_batch: the records of one call must stay below 2^34 - 4096 bases")

        def reserve(self, n):
            self.calls.append(("reserve", n))

        def generate(self, gid, n, first_ordinal, seed, sequence_type, gc_bias, out_first_pair):
            self.calls.append(("gen1", gid, n, first_ordinal, out_first_pair))
            self.last = n

        def mutations(self):
            rows = np.zeros(2, dtype=MUT_DTYPE)
            rows["pair"] = [0, self.last - 1]
            rows["ref"], rows["alt"] = ord("A"), ord("C")
            return rows

        def fastq_emit_batch(self, fd1, fd2, items, cpu):
            self.calls.append(("emit", list(items)))

    class FakeWorker:
        BATCH_PAIRS = 1000

        def __init__(self):
            self.engine = FakeEngine()
            self.ordinal, self.seed, self.cpu_number, self.store_mutations = 0, 1, 4, True

        def needs_room_for(self, record):
            return False

        def genome_id(self, record):
            return {"a": 0, "b": 1}[record.id]

    w = FakeWorker()
    vcf = io.StringIO()
    work = [(G.Record("A" * 500, id="a"), 40, "default"), (G.Record("C" * 500, id="b"), 25, "default")]
    G._simulate_work_batched(w, work, open(os.devnull, "wb"), open(os.devnull, "wb"), vcf, "metagenomics", False)
    assert [c for c in w.engine.calls if c[0] == "gen1"] == [("gen1", 0, 40, 0, 0), ("gen1", 1, 25, 40, 40)]
    assert [c for c in w.engine.calls if c[0] == "emit"] == [("emit", [("a", 0, 0, 40), ("b", 0, 40, 25)])]
    ids = [line.split("\t")[0] for line in vcf.getvalue().splitlines()]
    assert ids == ["a_0_4/1", "a_39_4/1", "b_0_4/1", "b_24_4/1"] and w.ordinal == 65


def test_mutation_rows_retry_on_overflow():
    """A --store_mutations batch that overflows its row buffer (ISS_E_NOMEM) is repeated ONCE with the reservation the call
    itself asked for (iss_mutations_download reports the slots needed) instead of aborting the worker (generation is a pure
    function of seed and ordinal); an engine that does not say how many doubles; other errors propagate."""
    import insilicoseq_amd.generator as G
    from insilicoseq_amd import _native

    class FakeEngine:
        def __init__(self, need, tells=True):
            self.mutations_capacity, self.need, self.calls, self.tells = 1 << 17, need, [], tells

        def mutations(self):
            if self.mutations_capacity < self.need:
                self.mutation_slots_needed = self.need if self.tells else 0
                raise _native.EngineError(_native.E_NOMEM, "mutation rows overflow")
            return "rows"

        def mutations_reserve(self, cap):
            self.calls.append(("reserve", cap))
            self.mutations_capacity = cap

    eng, regen = FakeEngine(1 << 19), []
    assert G.mutation_rows(eng, lambda: regen.append(1)) == "rows"
    assert len(eng.calls) == 1 and (1 << 19) <= eng.calls[0][1] <= (1 << 20) and len(regen) == 1
    eng, regen = FakeEngine(1 << 19, tells=False), []
    assert G.mutation_rows(eng, lambda: regen.append(1)) == "rows"
    assert len(regen) == len(eng.calls) >= 1 and eng.mutations_capacity >= 1 << 19

    class Broken(FakeEngine):
        def mutations(self):
            raise _native.EngineError(_native.E_HIP, "device lost")

    with pytest.raises(_native.EngineError):
        G.mutation_rows(Broken(0), lambda: None)


def test_worker_resolves_records_by_ordinal_not_by_id(tmp_path, monkeypatch):
    """Draft assemblies repeat contig ids: a worker must simulate every work item from ITS record (the reference hands
    the record objects to the pool, iss/app.py:99-106), not from the last record carrying the same id."""
    from insilicoseq_amd import app

    fasta = tmp_path / "g.fasta"
    fasta.write_text(">contig_1 first\n" + "ACGT" * 100 + "\n>contig_1 second\n" + "TTGGCCAA" * 60 + "\n>other\n" + "GATTACA" * 70 + "\n")
    seen = []

    def fake_worker_iterator(work, model, rank, prefix, seed, sequence_type, gc_bias, **kw):
        seen.extend((str(rec.seq), n) for rec, n, _ in work)

    monkeypatch.setattr(app, "worker_iterator", fake_worker_iterator)
    monkeypatch.setattr(app, "BasicErrorModel", lambda *a, **k: object())
    app._worker(0, 0, str(fasta), [(0, 5), (1, 7), (2, 9)], None, 1, str(tmp_path / "w"), "metagenomics", False, "mt", False,
                (None, None))
    assert [n for _, n in seen] == [5, 7, 9]
    assert seen[0][0].startswith("ACGTACGT") and seen[1][0].startswith("TTGGCCAA") and seen[2][0].startswith("GATTACA")


def test_two_block_fill_index_arithmetic():
    """mt_fill_two (two MT19937 blocks per barrier, both in terms of the old block -- tools/patches/mt_fill_two_blocks.patch:
    built, correct, slower, not merged): its index arithmetic, lane by lane in Python (tools/mt_two_block_model.py), yields
    numpy's own words."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("mt_two_block_model", os.path.join(root, "tools", "mt_two_block_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
