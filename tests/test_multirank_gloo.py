"""world_size > 1 path on CPU (gloo): model/genome broadcast, the reference's chunking with
cpus = world, per-rank temp files and rank-order concatenation.  The device is replaced by the CPU
oracle in MT mode (worker seed = seed + rank), so the assembled FASTQ must equal the REFERENCE's own
`iss generate --cpus N` output byte for byte (golden, tests/golden/generate/)."""
import os
import socket

import numpy as np
import pytest

from helpers import GOLDEN


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, tmpdir, golden):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.engine import fastq_write
    from insilicoseq_amd.generator import Record, lognormal_abundance, parse_fasta
    from insilicoseq_amd.model import DenseModel
    from oracle import oracle as O

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        seed, n_reads = 42, 600
        if rank == 0:
            dense = DenseModel.load(os.path.join(golden, "..", "..", "insilicoseq_amd", "profiles", "hiseq.dense.npz"))
            recs = list(parse_fasta(os.path.join(golden, "genomes.fasta")))
            ids = [r.id for r in recs]
            genomes = [r.seq.encode() for r in recs]
        else:
            dense, genomes, ids = None, None, None
        box = [ids]
        dist.broadcast_object_list(box, src=0)
        ids = box[0]
        dense, genomes = D.broadcast_model_and_genomes(dense, genomes, dist, device="cpu")
        ref = DenseModel.load(os.path.join(golden, "..", "..", "insilicoseq_amd", "profiles", "hiseq.dense.npz"))
        for k in DenseModel.FIELDS:
            assert np.array_equal(getattr(dense, k), getattr(ref, k)), k
        records = [Record(g.tobytes().decode(), id=i) for g, i in zip(genomes, ids)]
        # parent-process abundance draw: np.random.seed(seed); abundance.lognormal(...)  (generator.py:397-400)
        abundance = lognormal_abundance(ids, np.random.RandomState(seed))
        output = os.path.join(tmpdir, "out")
        work, chunk_size, n_chunks = D.rank_work(records, None, abundance, n_reads, None, None, dense, output, world,
                                                 rank)
        orc = O.Oracle(dense)
        rng = O.Rng().seed_mt(seed + rank)  # worker_iterator: seed + cpu_number
        prefix = D.temp_prefix(output, rank)
        with open(prefix + "_R1.fastq", "wb") as f1, open(prefix + "_R2.fastq", "wb") as f2:
            for rec, n, _ in (work or []):
                res = orc.simulate(rng, rec.seq, n)
                if res["status"] == O.SKIP_RECORD:
                    continue
                assert res["status"] == 0
                fastq_write(f1.fileno(), f2.fileno(), rec.id, 0, rank, res["n_done"], dense.read_length,
                            dense.read_length, res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], 2)
        dist.barrier()
        if rank == 0:
            D.concatenate_rank_files(output, world)
            z = np.load(os.path.join(golden, "generate", "genomes_hiseq_n600_seed42_cpus%d.npz" % world))
            assert open(output + "_R1.fastq", "rb").read() == z["r1"].tobytes()
            assert open(output + "_R2.fastq", "rb").read() == z["r2"].tobytes()
            lines = z["abundance"].tobytes().decode().split()
            for i, rid in enumerate(ids):
                assert lines[2 * i] == rid and lines[2 * i + 1] == str(abundance[rid])
            assert not os.path.exists(prefix + "_R1.fastq")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_reproduce_reference_generate(world, tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), GOLDEN), nprocs=world, join=True)


def test_baseline_configs0_as_written_on_the_oracle(tmp_path):
    """BASELINE configs[0] as written (`iss generate --genomes data/ecoli.fasta --mode basic -n 10000 --cpus 1`, seed 42; golden of
    tests/golden/tooling/make_golden_configs0.py): the reference's divider with one worker, the CPU oracle (MT streams, the
    BasicErrorModel's legacy-gauss phreds) standing in for the device -- 5 000 pairs, byte for byte."""
    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.engine import fastq_write
    from insilicoseq_amd.generator import lognormal_abundance, parse_fasta
    from insilicoseq_amd.model import DenseModel
    from oracle import oracle as O

    z = np.load(os.path.join(GOLDEN, "generate", "ecoli_basic_n10000_seed42_cpus1.npz"))
    fasta = str(tmp_path / "ecoli.fasta")
    with open(fasta, "wb") as fh:
        fh.write(z["fasta"].tobytes())
    dense = DenseModel.basic()
    records = list(parse_fasta(fasta))
    abundance = lognormal_abundance([r.id for r in records], np.random.RandomState(42))
    output = str(tmp_path / "out")
    work, _, n_chunks = D.rank_work(records, None, abundance, 10000, None, None, dense, output, 1, 0)
    orc, rng = O.Oracle(dense), O.Rng().seed_mt(42)
    prefix = D.temp_prefix(output, 0)
    with open(prefix + "_R1.fastq", "wb") as f1, open(prefix + "_R2.fastq", "wb") as f2:
        for rec, n, _ in work:
            res = orc.simulate(rng, rec.seq, n)
            assert res["status"] == 0
            fastq_write(f1.fileno(), f2.fileno(), rec.id, 0, 0, res["n_done"], dense.read_length, dense.read_length,
                        res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], 2)
    D.concatenate_rank_files(output, 1)
    assert open(output + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(output + "_R2.fastq", "rb").read() == z["r2"].tobytes()


@pytest.mark.parametrize("case,model,n_reads,seed", [("genomes_hiseq_n1600_seed42", "hiseq", 1600, 42),
                                                     ("syn3_novaseq_n3000_seed7", "novaseq", 3000, 7)])
def test_eight_workers_reproduce_reference_generate(case, model, n_reads, seed, tmp_path):
    """`iss generate --cpus 8` (goldens of tests/golden/tooling/make_golden_cpus8.py): chunk r of the reference's divider with
    cpus = 8, worker seed = seed + r, temp files concatenated in worker order -- the CPU oracle (MT streams) standing in for
    the device.  The same fixtures pin the device's W-workers-per-launch mode (tests/test_gpu_mt_compat.py)."""
    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.engine import fastq_write
    from insilicoseq_amd.generator import lognormal_abundance, parse_fasta
    from insilicoseq_amd.model import DenseModel
    from oracle import oracle as O

    z = np.load(os.path.join(GOLDEN, "generate", case + "_cpus8.npz"))
    fasta = os.path.join(GOLDEN, "genomes.fasta")
    if "fasta" in z.files:
        fasta = str(tmp_path / "in.fasta")
        with open(fasta, "wb") as fh:
            fh.write(z["fasta"].tobytes())
    dense = DenseModel.load(os.path.join(GOLDEN, "..", "..", "insilicoseq_amd", "profiles", model + ".dense.npz"))
    records = list(parse_fasta(fasta))
    abundance = lognormal_abundance([r.id for r in records], np.random.RandomState(seed))
    output = str(tmp_path / "out")
    world, n_files = 8, 0
    for rank in range(world):
        work, _, n_chunks = D.rank_work(records, None, abundance, n_reads, None, None, dense, output, world, rank)
        if work is None:
            continue
        orc, rng = O.Oracle(dense), O.Rng().seed_mt(seed + rank)
        prefix = D.temp_prefix(output, rank)
        with open(prefix + "_R1.fastq", "wb") as f1, open(prefix + "_R2.fastq", "wb") as f2:
            for rec, n, _ in work:
                res = orc.simulate(rng, rec.seq, n)
                if res["status"] == O.SKIP_RECORD:
                    continue
                assert res["status"] == 0
                fastq_write(f1.fileno(), f2.fileno(), rec.id, 0, rank, res["n_done"], dense.read_length, dense.read_length,
                            res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], 2)
        n_files += 1
    assert n_files == world
    D.concatenate_rank_files(output, world)
    assert open(output + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(output + "_R2.fastq", "rb").read() == z["r2"].tobytes()


@pytest.mark.parametrize("case,model,n_reads,seed", [("genomes_hiseq_n1600_seed42", "hiseq", 1600, 42),
                                                     ("syn3_novaseq_n3000_seed7", "novaseq", 3000, 7)])
def test_worker_set_iterator_host_logic_reproduces_cpus8(case, model, n_reads, seed, tmp_path, monkeypatch):
    """worker_set_iterator -- W reference workers in ONE engine: rounds of one piece per worker, short records (a draw, no
    rows), per-worker files -- with the device replaced by an engine whose generate_mt_workers runs the CPU oracle on every
    worker's own MT streams: the assembled files are `iss generate --cpus 8`'s (goldens of make_golden_cpus8.py).  The
    device's side of the same call is tests/test_gpu_mt_compat.py."""
    import insilicoseq_amd.generator as G
    from insilicoseq_amd import _native
    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.engine import fastq_write
    from insilicoseq_amd.model import DenseModel
    from oracle import oracle as O

    dense = DenseModel.load(os.path.join(GOLDEN, "..", "..", "insilicoseq_amd", "profiles", model + ".dense.npz"))

    class OracleEngine(object):
        def __init__(self, device):
            self.genomes, self.rows, self.calls = [], {}, 0

        read_length = dense.read_length

        def load_model(self, d):
            self.orc = O.Oracle(d)

        def seed_mt_workers(self, seeds):
            self.rngs = [O.Rng().seed_mt(int(sd)) for sd in seeds]

        def mt_set_fragment(self, a, b):
            assert a is None and b is None

        def add_genome(self, seq):
            self.genomes.append(seq)
            return len(self.genomes) - 1

        def generate_mt_workers(self, g, n, row, sequence_type="metagenomics", gc_bias=False):
            self.calls += 1
            done, status = np.zeros(len(n), dtype=np.int64), np.zeros(len(n), dtype=np.int32)
            for w in range(len(n)):
                if n[w] == 0:
                    continue
                res = self.orc.simulate(self.rngs[w], self.genomes[g[w]], int(n[w]), gc_bias=gc_bias)
                if res["status"] == O.SKIP_RECORD:
                    status[w] = _native.E_SHORT_RECORD
                    continue
                assert res["status"] == 0
                done[w] = res["n_done"]
                self.rows[int(row[w])] = res
            return done, status

        def fastq_emit(self, fd1, fd2, rid, first_i, cpu, first_pair, n_pairs, n_threads=1):
            res = self.rows.pop(int(first_pair))
            assert res["n_done"] == n_pairs
            fastq_write(fd1, fd2, rid, first_i, cpu, n_pairs, dense.read_length, dense.read_length, res["r1_base"], res["r1_qual"],
                        res["r2_base"], res["r2_qual"], 1)

        def fastq_flush(self):
            pass

        def close(self):
            pass

    made = []
    monkeypatch.setattr(G, "ReadEngine", lambda device: made.append(OracleEngine(device)) or made[-1])
    z = np.load(os.path.join(GOLDEN, "generate", case + "_cpus8.npz"))
    fasta = os.path.join(GOLDEN, "genomes.fasta")
    if "fasta" in z.files:
        fasta = str(tmp_path / "in.fasta")
        with open(fasta, "wb") as fh:
            fh.write(z["fasta"].tobytes())
    records = list(G.parse_fasta(fasta))
    abundance = G.lognormal_abundance([r.id for r in records], np.random.RandomState(seed))
    output = str(tmp_path / "out")
    world = 8
    works = [D.rank_work(records, None, abundance, n_reads, None, None, dense, output, world, r)[0] for r in range(world)]
    assert all(w is not None for w in works)
    G.worker_set_iterator(works, dense, list(range(world)), [D.temp_prefix(output, r) for r in range(world)], seed, "metagenomics",
                          False, device=0, batch_pairs=64)
    assert len(made) == 1 and made[0].calls > 2  # ONE engine for the eight workers, several rounds
    D.concatenate_rank_files(output, world)
    assert open(output + "_R1.fastq", "rb").read() == z["r1"].tobytes()
    assert open(output + "_R2.fastq", "rb").read() == z["r2"].tobytes()


def test_single_rank_is_a_noop_broadcast():
    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.model import DenseModel

    dense = DenseModel.load(os.path.join(GOLDEN, "..", "..", "insilicoseq_amd", "profiles", "ecoli.dense.npz"))
    d2, g2 = D.broadcast_model_and_genomes(dense, [b"ACGT" * 10], None)
    assert d2 is dense and g2[0].tobytes() == b"ACGT" * 10


def _forced_one_rank_main(rank, port):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.model import DenseModel

    dense = DenseModel.load(os.path.join(GOLDEN, "..", "..", "insilicoseq_amd", "profiles", "ecoli.dense.npz"))
    rng = np.random.RandomState(3)
    genomes = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, size=n)] for n in (1, 15, 16, 17, 4099)]
    genomes.append(np.frombuffer(b"ACGTNacgtRYKM", dtype=np.uint8)[rng.randint(0, 13, size=333)])  # (travels as ASCII)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        for as_refs in (False, True):
            d2, g2 = D.broadcast_model_and_genomes(dense, genomes, dist, device="cpu", as_refs=as_refs, force=True)
            for k in DenseModel.FIELDS:
                assert np.array_equal(getattr(d2, k), getattr(dense, k)), k
            got = [g.ascii() for g in g2] if as_refs else g2
            assert len(got) == len(genomes) and all(np.array_equal(np.asarray(a), b) for a, b in zip(got, genomes))
    finally:
        dist.destroy_process_group()


def test_forced_one_rank_broadcast_roundtrips_the_payload():
    """force=True sends the payload through the collectives even with one rank (the switch behind bench.py's
    ISS_BENCH_FORCE_DIST and the one-rank RCCL tests on the GPU box): header, tables, 2-bit packed and ASCII genomes -- incl. lengths
    around a 16-base word -- come back as they went in."""
    import torch.multiprocessing as mp

    mp.spawn(_forced_one_rank_main, args=(_free_port(),), nprocs=1, join=True)
