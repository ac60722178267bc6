"""The world_size > 1 path THROUGH THE DEVICE on a single-GPU box: two ranks (gloo; both on GPU 0) receive the one flat
broadcast of model tables + 2-bit packed genomes into device memory, upload their genomes straight from that buffer
(iss_genome_upload_packed), take chunk `rank` of the reference's divider (iss/app.py:81-83), generate their work list with
iss_generate_batch, and every rank's rows are compared with the CPU oracle (worker seed = seed + rank, ordinals running
across its work items); the per-rank FASTQ files concatenated in rank order equal the oracle's."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, tmpdir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist

    from helpers import dense_model, mixed_genome, random_genome
    from insilicoseq_amd import distributed as D
    from insilicoseq_amd.engine import ReadEngine, fastq_write
    from insilicoseq_amd.generator import Record, lognormal_abundance
    from oracle import oracle as O

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        seed, n_reads = 77, 60000
        # rank 0 owns the inputs; the other rank builds the same letters only to CHECK what it received
        letters = [random_genome(300 + i, 20000 + 3000 * i) for i in range(6)] + [mixed_genome(5, 9000)]
        dense0 = dense_model("hiseq")
        dense, refs = D.broadcast_model_and_genomes(dense0 if rank == 0 else None,
                                                    [g.encode() for g in letters] if rank == 0 else None, dist,
                                                    device=torch.device("cuda", 0), as_refs=True)
        assert [r.packed for r in refs] == [True] * 6 + [False]  # the mixed-case record travels as ASCII
        assert all(r.dev_ptr for r in refs)
        for k in type(dense).FIELDS:
            assert np.array_equal(getattr(dense, k), getattr(dense0, k)), k
        records = [Record(g, id="g%d" % i) for i, g in enumerate(letters)]
        abundance = lognormal_abundance([r.id for r in records], np.random.RandomState(seed))
        output = os.path.join(tmpdir, "out")
        work, chunk_size, n_chunks = D.rank_work(records, None, abundance, n_reads, None, None, dense, output, world, rank)
        assert n_chunks >= world and work
        prefix = D.temp_prefix(output, rank)
        with ReadEngine(0) as eng:
            eng.load_model(dense)
            gid = {id(r): g.upload(eng) for r, g in zip(records, refs)}  # from the broadcast buffer in HBM
            pairs = [n for _, n, _ in work]
            eng.reserve(sum(pairs))
            eng.generate_batch([gid[id(r)] for r, _, _ in work], pairs, first_ordinal=0, seed=seed + rank, out_first_pair=0)
            eng.synchronize()
            got = eng.download(0, sum(pairs))
        orc = O.Oracle(dense)
        first = 0
        with open(prefix + "_R1.fastq", "wb") as f1, open(prefix + "_R2.fastq", "wb") as f2, \
                open(prefix + "_exp_R1.fastq", "wb") as e1, open(prefix + "_exp_R2.fastq", "wb") as e2:
            for (rec, n, _) in work:
                res = orc.simulate(O.Rng().seed_philox(seed + rank), rec.seq, n, first_ordinal=first)
                assert res["status"] == 0
                for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                    assert np.array_equal(got[key][first:first + n], res[key]), (rank, rec.id, key)
                rows = [np.ascontiguousarray(got[key][first:first + n]) for key in ("r1_base", "r1_qual", "r2_base", "r2_qual")]
                fastq_write(f1.fileno(), f2.fileno(), rec.id, 0, rank, n, dense.read_length, dense.read_length, *rows, 1)
                fastq_write(e1.fileno(), e2.fileno(), rec.id, 0, rank, n, dense.read_length, dense.read_length,
                            res["r1_base"], res["r1_qual"], res["r2_base"], res["r2_qual"], 1)
                first += n
        dist.barrier()
        if rank == 0:
            exp = [b"".join(open(D.temp_prefix(output, r) + "_exp" + sfx, "rb").read() for r in range(world))
                   for sfx in ("_R1.fastq", "_R2.fastq")]
            D.concatenate_rank_files(output, world)
            assert open(output + "_R1.fastq", "rb").read() == exp[0]
            assert open(output + "_R2.fastq", "rb").read() == exp[1]
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_through_the_device(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_multirank_dry_run(world):
    """bench.py's own N > 1 code -- torch.distributed.run rendezvous on 127.0.0.1, the process group, the one broadcast of
    tables + packed genomes, chunk r of the reference's divider per rank (iss/app.py:81-83), the barrier-bracketed timed
    region, the all_gather of every rank's (pairs, seconds), elapsed = max over ranks -- as a dry run on ONE GPU (gloo;
    ISS_BENCH_SHARE_GPU=1).  The driver's 8-GPU run differs in the backend name ("nccl" = RCCL) and the device per rank; world 8
    is that run's shape: eight chunks of both legs, a rank uploading only the records its chunk names."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reads, strong_reads = 300000, 1_200_000
    env = dict(os.environ, ISS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world),
                          "--backend", "gloo", "--reads", str(reads), "--strong-reads", str(strong_reads), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-end-to-end", "--no-other-workloads"], cwd=root, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["n_ranks_seen"] == world and d["scaling"] == "weak" and d["steps"] == 3
    pairs = d["config"]["pairs_per_step_per_gpu"]
    # (the divider rounds every record's share and drops a surplus chunk like the reference: iss/app.py:104)
    assert len(pairs) == world and abs(sum(pairs) - reads * world // 2) <= 64 and min(pairs) > 0 and max(pairs) <= -(-reads * world // 2 // world)
    assert len(d["per_rank_pairs_per_sec"]) == world and all(v > 0 for v in d["per_rank_pairs_per_sec"])
    assert abs(d["value"] - sum(pairs) * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]  # whole-job pairs / the slowest rank's time
    assert str(d["parity_window"]).startswith("ok")
    # the default N > 1 line carries BOTH legs: configs[2] per GPU (weak) and configs[3] in total (strong), each with every rank's
    # share and own rate, the broadcast time, the ranks seen and a scaling efficiency against ONE GPU on the N = 1 shape
    weak, strong = d["weak_leg"], d["strong_leg"]
    assert weak["scaling"] == "weak" and weak["pairs_per_step_per_gpu"] == pairs and abs(weak["value"] - d["value"]) <= 1e-9 * d["value"]
    assert strong["scaling"] == "strong" and "configs[3]" in strong["workload"]
    sp = strong["pairs_per_step_per_gpu"]
    assert len(sp) == world and min(sp) > 0 and abs(sum(sp) - strong_reads // 2) <= 64  # (50 records: rounding + the dropped surplus chunk)
    assert abs(strong["one_gpu_whole_job"]["pairs_per_step"] - strong_reads // 2) <= 64
    for leg in (weak, strong):
        assert leg["n_ranks_seen"] == world and len(leg["per_rank_pairs_per_sec"]) == world and all(v > 0 for v in leg["per_rank_pairs_per_sec"])
        assert leg["model_broadcast_s"] > 0 and 0 < leg["scaling_efficiency"] < 2.0 and str(leg["parity_window"]).startswith("ok")
        # ONE payload: the tables (< 2 MB) + the records as 2-bit codes (a quarter of a byte per base, 16-byte aligned)
        n_g = leg["n_genomes"]
        assert leg["model_broadcast"]["packed_genomes"] == n_g and leg["model_broadcast_bytes"] == leg["model_broadcast"]["payload_bytes"]
        assert n_g * 1_250_000 <= leg["model_broadcast"]["genome_bytes"] <= n_g * 1_250_016 and 0 < leg["model_broadcast"]["model_bytes"] < 4 << 20
        # a rank uploads the records of ITS chunk only (contiguous in record order: neighbours share at most one record)
        up = leg["genomes_uploaded_per_rank"]
        assert len(up) == world and min(up) >= 1 and n_g - 3 <= sum(up) <= n_g + world - 1  # (a record whose share rounds to 0 pairs is nobody's)
        assert abs(leg["value"] - sum(leg["pairs_per_step_per_gpu"]) * leg["steps"] / (leg["ms_per_step"] * leg["steps"] * 1e-3)) <= 1e-6 * leg["value"]
    assert d["scaling_efficiency"] == weak["scaling_efficiency"]


def test_bench_refuses_a_mismatched_launch():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)


def test_bench_rccl_calls_with_one_rank():
    """Every RCCL call of bench.py's N > 1 path -- init_process_group("nccl", device_id=...), the barriers, the two broadcasts
    of CUDA tensors (size announcement, payload), uploads straight from the received device buffer, the all_gather of the
    ranks' (pairs, seconds) -- executed for real on the one GPU of this box by a ONE-rank group (ISS_BENCH_FORCE_DIST=1
    under torch.distributed.run).  An 8-GPU run differs by the number of ranks."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1",
                          "--reads", "400000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end",
                          "--no-other-workloads"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["n_ranks_seen"] == 1 and d["n_gpus"] == 1
    assert d["model_broadcast_s"] > 0 and str(d["parity_window"]).startswith("ok")
    assert len(d["per_rank_pairs_per_sec"]) == 1 and d["per_rank_pairs_per_sec"][0] > 0


def _rccl_one_rank_main(rank, port):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist

    from insilicoseq_amd.distributed import broadcast_model_and_genomes
    from insilicoseq_amd.engine import ReadEngine
    from insilicoseq_amd.model import DenseModel

    dense = DenseModel.load(os.path.join(root, "insilicoseq_amd", "profiles", "novaseq.dense.npz"))
    rng = np.random.RandomState(5)
    genomes = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, size=n)] for n in (30000, 4097)]
    genomes.append(np.frombuffer(b"ACGTNacgtRY", dtype=np.uint8)[rng.randint(0, 11, size=5000)])  # (travels as ASCII)
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got_dense, refs = broadcast_model_and_genomes(dense, genomes, dist, device=torch.device("cuda", 0), as_refs=True, force=True)
        for k in DenseModel.FIELDS:
            assert np.array_equal(getattr(got_dense, k), getattr(dense, k)), k
        assert [r.length for r in refs] == [g.size for g in genomes]
        outs = []
        for feed in ("buffer", "host"):
            with ReadEngine(0) as eng:
                eng.load_model(dense)
                gids = [r.upload(eng) for r in refs] if feed == "buffer" else [eng.add_genome(g) for g in genomes]
                eng.generate_batch(gids, [700, 300, 200], first_ordinal=0, seed=3)
                outs.append(eng.download(0, 1200))
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), k
    finally:
        dist.destroy_process_group()


def test_rccl_broadcast_payload_equals_the_local_one():
    """The same through the library call alone: a one-rank RCCL group (in a process of its own), force=True -- the tables
    and the 2-bit genomes that come back from the device buffer equal what went in, and an engine fed from the buffer
    generates the reads of one fed from the host arrays."""
    import torch.multiprocessing as mp

    mp.spawn(_rccl_one_rank_main, args=(_free_port(),), nprocs=1, join=True)
