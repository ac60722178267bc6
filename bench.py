#!/usr/bin/env python3
"""bench.py -- read-pairs/s of the MI355X read-generation path on BASELINE.json's headline config.

A "step" = one pass of the hot path over one batch of synthetic input: 10 M reads (5 M pairs) of
the NovaSeq KDE model (read_length 151) over 5 synthetic 5 Mbp genomes with log-normal abundances
(BASELINE.json configs[2]; SURVEY.md 8d "cfg 3").  Genomes, model tables and work list are resident
in HBM before the timed region; outputs stay in HBM (R1/R2 base + phred buffers).

N > 1 (one process per GPU, launched by torch.distributed.run): weak scaling -- every rank is one
reference worker (cpu_number = rank, worker seed = seed + rank) generating its own 5 M pairs;
the only collective is ONE RCCL broadcast of the model tables + packed genomes from rank 0 before
the timed region (the path itself has no exchange step).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READS_PER_STEP = 10_000_000
N_GENOMES = 5
GENOME_LEN = 5_000_000
SEED = 42
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def synthetic_genomes(n, length, seed):
    rng = np.random.RandomState(seed)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [letters[rng.randint(0, 4, size=length)] for _ in range(n)]


def algorithmic_bytes_per_pair(read_length):
    # 4*RL output bytes (R1+R2 bases and phreds) + two RL-base windows of the 2-bit genome
    return 4 * read_length + 2 * ((read_length + 3) // 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=READS_PER_STEP, help="reads per step per GPU")
    ap.add_argument("--model", default="novaseq")
    ap.add_argument("--n-genomes", type=int, default=N_GENOMES, help="records of the synthetic community (BASELINE configs[3]: 50)")
    ap.add_argument("--indel", type=float, nargs=2, default=None, metavar=("P_INS", "P_DEL"),
                    help="override every insertion / deletion probability (BASELINE configs[4]-like indel-heavy model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the FASTQ-on-tmpfs leg (rank 0, N = 1 only)")
    ap.add_argument("--e2e-pairs", type=int, default=20_000_000)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo: a dry run of "
                    "the multi-rank logic, e.g. with ISS_BENCH_SHARE_GPU=1 on a single-GPU box)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=1_500_000)
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU oracle as one process per host core "
                    "(<= 64; off by default: on a box with hundreds of cores the start-up alone takes a minute)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if os.environ.get("ISS_BENCH_SHARE_GPU") == "1":  # dry runs only: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()

    from insilicoseq_amd.engine import ReadEngine
    from insilicoseq_amd.generator import Record, generate_work_divider, lognormal_abundance
    from insilicoseq_amd.model import DenseModel

    # ---- inputs: rank 0 builds them; ONE RCCL broadcast to the other ranks (only when N > 1)
    from insilicoseq_amd.distributed import broadcast_model_and_genomes

    model_path = os.path.join(ROOT, "insilicoseq_amd", "profiles", args.model + ".dense.npz")
    dense, genomes = None, None
    if rank == 0:
        dense = DenseModel.load(model_path)
        if args.indel is not None:
            dense.ins[:] = args.indel[0]
            dense.dele[:] = args.indel[1]
        genomes = synthetic_genomes(args.n_genomes, GENOME_LEN, 123)
    t_b = time.time()
    dense, genomes = broadcast_model_and_genomes(dense, genomes, dist, device=torch.device("cuda", local_rank))
    torch.cuda.synchronize()
    bcast_s = time.time() - t_b if dist is not None else 0.0

    records = [Record(g, id="genome_%d" % i) for i, g in enumerate(genomes)]
    abundance = lognormal_abundance([r.id for r in records], np.random.RandomState(123))
    n_pairs_step = args.reads // 2
    work = []
    for chunk in generate_work_divider(records, None, abundance, args.reads, None, None, dense, "bench",
                                       chunk_size=n_pairs_step):
        work.extend(chunk)
    work = [(r, n) for r, n, _ in work]
    total_pairs_step = sum(n for _, n in work)

    eng = ReadEngine(local_rank)
    eng.load_model(dense)
    gids = {id(r): eng.add_genome(r.seq) for r in records}
    eng.reserve(total_pairs_step)
    worker_seed = SEED + rank
    ordinal = [0]

    item_ids = [gids[id(rec)] for rec, _ in work]
    item_pairs = [n for _, n in work]

    def step():  # the step's whole work list in one set of launches (iss_generate_batch)
        eng.generate_batch(item_ids, item_pairs, first_ordinal=ordinal[0], seed=worker_seed, out_first_pair=0)
        ordinal[0] += total_pairs_step

    def sync_all():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Warm-up: HIP events around every kernel (the per-kernel split reported below).  Timed region: events around
    # k_main only -- the roofline's kernel duration is measured live over the timed region, but every event is a
    # bubble in the stream and the five-kernel timing costs ~6 % of a step.
    eng.timing_read()
    eng.timing_enable(1)
    for _ in range(args.warmup):
        step()
    sync_all()
    tm_warm = eng.timing_read()
    eng.timing_enable(0 if os.environ.get("ISS_BENCH_NO_KERNEL_EVENTS") == "1" else 2)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    tm = eng.timing_read()
    eng.timing_enable(0)
    stats = eng.stats_read()
    if dist is not None:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    if rank == 0:
        RL = dense.read_length
        pairs_total = total_pairs_step * args.steps * world
        value = pairs_total / elapsed
        b_pair = algorithmic_bytes_per_pair(RL)
        main_s = tm["main_ms"] / 1e3
        n_main_launches = args.steps  # one k_main launch per step (all work items)
        achieved = (total_pairs_step * args.steps * b_pair) / main_s / 1e9 if main_s > 0 else 0.0
        # the other kernels' milliseconds come from the warm-up steps (per step)
        other = {k: (tm_warm[k] / args.warmup if args.warmup else None) for k in ("setup_ms", "indel_scan_ms", "indel_fixup_ms")}
        all_kernels_s = (tm["main_ms"] + sum(v or 0.0 for v in other.values()) * args.steps) / 1e3
        traffic, traffic_note = committed_traffic()
        out = {
            "metric": "read_pairs_per_sec", "value": value, "unit": "read-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: %d reads/step/GPU, %s KDE model (read_length %d), %d x %d bp "
                            "uniform ACGT genomes, log-normal abundance, seed %d; outputs left in HBM" % (
                                args.reads, args.model, RL, N_GENOMES, GENOME_LEN, SEED),
                "pairs_per_step_per_gpu": total_pairs_step, "read_length": RL, "work_items": len(work),
                "rng": "philox4x32-10", "parallelism": "1 worker/GPU, no data-path collective",
                "indel_override": args.indel,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_main", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_note": traffic_note, "algorithmic_bytes_per_launch": b_pair * total_pairs_step,
                "algorithmic_bytes_per_pair": b_pair, "avg_launch_ms": tm["main_ms"] / max(n_main_launches, 1),
                "launches": n_main_launches,
                # what actually bounds k_main: integer VALU issue.  Wavefront-level VALU instructions per launch from the
                # committed PMC pass; 4 cycles each on one of 1024 SIMDs (256 CUs x 4) at 2.4 GHz
                "valu": {"insts_per_launch": getattr(committed_traffic, "valu_insts", None),
                         "issue_busy_frac_est": (getattr(committed_traffic, "valu_insts", None) or 0) * 4.0 / (
                             1024 * 2.4e9 * (tm["main_ms"] / max(n_main_launches, 1)) / 1e3) if main_s > 0 else None,
                         "note": "excludes the second issue cycle pair of v_mad_u64_u32 (Philox), ~12 % more"},
            },
            "kernel_ms_per_step": dict(other, main_ms=tm["main_ms"] / args.steps,
                                       note="main_ms: HIP events over the timed region; the others: over the warm-up steps"),
            "all_kernels_GBps": (total_pairs_step * args.steps * b_pair) / all_kernels_s / 1e9 if all_kernels_s else 0,
            "indel_fixup_reads_per_step": stats["fixup_reads"] / max(args.steps, 1),
            "model_broadcast_s": bcast_s,
        }
        if world == 1 and not args.no_end_to_end:
            out["end_to_end"] = end_to_end(dense, records, abundance, args.e2e_pairs)
            out["end_to_end_gzip"] = end_to_end(dense, records, abundance, args.e2e_pairs, compress=True)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dense, work, args.cpu_sample_pairs)
            if args.cpu_all_cores and args.indel is None:  # opt-in: one oracle process per host core
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(model_path, args.n_genomes, work, args.cpu_sample_pairs // 6)
                except Exception as e:
                    out["cpu_baseline_all_cores"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def committed_traffic():
    """HBM-side bytes per k_main launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
    runs of this same command, corrected with the calibration kernels as MI355X_MICROARCH.md prescribes).  PMC
    collection cannot run inside the timed region, so the latest committed measurement is reported."""
    import glob

    import re

    def natural(path):  # r01_v10 sorts after r01_v9
        return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(path))]

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=natural)
    if not files:
        return None, "no PMC pass committed"
    with open(files[-1]) as fh:
        t = json.load(fh)
    committed_traffic.valu_insts = t.get("valu_insts_per_launch")
    return t["traffic_bytes_per_launch"], "bytes per launch (%d pairs), from %s" % (
        t.get("pairs_per_launch_avg", 0), os.path.basename(files[-1]))


def end_to_end(dense, records, abundance, n_pairs, compress=False):
    """SURVEY.md 8d "report both": one worker from genomes in HBM to FASTQ FILES (worker_iterator: generation, FASTQ text
    built on the device, copy, pwrite) on tmpfs.  Informational: `value` above stays the kernel-side rate.
    compress: `--compress`, the text is deflated on the device and only gzip members are copied and written."""
    import shutil
    import tempfile

    from insilicoseq_amd.generator import worker_iterator

    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(dir=base)
    try:
        # keep well inside the free space of the file system (2 files x ~330 B per pair as text)
        need = 2 * 340 * n_pairs * (0.4 if compress else 1.0)
        free = shutil.disk_usage(d).free
        if need > 0.5 * free:
            n_pairs = max(1000, int(n_pairs * 0.5 * free / need))
        work = [(r, int(n_pairs * abundance[r.id]), "default") for r in records]
        prefix = os.path.join(d, "w")
        worker_iterator([(records[0], 1000, "default")], dense, 0, prefix, SEED, "metagenomics", False, device=0,
                        compress=compress)  # warm-up
        t0 = time.perf_counter()
        worker_iterator(work, dense, 0, prefix, SEED, "metagenomics", False, device=0, compress=compress)
        dt = time.perf_counter() - t0
        size = os.path.getsize(prefix + "_R1.fastq") + os.path.getsize(prefix + "_R2.fastq")
        n = sum(k for _, k, _ in work)
        return {"value": n / dt, "unit": "read-pairs/s", "written_GB_per_s": size / dt / 1e9,
                "sample": "%d pairs -> %.2f GB of %s on %s in %.2f s (incl. engine start-up), one worker" % (
                    n, size / 1e9, "gzip members (text deflated on the device)" if compress else "FASTQ",
                    base or "the temp dir", dt)}
    except Exception as e:  # a leg of extra information must not take the benchmark line down
        return {"value": None, "error": repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(dense, work, sample_pairs):
    """The CPU oracle (oracle/iss_oracle.c, the reference's algorithm restated in C with the reference's
    two MT19937 streams) timed on ONE host core on a bounded sample of the same workload."""
    from oracle import oracle as O

    orc = O.Oracle(dense)
    rng = O.Rng().seed_mt(SEED)
    todo = sample_pairs
    t0 = time.perf_counter()
    done = 0
    for rec, n in work:
        k = min(n, max(1, int(round(sample_pairs * n / sum(x for _, x in work)))), todo)
        if k <= 0:
            continue
        res = orc.simulate(rng, rec.seq, k)
        assert res["status"] == 0
        done += res["n_done"]
        todo -= k
    dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "read-pairs/s", "cores": 1, "kind": "port",
        "sample": "%d pairs of the same work list (proportional per genome), MT19937 streams, %.1f s" % (done, dt),
        "note": "the Python reference itself measured 873 (1 process) / 3376 (8 processes) read-pairs/s "
                "end-to-end in the build container (BASELINE.md); it cannot run on the GPU box",
    }


def _cpu_worker_main(argv):
    """`python bench.py --cpu-worker <model> <n_genomes> <cpu> <pairs per genome ...>`: one process of the all-cores CPU
    leg -- the oracle on its share of the sample with its own pair of MT19937 streams (seed + cpu_number, like a reference
    worker).  Prints `pairs seconds`."""
    from insilicoseq_amd.model import DenseModel
    from oracle import oracle as O

    model_path, n_genomes, cpu = argv[0], int(argv[1]), int(argv[2])
    shares = [int(x) for x in argv[3:]]
    dense = DenseModel.load(model_path)
    genomes = synthetic_genomes(n_genomes, GENOME_LEN, 123)
    orc = O.Oracle(dense)
    rng = O.Rng().seed_mt(SEED + cpu)
    t0 = time.perf_counter()
    done = 0
    for g, k in zip(genomes, shares):
        if k > 0:
            res = orc.simulate(rng, g, k)
            assert res["status"] == 0
            done += res["n_done"]
    print("%d %.6f" % (done, time.perf_counter() - t0), flush=True)


def cpu_baseline_all_cores(model_path, n_genomes, work, pairs_per_core, limit_s=60.0):
    """The same CPU oracle on every host core at once (one process per core, each a reference-style worker with its own
    streams): the box-level CPU rate the GPU number stands beside.  Bounded: pairs_per_core pairs per process, and the
    whole leg is abandoned (its processes killed by pid) after limit_s seconds."""
    import subprocess

    cores = max(1, min(os.cpu_count() or 1, 64))
    total = sum(n for _, n in work)
    shares = [str(max(1, int(round(pairs_per_core * n / total)))) for _, n in work]
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", model_path, str(n_genomes), str(c)] + shares,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT) for c in range(cores)]
    res = []
    try:
        for p in procs:
            left = max(1.0, limit_s - (time.perf_counter() - t0))
            out, _ = p.communicate(timeout=left)
            if p.returncode != 0:
                raise RuntimeError("a CPU worker failed")
            a, b = out.decode().split()
            res.append((int(a), float(b)))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": done / busy, "unit": "read-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pairs on %d processes (%d each), slowest process %.1f s, %.1f s with start-up" % (
                done, cores, sum(int(x) for x in shares), busy, wall)}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        _cpu_worker_main(sys.argv[2:])
        sys.exit(0)
    main()
