#!/usr/bin/env python3
"""bench.py -- read-pairs/s of the MI355X read-generation path on BASELINE.json's headline config.

A "step" = one pass of the hot path over one batch of synthetic input: 10 M reads (5 M pairs) per GPU of
the NovaSeq KDE model (read_length 151) over 5 synthetic 5 Mbp genomes with log-normal abundances
(BASELINE.json configs[2]; SURVEY.md 8d "cfg 3").  Genomes, model tables and work list are resident
in HBM before the timed region; outputs stay in HBM (R1/R2 base + phred rows).

N > 1 (one process per GPU, launched by torch.distributed.run): the community run of 10 M x N reads is sharded the
reference's way -- the flattened (record, pairs) list cut into N contiguous chunks of ceil(pairs / N) (iss/app.py:81-83),
rank r = worker cpu_number r (worker seed = seed + r, read ids ..._r) -- so the per-GPU work is fixed (weak scaling)
and N = 1 is the same code path.  The only collective is ONE RCCL broadcast of the model tables + 2-bit packed genomes
from rank 0 before the timed region (the path itself has no exchange step); the ranks upload their genomes straight
from the received device buffer.  `--workload configs3`: BASELINE configs[3] instead -- 100 M HiSeq reads per step over
50 x 5 Mbp genomes, the TOTAL fixed and sharded over the N ranks (strong scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline` and `parity_window`
(windows of the last timed step recomputed by the CPU oracle after the timed region).
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


class _SeqLen(object):
    """What the work divider needs of a record's letters: how many there are."""

    def __init__(self, n):
        self.n = int(n)

    def __len__(self):
        return self.n


def algorithmic_bytes_per_pair(read_length):
    # 4*RL output bytes (R1+R2 bases and phreds) + two RL-base windows of the 2-bit genome
    return 4 * read_length + 2 * ((read_length + 3) // 4)


def run_leg(args, label, model, reads, n_genomes, strong, indel, steps, warmup, rank, local_rank, world, dist, force_dist):
    """One leg of the bench on this rank: rank 0 builds model + genomes, ONE broadcast to the other ranks (only when a process
    group exists), every rank takes chunk `rank` of the reference's divider (iss/app.py:81-83, 99-106), `warmup` untimed and
    `steps` timed steps bracketed by barriers; the ranks' (pairs, seconds) are gathered.  Returns everything the line needs."""
    import torch

    from insilicoseq_amd.distributed import broadcast_model_and_genomes, rank_work
    from insilicoseq_amd.engine import ReadEngine
    from insilicoseq_amd.generator import Record, lognormal_abundance
    from insilicoseq_amd.model import DenseModel

    dense, genomes = None, None
    if rank == 0:
        dense = DenseModel.load(os.path.join(ROOT, "insilicoseq_amd", "profiles", model + ".dense.npz"))
        if indel is not None:
            dense.ins[:] = indel[0]
            dense.dele[:] = indel[1]
        genomes = synthetic_genomes(n_genomes, GENOME_LEN, 123)
    t_b = time.time()
    binfo = {}
    dense, grefs = broadcast_model_and_genomes(dense, genomes, dist, device=torch.device("cuda", local_rank), as_refs=True,
                                               force=force_dist, info=binfo)
    torch.cuda.synchronize()
    bcast_s = time.time() - t_b if dist is not None else 0.0

    eng = ReadEngine(local_rank)
    eng.load_model(dense)
    # records: id + length for the work divider, letters only where the CPU legs need them (rank 0)
    records = [Record(_SeqLen(g.length), id="genome_%d" % i) for i, g in enumerate(grefs)]
    letters = {id(r): g for r, g in zip(records, genomes)} if rank == 0 else {}
    gref_of = {id(r): g for r, g in zip(records, grefs)}
    abundance = lognormal_abundance([r.id for r in records], np.random.RandomState(123))
    total_reads = reads if strong else reads * world

    def work_of(n_ranks, r):  # the reference's sharding: chunk r of the divider with cpus = n_ranks
        chunk, _, _ = rank_work(records, None, abundance, total_reads, None, None, dense, "bench", n_ranks, r)
        return [(rec, n) for rec, n, _ in (chunk or [])]

    work = work_of(world, rank)
    total_pairs_step = sum(n for _, n in work)
    eng.reserve(max(total_pairs_step, 1))
    worker_seed = SEED + rank
    ordinal = [0]
    # A rank uploads the records its chunk names and no others (the reference's workers receive only their own
    # chunk's records too: iss/app.py:99-106) -- N > 1: straight from the broadcast buffer in HBM (2-bit codes; the ASCII copy
    # is expanded on the device).  configs[3] on 8 ranks: ~7 of 50 records, 35 MB instead of 250 MB of ASCII per rank.
    gid_of = {}

    def gid(rec):
        if id(rec) not in gid_of:
            gid_of[id(rec)] = gref_of[id(rec)].upload(eng)
        return gid_of[id(rec)]

    item_ids = [gid(rec) for rec, _ in work]
    item_pairs = [n for _, n in work]
    genomes_uploaded = len(gid_of)

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
    for _ in range(warmup):
        step()
    sync_all()
    tm_warm = eng.timing_read()
    eng.timing_enable(0 if os.environ.get("ISS_BENCH_NO_KERNEL_EVENTS") == "1" else 2)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    tm = eng.timing_read()
    eng.timing_enable(0)
    stats = eng.stats_read()
    stats["main_kernel"] = eng.main_kernel()
    per_rank = [(total_pairs_step, elapsed)]
    if dist is not None:
        mine = torch.tensor([float(total_pairs_step), elapsed], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [(int(x[0].item()), float(x[1].item())) for x in allr]
    elapsed_max = max(e for _, e in per_rank)  # the slowest rank's time
    uploaded = [genomes_uploaded]
    if dist is not None:
        mine = torch.tensor([float(genomes_uploaded), bcast_s], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        uploaded = [int(x[0].item()) for x in allr]
        bcast_s = max(float(x[1].item()) for x in allr)  # the broadcast is over when the slowest rank holds the payload

    # ---- after the timed region: windows of the LAST timed step recomputed by the CPU oracle (rank 0's rows)
    parity = None
    if rank == 0 and total_pairs_step:
        parity = parity_window(eng, dense, work, letters, ordinal[0] - total_pairs_step, worker_seed)

    # ---- what ONE GPU does on the whole job (N > 1, strong leg): rank 0 alone, the other ranks wait at the barrier behind it.
    #      The weak leg needs no such run: a rank's own step IS the N = 1 shape.
    one_gpu = None
    if strong and world > 1:
        if rank == 0:
            w1 = work_of(1, 0)
            ids1, pairs1 = [gid(rec) for rec, _ in w1], [n for _, n in w1]  # (rank 0 now holds every record of the job)
            tot1 = sum(pairs1)
            eng.reserve(max(tot1, 1))
            eng.generate_batch(ids1, pairs1, first_ordinal=0, seed=SEED, out_first_pair=0)
            eng.synchronize()
            k = max(2, min(steps, 5))
            t0 = time.perf_counter()
            for i in range(k):
                eng.generate_batch(ids1, pairs1, first_ordinal=(i + 1) * tot1, seed=SEED, out_first_pair=0)
            eng.synchronize()
            one_gpu = {"value": tot1 * k / (time.perf_counter() - t0), "pairs_per_step": tot1, "steps": k}
        dist.barrier()
    return dict(label=label, model=model, reads=reads, n_genomes=n_genomes, strong=strong, indel=indel, steps=steps, warmup=warmup,
                dense=dense, genomes=genomes, records=records, letters=letters, abundance=abundance, work=work, eng=eng,
                total_pairs_step=total_pairs_step, elapsed=elapsed_max, per_rank=per_rank, tm=tm, tm_warm=tm_warm, stats=stats,
                bcast_s=bcast_s, parity=parity, one_gpu=one_gpu, bcast_info=binfo, genomes_uploaded=uploaded)


def leg_summary(L, world):
    """The scaling view of a leg (N > 1 lines carry one per leg): whole-job rate, every rank's share and own rate, and the
    scaling efficiency value_N / (N x value of ONE GPU on the N = 1 shape) -- weak: a rank's own step is that shape, so the
    denominator is the sum of the ranks' own rates; strong: rank 0 ran the whole job alone (`one_gpu`)."""
    pairs = [n for n, _ in L["per_rank"]]
    own = [n * L["steps"] / e if e > 0 else None for n, e in L["per_rank"]]
    value = sum(pairs) * L["steps"] / L["elapsed"]
    if L["strong"]:
        base = world * L["one_gpu"]["value"] if L["one_gpu"] else None
    else:
        base = sum(v for v in own if v)
    return {"workload": L["label"], "scaling": "strong" if L["strong"] else "weak", "value": value, "unit": "read-pairs/s",
            "ms_per_step": L["elapsed"] / L["steps"] * 1e3, "steps": L["steps"], "pairs_per_step_per_gpu": pairs,
            "per_rank_pairs_per_sec": own, "model_broadcast_s": L["bcast_s"], "model_broadcast_bytes": L["bcast_info"].get("payload_bytes"),
            "model_broadcast": L["bcast_info"], "genomes_uploaded_per_rank": L["genomes_uploaded"], "n_genomes": L["n_genomes"],
            "n_ranks_seen": len(L["per_rank"]),
            "one_gpu_whole_job": L["one_gpu"], "scaling_efficiency": (value / base) if base else None,
            "parity_window": L["parity"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=["configs2", "configs3"],
                    help="configs2: BASELINE configs[2], 10 M NovaSeq reads per step PER GPU over 5 genomes (weak scaling); "
                         "configs3: BASELINE configs[3], 100 M HiSeq reads per step IN TOTAL over 50 genomes (strong scaling).  "
                         "Default: configs2 is the line's metric; with --gpus N > 1 the line ALSO carries the configs3 leg (`strong_leg`)")
    ap.add_argument("--reads", type=int, default=None, help="reads per step (configs2: per GPU; configs3: in total)")
    ap.add_argument("--strong-reads", type=int, default=100_000_000, help="total reads per step of the strong leg of a default N > 1 run")
    ap.add_argument("--model", default=None)
    ap.add_argument("--n-genomes", type=int, default=None, help="records of the synthetic community")
    ap.add_argument("--indel", type=float, nargs=2, default=None, metavar=("P_INS", "P_DEL"),
                    help="override every insertion / deletion probability (BASELINE configs[4]-like indel-heavy model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short legs on the other shipped models and the indel-heavy model (rank 0, N = 1, default workload only)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the FASTQ-on-tmpfs legs (rank 0, N = 1 only)")
    ap.add_argument("--e2e-pairs", type=int, default=200_000_000)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo: a dry run of "
                    "the multi-rank logic, e.g. with ISS_BENCH_SHARE_GPU=1 on a single-GPU box)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=1_500_000)
    ap.add_argument("--cpu-threads", type=int, default=None, help="threads of the all-cores CPU leg (default: min(host cores, 64))")
    args = ap.parse_args()
    both_legs = args.workload is None  # the default invocation
    strong = args.workload == "configs3"
    if args.reads is None:
        args.reads = 100_000_000 if strong else READS_PER_STEP
    if args.model is None:
        args.model = "hiseq" if strong else "novaseq"
    if args.n_genomes is None:
        args.n_genomes = 50 if strong else N_GENOMES

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it as `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...` (one rank per GPU)" % (
                             args.gpus, world, args.gpus, args.gpus))
    if os.environ.get("ISS_BENCH_SHARE_GPU") == "1":  # dry runs only: every rank on GPU 0
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) are visible (one rank per GPU; ISS_BENCH_SHARE_GPU=1 puts "
                         "every rank of a dry run on GPU 0)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    # ISS_BENCH_FORCE_DIST=1: the multi-rank code path (process group bound to the GPU, barriers, the broadcast, the
    # all_gather) with however many ranks there are -- ONE rank on a single-GPU box runs every RCCL call of an 8-GPU run
    force_dist = os.environ.get("ISS_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29511")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL needs dmabuf IPC on this driver)
        try:
            if args.backend == "nccl":  # RCCL: the process group is bound to this rank's GPU (barriers and collectives run there)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
        except Exception as e:
            raise SystemExit("rank %d: torch.distributed.init_process_group(%r) failed: %r (MASTER_ADDR=%s MASTER_PORT=%s)" % (
                rank, args.backend, e, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")))
        if dist.get_world_size() != world:
            raise SystemExit("rank %d: the process group has %d ranks, WORLD_SIZE says %d" % (rank, dist.get_world_size(), world))

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()

    from insilicoseq_amd.generator import Record, lognormal_abundance

    label = ("BASELINE configs[3]: %d reads/step in total" % args.reads) if strong else (
        "BASELINE configs[2]: %d reads/step/GPU" % args.reads)
    L = run_leg(args, label, args.model, args.reads, args.n_genomes, strong, args.indel, args.steps, args.warmup, rank, local_rank,
                world, dist, force_dist)
    # N > 1, default invocation: the line ALSO carries BASELINE configs[3] -- 100 M HiSeq reads per step in TOTAL over 50
    # records, chunk r of the divider per rank (strong scaling); a weak curve alone says nothing about a path without a
    # data-path collective
    L3 = None
    if both_legs and dist is not None and world > 1:
        L["eng"].close()
        L["eng"] = None
        L3 = run_leg(args, "BASELINE configs[3]: %d reads/step in total" % args.strong_reads, "hiseq", args.strong_reads, 50, True,
                     None, max(3, min(args.steps, 10)), min(args.warmup, 2), rank, local_rank, world, dist, force_dist)
        L3["eng"].close()
        L3["eng"] = None
    eng, dense, work, letters, records, genomes, abundance = (L[k] for k in ("eng", "dense", "work", "letters", "records", "genomes", "abundance"))
    total_pairs_step, elapsed, per_rank, tm, tm_warm, stats = (L[k] for k in ("total_pairs_step", "elapsed", "per_rank", "tm", "tm_warm", "stats"))
    parity, bcast_s = L["parity"], L["bcast_s"]

    if rank == 0:
        RL = dense.read_length
        pairs_total = sum(n for n, _ in per_rank) * args.steps
        value = pairs_total / elapsed
        b_pair = algorithmic_bytes_per_pair(RL)
        main_s = tm["main_ms"] / 1e3
        n_main_launches = args.steps  # one k_main launch per step (all work items)
        achieved = (total_pairs_step * args.steps * b_pair) / main_s / 1e9 if main_s > 0 else 0.0
        # the other kernels' milliseconds come from the warm-up steps (per step)
        other = {k: (tm_warm[k] / args.warmup if args.warmup else None) for k in ("setup_ms", "indel_scan_ms", "indel_fixup_ms")}
        all_kernels_s = (tm["main_ms"] + sum(v or 0.0 for v in other.values()) * args.steps) / 1e3
        traffic = committed_traffic()
        out = {
            "metric": "read_pairs_per_sec", "value": value, "unit": "read-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "%s, %s KDE model (read_length %d), %d x %d bp uniform ACGT genomes, log-normal abundance, "
                            "seed %d; sharded by the reference's chunk rule (rank = cpu_number); outputs left in HBM" % (
                                label, args.model, RL, args.n_genomes, GENOME_LEN, SEED),
                "pairs_per_step_per_gpu": [n for n, _ in per_rank] if world > 1 else total_pairs_step, "read_length": RL,
                "work_items": len(work), "rng": "philox4x32 (7 rounds for the hot digit blocks K_QM, 10 for every other draw)",
                "parallelism": "1 worker/GPU, chunk r of ceil(pairs/N) per rank (iss/app.py:81-83), no data-path collective",
                "indel_override": args.indel,
            },
            "roofline": {
                "bound": "hbm", "kernel": stats.get("main_kernel") or "k_main", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic.get("traffic_bytes_per_launch"),
                "traffic_note": traffic.get("note"), "traffic_fresh": traffic.get("fresh"),
                "algorithmic_bytes_per_launch": b_pair * total_pairs_step,
                "algorithmic_bytes_per_pair": b_pair, "avg_launch_ms": tm["main_ms"] / max(n_main_launches, 1),
                "launches": n_main_launches,
                # what actually bounds k_main: integer VALU issue.  Wavefront-level VALU instructions per launch from the
                # committed PMC pass; measured issue cost 1.8 ns per wavefront instruction and SIMD (tools/instr_rate.hip:
                # 4 cycles), 1024 SIMDs
                "pmc": committed_pmc("a"),
                "valu": {"insts_per_launch": traffic.get("valu_insts_per_launch"),
                         "issue_busy_frac_est": (traffic.get("valu_insts_per_launch") or 0) * 1.8e-9 / 1024 / (
                             main_s / max(n_main_launches, 1)) if main_s > 0 else None},
            },
            "kernel_ms_per_step": dict(other, main_ms=tm["main_ms"] / args.steps,
                                       note="main_ms: HIP events over the timed region; the others: over the warm-up steps "
                                            "(indel_scan_ms: k_indel_scan + the two k_indel_script launches)"),
            "all_kernels_GBps": (total_pairs_step * args.steps * b_pair) / all_kernels_s / 1e9 if all_kernels_s else 0,
            "indel_fixup_reads_per_step": stats["fixup_reads"] / max(args.steps + args.warmup, 1),
            "indel_scripted_reads_per_step": stats["scripted_reads"] / max(args.steps + args.warmup, 1),
            "model_broadcast_s": bcast_s, "model_broadcast_bytes": L["bcast_info"].get("payload_bytes"),
            "genomes_uploaded_per_rank": L["genomes_uploaded"],
            "parity_window": parity,
            "library_build_id": library_build_id(),
        }
        out["n_ranks_seen"] = dist.get_world_size() if dist is not None else 1
        out["backend"] = (args.backend if dist is not None else None)
        if dist is not None:
            out["per_rank_pairs_per_sec"] = [n * args.steps / e if e > 0 else None for n, e in per_rank]
            leg = leg_summary(L, world)
            out["weak_leg" if not strong else "strong_leg"] = leg
            out["scaling_efficiency"] = leg["scaling_efficiency"]
            if L3 is not None:
                out["strong_leg"] = leg_summary(L3, world)
        if world == 1 and not args.no_other_workloads and not strong and args.model == "novaseq" and args.indel is None:
            # the same work list on the other models of the parity suite (a few steps each; BASELINE's metric stays the line above)
            out["other_workloads"] = {}
            for name, model, indel in (("indel_heavy", "novaseq", (0.001, 0.003)), ("hiseq", "hiseq", None),
                                       ("nextseq", "nextseq", None), ("miseq", "miseq", None)):
                try:
                    out["other_workloads"][name] = side_workload(local_rank, model, indel, genomes, records, abundance, args.reads)
                    out["other_workloads"][name]["pmc"] = committed_pmc("indel" if indel else model)
                except Exception as e:  # (never let a side leg take the line down)
                    out["other_workloads"][name] = {"error": repr(e)}
            try:  # BASELINE configs[3] at its real shape on ONE GPU: 100 M HiSeq reads per step over 50 records of 5 Mbp
                g50 = synthetic_genomes(50, GENOME_LEN, 123)
                r50 = [Record(_SeqLen(len(g)), id="genome_%d" % i) for i, g in enumerate(g50)]
                a50 = lognormal_abundance([r.id for r in r50], np.random.RandomState(123))
                out["other_workloads"]["configs3_one_gpu"] = side_workload(local_rank, "hiseq", None, g50, r50, a50, 100_000_000, steps=3, warmup=1)
                del g50
            except Exception as e:
                out["other_workloads"]["configs3_one_gpu"] = {"error": repr(e)}
            try:  # the reference-identical mode (rng="mt"): the one whose FASTQ equals `iss generate`'s byte for byte
                out["other_workloads"]["mt_mode"] = mt_mode_leg(local_rank, dense, genomes[0])
            except Exception as e:
                out["other_workloads"]["mt_mode"] = {"error": repr(e)}
        if world == 1 and not args.no_end_to_end:
            eng.close()  # (the legs below bring their own engines: the rows of the timed region go back to the allocator first)
            eng = None
            e2e_records = [Record(letters[id(r)], id=r.id) for r in records]
            if isinstance(out.get("other_workloads", {}).get("mt_mode"), dict) and "error" not in out["other_workloads"]["mt_mode"]:
                out["other_workloads"]["mt_mode"]["end_to_end"] = mt_end_to_end(letters[id(records[0])])
            out["end_to_end"] = end_to_end(dense, e2e_records, abundance, args.e2e_pairs)
            out["end_to_end_gzip"] = end_to_end(dense, e2e_records, abundance, args.e2e_pairs, compress=True)
            out["end_to_end_4_workers"] = end_to_end(dense, e2e_records, abundance, args.e2e_pairs, workers=4)
        if world == 1 and not args.no_cpu_baseline:
            cpu_work = [(letters[id(r)], n) for r, n in work]
            out["cpu_baseline"] = cpu_baseline(dense, cpu_work, args.cpu_sample_pairs)
            try:  # the same oracle on all host cores (threads: the C call releases the GIL), ~5 s
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(dense, cpu_work, args.cpu_threads, out["cpu_baseline"]["value"])
            except Exception as e:
                out["cpu_baseline_all_cores"] = {"value": None, "error": repr(e)}
        # Two lines: everything (per-leg counter summaries, samples, notes) behind "#detail ", then -- LAST -- the line the
        # driver parses, short enough for the tail it keeps: the metric, its roofline and CPU baseline, and one
        # {value, ms_per_step, k_main's fraction of the HBM peak} per side leg.
        print("#detail " + json.dumps(out))
        print(json.dumps(headline(out)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if eng is not None:
        eng.close()


def _sig(v, digits=5):
    return float("%.*g" % (digits, v)) if isinstance(v, float) else v


def headline(out):
    """The line printed last: `out` without what only a reader of profiles needs -- short enough (< 2000 characters for the
    default N = 1 run) to survive whole in the tail the driver keeps.  legs: per side leg [read-pairs/s, ms per step, k_main's
    fraction of the HBM peak, kernel]."""
    drop = ("other_workloads", "end_to_end", "end_to_end_gzip", "end_to_end_4_workers", "cpu_baseline_all_cores", "all_kernels_GBps",
            "indel_fixup_reads_per_step", "indel_scripted_reads_per_step")
    if out.get("n_gpus") == 1 and out.get("backend") is None:  # (no process group: nothing to say about ranks and broadcasts)
        drop += ("model_broadcast_s", "model_broadcast_bytes", "genomes_uploaded_per_rank", "backend", "n_ranks_seen")
    h = {k: v for k, v in out.items() if k not in drop}
    h["config"] = {k: v for k, v in out["config"].items() if k in ("workload", "pairs_per_step_per_gpu", "read_length", "indel_override")}
    h["config"]["workload"] = h["config"]["workload"].split("; sharded")[0]
    h["roofline"] = {k: _sig(v, 6) for k, v in out["roofline"].items() if k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                                                  "traffic_fresh", "avg_launch_ms", "launches")}
    h["kernel_ms_per_step"] = {k: _sig(v, 4) for k, v in out["kernel_ms_per_step"].items() if k != "note"}
    h["parity_window"] = str(out.get("parity_window"))[:60]
    if "cpu_baseline" in out:
        h["cpu_baseline"] = {k: _sig(v) for k, v in out["cpu_baseline"].items() if k != "note"}
        h["cpu_baseline"]["sample"] = str(h["cpu_baseline"].get("sample", "")).replace(" of the same work list (proportional per genome)", "")
    legs = {}
    for name, leg in (out.get("other_workloads") or {}).items():
        if name == "mt_mode":
            if "error" in leg:
                legs["mt"] = {"error": leg["error"][:80]}
                continue
            legs["mt"] = {"one_worker": _sig(leg.get("value"), 4)}
            for w, ws in (leg.get("worker_sets") or {}).items():
                if w != "1":
                    legs["mt"]["workers_%s" % w] = _sig(ws.get("value"), 4)
            if leg.get("end_to_end"):
                legs["mt"]["end_to_end"] = {k: _sig(v, 4) for k, v in leg["end_to_end"].items() if k in ("value", "cpus", "pairs", "seconds", "error")}
        elif "error" in leg:
            legs[name] = {"error": leg["error"][:80]}
        else:
            legs[name] = [_sig(leg.get("value"), 4), _sig(leg.get("ms_per_step"), 4), _sig(leg.get("k_main_frac_of_hbm_peak"), 3), leg.get("kernel")]
    for name in ("end_to_end", "end_to_end_gzip", "end_to_end_4_workers"):
        if name in out:
            legs[name] = [_sig(out[name].get("value"), 4), "%s GB/s" % _sig(out[name].get("written_GB_per_s"), 3)]
    if "cpu_baseline_all_cores" in out:
        legs["cpu_all_cores"] = [_sig(out["cpu_baseline_all_cores"].get("value"), 4), "%s threads" % out["cpu_baseline_all_cores"].get("cores")]
    if legs:
        h["legs"] = legs
    return h


def parity_window(eng, dense, work, letters, step_first_ordinal, worker_seed, n=64):
    """Rows of the last timed step against the CPU oracle (Philox provider; every pair is a pure function of seed,
    ordinal and genome): the first n pairs of the step, n pairs straddling each of (up to) two work-item boundaries, and
    the last n pairs.  "ok" or a description of the first mismatch."""
    from oracle import oracle as O

    orc = O.Oracle(dense)
    firsts = np.concatenate(([0], np.cumsum([k for _, k in work]))).astype(np.int64)
    total = int(firsts[-1])
    starts = {0, max(0, total - n)}
    for b in firsts[1:-1][:2]:
        starts.add(int(max(0, b - n // 2)))
    checked = 0
    for s0 in sorted(starts):
        m = int(min(n, total - s0))
        got = eng.download(s0, m)
        i = s0
        while i < s0 + m:  # one oracle call per (window, work item)
            k = int(np.searchsorted(firsts, i, side="right") - 1)
            cnt = int(min(s0 + m, firsts[k + 1]) - i)
            res = orc.simulate(O.Rng().seed_philox(worker_seed), letters[id(work[k][0])], cnt,
                               first_ordinal=step_first_ordinal + i)
            if res["status"] != 0:
                return "oracle status %d at pair %d" % (res["status"], i)
            for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
                if not np.array_equal(got[key][i - s0:i - s0 + cnt], res[key]):
                    return "mismatch: pairs %d..%d of the step (item %d), %s" % (i, i + cnt - 1, k, key)
            checked += cnt
            i += cnt
    return "ok (%d pairs of the last timed step, item boundaries included, bit-identical to the CPU oracle)" % checked


def side_workload(device, model, indel, genomes, records, abundance, reads, steps=6, warmup=2):
    """bench.py's step on another model (own engine, the same genomes, work divider and seed): read-pairs/s over `steps`
    steps without kernel events, then the kernel split and k_main's roofline fraction over three steps with events, and
    the oracle check of windows of the last step."""
    from insilicoseq_amd.distributed import rank_work
    from insilicoseq_amd.engine import ReadEngine
    from insilicoseq_amd.model import DenseModel

    dense = DenseModel.load(os.path.join(ROOT, "insilicoseq_amd", "profiles", model + ".dense.npz"))
    if indel is not None:
        dense.ins[:] = indel[0]
        dense.dele[:] = indel[1]
    eng = ReadEngine(device)
    try:
        eng.load_model(dense)
        gids = [eng.add_genome(g) for g in genomes]
        chunk, _, _ = rank_work(records, None, abundance, reads, None, None, dense, "bench", 1, 0)
        work = [(r, n) for r, n, _ in (chunk or [])]
        gid_of = {id(r): g for r, g in zip(records, gids)}
        letters = {id(r): g for r, g in zip(records, genomes)}
        ids, pairs = [gid_of[id(r)] for r, _ in work], [n for _, n in work]
        total = sum(pairs)
        eng.reserve(max(total, 1))
        ordinal = [0]

        def step():
            eng.generate_batch(ids, pairs, first_ordinal=ordinal[0], seed=SEED, out_first_pair=0)
            ordinal[0] += total

        eng.timing_enable(0)
        for _ in range(warmup):
            step()
        eng.synchronize()
        eng.timing_read()
        eng.timing_enable(2)  # (events around k_main only, as in the main leg)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        eng.synchronize()
        elapsed = time.perf_counter() - t0
        main_ms = eng.timing_read()["main_ms"] / steps
        eng.timing_enable(1)
        for _ in range(3):
            step()
        eng.synchronize()
        tm = eng.timing_read()
        eng.timing_enable(0)
        parity = parity_window(eng, dense, work, letters, ordinal[0] - total, SEED)
        return {
            "value": total * steps / elapsed, "unit": "read-pairs/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "model": model, "read_length": dense.read_length, "indel_override": indel, "pairs_per_step": total,
            "kernel_ms_per_step": {"main_ms": main_ms, "setup_ms": tm["setup_ms"] / 3, "indel_scan_ms": tm["indel_scan_ms"] / 3,
                                   "indel_fixup_ms": tm["indel_fixup_ms"] / 3,
                                   "note": "main_ms: HIP events over the timed steps; the others: events around every kernel over three further steps"},
            "k_main_frac_of_hbm_peak": (total * algorithmic_bytes_per_pair(dense.read_length)) / (main_ms / 1e3) / 1e9 / HBM_PEAK_GBPS
            if main_ms > 0 else None,
            "kernel": eng.main_kernel(),
            "parity_window": parity,
        }
    finally:
        eng.close()


def library_build_id():
    """iss_build_id() of the library this process loaded: a hash of the sources it was COMPILED from (set by
    __graft_entry__.build()); "unknown" for a library built some other way."""
    from insilicoseq_amd import _native

    try:
        return _native.lib().iss_build_id().decode()
    except Exception:  # noqa: BLE001
        return "unknown"


def committed_traffic():
    """HBM-side bytes per k_main launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
    runs of this same command, corrected with the calibration kernels as MI355X_MICROARCH.md prescribes).  PMC
    collection cannot run inside the timed region, so the latest committed measurement is reported, with `fresh` =
    whether it was taken on the very library build this run has loaded (`library_build_id` of the summary against
    iss_build_id() of the loaded binary -- not a re-hash of whatever sources lie beside it)."""
    import glob

    import re

    def natural(path):  # r01_v10 sorts after r01_v9
        return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(path))]

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=natural)
    if not files:
        return {"note": "no PMC pass committed"}
    with open(files[-1]) as fh:
        t = json.load(fh)
    bid = library_build_id()
    t["fresh"] = bid != "unknown" and t.get("library_build_id") == bid
    t["note"] = "bytes per launch (%d pairs), from %s%s" % (
        t.get("pairs_per_launch_avg", 0), os.path.basename(files[-1]),
        "" if t["fresh"] else " -- STALE: measured on another build of the library (%s) than the one loaded (%s)" % (
            t.get("library_build_id", "?"), bid))
    return t


def committed_pmc(leg):
    """What the counters say about a side leg's k_main (tools/prof_model.sh -> tools/make_profile_summary.py ->
    profiles/<round>_<leg>_summary.json): the clock the chip ran the kernel at, VALU issue / wait / LDS-conflict fractions, HBM-side
    traffic against the algorithmic bytes.  PMC passes cannot run inside a timed region: the newest committed summary is reported,
    with `fresh` = taken on the very library build this run has loaded."""
    import glob
    import re

    def natural(path):
        return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(path))]

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_summary.json" % leg)), key=natural)
    if not files:
        return None
    with open(files[-1]) as fh:
        t = json.load(fh)
    keep = ("avg_launch_ms", "effective_clock_ghz", "valu_issue_busy_frac", "wait_any_frac", "wait_inst_any_frac", "lds_bank_conflict_ratio",
            "lds_active_frac", "traffic_over_algorithmic", "frac_of_hbm_peak", "k_main_launches_per_step")
    kern = {k.replace("void iss::", ""): {f: v.get(f) for f in keep if v.get(f) is not None} for k, v in t.get("kernels", {}).items()}
    return {"from": os.path.basename(files[-1]), "fresh": t.get("library_build_id") == library_build_id(), "k_main": kern}


def mt_mode_leg(device, dense, genome, n_pairs=1_000_000, worker_sets=(1, 8, 64, 256), budget_s=2.5):
    """Pairs per second in the reference-identical mode (rng="mt": the device consumes the reference's MT19937 streams in the
    reference's order; tests/test_gpu_mt_compat.py holds the byte-for-byte comparisons).  ONE worker through iss_generate_mt
    (the figure of the rounds before), then W workers side by side in one context (iss_generate_mt_workers: the reference's own
    `--cpus W` parallelism -- seeds seed + cpu_number, iss/generator.py:234-236 -- one workgroup per worker and kernel)."""
    from insilicoseq_amd.engine import ReadEngine

    eng = ReadEngine(device)
    try:
        eng.load_model(dense)
        gid = eng.add_genome(genome)
        eng.seed_mt(SEED)
        batch = 1 << 18
        eng.reserve(batch)
        assert eng.generate_mt(gid, batch) == batch  # warm-up
        eng.synchronize()
        t0 = time.perf_counter()
        done = 0
        while done < n_pairs:
            done += eng.generate_mt(gid, batch)
        eng.synchronize()
        dt = time.perf_counter() - t0
        out = {"value": done / dt, "unit": "read-pairs/s", "workers": 1,
               "sample": "%d pairs in %.2f s, rows left in HBM" % (done, dt), "worker_sets": {}}
        for W in worker_sets:
            try:
                per = max(2048, (1 << 21) // W)  # pairs per worker and call: every call is several turns of every worker
                eng.seed_mt_workers([SEED + w for w in range(W)])
                gids, ns, rows = [gid] * W, [per] * W, [w * per for w in range(W)]
                d, st = eng.generate_mt_workers(gids, ns, rows)  # warm-up (stream buffers, tables)
                assert int(d.sum()) == per * W and not st.any()
                eng.synchronize()
                t0 = time.perf_counter()
                total, calls = 0, 0
                while time.perf_counter() - t0 < budget_s or calls < 2:
                    d, st = eng.generate_mt_workers(gids, ns, rows)
                    total += int(d.sum())
                    calls += 1
                eng.synchronize()
                dt = time.perf_counter() - t0
                out["worker_sets"][str(W)] = {"workers": W, "value": total / dt, "per_worker": total / dt / W, "pairs": total,
                                              "seconds": dt, "calls": calls, "pairs_per_worker_and_call": per}
            except Exception as e:  # noqa: BLE001
                out["worker_sets"][str(W)] = {"workers": W, "error": repr(e)}
        best = max((v for v in out["worker_sets"].values() if v.get("value")), key=lambda v: v["value"], default=None)
        if best:
            out["best"] = {"workers": best["workers"], "value": best["value"]}
        out["path_counts"] = dict(zip(("resolved", "walked"), eng.mt_path_counts()))
        return out
    finally:
        eng.close()


def mt_end_to_end(genome, pairs=64_000_000, cpus=64):
    """The byte-identical mode END TO END: wall time of the whole command `python -m insilicoseq_amd generate --rng mt --cpus 64
    --devices 1` (interpreter start, FASTA parse, engine, the 64 reference workers side by side on one GPU, their text written at
    its place of the final files) on one 5 Mbp record, NovaSeq, FASTQ on tmpfs -- the files `iss generate --seed 7 --cpus 64`
    writes, byte for byte (tests/test_gpu_mt_compat.py holds the comparisons).  value = pairs / seconds of the command."""
    import shutil
    import subprocess
    import tempfile

    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(dir=base)
    try:
        need = 2 * 340 * pairs
        free = shutil.disk_usage(d).free
        if need > 0.4 * free:
            pairs = max(cpus * 64, int(pairs * 0.4 * free / need))
        fasta = os.path.join(d, "g.fasta")
        with open(fasta, "wb") as fh:
            fh.write(b">rec0\n" + (genome if isinstance(genome, bytes) else bytes(genome)) + b"\n")

        def run(n_pairs, tag):
            out = os.path.join(d, tag)
            t0 = time.perf_counter()
            subprocess.check_call([sys.executable, "-m", "insilicoseq_amd", "generate", "--genomes", fasta, "--model", "novaseq", "-n",
                                   str(2 * n_pairs), "--seed", "7", "--cpus", str(cpus), "--devices", "1", "--rng", "mt", "-o", out, "--quiet"],
                                  cwd=ROOT)
            dt = time.perf_counter() - t0
            size = sum(os.path.getsize(out + sfx) for sfx in ("_R1.fastq", "_R2.fastq"))
            for sfx in ("_R1.fastq", "_R2.fastq", "_abundance.txt"):
                os.remove(out + sfx)
            return dt, size

        t_small, _ = run(64 * cpus, "tiny")
        t_big, size = run(pairs, "big")
        return {"value": pairs / t_big, "unit": "read-pairs/s", "cpus": cpus, "pairs": pairs, "seconds": t_big, "startup_alone_s": t_small,
                "written_GB": size / 1e9,
                "sample": "the whole command `generate --rng mt --cpus %d --devices 1 -n %d` -> %.1f GB of FASTQ on %s in %.2f s "
                          "(the same command for %d pairs: %.2f s)" % (cpus, 2 * pairs, size / 1e9, base or "the temp dir", t_big, 64 * cpus, t_small)}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "error": repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def end_to_end(dense, records, abundance, n_pairs, compress=False, workers=1):
    """SURVEY.md 8d "report both": from genomes in HBM to FASTQ FILES (worker_iterator: generation, FASTQ text built on the
    device, copy, pwrite) on tmpfs.  Informational: `value` above stays the kernel-side rate.  `value` here is the STEADY
    STATE -- from the moment the first batch has been handed to the files to the end -- and the start-up (engine, model,
    pinned buffers, first batch) is reported beside it.  compress: `--compress`, the text is deflated on the device and only
    gzip members are copied and written.  workers > 1: that many reference workers (`--cpus N`: one temp file pair each,
    iss/app.py:73, 123-127) as threads sharing the GPU -- tmpfs serialises the writes to ONE file on its inode."""
    import shutil
    import tempfile
    import threading

    from insilicoseq_amd.generator import worker_iterator

    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(dir=base)
    try:
        # keep well inside the free space of the file system (2 files x ~330 B per pair as text)
        need = 2 * 340 * n_pairs * (0.4 if compress else 1.0)
        free = shutil.disk_usage(d).free
        if need > 0.4 * free:
            n_pairs = max(1000, int(n_pairs * 0.4 * free / need))
        work = [(r, int(n_pairs * abundance[r.id]), "default") for r in records]
        # the reference's chunks of the flattened list (iss/app.py:81-83), one per worker
        flat = [(r, 1 << 18, m) for r, n, m in work for _ in range(n >> 18)] + [(r, n & ((1 << 18) - 1), m) for r, n, m in work if n & ((1 << 18) - 1)]
        per = -(-len(flat) // workers)
        chunks = [flat[k * per:(k + 1) * per] for k in range(workers)]
        worker_iterator([(records[0], 1000, "default")], dense, 0, os.path.join(d, "warm"), SEED, "metagenomics", False, device=0,
                        compress=compress)  # warm-up (library, kernels)
        timings = [dict() for _ in range(workers)]
        errors = []

        def run(k):
            try:
                worker_iterator(chunks[k], dense, k, os.path.join(d, "w%d" % k), SEED, "metagenomics", False, device=0,
                                compress=compress, timings=timings[k])
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        t0 = time.perf_counter()
        ts = [threading.Thread(target=run, args=(k,)) for k in range(workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        if errors:
            raise RuntimeError(errors[0])
        size = sum(os.path.getsize(os.path.join(d, "w%d_R%d.fastq" % (k, m))) for k in range(workers) for m in (1, 2))
        n = sum(k for c in chunks for _, k, _ in c)
        # steady state: behind the moment every worker has handed its first batch to the files
        first = max(t["batches"][0][0] for t in timings if t.get("batches"))
        after = sum(p for t in timings for ts_, p in t.get("batches", []) if ts_ > first)
        t_end = max(t["t_end"] for t in timings)
        steady = after / (t_end - first) if t_end > first and after else None
        return {"value": steady, "unit": "read-pairs/s", "workers": workers, "files": 2 * workers,
                "written_GB_per_s": (size / n * steady / 1e9) if steady else None,
                "incl_startup": {"value": n / dt, "seconds": dt, "engine_startup_s": max(t["t_ready"] - t["t_start"] for t in timings),
                                 "first_batch_queued_after_s": first - t0},
                "sample": "%d pairs -> %.2f GB of %s on %s in %.2f s; steady state = the %d pairs queued after every worker's first "
                          "batch, %.2f s" % (n, size / 1e9, "gzip members (text deflated on the device)" if compress else "FASTQ",
                                             base or "the temp dir", dt, after, t_end - first)}
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
    for seq, n in work:
        k = min(n, max(1, int(round(sample_pairs * n / sum(x for _, x in work)))), todo)
        if k <= 0:
            continue
        res = orc.simulate(rng, seq, k)
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


def cpu_baseline_all_cores(dense, work, threads, one_core_rate, budget_s=5.0):
    """The same CPU oracle on every host core at once: one THREAD per core (the ctypes call into the C oracle releases
    the GIL), each a reference-style worker with its own pair of MT19937 streams (seed + cpu_number) on its share of the
    work list -- the box-level CPU rate the GPU number stands beside (the reference's pool: iss/app.py:99-106).  Sized
    from the one-core rate to take about budget_s seconds."""
    import threading

    from oracle import oracle as O

    cores = max(1, min(os.cpu_count() or 1, 64)) if not threads else int(threads)
    per_thread = max(1000, int(one_core_rate * budget_s / 4))  # (a thread of a full box runs at about a quarter of the lone core's rate)
    total = sum(n for _, n in work)
    shares = [max(1, int(round(per_thread * n / total))) for _, n in work]
    done = [0] * cores
    secs = [0.0] * cores
    errors = []

    def run(c):
        try:
            orc = O.Oracle(dense)
            rng = O.Rng().seed_mt(SEED + c)
            t0 = time.perf_counter()
            for (seq, _), k in zip(work, shares):
                res = orc.simulate(rng, seq, k)
                assert res["status"] == 0
                done[c] += res["n_done"]
            secs[c] = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t0 = time.perf_counter()
    ts = [threading.Thread(target=run, args=(c,)) for c in range(cores)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = time.perf_counter() - t0
    if errors:
        raise RuntimeError(errors[0])
    quota = None
    try:  # (a container's CPU quota, when there is one: the threads may have fewer cores than os.cpu_count() says)
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            quota = None if q == "max" else float(q) / float(period)
    except Exception:  # noqa: BLE001
        pass
    return {"value": sum(done) / wall, "unit": "read-pairs/s", "cores": cores, "kind": "port",
            "speedup_over_one_core": (sum(done) / wall) / one_core_rate if one_core_rate else None, "host_cpu_count": os.cpu_count(),
            "cgroup_cpu_quota_cores": quota,
            "sample": "%d pairs on %d threads (%d each), %.1f s wall (slowest thread %.1f s)" % (
                sum(done), cores, sum(shares), wall, max(secs)),
            "note": "reference-vs-port ratio measured in the build container: the Python reference 873 pairs/s per process, "
                    "this port ~9e4 per core"}


if __name__ == "__main__":
    main()
