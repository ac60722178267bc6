"""`generate`: the reference's ``iss generate`` flow (iss/app.py:23-144) on GPUs.

Same steps, same file names, same flag names for the options this engine supports: load the error
model, concatenate the genome FASTA files, draw / read abundances (written to ``<out>_abundance.txt``),
cut the work into ``ceil(n_pairs / workers)``-sized chunks (iss/app.py:81-83), run one worker per GPU
(``worker_iterator``; a process pool like the reference's, each process owning one device), concatenate
``<out>.iss.tmp.<k>_R{1,2}.fastq`` in worker order and clean up (iss/app.py:119-143).

Differences a user must know: uniforms come from Philox keyed by ``seed + worker``
(not the reference's Mersenne Twisters), and the reference's dropped surplus chunk / missing-temp-file
failure modes are reproduced deliberately so outputs stay comparable.
"""
import argparse
import gzip
import random
import logging
import multiprocessing as mp
import os
import shutil
import sys
import time

import numpy as np

from .distributed import VCF_HEADER, concatenate_rank_files, temp_prefix
from .generator import generate_work_divider, parse_fasta, worker_iterator, worker_set_iterator
from .model import BasicErrorModel, KDErrorModel

PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
# names of iss/generator.py:377-387 -> dense files converted from the reference's profiles
PRECOMPUTED = {"hiseq": "hiseq", "novaseq": "novaseq", "miseq": "miseq", "miseq-20": "miseq-20", "miseq-24": "miseq-24",
               "miseq-28": "miseq-28", "miseq-32": "miseq-32", "miseq-36": "miseq-36", "nextseq": "nextseq"}


def convert_n_reads(unit):
    """iss/util.py:137-161: 'k', 'm', 'g' suffixes."""
    suffixes = {"k": 3, "m": 6, "g": 9}
    unit = str(unit)
    if unit[-1].lower() in suffixes:
        return int(float(unit[:-1]) * 10 ** suffixes[unit[-1].lower()])
    return int(unit)


def load_error_model(mode, seed, model, fragment_length, fragment_length_sd, store_mutations, rng="philox"):
    """iss/generator.py:359-421 for what the device path covers (kde)."""
    logger = logging.getLogger(__name__)
    if mode == "perfect":  # (the reference's PerfectErrorModel fails at its first error draw: SURVEY.md Appendix A-8)
        logger.error("--mode perfect is not available on the GPU path")
        sys.exit(1)
    if fragment_length is not None and fragment_length_sd is not None:
        logger.info("Using custom fragment length %s and default fragment length sd %s" % (fragment_length,
                                                                                           fragment_length_sd))
    elif bool(fragment_length) ^ bool(fragment_length_sd):  # generator.py:393-395
        logger.error("fragment_length and fragment_length_sd must be specified together")
        sys.exit(1)
    if seed:  # generator.py:397-400 (seed 0 leaves the parent unseeded)
        random.seed(seed)
        np.random.seed(seed)
    if mode == "basic":  # generator.py:412-416
        if model is not None:
            logger.warning("--model %s will be ignored in --mode %s" % (model, mode))
        return BasicErrorModel(fragment_length, fragment_length_sd, store_mutations)
    if model is None:
        logger.error("--model is required in --mode kde")
        sys.exit(1)
    if model.lower() in PRECOMPUTED:
        npz = os.path.join(PROFILES, PRECOMPUTED[model.lower()] + ".dense.npz")
    else:
        npz = model
    return KDErrorModel(npz, fragment_length, fragment_length_sd, store_mutations)


def parse_abundance_file(path):
    """iss/abundance.py:13-44: tab separated `record<TAB>abundance`."""
    logger = logging.getLogger(__name__)
    out = {}
    try:
        with open(path) as fh:
            for line in fh:
                if not line.strip():
                    continue
                rid, val = line.split()[0], float(line.split()[1])
                out[rid] = val
    except (IOError, IndexError, ValueError) as e:
        logger.error("Failed to read abundance file: %s" % e)
        sys.exit(1)
    return out


def parse_readcount_file(path):
    return {k: int(v) for k, v in parse_abundance_file(path).items()}


def lognormal(record_list):
    """iss/abundance.py:137-154 (global numpy stream, seeded by load_error_model like the reference)."""
    dist = np.random.lognormal(size=len(record_list))
    scaled = dist / sum(dist)
    return {r: a for r, a in zip(record_list, scaled)}


def uniform(record_list):
    return {r: 1 / len(record_list) for r in record_list}


def exponential(record_list):
    dist = np.random.exponential(size=len(record_list))
    scaled = dist / sum(dist)
    return {r: a for r, a in zip(record_list, scaled)}


def halfnormal(record_list):
    """iss/abundance.py:97-114: scipy's half-normal variates (drawn from numpy's global stream, like the reference's)."""
    from scipy import stats

    dist = stats.halfnorm.rvs(loc=0.00, scale=1.00, size=len(record_list))
    scaled = dist / sum(dist)
    return {r: a for r, a in zip(record_list, scaled)}


def zero_inflated_lognormal(record_list):
    """iss/abundance.py:157-175: a fifth of the records (Bernoulli 0.2) get no reads at all."""
    from scipy import stats

    zero_inflated = stats.bernoulli.rvs(p=0.2, size=len(record_list))
    dist = (1 - zero_inflated) * np.random.lognormal(size=len(record_list))
    scaled = dist / sum(dist)
    return {r: a for r, a in zip(record_list, scaled)}


ABUNDANCE = {"lognormal": lognormal, "uniform": uniform, "exponential": exponential, "halfnormal": halfnormal,
             "zero_inflated_lognormal": zero_inflated_lognormal}


def coverage_scaling(total_n_reads, coverage_dic, records, read_length):
    """iss/abundance.py:196-228: scale a coverage distribution so that it adds up to the requested number of reads."""
    logger = logging.getLogger(__name__)
    total_reads = 0
    for record in records:
        if record.id not in coverage_dic:
            logger.error("Fasta record not found in abundance file: %r" % record.id)
            sys.exit(1)
        total_reads += coverage_dic[record.id] * len(record.seq) / read_length / 2
    scale_factor = total_n_reads / total_reads
    for key in coverage_dic:
        coverage_dic[key] *= scale_factor
    return coverage_dic


def _write_distribution(dic, output, mode):
    """abundance.to_file (iss/abundance.py:231-251): <out>_abundance.txt or <out>_coverage.txt"""
    with open(output + ("_abundance.txt" if mode == "abundance" else "_coverage.txt"), "w") as fh:
        for rid, a in dic.items():
            fh.write("%s\t%s\n" % (rid, a))


def compress_file(path, block_bytes=32 << 20, threads=None):
    """gzip `path` to `path + ".gz"` and remove it, like iss/util.py:255-268, but block-parallel: every block of the
    file becomes one gzip member compressed on its own thread (zlib releases the GIL); concatenated members are
    one valid gzip stream with the same content as the reference's single-member file."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    threads = threads or min(32, os.cpu_count() or 1)

    def member(block):
        c = zlib.compressobj(6, zlib.DEFLATED, 31)  # wbits 31: gzip container
        return c.compress(block) + c.flush()

    with open(path, "rb") as fi, open(path + ".gz", "wb") as fo, ThreadPoolExecutor(threads) as pool:
        pending = []
        while True:
            block = fi.read(block_bytes)
            if block:
                pending.append(pool.submit(member, block))
            while pending and (not block or len(pending) >= 2 * threads):
                fo.write(pending.pop(0).result())
            if not block:
                break
        if os.path.getsize(path) == 0:
            fo.write(member(b""))
    os.remove(path)
    return path + ".gz"


def _worker(rank, device, genome_file, work_spec, npz, seed, prefix, sequence_type, gc_bias, rng, store_mutations,
            fragment, compress=False, records=None):
    """One pool process == one GPU.  Records are re-read from the concatenated FASTA (the reference
    pickles them; same content) unless the caller runs in this process and hands its own over."""
    logging.basicConfig(level=logging.WARNING)
    # (records are identified by their ordinal in the concatenated FASTA, not by id: draft assemblies repeat ids)
    records = list(records if records is not None else parse_fasta(genome_file))
    if npz is None:  # --mode basic
        model = BasicErrorModel(fragment[0], fragment[1], store_mutations)
    else:
        model = KDErrorModel(npz, fragment[0], fragment[1], store_mutations)
    work = [(records[idx], n, "default") for idx, n in work_spec]
    worker_iterator(work, model, rank, prefix, seed, sequence_type, gc_bias, device=device, rng=rng, compress=compress)


def _run_worker_set(jobs, records, error_model, args, device_gzip, workers):
    """The reference's N workers as N chains side by side on ONE GPU (worker_set_iterator; up to 1024 of them).  Returns whether
    the FINAL files were written (else the temp files were), or None when the set cannot be set up -- more workers than the engine takes, not enough memory for their stream buffers (three turns of
    stream words per worker and buffer: tens of GB from W = 512 on with long reads) -- BEFORE anything was written: the caller
    then takes the process pool, which has no such limit.  (ISS_HOST_FASTQ=1, the host formatter, is a Worker switch: the pool.)"""
    from ._native import E_INVALID, E_NOMEM, EngineError

    logger = logging.getLogger(__name__)
    works = [[(records[idx], n, "default") for idx, n in j[3]] for j in jobs]
    try:
        # text mode: the workers write at their places of the FINAL files -- the concatenation has nothing left to do (fewer
        # chunks than workers: the reference fails on the missing temp file, util.py:233 -- that path keeps the temp files)
        return worker_set_iterator(
            works, error_model, [j[0] for j in jobs], [j[6] for j in jobs], args.seed, args.sequence_type, args.gc_bias, device=0,
            compress=device_gzip,
            final_prefix=args.output if len(jobs) == workers and os.environ.get("ISS_SET_TEMP_FILES", "") != "1" else None)
    except EngineError as e:
        if e.code in (E_INVALID, E_NOMEM) and getattr(e, "before_output", False):
            logger.warning("%d workers side by side do not fit the device (%s): one process per worker instead" % (workers, e))
            return None
        raise


def generate_reads(args):
    logger = logging.getLogger(__name__)
    error_model = load_error_model(args.mode, args.seed, args.model, args.fragment_length, args.fragment_length_sd,
                                   args.store_mutations, args.rng)
    if not args.genomes:
        logger.error("One of --genomes/-g is required")
        sys.exit(1)
    genome_file = args.output + ".iss.tmp.genomes.fasta"  # generator.py:468-469
    with open(genome_file, "wb") as out:
        for g in args.genomes:
            with open(g, "rb") as fh:
                shutil.copyfileobj(fh, out)
    records = list(parse_fasta(genome_file))
    if not records:
        logger.error("Genome(s) file seems empty: %s" % genome_file)
        sys.exit(1)
    ids = [r.id for r in records]
    # load_readcount_or_abundance (iss/generator.py:497-594), in the reference's order of precedence
    readcount_dic = abundance_dic = None
    if args.readcount_file:
        logger.warning("--readcount_file disables --n_reads, n_reads will be calculated from the readcount file")
        readcount_dic = parse_readcount_file(args.readcount_file)
        n_reads = sum(readcount_dic.values())
    else:
        n_reads = convert_n_reads(args.n_reads)
        if args.abundance_file:
            abundance_dic = parse_abundance_file(args.abundance_file)
        elif args.coverage_file:  # coverages instead of shares: the reads per record no longer depend on --n_reads
            logger.warning("--coverage_file disables --n_reads")
            abundance_dic = parse_abundance_file(args.coverage_file)
        elif args.coverage in ABUNDANCE:
            abundance_dic = ABUNDANCE[args.coverage](ids)
            if args.n_reads:
                abundance_dic = coverage_scaling(n_reads, abundance_dic, records, error_model.read_length)
            _write_distribution(abundance_dic, args.output, "coverage")
        elif args.abundance in ABUNDANCE:
            abundance_dic = ABUNDANCE[args.abundance](ids)
            _write_distribution(abundance_dic, args.output, "abundance")
        else:
            logger.error("Could not get abundance, or coverage or readcount information")
            sys.exit(1)
    workers = args.gpus
    # --compress: the workers' FASTQ files already hold gzip members built on the device (one per batch); concatenated
    # they are the .gz files util.compress would have made from the text (iss/util.py:255-268), which never exists
    device_gzip = bool(args.compress) and os.environ.get("ISS_HOST_FASTQ", "") != "1"
    gz = {"_R1.fastq": "_R1.fastq.gz", "_R2.fastq": "_R2.fastq.gz"} if device_gzip else None
    chunk_size = -((n_reads // 2) // -workers)  # ceildiv, app.py:82
    chunks = list(generate_work_divider(records, readcount_dic, abundance_dic, n_reads, args.coverage, args.coverage_file,
                                        error_model, args.output, chunk_size))
    jobs = []
    ordinal_of = {id(r): i for i, r in enumerate(records)}
    for rank, chunk in enumerate(chunks[:workers]):  # zip(work_chunks, temp_file_list), app.py:104
        spec = [(ordinal_of[id(rec)], n) for rec, n, _ in chunk]
        jobs.append((rank, rank % max(args.devices, 1), genome_file, spec, error_model.npz_path, args.seed,
                     temp_prefix(args.output, rank), args.sequence_type, args.gc_bias, args.rng, args.store_mutations,
                     (args.fragment_length, args.fragment_length_sd), device_gzip))
    t_gen = time.perf_counter()
    in_place = None
    if workers == 1:
        for j in jobs:
            _worker(*j, records=records)
    elif args.rng == "mt" and args.devices == 1 and not args.store_mutations and args.seed is not None and workers <= 1024 \
            and os.environ.get("ISS_HOST_FASTQ", "") != "1":
        in_place = _run_worker_set(jobs, records, error_model, args, device_gzip, workers)
    if workers > 1 and in_place is None:  # one process per worker (and what the set could not take)
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.starmap(_worker, jobs)
    t_cat = time.perf_counter()
    if in_place:
        pass
    elif args.store_mutations:  # app.py:128-133
        concatenate_rank_files(args.output, workers, suffixes=("_R1.fastq", "_R2.fastq", ".vcf"),
                               headers={".vcf": VCF_HEADER}, out_suffixes=gz)
    else:
        concatenate_rank_files(args.output, workers, out_suffixes=gz)  # raises if a worker had no chunk (util.py:233)
    logger.info("Workers %.2f s, concatenation of their files %.2f s" % (t_cat - t_gen, time.perf_counter() - t_cat))
    os.remove(genome_file)
    if args.compress:  # util.compress (iss/util.py:255-268): <file>.gz next to the file, original removed
        for suffix in (() if device_gzip else ("_R1.fastq", "_R2.fastq")) + ((".vcf",) if args.store_mutations else ()):
            compress_file(args.output + suffix)
    logger.info("Read generation complete")


def main(argv=None):
    p = argparse.ArgumentParser(prog="insilicoseq_amd", description="iss generate on MI355X")
    sub = p.add_subparsers(dest="cmd")
    g = sub.add_parser("generate")
    g.add_argument("--genomes", "-g", nargs="+")
    g.add_argument("--model", "-m")
    g.add_argument("--mode", "-e", default="kde", choices=["kde", "basic", "perfect"])
    g.add_argument("--n_reads", "-n", default="1000000")
    g.add_argument("--seed", type=int, default=None)
    g.add_argument("--gpus", "--cpus", "-p", type=int, default=1, dest="gpus", help="workers (one per GPU)")
    g.add_argument("--devices", type=int, default=0, help="visible GPUs (default: one per worker)")
    g.add_argument("--abundance", "-a", default="lognormal", choices=sorted(ABUNDANCE))
    g.add_argument("--abundance_file", "-b")
    g.add_argument("--coverage", "-C", default=None, choices=sorted(ABUNDANCE))
    g.add_argument("--coverage_file", "-D")
    g.add_argument("--readcount_file", "-R")
    g.add_argument("--gc_bias", "-c", action="store_true")
    g.add_argument("--sequence_type", "-t", default="metagenomics", choices=["metagenomics", "amplicon"])
    g.add_argument("--fragment-length", "-l", type=int, default=None, dest="fragment_length")
    g.add_argument("--fragment-length-sd", "-s", type=int, default=None, dest="fragment_length_sd")
    g.add_argument("--store_mutations", "-M", action="store_true")
    g.add_argument("--compress", "-z", action="store_true")
    g.add_argument("--rng", default="philox", choices=["philox", "mt"],
                   help="philox: parallel counter-based streams (default); mt: the reference's Mersenne-Twister "
                        "streams consumed sequentially on the GPU -- output identical to `iss generate` for the same --seed")
    g.add_argument("--output", "-o", required=True)
    g.add_argument("--quiet", "-q", action="store_true")
    args = p.parse_args(argv)
    if args.cmd != "generate":
        p.print_help()
        return 1
    if not args.devices:
        args.devices = args.gpus
    logging.basicConfig(level=logging.ERROR if args.quiet else logging.INFO)
    generate_reads(args)
    return 0
