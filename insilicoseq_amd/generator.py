"""Host-side mirror of the reference's per-worker driver, on top of the HIP engine.

Same function names, argument order and error behaviour as iss/generator.py so the parity
tests read like the reference's:

* ``worker_iterator(work, error_model, cpu_number, worker_prefix, seed, sequence_type, gc_bias)``
  -- iss/generator.py:223-251: creates ``{prefix}_R1.fastq``, ``{prefix}_R2.fastq``,
  ``{prefix}.vcf``; worker seed = ``seed + cpu_number`` when ``seed is not None``; work items in
  order; pair ids restart at 0 for every work item (:72-75); read ids
  ``{record.id}_{i}_{cpu_number}/1|2`` (:150, 181).
* ``simulate_reads`` -- iss/generator.py:21-66.
* ``generate_work_divider`` -- iss/generator.py:254-356 (rounding correction :299-306, chunking
  :333-356); ``to_coverage`` -- iss/abundance.py:178-193.

One worker == one GPU (``cpu_number`` == rank).  The uniforms come from Philox addressed by
(worker seed, running pair ordinal within the worker), so the output is a pure function of
(seed, cpu_number, work list) -- independent of launch geometry.  It is bit-identical to the
CPU oracle in Philox mode, which in MT mode is bit-identical to the reference.
"""
import logging
import os
import sys
import time

import numpy as np

from . import _native
from .engine import ReadEngine, fastq_write
from .model import DenseModel


class Record(object):
    """The two attributes of a Bio.SeqRecord the hot path reads: ``id`` and ``seq``."""

    def __init__(self, seq, id="<unknown id>", description=""):
        self.seq = seq
        self.id = id
        self.description = description

    def __len__(self):
        return len(self.seq)


def _parse_fasta_lines(fh):
    """Line by line (any text handle)."""
    header, chunks = None, []
    for line in fh:
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if header is not None:
                yield Record("".join(chunks), id=(header.split(None, 1) or [""])[0], description=header)
            header, chunks = line[1:], []
        elif header is not None:
            chunks.append(line.strip())
    if header is not None:
        yield Record("".join(chunks), id=(header.split(None, 1) or [""])[0], description=header)


def parse_fasta(path_or_handle):
    """FASTA -> Records; id = first whitespace-separated token of the header, sequence case-preserved
    (what Bio.SeqIO.parse(..., 'fasta') yields; pinned by iss/test/test_util.py:41-45).  A file given by path is cut
    at its header lines and every record's line ends are removed in one pass (a few hundred Mbp: seconds line by
    line); records with unusual white space inside go through the line-by-line form."""
    if not isinstance(path_or_handle, (str, bytes, os.PathLike)):
        yield from _parse_fasta_lines(path_or_handle)
        return
    with open(path_or_handle, "rb") as fh:
        data = fh.read()
    starts = []  # offsets of the '>' of every header line
    at = 0 if data.startswith(b">") else data.find(b"\n>") + 1
    while at > 0 or (at == 0 and data.startswith(b">") and not starts):
        starts.append(at)
        nxt = data.find(b"\n>", at)
        if nxt < 0:
            break
        at = nxt + 1
    for k, start in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(data)
        eol = data.find(b"\n", start, end)
        if eol < 0:
            eol = end
        header = data[start + 1:eol].decode().rstrip("\r\n")
        body = data[eol + 1:end]
        if b" " in body or b"\t" in body or b"\x0b" in body or b"\x0c" in body:
            seq = "".join(line.strip() for line in body.decode().splitlines())
        else:
            seq = body.translate(None, b"\r\n").decode()
        yield Record(seq, id=(header.split(None, 1) or [""])[0], description=header)


def to_coverage(total_n_reads, species_abundance, read_length, genome_size):
    """iss/abundance.py:178-193"""
    n_reads = total_n_reads * species_abundance
    coverage = (n_reads * read_length) / genome_size
    return coverage


def _pairs_per_record(records, readcount_dic, abundance_dic, n_reads, coverage_mode, read_length):
    """(record, read pairs) for every record that gets reads -- the arithmetic of iss/generator.py:288-311: pairs =
    round(reads / 2) per record, plus one whenever the running sum of the rounded counts falls behind the rounded
    running sum of the exact ones."""
    logger = logging.getLogger(__name__)
    if readcount_dic is None and abundance_dic is None:
        raise RuntimeError("No readcount or abundance file provided")
    table, what = (readcount_dic, "readcount") if readcount_dic is not None else (abundance_dic, "abundance")
    exact_sum, given = 0.0, 0
    for record in records:
        if record.id not in table:
            logger.warning("Record %s not found in %s file" % (record.id, what))
            continue
        if readcount_dic is not None:
            exact = table[record.id] / 2
        else:
            share, size = table[record.id], len(record.seq)
            cov = share if coverage_mode else to_coverage(n_reads, share, read_length, size)
            exact = ((cov * size) / read_length) / 2
        n_pairs = round(exact)
        exact_sum += exact
        given += n_pairs
        if round(exact_sum) > given:  # rounding correction
            n_pairs += 1
            given += 1
        logger.debug("%s: %s read pairs" % (record.id, n_pairs))
        if n_pairs:
            yield record, n_pairs


def generate_work_divider(fasta_file, readcount_dic, abundance_dic, n_reads, coverage, coverage_file, error_model,
                          output, chunk_size):
    """Lists of (record, n_pairs, mode), each list worth ``chunk_size`` pairs (the last one possibly less): the
    flattened (record, pairs) sequence cut into consecutive chunks, a record straddling a boundary being split --
    iss/generator.py:254-356.  ``mode`` is always "default": the > 2 GiB pickle spill (:313-331) is a transport
    workaround of the reference's process pool and has no counterpart here (genomes go to HBM, not through pickle)."""
    chunk, room = [], chunk_size
    for record, todo in _pairs_per_record(fasta_file, readcount_dic, abundance_dic, n_reads, bool(coverage or coverage_file),
                                          error_model.read_length):
        while todo:
            take = min(todo, room)
            chunk.append((record, take, "default"))
            todo -= take
            room -= take
            if room == 0:
                yield chunk
                chunk, room = [], chunk_size
    if chunk:
        yield chunk


def _dense_of(error_model):
    if isinstance(error_model, DenseModel):
        return error_model
    if hasattr(error_model, "dense"):
        return error_model.dense()
    raise TypeError("error_model must be an insilicoseq_amd KDErrorModel or DenseModel")


def worker_seed(seed, cpu_number):
    """``seed + cpu_number`` when seeded (iss/generator.py:234-236); OS entropy otherwise."""
    if seed is not None:
        return (int(seed) + int(cpu_number)) & (2**64 - 1)
    return int.from_bytes(os.urandom(8), "little")


class Worker(object):
    """State of one reference worker on one GPU: engine + uploaded genomes + running ordinal."""

    # rows generated / formatted / written per step of the streaming loop (ISS_BATCH_PAIRS: tuning aid).  2^18: the
    # kernels still run at their full rate, the (pinned) buffers are allocated in a quarter of the time of 2^20
    BATCH_PAIRS = int(os.environ.get("ISS_BATCH_PAIRS", 1 << 18))
    GENOME_BUDGET = 96 << 30  # letters kept resident in HBM (1.4 B each incl. the packed copies) before all are dropped

    def __init__(self, error_model, cpu_number, seed, device=None, rng="philox", compress=False):
        if rng not in ("philox", "mt"):
            raise ValueError("rng must be 'philox' (parallel, default) or 'mt' (reference-identical, sequential)")
        self.rng = rng
        self.cpu_number = int(cpu_number)
        self.seed = worker_seed(seed, cpu_number)
        self.engine = ReadEngine(self.cpu_number if device is None else device)
        self.dense = _dense_of(error_model)
        self.engine.load_model(self.dense)
        self.store_mutations = False
        self.device_fastq = os.environ.get("ISS_HOST_FASTQ", "") != "1"  # ISS_HOST_FASTQ=1: host formatter (iss_fastq_write)
        self.compress = bool(compress)
        if self.compress:
            if not self.device_fastq:
                raise ValueError("compress=True needs the device FASTQ path (unset ISS_HOST_FASTQ)")
            self.engine.fastq_compress(True)  # the FASTQ handles receive gzip members instead of text
        if rng == "mt":
            # random.seed(seed + cpu_number); np.random.seed(seed + cpu_number)  (generator.py:234-236);
            # unseeded workers draw an OS-entropy seed (the reference is then not reproducible either)
            self.engine.seed_mt(self.seed & 0xFFFFFFFF if seed is None else self.seed)
            self.engine.mt_set_fragment(getattr(error_model, "fragment_length", None),
                                        getattr(error_model, "fragment_sd", None))
        else:
            self.engine.set_fragment(getattr(error_model, "fragment_length", None),
                                     getattr(error_model, "fragment_sd", None))
        self.has_fragment = (getattr(error_model, "fragment_length", None) is not None and
                             getattr(error_model, "fragment_sd", None) is not None)
        self.ordinal = 0
        self.timings = None
        self._gids = {}  # id(record) -> (record, genome id on the device)
        self._resident = 0

    def close(self):
        self.engine.close()

    def needs_room_for(self, record):
        """Would uploading this record drop the resident genomes (GENOME_BUDGET)?"""
        hit = self._gids.get(id(record))
        return (hit is None or hit[0] is not record) and bool(self._gids) and self._resident + len(record.seq) > self.GENOME_BUDGET

    def genome_id(self, record):
        key = id(record)
        hit = self._gids.get(key)
        if hit is None or hit[0] is not record:
            seq = record.seq  # str / bytes / uint8 array as they are; Bio.Seq and the like through str()
            if not isinstance(seq, (str, bytes, bytearray, np.ndarray)):
                seq = str(seq)
            if self._gids and self._resident + len(seq) > self.GENOME_BUDGET:
                # a work list visits a record in one or two consecutive items: nothing uploaded so far is needed again
                self.engine.clear_genomes()  # (waits for the device and the FASTQ pipeline first)
                self._gids.clear()
                self._resident = 0
            hit = self._gids[key] = (record, self.engine.add_genome(seq))
            self._resident += len(seq)
        return hit[1]

    def simulate_reads(self, record, n_pairs, forward_handle, reverse_handle, mutations_handle, sequence_type,
                       gc_bias=False, writer_threads=4, flush=True):
        """iss/generator.py:21-66 for one work item, streamed in batches: generate on the GPU, copy
        back, format FASTQ on host threads."""
        logger = logging.getLogger(__name__)
        logger.debug("Cpu #%s: Generating %s read pairs" % (self.cpu_number, n_pairs))
        eng = self.engine
        if not (eng.read_length < len(record.seq)):
            # AssertionError in simulate_read -> warning + record skipped (generator.py:77-80)
            logger.warning("%s shorter than read length for this ErrorModel" % record.id)
            logger.warning("Skipping %s. You will have less reads than specified" % record.id)
            if self.rng == "mt" and n_pairs > 0:
                # the reference has already drawn the insert size (or, with --fragment-length, its gaussian) when its
                # assertion fails (generator.py:121-130): the engine consumes the same draw and reports the short record
                gid = self.genome_id(record)  # (upload errors -- letters outside the alphabet, an empty record -- propagate)
                try:
                    eng.generate_mt(gid, 1)
                except _native.EngineError as e:
                    if e.code != _native.E_SHORT_RECORD:
                        raise
            return 0
        gid = self.genome_id(record)
        for fh in (forward_handle, reverse_handle):
            fh.flush()
        done = 0
        while done < n_pairs:
            n = min(self.BATCH_PAIRS, n_pairs - done)
            if self.rng == "mt":
                assert eng.generate_mt(gid, n, sequence_type=sequence_type, gc_bias=gc_bias, out_first_pair=0) == n
                if self.store_mutations:
                    write_mutations(eng.mt_mutations(), mutations_handle, record.id, done, self.cpu_number)
            else:
                def gen():
                    eng.generate(gid, n, first_ordinal=self.ordinal, seed=self.seed, sequence_type=sequence_type,
                                 gc_bias=gc_bias, out_first_pair=0)
                gen()
                if self.store_mutations:
                    write_mutations(mutation_rows(eng, gen), mutations_handle, record.id, done, self.cpu_number)
            if self.device_fastq:
                # text built on the device, copied and written behind the next batch's generation
                # (one pwrite stream per file: tmpfs gets slower with concurrent writers to one file)
                eng.fastq_emit(forward_handle.fileno(), reverse_handle.fileno(), record.id, done, self.cpu_number, 0, n,
                               n_threads=1)
            else:
                eng.synchronize()
                rows = eng.download(0, n)["_pitched"]
                fastq_write(forward_handle.fileno(), reverse_handle.fileno(), record.id, done, self.cpu_number, n,
                            eng.read_length, eng.pitch, rows[0], rows[1], rows[2], rows[3], n_threads=writer_threads)
            self.ordinal += n
            done += n
        if self.device_fastq and flush:
            eng.fastq_flush()  # the handles are the caller's again
        return done


def mutation_rows(eng, regenerate):
    """--store_mutations rows of the generate call just made (Philox path).  The row buffer is sized from the model's
    expected rows; when a batch overflows it (ISS_E_NOMEM: a heavy-indel or edited model, rows written twice for reads
    the indel kernels rebuild) the reservation doubles and ``regenerate()`` repeats the call -- generation is a pure
    function of seed and ordinal, so the rows and the reads are the same ones."""
    while True:
        try:
            return eng.mutations()
        except _native.EngineError as e:
            if e.code != _native.E_NOMEM:
                raise
            cap = eng.mutations_capacity
            if cap >= 0x7fffffff:
                raise
            # the call says how many slots it asked for: reserve them (+ 1/8, the kernels hand slots out in 256-row chunks
            # per wavefront) in ONE step and repeat the call once
            need = int(getattr(eng, "mutation_slots_needed", 0)) or 2 * max(cap, 1 << 16)
            new_cap = min(need + need // 8 + (1 << 16), 0x7fffffff)
            logging.getLogger(__name__).info("mutation rows: %d slots reserved, the call needs %d: reserving %d and repeating it" % (
                cap, need, new_cap))
            eng.mutations_reserve(new_cap)
            regenerate()


def _simulate_work_batched(w, work, forward_handle, reverse_handle, mutations_handle, sequence_type, gc_bias):
    """The worker's loop over its work items (iss/generator.py:245-249) with the parallel path's batches cut across
    items: up to BATCH_PAIRS pairs of consecutive items go through ONE set of launches (engine.generate_batch; the
    rows are those of one call per item), then every item's rows are handed to the FASTQ pipeline under its own record
    id.  Same files as simulate_reads item by item."""
    logger = logging.getLogger(__name__)
    eng = w.engine
    pending, cur = [], 0  # (record id, genome id, pairs, id of the item's first pair in this piece)

    def run():
        nonlocal pending, cur
        if not pending:
            return
        rows = None
        try:
            def gen():
                eng.generate_batch([p[1] for p in pending], [p[2] for p in pending], first_ordinal=w.ordinal, seed=w.seed,
                                   sequence_type=sequence_type, gc_bias=gc_bias, out_first_pair=0)
            gen()
            if w.store_mutations:
                rows = mutation_rows(eng, gen)
        except _native.EngineError as e:
            if e.code != _native.E_INVALID or "records of one call must stay below" not in str(e):
                raise
            # records too long to stand side by side in one arena: the same rows from one call per item
            eng.reserve(sum(p[2] for p in pending))
            parts, row, ordinal = [], 0, w.ordinal
            for _rid, gid, n, _first_i in pending:
                def gen1(gid=gid, n=n, ordinal=ordinal, row=row):
                    eng.generate(gid, n, first_ordinal=ordinal, seed=w.seed, sequence_type=sequence_type, gc_bias=gc_bias,
                                 out_first_pair=row)
                gen1()
                if w.store_mutations:
                    part = mutation_rows(eng, gen1)
                    part["pair"] += row
                    parts.append(part)
                row += n
                ordinal += n
            if w.store_mutations:
                rows = np.concatenate(parts) if parts else None
        pairs = rows["pair"] if rows is not None else None  # ascending: the rows come back in (pair, mate, ...) order
        row, emit = 0, []
        for rid, _gid, n, first_i in pending:
            if rows is not None:  # this item's rows are contiguous
                lo, hi = np.searchsorted(pairs, row, "left"), np.searchsorted(pairs, row + n, "left")
                sel = rows[lo:hi].copy()
                sel["pair"] -= row
                write_mutations(sel, mutations_handle, rid, first_i, w.cpu_number)
            emit.append((rid, first_i, row, n))
            row += n
        eng.fastq_emit_batch(forward_handle.fileno(), reverse_handle.fileno(), emit, w.cpu_number)  # one text job
        w.ordinal += row
        if getattr(w, "timings", None) is not None:  # (measurement: when each batch was handed to the FASTQ pipeline, and how many pairs it held)
            w.timings.setdefault("batches", []).append((time.perf_counter(), row))
        pending, cur = [], 0

    for fh in (forward_handle, reverse_handle):
        fh.flush()
    for record, n_pairs, _mode in work:
        logger.debug("Cpu #%s: Generating %s read pairs" % (w.cpu_number, n_pairs))
        if not (eng.read_length < len(record.seq)):  # AssertionError in simulate_read -> record skipped (generator.py:77-80)
            logger.warning("%s shorter than read length for this ErrorModel" % record.id)
            logger.warning("Skipping %s. You will have less reads than specified" % record.id)
            continue
        if w.needs_room_for(record):
            run()  # (the genomes about to be dropped are still named by the pending items)
        gid = w.genome_id(record)
        done = 0
        while done < n_pairs:
            take = min(n_pairs - done, w.BATCH_PAIRS - cur)
            pending.append((record.id, gid, take, done))
            cur += take
            done += take
            if cur >= w.BATCH_PAIRS:
                run()
    run()


def write_mutations(rows, mutations_handle, record_id, first_i, cpu_number):
    """iss/generator.py:598-620 for the device's mutation records (columns converted once: the per-row cost matters
    at millions of rows per batch)."""
    if len(rows) == 0:
        return
    pair = (np.asarray(rows["pair"], dtype=np.int64) + int(first_i)).tolist()
    mate = (np.asarray(rows["mate"], dtype=np.int64) + 1).tolist()
    typ = np.asarray(rows["type"]).tolist()
    pos = (np.asarray(rows["position"], dtype=np.int64) + 1).tolist()
    ref = np.asarray(rows["ref"], dtype=np.uint8).tobytes().decode("latin-1")
    alt = np.asarray(rows["alt"], dtype=np.uint8).tobytes().decode("latin-1")
    qual = np.asarray(rows["quality"]).tolist()
    head = "%s_" % record_id
    tail = "_%d/" % cpu_number
    # insertion: alt = ref + inserted letter (__init__.py:203); only substitutions carry a quality
    mutations_handle.write("".join(
        "%s%d%s%d\t%d\t.\t%s\t%s\t%s\t\t\n" % (head, p, tail, m, x, r, r + a if t == 1 else a, q if t == 0 else ".")
        for p, m, t, x, r, a, q in zip(pair, mate, typ, pos, ref, alt, qual)))


def simulate_reads(record, error_model, n_pairs, cpu_number, forward_handle, reverse_handle, mutations_handle,
                   sequence_type, gc_bias=False, mode="default", seed=None):
    """Signature of iss/generator.py:21-32 (+ ``seed``: the reference reads the global RNG state)."""
    w = Worker(error_model, cpu_number, seed, device=0)
    try:
        return w.simulate_reads(record, n_pairs, forward_handle, reverse_handle, mutations_handle, sequence_type,
                                gc_bias)
    finally:
        w.close()


def worker_iterator(work, error_model, cpu_number, worker_prefix, seed, sequence_type, gc_bias, device=None,
                    rng="philox", compress=False, timings=None):
    """iss/generator.py:223-251 on GPU ``device`` (default: ``cpu_number``).  ``rng="mt"`` consumes the
    reference's two Mersenne-Twister streams on the device: the files then equal the reference's byte for
    byte (sequential, ~1e5 pairs/s); ``rng="philox"`` is the parallel path.  ``compress=True``: the two FASTQ files
    (same names) hold gzip members built on the device instead of text -- `--compress` without the text ever leaving
    the GPU; gunzipped they are the files ``compress=False`` writes.  ``timings``: a dict that receives ``t_start``, ``t_ready``
    (engine created, model uploaded), ``batches`` [(time a batch was queued for the files, its pairs)] and ``t_end`` (files
    complete) -- bench.py's end-to-end legs report the steady state apart from the start-up."""
    logger = logging.getLogger(__name__)
    if timings is not None:
        timings["t_start"] = time.perf_counter()
    store_mutations = bool(getattr(error_model, "store_mutations", False))
    if sequence_type not in _native.SEQ_TYPES:
        raise RuntimeError("sequence type '%s' is not supported" % sequence_type)  # generator.py:139
    try:
        forward_handle = open("%s_R1.fastq" % worker_prefix, "w")
        reverse_handle = open("%s_R2.fastq" % worker_prefix, "w")
        mutation_handle = open("%s.vcf" % worker_prefix, "w")
    except PermissionError as e:
        logger.error("Failed to write temporary output file(s): %s" % e)
        sys.exit(1)
    w = Worker(error_model, cpu_number, seed, device=device, rng=rng, compress=compress)
    w.timings = timings
    if timings is not None:
        timings["t_ready"] = time.perf_counter()
    if store_mutations:
        w.store_mutations = True
        # row buffers of a batch, from the model's own error rates (twice the expectation + slack; the Philox kernels
        # reserve 256-row chunks per wavefront on top)
        per_pair = 2.0 * _dense_of(error_model).expected_mutation_rows_per_pair() + 4.0
        if rng == "mt":
            w.engine.mt_mutations_reserve(int(Worker.BATCH_PAIRS * per_pair))
        else:
            w.engine.mutations_reserve(int(Worker.BATCH_PAIRS * per_pair) + (1 << 21))
    try:
        with forward_handle, reverse_handle, mutation_handle:
            if rng == "philox" and w.device_fastq and os.environ.get("ISS_ITEMWISE", "") != "1":
                _simulate_work_batched(w, work, forward_handle, reverse_handle, mutation_handle, sequence_type, gc_bias)
            else:
                for record, n_pairs, _mode in work:
                    w.simulate_reads(record, n_pairs, forward_handle, reverse_handle, mutation_handle, sequence_type,
                                     gc_bias, flush=False)  # keep the text pipeline running across work items
            w.engine.fastq_flush()
            if timings is not None:
                timings["t_end"] = time.perf_counter()
    finally:
        w.close()


def _digits_before(x):
    """Characters of the decimal numbers 0 .. x-1 written one after the other (iss_fastq.hip.h: digits_before)."""
    total, d, p = 0, 1, 1  # p = 10^(d-1): the numbers with d digits are p .. 10 p - 1 (and 0 has one)
    while x >= p * 10:
        total += d * (p * 10 - (p if d > 1 else 0))
        p *= 10
        d += 1
    return total + d * (x - (p if d > 1 else 0))


def fastq_text_bytes(record_id, n_pairs, cpu_number, read_length):
    """Bytes of the FASTQ text of pairs 0 .. n_pairs-1 of one work item in EITHER file: per record "@{id}_{i}_{cpu}/m\n" + SEQ +
    "\n+\n" + QUAL + "\n" (iss/generator.py:64-65, 150, 181) = len(id) + len(cpu) + 2 RL + 10 + digits(i)."""
    return n_pairs * (len(str(record_id).encode()) + len(str(int(cpu_number))) + 2 * read_length + 10) + _digits_before(n_pairs)


def worker_set_iterator(works, error_model, cpu_numbers, worker_prefixes, seed, sequence_type, gc_bias, device=None,
                        compress=False, batch_pairs=None, final_prefix=None):
    """W reference workers (``rng="mt"``) on ONE GPU, side by side: the files of ``worker_iterator(works[k], error_model,
    cpu_numbers[k], worker_prefixes[k], seed, ..., rng="mt")`` for every k -- byte for byte the reference's
    ``iss generate --cpus W`` temp files (iss/app.py:99-106, iss/generator.py:223-251) -- but the workers' chains run in the
    same kernel launches, one workgroup per worker (ReadEngine.generate_mt_workers).  A worker is a sequential chain over its
    two MT19937 streams (seed + cpu_number, generator.py:234-236); W of them are what the reference itself runs in parallel.
    ``--store_mutations`` rows are per engine: such a run takes one worker after the other through worker_iterator.

    ``final_prefix`` (text mode): the workers' text goes straight to ``{final_prefix}_R1.fastq`` / ``_R2.fastq`` -- what the
    parent's concatenation of the temp files in worker order would hold (iss/app.py:123-127, iss/util.py:213-234): a worker's
    text size is arithmetic (fastq_text_bytes), so worker k starts where workers 0 .. k-1 end, a round is ONE text job whose
    pieces are written at their places (ReadEngine.fastq_emit_scatter), and no temp file is made.  Returns True when the final
    files were written, False when the temp files were (the caller concatenates them)."""
    logger = logging.getLogger(__name__)
    W = len(works)
    if not (W == len(cpu_numbers) == len(worker_prefixes)) or W < 1:
        raise ValueError("worker_set_iterator: one work list, cpu number and file prefix per worker")
    if sequence_type not in _native.SEQ_TYPES:
        raise RuntimeError("sequence type '%s' is not supported" % sequence_type)  # generator.py:139
    if bool(getattr(error_model, "store_mutations", False)) or seed is None:
        # (unseeded workers draw their seeds from the OS one by one, like the reference's processes)
        for work, cpu, prefix in zip(works, cpu_numbers, worker_prefixes):
            worker_iterator(work, error_model, cpu, prefix, seed, sequence_type, gc_bias, device=device, rng="mt", compress=compress)
        return False
    final = final_prefix is not None and not compress
    handles = []
    try:
        if final:
            handles.append((open("%s_R1.fastq" % final_prefix, "w"), open("%s_R2.fastq" % final_prefix, "w")))
        else:
            for prefix in worker_prefixes:
                handles.append((open("%s_R1.fastq" % prefix, "w"), open("%s_R2.fastq" % prefix, "w"), open("%s.vcf" % prefix, "w")))
    except PermissionError as e:
        logger.error("Failed to write %s output file(s): %s" % ("the" if final else "temporary", e))
        sys.exit(1)
    eng = ReadEngine(0 if device is None else device)
    try:
        dense = _dense_of(error_model)
        eng.load_model(dense)
        at = [0] * W  # final files: where worker k's next byte goes
        if final:
            RL, total = dense.read_length, 0
            ends = [0] * W
            for k, (work, cpu) in enumerate(zip(works, cpu_numbers)):
                at[k] = total
                total += sum(fastq_text_bytes(rec.id, n, cpu, RL) for rec, n, _m in work if RL < len(rec.seq))
                ends[k] = total
            for fh in handles[0]:
                fh.flush()
                os.ftruncate(fh.fileno(), total)
        if compress:
            eng.fastq_compress(True)
        began = [False]  # (an engine error before the first round's rows exist leaves no output: the caller may take another path)
        try:
            eng.seed_mt_workers([worker_seed(seed, c) for c in cpu_numbers])
        except _native.EngineError as e:
            e.before_output = True
            raise
        eng.mt_set_fragment(getattr(error_model, "fragment_length", None), getattr(error_model, "fragment_sd", None))
        # rows per worker and round (2^20 pairs per round for all workers together; 2^22 and 2^24 measured the same end to end:
        # 16 M pairs at W = 64 in 2.6-2.8 s incl. 0.5 s of engine start -- generation and text take turns)
        per = int(batch_pairs or max(1024, min(Worker.BATCH_PAIRS, (1 << 20) // W)))
        gids, resident = {}, [0]

        # letters resident in HBM before all are dropped: Worker's budget less what the set itself holds on the device (stream
        # buffers, rows of a round -- up to a third of the memory)
        budget = max(Worker.GENOME_BUDGET // 2, 1 << 30)
        over = [False]

        def gid_of(record):
            hit = gids.get(id(record))
            if hit is None or hit[0] is not record:
                seq = record.seq
                if not isinstance(seq, (str, bytes, bytearray, np.ndarray)):
                    seq = str(seq)
                if gids and resident[0] + len(seq) > budget:
                    over[0] = True  # (this round's pieces name the resident records: they go at the round's end)
                hit = gids[id(record)] = (record, eng.add_genome(seq))
                resident[0] += len(seq)
            return hit[1]

        def pieces(work, cpu):  # (record, genome id, pairs, id of the piece's first pair, short record?)
            for record, n_pairs, _mode in work:
                logger.debug("Cpu #%s: Generating %s read pairs" % (cpu, n_pairs))
                if not (eng.read_length < len(record.seq)):
                    # AssertionError in simulate_read -> warning + record skipped (generator.py:77-80); the reference has
                    # drawn the insert size (or its gaussian) by then (generator.py:121-130): the engine consumes that draw
                    logger.warning("%s shorter than read length for this ErrorModel" % record.id)
                    logger.warning("Skipping %s. You will have less reads than specified" % record.id)
                    if n_pairs > 0:
                        yield record, gid_of(record), 1, 0, True
                    continue
                done = 0
                while done < n_pairs:
                    n = min(per, n_pairs - done)
                    yield record, gid_of(record), n, done, False
                    done += n

        its = [pieces(work, cpu) for work, cpu in zip(works, cpu_numbers)]
        for fh3 in handles:
            for fh in fh3[:2]:
                fh.flush()
        while True:
            if over[0] or resident[0] > budget:
                over[0] = False
                # between rounds nothing names an uploaded record: drop them all (a round uploads what its pieces need again)
                eng.clear_genomes()  # (waits for the device and the FASTQ pipeline first)
                gids.clear()
                resident[0] = 0
            cur = [next(it, None) for it in its]
            if all(c is None for c in cur):
                break
            g = [c[1] if c else 0 for c in cur]
            n = [c[2] if c else 0 for c in cur]
            row = np.concatenate(([0], np.cumsum(n)[:-1])).astype(np.int64)
            try:
                done, status = eng.generate_mt_workers(g, n, row, sequence_type=sequence_type, gc_bias=gc_bias)
            except _native.EngineError as e:
                e.before_output = not began[0]  # (the first call reserves the stream buffers: ISS_E_NOMEM comes from there)
                raise
            began[0] = True
            scattered = []
            for k, c in enumerate(cur):
                if c is None:
                    continue
                if c[4]:
                    assert status[k] == _native.E_SHORT_RECORD and done[k] == 0, (k, int(status[k]), int(done[k]))
                    continue
                assert status[k] == 0 and done[k] == c[2], (k, int(status[k]), int(done[k]), c[2])
                if final:  # (the piece's bytes: pairs c[3] .. c[3] + c[2] - 1 of the work item)
                    scattered.append((c[0].id, c[3], int(row[k]), c[2], cpu_numbers[k], at[k]))
                    at[k] += fastq_text_bytes(c[0].id, c[3] + c[2], cpu_numbers[k], dense.read_length) - \
                        fastq_text_bytes(c[0].id, c[3], cpu_numbers[k], dense.read_length)
                else:
                    eng.fastq_emit(handles[k][0].fileno(), handles[k][1].fileno(), c[0].id, c[3], cpu_numbers[k], int(row[k]), c[2],
                                   n_threads=1)
            if scattered:  # ONE text job per round: the next round's kernels run beside its copy and its writes
                eng.fastq_emit_scatter(handles[0][0].fileno(), handles[0][1].fileno(), scattered, n_threads=1)
        eng.fastq_flush()
        if final and at != ends:  # every worker's text ends where the next one's starts
            raise RuntimeError("worker_set_iterator: a worker's text is not the size computed for it: %r / %r" % (at, ends))
        return final
    finally:
        eng.close()
        for fh3 in handles:
            for fh in fh3:
                fh.close()


def lognormal_abundance(record_ids, rng):
    """iss/abundance.py:137-154 with an explicit RandomState."""
    dist = rng.lognormal(size=len(record_ids))
    dist_scaled = dist / sum(dist)
    return {r: a for r, a in zip(record_ids, dist_scaled)}
