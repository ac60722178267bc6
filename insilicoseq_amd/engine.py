"""ReadEngine: one MI355X context = one reference worker (host-side driver over the C ABI).

Uploads the model tables and genomes once to HBM, launches the read-generation kernels for
(record, n_pairs) work items and brings the R1/R2 base + phred buffers back (or leaves them in
HBM for a device-side consumer).  Everything numeric happens in the HIP library; this file is
plumbing.  Reference counterparts: the body of ``worker_iterator`` / ``simulate_reads`` /
``reads_generator`` (iss/generator.py:223-251, 21-66, 69-95)."""
import ctypes as C

import numpy as np

from . import _native
from ._native import EngineError, SEQ_TYPES, check


MUT_DTYPE = np.dtype([("pair", "<i4"), ("mate", "i1"), ("type", "i1"), ("position", "<i2"), ("ref", "u1"),
                      ("alt", "u1"), ("quality", "<i2")], align=True)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class ReadEngine(object):
    def __init__(self, device=0):
        self._lib = _native.lib()
        self._ctx = C.c_void_p()
        check(None, self._lib.iss_ctx_create(int(device), C.byref(self._ctx)))
        self.device = int(device)
        self.read_length = None
        self.pitch = None
        self._capacity = 0
        self._genome_lengths = []

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.iss_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        return check(self._ctx, rc)

    # ------------------------------------------------------------------ uploads
    def load_model(self, dense):
        """Upload a DenseModel (insilicoseq_amd.model) -- replaces pickling the ErrorModel to the worker."""
        t = dense.device_tables()
        keep = {
            "isize_thr": _u64(t["isize_thr"]), "bin_thr": _u64(t["bin_thr"]), "q_thr": _u64(t["q_thr"]),
            "subst_thr": _u64(t["subst_thr"]), "ins_thr": _u64(t["ins_thr"]), "del_thr": _u64(t["del_thr"]),
            "mut_thr": _u64(t["mut_thr"]),
            "bin_nonempty": np.ascontiguousarray(dense.bin_nonempty, dtype=np.uint8),
            "subst_alt": np.ascontiguousarray(dense.subst_alt, dtype=np.uint8),
            "ins_letter": np.ascontiguousarray(dense.ins_letter, dtype=np.uint8),
        }
        mt = _native.ModelTables()
        mt.read_length = dense.read_length
        mt.n_isize = dense.n_isize
        mt.n_q = dense.n_q
        for k, v in keep.items():
            setattr(mt, k, v.ctypes.data)
        mt.quality_mode = int(getattr(dense, "quality_mode", 0))
        mt.basic_insert_size = int(getattr(dense, "basic_insert_size", 200))
        # np.random.normal(util.phred_to_prob(mean_quality), 0.01, read_length); min(q, 0.9999)  (basic.py:52)
        from .model import phred_to_prob
        mt.basic_mean = float(phred_to_prob(int(getattr(dense, "basic_mean_quality", 30))))
        mt.basic_sd = 0.01
        mt.basic_cap = 0.9999
        self._check(self._lib.iss_model_upload(self._ctx, C.byref(mt)))
        self.quality_mode = mt.quality_mode
        self.read_length = dense.read_length
        self.pitch = self._lib.iss_output_pitch(self._ctx)
        self._capacity = 0
        return self

    def add_genome(self, seq):
        """Upload one record's sequence (str / bytes / uint8 array); returns its genome id."""
        if isinstance(seq, str):
            seq = seq.encode("ascii")
        a = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.ascontiguousarray(
            seq, dtype=np.uint8)
        gid = C.c_int32(-1)
        self._check(self._lib.iss_genome_upload(self._ctx, a.ctypes.data, a.size, C.byref(gid)))
        self._genome_lengths.append(int(a.size))
        return gid.value

    def add_genome_packed(self, codes, length, device_ptr=None):
        """Upload a record of plain A/C/G/T given as 2-bit codes (see distributed.pack_2bit): ``codes`` a uint32 array
        on the host, or ``device_ptr`` the address of the words in this GPU's memory (a slice of the broadcast buffer)."""
        gid = C.c_int32(-1)
        if device_ptr is not None:
            self._check(self._lib.iss_genome_upload_packed(self._ctx, C.c_void_p(int(device_ptr)), int(length), 1, C.byref(gid)))
        else:
            a = np.ascontiguousarray(codes, dtype=np.uint32)
            assert a.size >= (int(length) + 15) // 16
            self._check(self._lib.iss_genome_upload_packed(self._ctx, a.ctypes.data, int(length), 0, C.byref(gid)))
        self._genome_lengths.append(int(length))
        return gid.value

    def clear_genomes(self):
        self._check(self._lib.iss_genome_clear(self._ctx))
        self._genome_lengths = []

    def genome_length(self, gid):
        return self._genome_lengths[gid]

    def reserve(self, n_pairs):
        if n_pairs > self._capacity:
            self._check(self._lib.iss_output_reserve(self._ctx, int(n_pairs)))
            self._capacity = int(n_pairs)

    # ------------------------------------------------------------------ the hot path
    def generate(self, genome_id, n_pairs, first_ordinal=0, seed=0, sequence_type="metagenomics", gc_bias=False,
                 out_first_pair=0):
        """Asynchronously generate n_pairs pairs into rows [out_first_pair, +n_pairs).  Raises
        EngineError(code=E_SHORT_RECORD) when read_length >= len(record) (the reference's
        AssertionError, iss/generator.py:130)."""
        if sequence_type not in SEQ_TYPES:
            raise ValueError("Sequence type %s not known" % sequence_type)  # generator.py:171
        self.reserve(out_first_pair + n_pairs)
        self._check(self._lib.iss_generate(self._ctx, int(genome_id), int(n_pairs), int(first_ordinal) & (2**64 - 1),
                                           int(seed) & (2**64 - 1), SEQ_TYPES[sequence_type], int(bool(gc_bias)),
                                           int(out_first_pair)))

    def generate_batch(self, genome_ids, n_pairs, first_ordinal=0, seed=0, sequence_type="metagenomics", gc_bias=False,
                       out_first_pair=0):
        """A whole work list -- items (genome_ids[k], n_pairs[k]) -- in one set of launches; the rows equal those of
        consecutive generate() calls with running ordinals and rows.  No custom fragment lengths here."""
        if sequence_type not in SEQ_TYPES:
            raise ValueError("Sequence type %s not known" % sequence_type)
        ids = np.ascontiguousarray(genome_ids, dtype=np.int32)
        cnt = np.ascontiguousarray(n_pairs, dtype=np.int64)
        assert ids.ndim == 1 and ids.shape == cnt.shape
        self.reserve(out_first_pair + int(cnt.sum()))
        self._check(self._lib.iss_generate_batch(self._ctx, int(ids.size), ids.ctypes.data, cnt.ctypes.data,
                                                 int(first_ordinal) & (2**64 - 1), int(seed) & (2**64 - 1),
                                                 SEQ_TYPES[sequence_type], int(bool(gc_bias)), int(out_first_pair)))

    # ------------------------------------------------------------------ the ErrorModel methods, batched (inner plugin surface)
    def gen_phred_scores(self, orientation, n, first_ordinal=0, seed=0):
        """KDErrorModel.gen_phred_scores for n reads (iss/error_models/kde.py:52-86): uint8 [n, read_length]; read i draws
        at ordinal first_ordinal + i of the worker stream `seed` -- the phreds iss_generate gives that pair's mate."""
        out = np.empty((int(n), self.read_length), dtype=np.uint8)
        self._check(self._lib.iss_gen_phred_scores(self._ctx, int(orientation), int(n), int(first_ordinal) & (2**64 - 1),
                                                   int(seed) & (2**64 - 1), out.ctypes.data))
        return out

    def mut_sequence(self, orientation, seqs, quals, first_ordinal=0, seed=0):
        """ErrorModel.mut_sequence (iss/error_models/__init__.py:69-112) for uint8 [n, read_length] letters and phreds:
        (mutated letters, status per read -- 2: the reference's KeyError)."""
        s = np.array(seqs, dtype=np.uint8, order="C", copy=True).reshape(-1, self.read_length)
        q = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1, self.read_length)
        assert s.shape == q.shape
        st = np.zeros(s.shape[0], dtype=np.int32)
        self._check(self._lib.iss_mut_sequence(self._ctx, int(orientation), s.shape[0], int(first_ordinal) & (2**64 - 1),
                                               int(seed) & (2**64 - 1), s.ctypes.data, q.ctypes.data, st.ctypes.data))
        return s, st

    def random_insert_size(self, n, first_ordinal=0, seed=0):
        """KDErrorModel.random_insert_size (iss/error_models/kde.py:88-98) for n pairs: int64 [n]."""
        out = np.empty(int(n), dtype=np.int64)
        self._check(self._lib.iss_random_insert_size(self._ctx, int(n), int(first_ordinal) & (2**64 - 1), int(seed) & (2**64 - 1),
                                                     out.ctypes.data))
        return out

    def introduce_indels(self, orientation, seqs, lengths, full_seq, bounds, first_ordinal=0, seed=0):
        """ErrorModel.introduce_indels incl. adjust_seq_length (iss/error_models/__init__.py:158-228, 114-156): seqs uint8
        [n, read_length] holding lengths[i] letters each (read direction), bounds int64 [n, 2] = (read_start, read_end) in
        full_seq.  Returns (uint8 [n, read_length], status per read: 0, 2 KeyError, 3 IndexError)."""
        s = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1, self.read_length)
        ln = np.ascontiguousarray(lengths, dtype=np.int32)
        b = np.ascontiguousarray(bounds, dtype=np.int64).reshape(-1, 2)
        g = np.frombuffer(full_seq.encode("ascii") if isinstance(full_seq, str) else bytes(full_seq), dtype=np.uint8)
        assert s.shape[0] == ln.size == b.shape[0]
        out = np.zeros_like(s)
        st = np.zeros(s.shape[0], dtype=np.int32)
        self._check(self._lib.iss_introduce_indels(self._ctx, int(orientation), s.shape[0], int(first_ordinal) & (2**64 - 1),
                                                   int(seed) & (2**64 - 1), s.ctypes.data, ln.ctypes.data, g.ctypes.data, g.size,
                                                   b.ctypes.data, out.ctypes.data, st.ctypes.data))
        return out, st

    def ev_step(self, orientation, cur, m53, v53=None):
        """One draw of the indel event process (the sampler behind the tests of introduce_indels,
        iss/error_models/__init__.py:193-196, :209) for arrays of (state, uniform numerator[, numerator of the deletion
        sub-draw]): (next state, slot that fires or -1, event mask).  A test hook (include/iss_mi355x.h: iss_ev_step)."""
        cur = np.ascontiguousarray(cur, dtype=np.int32)
        m53 = np.ascontiguousarray(m53, dtype=np.uint64)
        v53 = np.zeros_like(m53) if v53 is None else np.ascontiguousarray(v53, dtype=np.uint64)
        assert cur.shape == m53.shape == v53.shape
        nxt, slot, mask = np.zeros_like(cur), np.zeros_like(cur), np.zeros(cur.shape, dtype=np.uint8)
        self._check(self._lib.iss_ev_step(self._ctx, int(orientation), cur.size, cur.ctypes.data, m53.ctypes.data, v53.ctypes.data,
                                          nxt.ctypes.data, slot.ctypes.data, mask.ctypes.data))
        return nxt, slot, mask

    def set_fragment(self, fragment_length=None, fragment_sd=None):
        """Custom fragment length for generate() (None, None: the model's insert sizes)."""
        on = fragment_length is not None and fragment_sd is not None
        self._check(self._lib.iss_set_fragment(self._ctx, int(on), float(fragment_length or 0.0), float(fragment_sd or 0.0)))

    def mutations_reserve(self, capacity):
        """Enable (capacity > 0) / disable --store_mutations row capture of generate() (Philox path)."""
        self._check(self._lib.iss_mutations_reserve(self._ctx, int(capacity)))
        self._pmut_cap = int(capacity)

    def mutations(self):
        """Rows of the last generate() call, in the reference's order (structured array, see iss_mutation)."""
        cap = getattr(self, "_pmut_cap", 0)
        buf = getattr(self, "_pmut_buf", None)
        if buf is None or buf.size < cap:  # (one landing buffer per engine: tens of MB, not per call)
            buf = self._pmut_buf = np.empty(cap, dtype=MUT_DTYPE)
        n = C.c_int64(0)
        rc = self._lib.iss_mutations_download(self._ctx, buf.ctypes.data, cap, C.byref(n))
        self.mutation_slots_needed = n.value if rc == _native.E_NOMEM else 0  # (what a retry has to reserve)
        self._check(rc)
        return buf[: n.value].copy()

    @property
    def mutations_capacity(self):
        """Row slots reserved by mutations_reserve()."""
        return int(getattr(self, "_pmut_cap", 0))

    # ------------------------------------------------------------------ reference-compatible MT mode
    def seed_mt(self, seed):
        """random.seed(seed); np.random.seed(seed) -- on the device (iss/generator.py:234-236)."""
        self._check(self._lib.iss_mt_seed(self._ctx, int(seed)))

    def generate_mt(self, genome_id, n_pairs, sequence_type="metagenomics", gc_bias=False, out_first_pair=0):
        """Sequential, reference-identical generation; returns the number of pairs emitted."""
        if sequence_type not in SEQ_TYPES:
            raise ValueError("Sequence type %s not known" % sequence_type)
        self.reserve(out_first_pair + n_pairs)
        done = C.c_int64(0)
        self._check(self._lib.iss_generate_mt(self._ctx, int(genome_id), int(n_pairs), SEQ_TYPES[sequence_type],
                                              int(bool(gc_bias)), int(out_first_pair), C.byref(done)))
        return done.value

    def seed_mt_workers(self, seeds):
        """W reference workers side by side in this context: worker w == seed_mt(seeds[w]) in a context of its own
        (its two MT19937 streams; iss/generator.py:234-236 with seed + cpu_number)."""
        a = np.ascontiguousarray(seeds, dtype=np.uint64)
        self._check(self._lib.iss_mt_workers_seed(self._ctx, int(a.size), a.ctypes.data))
        self._mt_workers = int(a.size)

    def generate_mt_workers(self, genome_ids, n_pairs, out_first_pair, sequence_type="metagenomics", gc_bias=False):
        """One work item (or piece of one) per worker, every kernel of the path launched once for all workers: worker w
        generates n_pairs[w] pairs of genome genome_ids[w] into rows [out_first_pair[w], +n_pairs[w]) from ITS streams --
        the rows and stream positions of generate_mt per worker.  Returns (pairs emitted per worker, status per worker:
        0 or E_SHORT_RECORD)."""
        if sequence_type not in SEQ_TYPES:
            raise ValueError("Sequence type %s not known" % sequence_type)
        g = np.ascontiguousarray(genome_ids, dtype=np.int32)
        n = np.ascontiguousarray(n_pairs, dtype=np.int64)
        r = np.ascontiguousarray(out_first_pair, dtype=np.int64)
        if not (g.size == n.size == r.size == getattr(self, "_mt_workers", -1)):
            raise ValueError("generate_mt_workers: one genome id, pair count and first row per seeded worker")
        self.reserve(int((r + n).max()) if n.size else 0)
        done = np.zeros(n.size, dtype=np.int64)
        status = np.zeros(n.size, dtype=np.int32)
        self._check(self._lib.iss_generate_mt_workers(self._ctx, int(n.size), g.ctypes.data, n.ctypes.data, r.ctypes.data,
                                                      SEQ_TYPES[sequence_type], int(bool(gc_bias)), done.ctypes.data,
                                                      status.ctypes.data))
        return done, status

    def mt_workers_peek(self, worker, n=8):
        """mt_peek for worker ``worker`` of the set."""
        a = np.zeros(n, dtype=np.uint32)
        b = np.zeros(n, dtype=np.uint32)
        self._check(self._lib.iss_mt_workers_peek(self._ctx, int(worker), a.ctypes.data, b.ctypes.data, int(n)))
        return a, b

    def mt_set_fragment(self, fragment_length=None, fragment_sd=None):
        """Custom fragment length for generate_mt (None, None switches back to the model's insert sizes)."""
        on = fragment_length is not None and fragment_sd is not None
        self._check(self._lib.iss_mt_set_fragment(self._ctx, int(on), float(fragment_length or 0.0),
                                                  float(fragment_sd or 0.0)))

    def mt_mutations_reserve(self, capacity):
        """Enable (capacity > 0) / disable --store_mutations row capture of generate_mt."""
        self._check(self._lib.iss_mt_mutations_reserve(self._ctx, int(capacity)))
        self._mut_cap = int(capacity)

    def mt_mutations(self):
        """Rows of the last generate_mt call as a structured array (see iss_mutation in the C header)."""
        cap = getattr(self, "_mut_cap", 0)
        out = np.zeros(cap, dtype=MUT_DTYPE)
        n = C.c_int64(0)
        self._check(self._lib.iss_mt_mutations_download(self._ctx, out.ctypes.data, cap, C.byref(n)))
        if n.value > cap:
            raise EngineError(_native.E_INVALID, "mutation buffer too small: %d rows, capacity %d" % (n.value, cap))
        return out[: n.value]

    def mt_peek(self, n=8):
        """The next n 32-bit words of (CPython random, numpy) -- not consumed."""
        a = np.zeros(n, dtype=np.uint32)
        b = np.zeros(n, dtype=np.uint32)
        self._check(self._lib.iss_mt_peek(self._ctx, a.ctypes.data, b.ctypes.data, int(n)))
        return a, b

    def fastq_emit(self, fd_r1, fd_r2, record_id, first_i, cpu_number, first_pair, n_pairs, n_threads=1):
        """Format rows [first_pair, +n_pairs) as FASTQ on the device and append them to the two file
        descriptors (asynchronous; ``fastq_flush`` before the files are used)."""
        self._check(self._lib.iss_fastq_emit(self._ctx, int(fd_r1), int(fd_r2), str(record_id).encode(), int(first_i),
                                             int(cpu_number), int(first_pair), int(n_pairs), int(n_threads)))

    def fastq_emit_batch(self, fd_r1, fd_r2, items, cpu_number):
        """items: (record id, first pair id, first output row, pairs) per work item -- one text job for all of them."""
        n = len(items)
        ids = (C.c_char_p * n)(*[str(it[0]).encode() for it in items])
        first_i = np.array([it[1] for it in items], dtype=np.int64)
        first_pair = np.array([it[2] for it in items], dtype=np.int64)
        n_pairs = np.array([it[3] for it in items], dtype=np.int64)
        self._check(self._lib.iss_fastq_emit_batch(self._ctx, int(fd_r1), int(fd_r2), n, ids, first_i.ctypes.data,
                                                   first_pair.ctypes.data, n_pairs.ctypes.data, int(cpu_number)))

    def fastq_emit_scatter(self, fd_r1, fd_r2, items, n_threads=1):
        """items: (record id, first pair id, first output row, pairs, cpu number, byte offset in both files) -- one text job, every
        item's text written at its own place (the workers of a set straight into the final files; text mode only)."""
        n = len(items)
        if not n:
            return
        ids = (C.c_char_p * n)(*[str(it[0]).encode() for it in items])
        cols = [np.array([it[k] for it in items], dtype=np.int64) for k in (1, 2, 3, 5)]
        cpus = np.array([it[4] for it in items], dtype=np.int32)
        self._check(self._lib.iss_fastq_emit_scatter(self._ctx, int(fd_r1), int(fd_r2), n, ids, cols[0].ctypes.data, cols[1].ctypes.data,
                                                     cols[2].ctypes.data, cpus.ctypes.data, cols[3].ctypes.data, int(n_threads)))

    def fastq_compress(self, on=True):
        """`--compress` on the device: every fastq_emit appends one gzip member per file instead of text."""
        self._check(self._lib.iss_fastq_compress(self._ctx, 1 if on else 0))

    def fastq_flush(self):
        self._check(self._lib.iss_fastq_flush(self._ctx))

    def mt_path_counts(self):
        """(pairs resolved in parallel, pairs walked sequentially) by generate_mt so far."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.iss_mt_path_counts(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def synchronize(self):
        self._check(self._lib.iss_synchronize(self._ctx))

    def download(self, first_pair, n_pairs):
        """Rows [first_pair, +n_pairs) as four uint8 arrays [n_pairs, read_length] (views on pitch-wide rows)."""
        outs = [np.empty((n_pairs, self.pitch), dtype=np.uint8) for _ in range(4)]
        self._check(self._lib.iss_output_download(self._ctx, int(first_pair), int(n_pairs),
                                                  *[o.ctypes.data for o in outs]))
        RL = self.read_length
        return {"r1_base": outs[0][:, :RL], "r1_qual": outs[1][:, :RL], "r2_base": outs[2][:, :RL],
                "r2_qual": outs[3][:, :RL], "_pitched": outs}

    def coords(self, first_pair, n_pairs):
        c = np.empty((n_pairs, 4), dtype=np.int64)
        self._check(self._lib.iss_output_download_coords(self._ctx, int(first_pair), int(n_pairs), c.ctypes.data))
        return c

    def device_ptrs(self):
        p = [C.c_void_p() for _ in range(4)]
        self._check(self._lib.iss_output_device_ptrs(self._ctx, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.iss_ctx_set_stream(self._ctx, C.c_void_p(hip_stream_ptr)))

    # ------------------------------------------------------------------ measurement
    def timing_enable(self, on=True):
        """False / 0: off; True / 1: HIP events around every kernel; 2: around k_main only (cheaper)."""
        self._check(self._lib.iss_timing_enable(self._ctx, int(on)))

    def timing_read(self):
        ms = (C.c_double * 4)()
        n = C.c_int64(0)
        self._check(self._lib.iss_timing_read(self._ctx, C.byref(ms), C.byref(n)))
        return {"setup_ms": ms[0], "main_ms": ms[1], "indel_scan_ms": ms[2], "indel_fixup_ms": ms[3],
                "launches": n.value}

    def main_kernel(self):
        """Name of the kernel the last generate() / generate_batch() launched for the hot path ("k_main<...>" / "k_main_g<NI, NP>")."""
        buf = C.create_string_buffer(64)
        self._check(self._lib.iss_main_kernel(self._ctx, buf, 64))
        return buf.value.decode()

    def stats_read(self):
        n, m = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.iss_stats_read(self._ctx, C.byref(n), C.byref(m)))
        return {"fixup_reads": n.value, "scripted_reads": m.value}


def fastq_write(fd_r1, fd_r2, record_id, first_i, cpu_number, n_pairs, read_length, pitch, r1_base, r1_qual, r2_base,
                r2_qual, n_threads=4):
    """FASTQ text for rows of pitched uint8 arrays (SeqIO.write(..., 'fastq-sanger'), generator.py:64-65)."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (r1_base, r1_qual, r2_base, r2_qual)]
    rc = _native.lib().iss_fastq_write(int(fd_r1), int(fd_r2), str(record_id).encode(), int(first_i), int(cpu_number),
                                       int(n_pairs), int(read_length), int(pitch), *[a.ctypes.data for a in arrs],
                                       int(n_threads))
    check(None, rc)


__all__ = ["ReadEngine", "EngineError", "fastq_write"]
