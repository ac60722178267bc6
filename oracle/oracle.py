"""ctypes front-end of oracle/iss_oracle.c (the CPU restatement used as the parity checker).

TEST INFRASTRUCTURE ONLY -- see the header of iss_oracle.c for the reference file:line map and
how the oracle itself is pinned to the reference."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libiss_oracle.so")

OK, SKIP_RECORD, ERR_KEY, ERR_INDEX, ERR_UNSUPPORTED = 0, 1, 2, 3, 4


def build(force=False):
    src = os.path.join(_HERE, "iss_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libiss_oracle.so"])
    return _LIB_PATH


class _Model(C.Structure):
    _fields_ = [
        ("read_length", C.c_int32), ("n_isize", C.c_int32), ("n_q", C.c_int32), ("quality_mode", C.c_int32),
        ("isize_cdf", C.c_void_p), ("bin_cdf", C.c_void_p), ("qcdf", C.c_void_p), ("subst_cdf", C.c_void_p),
        ("subst_alt", C.c_void_p), ("ins", C.c_void_p), ("ins_letter", C.c_void_p), ("dele", C.c_void_p),
        ("phred_thr", C.c_void_p), ("basic_insert_size", C.c_int32), ("basic_mean_quality", C.c_int32),
    ]


class _RunParams(C.Structure):
    _fields_ = [("sequence_type", C.c_int32), ("has_fragment", C.c_int32), ("fragment_length", C.c_double),
                ("fragment_sd", C.c_double), ("gc_bias", C.c_int32)]


MUT_DTYPE = np.dtype([("pair", "<i4"), ("mate", "i1"), ("type", "i1"), ("position", "<i2"), ("ref", "u1"),
                      ("alt", "u1"), ("quality", "<i2")], align=True)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.iss_oracle_rng_new.restype = C.c_void_p
        L.iss_oracle_rng_free.argtypes = [C.c_void_p]
        L.iss_oracle_rng_seed_mt.argtypes = [C.c_void_p, C.c_uint64]
        L.iss_oracle_rng_seed_py.argtypes = [C.c_void_p, C.c_uint64]
        L.iss_oracle_rng_seed_np.argtypes = [C.c_void_p, C.c_uint32]
        L.iss_oracle_rng_seed_philox.argtypes = [C.c_void_p, C.c_uint64]
        L.iss_oracle_rng_set_address.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        for f in ("iss_oracle_py_random", "iss_oracle_np_random"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("iss_oracle_py_word", "iss_oracle_np_word"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("iss_oracle_py_words_used", "iss_oracle_np_words_used"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.iss_oracle_np_normal.restype = C.c_double
        L.iss_oracle_np_normal.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.iss_oracle_py_randbelow.restype = C.c_uint64
        L.iss_oracle_py_randbelow.argtypes = [C.c_void_p, C.c_uint64]
        L.iss_oracle_philox4x32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.iss_oracle_simulate.restype = C.c_int
        L.iss_oracle_simulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                          C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.iss_oracle_introduce_indels.restype = C.c_int
        L.iss_oracle_introduce_indels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        L.iss_oracle_gen_phred_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.iss_oracle_mut_sequence.restype = C.c_int
        L.iss_oracle_mut_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.iss_oracle_random_insert_size.restype = C.c_int64
        L.iss_oracle_random_insert_size.argtypes = [C.c_void_p, C.c_void_p]
        L.iss_oracle_rev_comp.restype = C.c_int
        L.iss_oracle_rev_comp.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def philox4x32(ctr, key, rounds=10):
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    lib().iss_oracle_philox4x32(c.ctypes.data, k.ctypes.data, int(rounds), out.ctypes.data)
    return out


def philox4x32_10(ctr, key):
    return philox4x32(ctr, key, 10)


def _as_bytes(seq):
    if isinstance(seq, np.ndarray) and seq.dtype == np.uint8 and seq.flags["C_CONTIGUOUS"]:
        return seq  # (as it is: a record of 2^31 bases is not copied twice)
    if isinstance(seq, str):
        seq = seq.encode("ascii")
    return np.frombuffer(bytes(seq), dtype=np.uint8).copy()


class Rng(object):
    """The uniform-stream provider: two MT19937 streams (reference-compatible) or Philox."""

    def __init__(self):
        self._h = lib().iss_oracle_rng_new()

    def __del__(self):
        try:
            lib().iss_oracle_rng_free(self._h)
        except Exception:
            pass

    def seed_mt(self, seed):
        """random.seed(seed); np.random.seed(seed)"""
        lib().iss_oracle_rng_seed_mt(self._h, int(seed))
        return self

    def seed_py(self, seed):
        lib().iss_oracle_rng_seed_py(self._h, int(seed))
        return self

    def seed_np(self, seed):
        lib().iss_oracle_rng_seed_np(self._h, int(seed))
        return self

    def seed_philox(self, seed):
        lib().iss_oracle_rng_seed_philox(self._h, int(seed) & (2**64 - 1))
        return self

    def set_address(self, ordinal, attempt=0):
        lib().iss_oracle_rng_set_address(self._h, int(ordinal), int(attempt))

    def py_random(self):
        return lib().iss_oracle_py_random(self._h)

    def np_random(self):
        return lib().iss_oracle_np_random(self._h)

    def py_word(self):
        return lib().iss_oracle_py_word(self._h)

    def np_word(self):
        return lib().iss_oracle_np_word(self._h)

    def np_normal(self, loc, scale):
        return lib().iss_oracle_np_normal(self._h, loc, scale)

    def py_randbelow(self, n):
        return lib().iss_oracle_py_randbelow(self._h, int(n))

    def words_used(self):
        return (lib().iss_oracle_py_words_used(self._h), lib().iss_oracle_np_words_used(self._h))


class Oracle(object):
    """The semantic function on dense tables (insilicoseq_amd.model.DenseModel-shaped object)."""

    def __init__(self, dense, quality_mode=None, basic_insert_size=None, basic_mean_quality=None):
        # defaults: what the dense model says (DenseModel.basic() marks itself)
        quality_mode = getattr(dense, "quality_mode", 0) if quality_mode is None else quality_mode
        basic_insert_size = getattr(dense, "basic_insert_size", 200) if basic_insert_size is None else basic_insert_size
        basic_mean_quality = (getattr(dense, "basic_mean_quality", 30) if basic_mean_quality is None
                              else basic_mean_quality)
        self.d = dense
        self._keep = [dense.isize_cdf, dense.bin_cdf, dense.qcdf, dense.subst_cdf, dense.subst_alt, dense.ins,
                      dense.ins_letter, dense.dele, dense.phred_thr]
        m = _Model()
        m.read_length = dense.read_length
        m.n_isize = dense.isize_cdf.shape[0]
        m.n_q = dense.qcdf.shape[3]
        m.quality_mode = quality_mode
        m.isize_cdf, m.bin_cdf, m.qcdf, m.subst_cdf, m.subst_alt, m.ins, m.ins_letter, m.dele, m.phred_thr = [
            a.ctypes.data for a in self._keep]
        m.basic_insert_size = basic_insert_size
        m.basic_mean_quality = basic_mean_quality
        self._m = m
        self.read_length = dense.read_length

    def simulate(self, rng, genome, n_pairs, first_ordinal=0, sequence_type="metagenomics", fragment_length=None,
                 fragment_sd=None, gc_bias=False, pitch=None, store_mutations=False, want_coords=False):
        g = _as_bytes(genome)
        RL = self.read_length
        pitch = RL if pitch is None else pitch
        rp = _RunParams()
        rp.sequence_type = {"metagenomics": 0, "amplicon": 1}[sequence_type]
        rp.has_fragment = int(fragment_length is not None and fragment_sd is not None)
        rp.fragment_length = float(fragment_length or 0.0)
        rp.fragment_sd = float(fragment_sd or 0.0)
        rp.gc_bias = int(bool(gc_bias))
        outs = [np.zeros((n_pairs, pitch), dtype=np.uint8) for _ in range(4)]
        coords = np.zeros((n_pairs, 4), dtype=np.int64) if want_coords else None
        mut_cap = (4 * RL * 2 * max(n_pairs, 1)) if store_mutations else 0
        muts = np.zeros(mut_cap, dtype=MUT_DTYPE) if store_mutations else None
        n_mut = C.c_int64(0)
        n_done = C.c_int64(0)
        rc = lib().iss_oracle_simulate(
            C.byref(self._m), rng._h, C.byref(rp), g.ctypes.data, len(g), n_pairs, int(first_ordinal), pitch,
            outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, outs[3].ctypes.data,
            coords.ctypes.data if want_coords else None, muts.ctypes.data if store_mutations else None, mut_cap,
            C.byref(n_mut), C.byref(n_done))
        res = {"status": rc, "n_done": n_done.value, "r1_base": outs[0], "r1_qual": outs[1], "r2_base": outs[2],
               "r2_qual": outs[3]}
        if want_coords:
            res["coords"] = coords
        if store_mutations:
            res["mutations"] = muts[: min(n_mut.value, mut_cap)]
        return res

    def introduce_indels(self, rng, seq, orientation, genome, bounds):
        s, g = _as_bytes(seq), _as_bytes(genome)
        out = np.zeros(self.read_length, dtype=np.uint8)
        rc = lib().iss_oracle_introduce_indels(C.byref(self._m), rng._h, int(orientation), s.ctypes.data, len(s),
                                               g.ctypes.data, len(g), bounds[0], bounds[1], out.ctypes.data)
        return rc, out.tobytes().decode("ascii")

    def indel_event_masks(self, rng, orientation, ordinals):
        """Position-addressable mode: the event masks (bits 0-3 insertion of letter slot x fires, bits 4-7 the deletion
        fires if the token is base b) of the reads at the given ordinals: [len(ordinals)][read_length]."""
        L = lib()
        L.iss_oracle_indel_event_masks.restype = C.c_int
        L.iss_oracle_indel_event_masks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        out = np.zeros((len(ordinals), self.read_length), dtype=np.uint8)
        for i, k in enumerate(ordinals):
            rng.set_address(int(k))
            assert L.iss_oracle_indel_event_masks(C.byref(self._m), rng._h, int(orientation), out[i].ctypes.data) == 0
        return out

    def ev_step(self, orientation, cur, m53, v53=None):
        """One draw of the indel event process (iss_oracle.c: ev_step) for arrays of (state, uniform numerator[, numerator
        of the deletion sub-draw]): (next state, fired slot or -1, event mask)."""
        L = lib()
        L.iss_oracle_ev_step.restype = C.c_int
        L.iss_oracle_ev_step.argtypes = [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 6
        cur = np.ascontiguousarray(cur, dtype=np.int32)
        m53 = np.ascontiguousarray(m53, dtype=np.uint64)
        v53 = np.zeros_like(m53) if v53 is None else np.ascontiguousarray(v53, dtype=np.uint64)
        assert cur.shape == m53.shape == v53.shape
        nxt, slot, mask = np.zeros_like(cur), np.zeros_like(cur), np.zeros(cur.shape, dtype=np.uint8)
        rc = L.iss_oracle_ev_step(C.byref(self._m), int(orientation), cur.size, cur.ctypes.data, m53.ctypes.data, v53.ctypes.data,
                                  nxt.ctypes.data, slot.ctypes.data, mask.ctypes.data)
        assert rc == 0, "state or numerator out of range"
        return nxt, slot, mask

    def ev_segments(self, orientation):
        """Last slot of every slot's segment of the sampler's survival tables ([5 (RL - 1)])."""
        L = lib()
        L.iss_oracle_ev_segments.restype = C.c_int
        L.iss_oracle_ev_segments.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        out = np.zeros(5 * max(self.read_length - 1, 0), dtype=np.int32)
        assert L.iss_oracle_ev_segments(C.byref(self._m), int(orientation), out.ctypes.data) == out.size
        return out

    def gen_phred_scores(self, rng, orientation):
        q = np.zeros(self.read_length, dtype=np.uint8)
        lib().iss_oracle_gen_phred_scores(C.byref(self._m), rng._h, int(orientation), q.ctypes.data)
        return q

    def mut_sequence(self, rng, seq, qual, orientation):
        s = _as_bytes(seq)
        q = np.ascontiguousarray(qual, dtype=np.uint8)
        assert len(s) == self.read_length and len(q) == self.read_length
        rc = lib().iss_oracle_mut_sequence(C.byref(self._m), rng._h, int(orientation), s.ctypes.data, q.ctypes.data)
        return rc, s.tobytes().decode("ascii")

    def random_insert_size(self, rng):
        return lib().iss_oracle_random_insert_size(C.byref(self._m), rng._h)


def rev_comp(s):
    a = _as_bytes(s)
    out = np.zeros_like(a)
    rc = lib().iss_oracle_rev_comp(a.ctypes.data, len(a), out.ctypes.data)
    if rc:
        raise KeyError("non-IUPAC letter")
    return out.tobytes().decode("ascii")
