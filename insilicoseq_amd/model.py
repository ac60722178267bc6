"""Error-model tables: the reference's pickled ``.npz`` schema -> dense, pickle-free arrays.

Reference behaviour mirrored here (InSilicoSeq v2.0.1):

* ``ErrorModel.load_npz``                 iss/error_models/__init__.py:27-50
* ``KDErrorModel.__init__`` (13 keys)     iss/error_models/kde.py:24-50
* schema written by ``bam.write_to_file`` iss/bam.py:82-97
* bin choice normalisation                iss/error_models/kde.py:72-74
* ``np.random.choice(p=)`` normalisation  (numpy legacy: ``cdf = p.cumsum(); cdf /= cdf[-1]``)
* ``util.phred_to_prob``                  iss/util.py:16-29

The dense layout is what is uploaded to HBM; every derived
table (cumsum/normalise, ``1 - 10**(-q/10)``) is evaluated HERE on the host with numpy
in the reference's expression order and never recomputed on the device.
"""
import logging
import sys

import numpy as np

BASES = "ATCG"  # dict order in every shipped model; 2-bit code = index, complement = code ^ 1
_BASE_INDEX = {b: i for i, b in enumerate(BASES)}
N_BINS = 4
TWO53 = float(2**53)


class ModelError(ValueError):
    """The model file is readable but cannot be used by the engine."""


def phred_to_prob(q):
    """``1 - 10 ** (-q / 10)`` with q a numpy int64, as the reference evaluates it
    (searchsorted returns np.int64; iss/util.py:28-29)."""
    p = 10 ** (-np.int64(q) / 10)
    return 1 - p


def basic_prob_to_phred(x, cap=0.9999):
    """One phred score of ``BasicErrorModel.gen_phred_scores`` from its normal deviate ``x``:
    ``prob_to_phred(min(x, 0.9999))`` = ``int(round(-10 * np.log10(1 - p)))`` (iss/error_models/basic.py:52-53,
    iss/util.py:44)."""
    return int(round(-10 * np.log10(1 - min(x, cap))))


def basic_phred_cdf(mean_quality=30, sd=0.01, cap=0.9999, n_q=41):
    """``P(phred <= k)``, k = 0 .. n_q-1, of one position of ``BasicErrorModel.gen_phred_scores``
    (iss/error_models/basic.py:52-53): x ~ Normal(phred_to_prob(mean_quality), sd), phred = basic_prob_to_phred(x).
    The score is a non-decreasing step function of x, so ``P(phred <= k) = Phi((b_k - mean) / sd)`` with b_k the
    largest double whose score is <= k -- found by bisection on the reference's own expression (not on its algebraic
    inverse ``1 - 10 ** (-(k + 0.5) / 10)``: the steps are where the FLOAT expression rounds).  This is the table the
    position-addressable (Philox) path inverts, like a KDE quality row; the reference-compatible mode draws the normal
    deviates themselves."""
    import math
    mean = float(phred_to_prob(mean_quality))
    top = basic_prob_to_phred(cap, cap)  # the score of every x >= cap (40)
    cdf = np.ones(n_q, dtype=np.float64)
    for k in range(min(n_q, top)):
        lo, hi = -1.0, float(cap)  # score(lo) = 0 <= k < score(hi) = top (-1.0: 1 - x = 2, log10 > 0, rounds to -3 .. 0)
        if basic_prob_to_phred(lo, cap) > k:
            cdf[k] = 0.0
            continue
        while True:
            mid = 0.5 * (lo + hi)
            if mid <= lo or mid >= hi:
                break
            if basic_prob_to_phred(mid, cap) <= k:
                lo = mid
            else:
                hi = mid
        cdf[k] = 0.5 * math.erfc(-((lo - mean) / sd) / math.sqrt(2.0)) if sd > 0 else float(lo >= mean)
    return cdf


def _choice_cdf(p):
    """The CDF ``np.random.choice(a, p=p)`` inverts (legacy RandomState.choice)."""
    p = np.array(p, dtype=np.float64)
    if p.ndim != 1 or p.size == 0:
        raise ModelError("probabilities must be a non-empty 1-d vector")
    if np.isnan(p).any() or (p < 0).any():
        raise ModelError("probabilities contain NaN or negative entries (np.random.choice raises)")
    if abs(float(np.sum(p)) - 1.0) > np.sqrt(np.finfo(np.float64).eps):
        raise ModelError("probabilities do not sum to 1 (np.random.choice raises)")
    cdf = p.cumsum()
    cdf /= cdf[-1]
    return cdf


class DenseModel(object):
    """Dense f64/u8 tables of one KDE error model (orientation 0 = forward, 1 = reverse)."""

    FIELDS = (
        "isize_cdf", "bin_cdf", "bin_nonempty", "qcdf", "subst_cdf", "subst_alt",
        "ins", "ins_letter", "dele", "phred_thr",
    )

    def __init__(self, read_length, isize_cdf, bin_cdf, bin_nonempty, qcdf, subst_cdf, subst_alt, ins,
                 ins_letter, dele, phred_thr):
        self.read_length = int(read_length)
        self.isize_cdf = np.ascontiguousarray(isize_cdf, dtype=np.float64)
        self.bin_cdf = np.ascontiguousarray(bin_cdf, dtype=np.float64)
        self.bin_nonempty = np.ascontiguousarray(bin_nonempty, dtype=np.uint8)
        self.qcdf = np.ascontiguousarray(qcdf, dtype=np.float64)
        self.subst_cdf = np.ascontiguousarray(subst_cdf, dtype=np.float64)
        self.subst_alt = np.ascontiguousarray(subst_alt, dtype=np.uint8)
        self.ins = np.ascontiguousarray(ins, dtype=np.float64)
        self.ins_letter = np.ascontiguousarray(ins_letter, dtype=np.uint8)
        self.dele = np.ascontiguousarray(dele, dtype=np.float64)
        self.phred_thr = np.ascontiguousarray(phred_thr, dtype=np.float64)
        # 0: KDE tables.  1: BasicErrorModel (iss/error_models/basic.py) -- constant insert size (no draw), phred scores
        # from a normal distribution around phred_to_prob(mean quality): the reference-compatible mode draws the normal
        # deviates, the Philox path inverts the quality rows (basic_phred_cdf); bin / insert-size tables are unused.
        self.quality_mode = 0
        self.basic_insert_size = 200
        self.basic_mean_quality = 30
        self.validate()

    # ------------------------------------------------------------------ shape
    @property
    def n_q(self):
        return int(self.qcdf.shape[3])

    @property
    def n_isize(self):
        return int(self.isize_cdf.shape[0])

    def validate(self):
        RL = self.read_length
        if RL < 2:
            raise ModelError("read_length must be >= 2")
        if self.bin_cdf.shape != (2, N_BINS) or self.bin_nonempty.shape != (2, N_BINS):
            raise ModelError("bin tables must be [2][4]")
        if self.qcdf.ndim != 4 or self.qcdf.shape[:3] != (2, N_BINS, RL):
            raise ModelError("qcdf must be [2][4][read_length][n_q]")
        if self.n_q < 1 or self.n_q > 255:
            raise ModelError("per-position quality CDFs must have 1..255 entries")
        if self.subst_cdf.shape != (2, RL, 4, 3) or self.subst_alt.shape != (2, RL, 4, 3):
            raise ModelError("substitution tables must be [2][read_length][4][3]")
        for name in ("ins", "ins_letter", "dele"):
            if getattr(self, name).shape != (2, RL, 4):
                raise ModelError("%s must be [2][read_length][4]" % name)
        if self.phred_thr.shape != (self.n_q + 1,):
            raise ModelError("phred_thr must have n_q + 1 entries")
        if self.isize_cdf.ndim != 1 or self.isize_cdf.size < 1:
            raise ModelError("insert-size CDF must be a non-empty vector")
        # np.searchsorted on a non-monotone array is a binary search with arbitrary (if
        # deterministic) results; the engine inverts CDFs by counting, so insist on monotone.
        if np.isnan(self.isize_cdf).any() or (np.diff(self.isize_cdf) < 0).any():
            raise ModelError("insert-size CDF is not monotone")
        for o in range(2):
            for b in range(N_BINS):
                if self.bin_nonempty[o, b]:
                    rows = self.qcdf[o, b]
                    if np.isnan(rows).any() or (np.diff(rows, axis=1) < 0).any():
                        raise ModelError("quality CDF (orientation %d, bin %d) is not monotone" % (o, b))
            prob = np.diff(np.concatenate(([0.0], self.bin_cdf[o])))
            if ((prob > 0) & (self.bin_nonempty[o] == 0)).any():
                raise ModelError("a mean-quality bin with non-zero probability has no histograms")

    # ------------------------------------------------- reference schema -> dense
    @classmethod
    def from_attributes(cls, read_length, i_size_cdf, mean_forward, mean_reverse, quality_forward,
                        quality_reverse, subst_choices_for, subst_choices_rev, ins_for, ins_rev, del_for,
                        del_rev):
        """Build from objects shaped like ``KDErrorModel``'s attributes (kde.py:30-48)."""
        RL = int(read_length)
        isize = np.asarray(i_size_cdf, dtype=np.float64)
        bin_cdf = np.zeros((2, N_BINS))
        nonempty = np.zeros((2, N_BINS), dtype=np.uint8)
        n_q = None
        hists = (quality_forward, quality_reverse)
        for o in range(2):
            if len(hists[o]) != N_BINS:
                raise ModelError("quality_hist must hold %d mean-quality bins" % N_BINS)
            for b in range(N_BINS):
                rows = hists[o][b]
                if len(rows) == 0:
                    continue
                if len(rows) != RL:
                    raise ModelError("quality_hist bin has %d positions, read_length is %d" % (len(rows), RL))
                nonempty[o, b] = 1
                for row in rows:
                    n = len(row)
                    if n_q is None:
                        n_q = n
                    elif n != n_q:
                        raise ModelError("per-position quality CDFs differ in length")
        if n_q is None:
            raise ModelError("model has no quality histograms at all")
        qcdf = np.ones((2, N_BINS, RL, n_q))
        for o, mean in enumerate((mean_forward, mean_reverse)):
            mean = np.asarray(mean)
            if mean.shape != (N_BINS,):
                raise ModelError("mean_count must have %d entries" % N_BINS)
            norm_mean = mean / sum(mean)  # kde.py:72 (builtin sum, then true division)
            bin_cdf[o] = _choice_cdf(norm_mean)  # kde.py:74
            for b in range(N_BINS):
                if nonempty[o, b]:
                    qcdf[o, b] = np.array([np.asarray(r, dtype=np.float64) for r in hists[o][b]])
        subst_cdf = np.zeros((2, RL, 4, 3))
        subst_alt = np.zeros((2, RL, 4, 3), dtype=np.uint8)
        ins = np.zeros((2, RL, 4))
        ins_letter = np.zeros((2, RL, 4), dtype=np.uint8)
        dele = np.zeros((2, RL, 4))
        for o, (sub, i_, d_) in enumerate(((subst_choices_for, ins_for, del_for),
                                            (subst_choices_rev, ins_rev, del_rev))):
            if len(sub) < RL or len(i_) < RL or len(d_) < RL:
                raise ModelError("substitution/indel tables shorter than read_length")
            for p in range(RL):
                for base, bi in _BASE_INDEX.items():
                    letters, probs = sub[p][base]
                    if len(letters) != 3 or len(probs) != 3:
                        raise ModelError("substitution choices must have 3 alternatives")
                    subst_cdf[o, p, bi] = _choice_cdf(probs)
                    subst_alt[o, p, bi] = [ord(str(x)) for x in letters]
                    dele[o, p, bi] = float(d_[p][base])
                items = list(i_[p].items())  # iteration order is the draw order (__init__.py:193)
                if sorted(str(k) for k, _ in items) != sorted(BASES):
                    raise ModelError("insertion table keys must be A, T, C, G")
                for x, (letter, prob) in enumerate(items):
                    ins[o, p, x] = float(prob)
                    ins_letter[o, p, x] = ord(str(letter))
        thr = np.array([phred_to_prob(q) for q in range(n_q + 1)], dtype=np.float64)
        return cls(RL, isize, bin_cdf, nonempty, qcdf, subst_cdf, subst_alt, ins, ins_letter, dele, thr)

    @classmethod
    def from_reference_npz(cls, npz_path):
        """Load the reference's pickled-object ``.npz`` (np.load(allow_pickle=True), __init__.py:40)."""
        prof = np.load(npz_path, allow_pickle=True)
        if str(prof["model"]) != "kde":
            raise ModelError("Trying to load a %s ErrorModel in kde mode" % prof["model"])
        return cls.from_attributes(
            prof["read_length"], prof["insert_size"], prof["mean_count_forward"], prof["mean_count_reverse"],
            prof["quality_hist_forward"], prof["quality_hist_reverse"], prof["subst_choices_forward"],
            prof["subst_choices_reverse"], prof["ins_forward"], prof["ins_reverse"], prof["del_forward"],
            prof["del_reverse"],
        )

    @classmethod
    def basic(cls, read_length=125, insert_size=200, mean_quality=30):
        """The tables of ``BasicErrorModel`` (iss/error_models/basic.py:18-38): read length 125, insert size 200,
        every substitution equally likely (``np.random.choice`` over [1/3, 1/3, 1/3]), no indels."""
        RL = int(read_length)
        cdf = _choice_cdf([1 / 3, 1 / 3, 1 / 3])
        alts = {"A": "TCG", "T": "ACG", "C": "ATG", "G": "ATC"}  # basic.py:27-32
        subst_alt = np.zeros((2, RL, 4, 3), dtype=np.uint8)
        for bi, b in enumerate(BASES):
            subst_alt[:, :, bi, :] = [ord(c) for c in alts[b]]
        # the quality rows: the distribution of one basic phred score, the same at every position, in every bin and for
        # both mates (inverted by the Philox path; the reference-compatible mode draws np.random.normal itself)
        qrow = basic_phred_cdf(mean_quality)
        d = cls(RL, np.array([1.0]), np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (2, 1)),
                np.tile(np.array([0, 0, 0, 1], dtype=np.uint8), (2, 1)), np.tile(qrow, (2, N_BINS, RL, 1)),
                np.tile(cdf, (2, RL, 4, 1)), subst_alt, np.zeros((2, RL, 4)),
                np.tile(np.frombuffer(BASES.encode(), dtype=np.uint8), (2, RL, 1)), np.zeros((2, RL, 4)),
                np.array([phred_to_prob(q) for q in range(42)]))
        d.quality_mode = 1
        d.basic_insert_size = int(insert_size)
        d.basic_mean_quality = int(mean_quality)
        return d

    # ----------------------------------------------------- dense (pickle-free) io
    def save(self, path):
        np.savez_compressed(
            path, format=np.array("iss-dense-1"), read_length=np.int64(self.read_length),
            quality_mode=np.int64(self.quality_mode), basic_insert_size=np.int64(self.basic_insert_size),
            basic_mean_quality=np.int64(self.basic_mean_quality),
            **{k: getattr(self, k) for k in self.FIELDS}
        )

    @classmethod
    def load(cls, path):
        d = np.load(path, allow_pickle=False)
        if "format" not in d.files or str(d["format"]) != "iss-dense-1":
            raise ModelError("%s is not a dense iss model" % path)
        m = cls(int(d["read_length"]), *[d[k] for k in cls.FIELDS])
        if "quality_mode" in d.files:
            m.quality_mode = int(d["quality_mode"])
            m.basic_insert_size = int(d["basic_insert_size"])
            m.basic_mean_quality = int(d["basic_mean_quality"])
        return m

    @classmethod
    def load_any(cls, path):
        """Dense file if it is one, else the reference's pickled schema."""
        try:
            d = np.load(path, allow_pickle=False)
            if "format" in d.files:
                return cls.load(path)
        except ValueError:
            pass  # object arrays -> reference schema
        return cls.from_reference_npz(path)

    # -------------------------------------------------------- integer thresholds
    def expected_mutation_rows_per_pair(self):
        """Expected --store_mutations rows of one read pair: substitutions (sum over positions of
        P(phred) * 10**(-phred/10), bins weighted by their probability) + insertions + deletions (an upper estimate: the
        largest deletion probability of a position).  Sizes the device's row buffers (generator.py:worker_iterator)."""
        total = 0.0
        if int(getattr(self, "quality_mode", 0)) == 1:  # BasicErrorModel: phreds around basic_mean_quality, floor 20 (basic.py:52)
            total = 2.0 * self.read_length * 10.0 ** (-(int(getattr(self, "basic_mean_quality", 30)) - 10) / 10.0)
        for o in range(2):
            if int(getattr(self, "quality_mode", 0)) == 1:
                ins = np.nan_to_num(np.clip(self.ins[o], 0.0, 1.0))
                dele = np.nan_to_num(np.clip(self.dele[o], 0.0, 1.0))
                total += float(ins.sum() + dele.max(axis=1).sum())
                continue
            w = np.diff(np.concatenate(([0.0], self.bin_cdf[o])))
            for b in range(4):
                if not self.bin_nonempty[o][b] or w[b] <= 0:
                    continue
                pm = np.diff(np.concatenate((np.zeros((self.read_length, 1)), self.qcdf[o, b]), axis=1), axis=1)
                rest = 1.0 - self.qcdf[o, b][:, -1]  # phred n_q
                total += w[b] * float((pm * (1.0 - self.phred_thr[:self.n_q])[None, :]).sum() + (rest * (1.0 - self.phred_thr[self.n_q])).sum())
            ins = np.nan_to_num(np.clip(self.ins[o], 0.0, 1.0))
            dele = np.nan_to_num(np.clip(self.dele[o], 0.0, 1.0))
            total += float(ins.sum() + dele.max(axis=1).sum())
        return total

    def device_tables(self):
        """Integer restatement of every f64 comparison on the path ("integer thresholds").

        A uniform is u = m / 2**53 with m a 53-bit integer, so for a table value c (f64):
          c <  u  <=>  floor(c * 2**53) <  m      (searchsorted side='left', ``u > thr``)
          c <= u  <=>  ceil (c * 2**53) <= m      (searchsorted side='right')
          u <  c  <=>  m < ceil(c * 2**53)        (indel tests)
        c * 2**53 is exact in f64 (power-of-two scaling), so floor/ceil are exact.
        """
        def fl(x):
            x = np.clip(np.nan_to_num(np.asarray(x, dtype=np.float64), nan=0.0), 0.0, 1.0)
            return np.floor(x * TWO53).astype(np.uint64)

        def ce(x):
            x = np.clip(np.nan_to_num(np.asarray(x, dtype=np.float64), nan=0.0), 0.0, 1.0)
            return np.ceil(x * TWO53).astype(np.uint64)

        return {
            "isize_thr": fl(self.isize_cdf),            # count(thr < m)
            "bin_thr": ce(self.bin_cdf),                # count(thr <= m)
            "q_thr": fl(self.qcdf),                     # count(thr < m)
            "subst_thr": ce(self.subst_cdf),            # count(thr <= m)
            "ins_thr": ce(self.ins),                    # event iff m < thr (NaN/<=0 -> never)
            "del_thr": ce(self.dele),
            "mut_thr": fl(self.phred_thr),              # event iff m > thr
        }


class KDErrorModel(object):
    """Host-side mirror of ``iss.error_models.kde.KDErrorModel`` (kde.py:9-98).

    Same constructor signature, attribute names and error behaviour: unreadable file or a
    non-"kde" model logs an error and ``sys.exit(1)`` (error_models/__init__.py:39-47).  The
    attributes hold the reference schema objects so callers may edit them in place (the
    reference's tests do, e.g. ``err_mod.del_for[0]["A"] = 1.0``); ``dense()`` flattens the
    current attribute values for upload.
    """

    def __init__(self, npz_path, fragment_length=None, fragment_sd=None, store_mutations=False):
        self.npz_path = npz_path
        self.store_mutations = store_mutations
        self.fragment_length = fragment_length
        self.fragment_sd = fragment_sd
        self._dense_file = None
        logger = self.logger
        try:
            dense = None
            try:
                probe = np.load(npz_path, allow_pickle=False)
                if "format" in probe.files:
                    dense = DenseModel.load(npz_path)
            except ValueError:
                dense = None
            if dense is not None:
                self._init_from_dense(dense)
            else:
                import _pickle

                try:
                    prof = np.load(npz_path, allow_pickle=True)
                    model = prof["model"]
                except (OSError, IOError, EOFError, _pickle.UnpicklingError) as e:
                    raise OSError(e)
                if str(model) != "kde":
                    logger.error("Trying to load a %s ErrorModel in %s mode" % (model, "kde"))
                    sys.exit(1)
                self.read_length = prof["read_length"]
                self.i_size_cdf = prof["insert_size"]
                self.mean_forward = prof["mean_count_forward"]
                self.mean_reverse = prof["mean_count_reverse"]
                self.quality_forward = prof["quality_hist_forward"]
                self.quality_reverse = prof["quality_hist_reverse"]
                self.subst_choices_for = prof["subst_choices_forward"]
                self.subst_choices_rev = prof["subst_choices_reverse"]
                self.ins_for = prof["ins_forward"]
                self.ins_rev = prof["ins_reverse"]
                self.del_for = prof["del_forward"]
                self.del_rev = prof["del_reverse"]
        except (OSError, IOError, EOFError) as e:
            logger.error("Failed to read ErrorModel file: %s" % e)
            sys.exit(1)
        else:
            logger.debug("Loaded ErrorProfile: %s" % npz_path)

    @property
    def logger(self):
        component = "{}.{}".format(type(self).__module__, type(self).__name__)
        return logging.getLogger(component)

    def _init_from_dense(self, d):
        """Rebuild reference-shaped attributes from a dense file (lossless for the engine)."""
        RL = d.read_length
        self.read_length = np.int64(RL)
        self.i_size_cdf = d.isize_cdf
        means = []
        for o in range(2):
            prob = np.diff(np.concatenate(([0.0], d.bin_cdf[o])))
            means.append(prob)
        self.mean_forward, self.mean_reverse = means
        self._dense_file = d

        def hist(o):
            return [[d.qcdf[o, b, p] for p in range(RL)] if d.bin_nonempty[o, b] else [] for b in range(N_BINS)]

        self.quality_forward, self.quality_reverse = hist(0), hist(1)

        def sub(o):
            out = []
            for p in range(RL):
                row = {}
                for bi, base in enumerate(BASES):
                    cdf = d.subst_cdf[o, p, bi]
                    row[base] = ([chr(c) for c in d.subst_alt[o, p, bi]], list(np.diff(np.concatenate(([0.0], cdf)))))
                out.append(row)
            return out

        self.subst_choices_for, self.subst_choices_rev = sub(0), sub(1)
        self.ins_for = [{chr(d.ins_letter[0, p, x]): d.ins[0, p, x] for x in range(4)} for p in range(RL)]
        self.ins_rev = [{chr(d.ins_letter[1, p, x]): d.ins[1, p, x] for x in range(4)} for p in range(RL)]
        self.del_for = [{b: d.dele[0, p, i] for i, b in enumerate(BASES)} for p in range(RL)]
        self.del_rev = [{b: d.dele[1, p, i] for i, b in enumerate(BASES)} for p in range(RL)]
        self._pristine = self._fingerprint()

    def _fingerprint(self):
        """What dense() may hand back from the stored dense file as long as nobody edited it: the attributes rebuilt from
        the file (substitution choices, bin means, identity of the insert-size / histogram arrays) + the stored tables."""
        import hashlib

        h = hashlib.sha1()
        d = self._dense_file
        for a in (self.mean_forward, self.mean_reverse, self.i_size_cdf, d.qcdf, d.bin_cdf, d.subst_cdf, d.subst_alt, d.isize_cdf):
            h.update(np.ascontiguousarray(a).tobytes())
        for t in (self.subst_choices_for, self.subst_choices_rev):
            h.update(repr(t).encode())
        for t in (self.quality_forward, self.quality_reverse):
            h.update(repr([[id(x) for x in b] for b in t]).encode())
        return h.hexdigest()

    def dense(self):
        """Flatten the CURRENT attribute values (so in-place edits are honoured: a model loaded from a dense file hands its
        stored tables back only while every attribute still is what the file gave -- histogram rows are views on the
        stored arrays, indel tables are re-read below, anything else that changed sends the model through
        from_attributes)."""
        if self._dense_file is not None and not getattr(self, "_edited", False) and self._fingerprint() == self._pristine:
            # a dense file stores post-normalisation CDFs; re-deriving them from differences
            # would not be bit-exact, so hand the stored tables back unless the caller edited
            d = self._dense_file
            ins = np.array([[[row[chr(d.ins_letter[o, p, x])] for x in range(4)]
                             for p, row in enumerate(t)] for o, t in enumerate((self.ins_for, self.ins_rev))])
            dele = np.array([[[row[b] for b in BASES] for row in t] for t in (self.del_for, self.del_rev)])
            return DenseModel(d.read_length, d.isize_cdf, d.bin_cdf, d.bin_nonempty, d.qcdf, d.subst_cdf,
                              d.subst_alt, ins, d.ins_letter, dele, d.phred_thr)
        return DenseModel.from_attributes(
            self.read_length, self.i_size_cdf, self.mean_forward, self.mean_reverse, self.quality_forward,
            self.quality_reverse, self.subst_choices_for, self.subst_choices_rev, self.ins_for, self.ins_rev,
            self.del_for, self.del_rev,
        )


    # ---- the reference's per-read methods, on the GPU (inner plugin surface) -----------------------
    # ``bind(engine, seed)`` attaches an engine holding this model's tables; every method then makes one (batch of 1) call
    # of the batched C-ABI entry of the same name.  ``self.ordinal`` is the Philox address of the read being built -- the
    # methods of one read share it (as they share the reference's stream position), ``next_read()`` advances it.
    _ORIENT = {"forward": 0, "reverse": 1}

    def bind(self, engine, seed=0, ordinal=0):
        engine.load_model(self.dense())
        self._engine, self._seed, self.ordinal = engine, int(seed), int(ordinal)
        return self

    def next_read(self):
        self.ordinal += 1

    def _orientation(self, orientation):
        if orientation not in self._ORIENT:
            raise ValueError("orientation must be 'forward' or 'reverse'")
        return self._ORIENT[orientation]

    def gen_phred_scores(self, cdfs, orientation):
        """kde.py:52-86.  ``cdfs`` is accepted for signature parity: the engine reads the uploaded tables."""
        return [int(q) for q in self._engine.gen_phred_scores(self._orientation(orientation), 1, self.ordinal, self._seed)[0]]

    def introduce_error_scores(self, record, orientation):
        """__init__.py:52-67: phred scores into record.letter_annotations["phred_quality"]."""
        cdfs = self.quality_forward if orientation == "forward" else self.quality_reverse
        record.letter_annotations = dict(getattr(record, "letter_annotations", None) or {})
        record.letter_annotations["phred_quality"] = self.gen_phred_scores(cdfs, orientation)
        return record

    def mut_sequence(self, record, orientation):
        """__init__.py:69-112: substitutions according to record.letter_annotations["phred_quality"]; returns the sequence."""
        RL = int(self.read_length)
        seq = np.frombuffer(str(record.seq).encode("ascii"), dtype=np.uint8)
        qual = np.asarray(record.letter_annotations["phred_quality"], dtype=np.uint8)
        if seq.size != RL or qual.size != RL:
            raise ValueError("mut_sequence needs a read of read_length letters and phreds")
        out, st = self._engine.mut_sequence(self._orientation(orientation), seq[None, :], qual[None, :], self.ordinal, self._seed)
        if st[0] == 2:
            raise KeyError("letter outside the substitution table")
        return out[0].tobytes().decode("ascii")

    def introduce_indels(self, record, orientation, full_seq, bounds):
        """__init__.py:158-228 (+ adjust_seq_length :114-156): record.seq replaced by the read with indels."""
        RL = int(self.read_length)
        seq = np.frombuffer(str(record.seq).encode("ascii"), dtype=np.uint8)
        if seq.size > RL:
            raise ValueError("a read longer than read_length")
        row = np.zeros((1, RL), dtype=np.uint8)
        row[0, :seq.size] = seq
        out, st = self._engine.introduce_indels(self._orientation(orientation), row, [seq.size], str(full_seq), [list(bounds)],
                                                self.ordinal, self._seed)
        if st[0] == 2:
            raise KeyError("letter outside the deletion table")
        if st[0] == 3:
            raise IndexError("padding index outside the reference sequence")
        record.seq = out[0].tobytes().decode("ascii")
        return record

    def random_insert_size(self):
        """kde.py:88-98."""
        return int(self._engine.random_insert_size(1, self.ordinal, self._seed)[0])


class BasicErrorModel(object):
    """Host-side mirror of ``iss.error_models.basic.BasicErrorModel`` (basic.py:10-63): same constructor and
    attributes; ``dense()`` gives the tables the engine uploads.  On the device it runs in the
    reference-compatible RNG mode (``rng="mt"``)."""

    def __init__(self, fragment_length=None, fragment_sd=None, store_mutations=False):
        self.read_length = 125
        self.insert_size = 200
        self.fragment_length = fragment_length
        self.fragment_sd = fragment_sd
        self.store_mutations = store_mutations
        self.quality_forward = self.quality_reverse = 30
        self.npz_path = None

    def dense(self):
        return DenseModel.basic(self.read_length, self.insert_size, self.quality_forward)

    def random_insert_size(self):
        return self.insert_size
