"""A stand-in for the part of pysam that InSilicoSeq's `iss model` touches (iss/bam.py, iss/modeller.py) -- golden TOOLING only,
like bio_shim: pysam is absent from the build container, and BASELINE configs[4] names "a custom .npz from data/ecoli.bam via
iss model".  With it the reference's own modeller builds that model from the reference's own BAM file
(tests/golden/tooling/make_golden_bam_model.py); nothing of this travels into the product.

Covered (pysam 0.21 semantics): idxstats(path); AlignmentFile(path, "rb") as a context manager with fetch(); per read
is_unmapped / is_paired / is_read1 / is_read2 / is_reverse, template_length, seq == query_sequence, query_alignment_sequence,
query_qualities (array('B')), cigartuples, get_aligned_pairs(matches_only=True, with_seq=True) from the MD tag (reference base in
lower case where it differs from the read); utils.SamtoolsError.  BAM: BGZF members are gzip members (RFC 1952), records as in
the SAM/BAM specification section 4.2."""
import array
import gzip
import struct


class _Utils(object):
    class SamtoolsError(Exception):
        pass


utils = _Utils()
_SEQ = "=ACMGRSVTWYHKDBN"


def _parse(path):
    try:
        data = gzip.open(path, "rb").read()
    except (OSError, EOFError) as e:
        raise IOError("could not open alignment file `%s`: %s" % (path, e))
    if data[:4] != b"BAM\x01":
        raise ValueError("file `%s` does not have a valid BAM header" % path)
    off = 4
    (l_text,) = struct.unpack_from("<i", data, off)
    off += 4 + l_text
    (n_ref,) = struct.unpack_from("<i", data, off)
    off += 4
    refs = []
    for _ in range(n_ref):
        (ln,) = struct.unpack_from("<i", data, off)
        off += 4
        name = data[off:off + ln - 1].decode()
        off += ln
        (length,) = struct.unpack_from("<i", data, off)
        off += 4
        refs.append((name, length))
    reads = []
    while off < len(data):
        (bs,) = struct.unpack_from("<i", data, off)
        off += 4
        reads.append(AlignedSegment(data[off:off + bs]))
        off += bs
    return refs, reads


def idxstats(path):
    """samtools idxstats: reference, length, mapped, unmapped per line (+ the '*' line)."""
    try:
        refs, reads = _parse(path)
    except (IOError, ValueError) as e:
        raise utils.SamtoolsError(str(e))
    lines = []
    for k, (name, length) in enumerate(refs):
        mapped = sum(1 for r in reads if r.reference_id == k and not r.is_unmapped)
        unmapped = sum(1 for r in reads if r.reference_id == k and r.is_unmapped)
        lines.append("%s\t%d\t%d\t%d" % (name, length, mapped, unmapped))
    lines.append("*\t0\t0\t%d" % sum(1 for r in reads if r.reference_id < 0))
    return "\n".join(lines) + "\n"


class AlignedSegment(object):
    def __init__(self, rec):
        (self.reference_id, self.reference_start, l_name, self.mapping_quality, _bin, n_cigar, self.flag, l_seq, self.next_reference_id,
         self.next_reference_start, self.template_length) = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        p = 32
        self.query_name = rec[p:p + l_name - 1].decode()
        p += l_name
        cig = struct.unpack_from("<%dI" % n_cigar, rec, p)
        p += 4 * n_cigar
        self.cigartuples = [(c & 15, c >> 4) for c in cig]
        sb = rec[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        self.query_sequence = "".join(_SEQ[(sb[i // 2] >> (4 * (1 - i % 2))) & 15] for i in range(l_seq))
        self.query_qualities = array.array("B", rec[p:p + l_seq])
        p += l_seq
        self.tags = {}
        while p < len(rec):
            tag, typ = rec[p:p + 2].decode(), chr(rec[p + 2])
            p += 3
            if typ in "cCsSiIf":
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ]
                (val,) = struct.unpack_from("<" + fmt, rec, p)
                p += struct.calcsize(fmt)
            elif typ == "A":
                val = chr(rec[p])
                p += 1
            elif typ in "ZH":
                end = rec.index(b"\x00", p)
                val = rec[p:end].decode()
                p = end + 1
            elif typ == "B":
                sub = chr(rec[p])
                (cnt,) = struct.unpack_from("<i", rec, p + 1)
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
                val = list(struct.unpack_from("<%d%s" % (cnt, fmt), rec, p + 5))
                p += 5 + cnt * struct.calcsize(fmt)
            else:
                raise ValueError("unknown tag type %r" % typ)
            self.tags[tag] = val

    seq = property(lambda self: self.query_sequence)
    is_paired = property(lambda self: bool(self.flag & 1))
    is_unmapped = property(lambda self: bool(self.flag & 4))
    is_reverse = property(lambda self: bool(self.flag & 16))
    is_read1 = property(lambda self: bool(self.flag & 64))
    is_read2 = property(lambda self: bool(self.flag & 128))

    @property
    def query_alignment_sequence(self):  # without soft-clipped ends
        lo = self.cigartuples[0][1] if self.cigartuples and self.cigartuples[0][0] == 4 else 0
        hi = self.cigartuples[-1][1] if len(self.cigartuples) > 1 and self.cigartuples[-1][0] == 4 else 0
        return self.query_sequence[lo:len(self.query_sequence) - hi]

    def get_aligned_pairs(self, matches_only=False, with_seq=False):
        if not (matches_only and with_seq):
            raise NotImplementedError("the shim covers get_aligned_pairs(matches_only=True, with_seq=True)")
        md = self.tags.get("MD")
        if md is None:
            raise ValueError("MD tag not present")
        # the MD tag as a stream of reference letters over the aligned (M / = / X) columns: digits = that many matches
        ref_letters, k = [], 0
        while k < len(md):
            if md[k].isdigit():
                j = k
                while j < len(md) and md[j].isdigit():
                    j += 1
                ref_letters.extend([None] * int(md[k:j]))
                k = j
            elif md[k] == "^":  # deleted reference letters: no aligned column
                k += 1
                while k < len(md) and md[k].isalpha():
                    k += 1
            else:
                ref_letters.append(md[k].lower())  # a mismatch: the reference letter, lower case
                k += 1
        out, q, r, col = [], 0, self.reference_start, 0
        for op, ln in self.cigartuples:
            if op in (0, 7, 8):  # M, =, X
                for _ in range(ln):
                    ref = ref_letters[col]
                    out.append((q, r, self.query_sequence[q] if ref is None else ref))
                    q += 1
                    r += 1
                    col += 1
            elif op in (1, 4):  # I, S: query only
                q += ln
            elif op in (2, 3):  # D, N: reference only
                r += ln
        return out


class AlignmentFile(object):
    def __init__(self, path, mode="rb"):
        self.refs, self._reads = _parse(path)

    def fetch(self):
        return iter(self._reads)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
