"""Multi-GPU plumbing: one process per GPU, rank r == the reference's worker ``cpu_number`` r.

The path shards with NO data-path collective (pairs are independent given genome, tables and the
RNG address): the only collective is ONE broadcast of the dense model tables and the 2-bit packed genomes
from rank 0 before generation (RCCL over xGMI with backend "nccl"; "gloo" on CPU in the tests), issued
only when world_size > 1.  Work division and output assembly mirror the reference:

* chunk size ``ceil((n_reads // 2) / world)`` and ``zip(work_chunks, temp_file_list)`` -- a surplus
  (world+1)-th rounding chunk is dropped exactly like the reference does (iss/app.py:81-83, 99-106);
* per-rank temp files ``{output}.iss.tmp.{rank}_R1.fastq`` ... concatenated in rank order
  (iss/app.py:73, 123-127; iss/util.py:213-234) and removed.
"""
import os
import shutil

import numpy as np

from .generator import generate_work_divider
from .model import DenseModel

_U8_FIELDS = ("bin_nonempty", "subst_alt", "ins_letter")
_CODE_OF = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ATCG"):  # the engine's 2-bit alphabet: A,T,C,G = 0..3 (complement = code ^ 1)
    _CODE_OF[_c] = _i


def pack_2bit(ascii_u8, block=1 << 24):
    """uint8 letters -> uint32 words of 2-bit codes (16 bases per word, base i in bits 2*(i % 16)...), or None when the
    record holds anything but plain A/C/G/T (it is shipped as ASCII then).  Packed ``block`` bases at a time: the
    temporaries stay at a few bytes per base of ONE block whatever the record's length (the engine takes 1.2 Gbp records)."""
    a = np.asarray(ascii_u8, dtype=np.uint8).reshape(-1)
    if a.size == 0:
        return None
    block = max(16, int(block) & ~15)
    out = np.zeros((a.size + 15) // 16, dtype=np.uint32)
    shifts = 2 * np.arange(16, dtype=np.uint32)
    for at in range(0, a.size, block):
        codes = _CODE_OF[a[at:at + block]]
        if int(codes.max()) > 3:
            return None
        n = (codes.size + 15) // 16
        buf = np.zeros(n * 16, dtype=np.uint32)
        buf[:codes.size] = codes
        buf = buf.reshape(n, 16)
        buf <<= shifts
        out[at // 16:at // 16 + n] = np.bitwise_or.reduce(buf, axis=1)
    return out


def unpack_2bit(words, length):
    """Inverse of pack_2bit (host side: tests, CPU consumers)."""
    w = np.asarray(words, dtype=np.uint32)
    codes = ((w[:, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(-1)[:length]
    return np.frombuffer(b"ATCG", dtype=np.uint8)[codes]


class BroadcastGenome(object):
    """One record as a rank holds it after the broadcast: ``length`` letters, either 2-bit codes (``packed`` True) or
    ASCII, as a uint8 view ``raw`` on the host and -- when the payload lives on a GPU -- the device address ``dev_ptr``."""

    def __init__(self, length, packed, raw, dev_ptr=None, dev_bytes=None):
        self.length, self.packed, self.raw, self.dev_ptr = int(length), bool(packed), raw, dev_ptr
        self._dev_bytes = dev_bytes  # the record's slice of the device payload (a torch uint8 tensor), when raw is None

    def _host_raw(self):
        if self.raw is None:
            if self._dev_bytes is None:
                raise ValueError("BroadcastGenome: neither a host copy nor a device slice of the record is held")
            self.raw = self._dev_bytes.cpu().numpy()  # (a packed record left on the GPU by as_refs=True: fetched on demand)
        return self.raw

    def ascii(self):
        if not self.packed:
            return np.asarray(self._host_raw(), dtype=np.uint8)[:self.length]
        return unpack_2bit(np.asarray(self._host_raw()).view(np.uint32), self.length)

    def upload(self, engine):
        """To the engine's HBM: straight from the broadcast buffer on the device when there is one."""
        if self.packed:
            return engine.add_genome_packed(None if self.dev_ptr is not None else np.asarray(self._host_raw()).view(np.uint32),
                                            self.length, device_ptr=self.dev_ptr)
        return engine.add_genome(np.asarray(self._host_raw(), dtype=np.uint8)[:self.length])


def _align16(n):
    return (n + 15) & ~15


def broadcast_model_and_genomes(dense, genomes, dist=None, device="cpu", src=0, as_refs=False, force=False, info=None):
    """Rank ``src`` passes (DenseModel, list of uint8 arrays / bytes); other ranks pass (None, None).
    Returns (DenseModel, genomes) on every rank; no-op when ``dist`` is None / world == 1 (``force``: go through the
    collectives even then -- a one-rank RCCL group on a single-GPU box runs the very calls of an 8-GPU run).

    ONE payload: [header][model tables][genomes], the genomes as 2-bit codes (plain A/C/G/T records: a quarter of the
    letters -- 62.5 MB for BASELINE configs[3]'s 250 Mbp) or ASCII (records with IUPAC / lower-case letters), sent by one
    broadcast (RCCL over xGMI with backend "nccl") after an 8-byte size announcement.  ``as_refs``: return
    BroadcastGenome objects -- on a GPU they point INTO the received device buffer, which ``upload`` hands to
    iss_genome_upload_packed without a detour through the host; otherwise uint8 ASCII arrays (host consumers).
    ``info``: an optional dict that receives what went over the wire (payload_bytes, model_bytes, genome_bytes, collective)."""
    def as_u8(g):
        return np.frombuffer(g, dtype=np.uint8) if isinstance(g, (bytes, bytearray)) else np.ascontiguousarray(g, dtype=np.uint8)

    if dist is None or (dist.get_world_size() == 1 and not force):
        if info is not None:
            info.update(payload_bytes=0, model_bytes=0, genome_bytes=0, collective=None)
        glist = [as_u8(g) for g in genomes]
        return dense, ([BroadcastGenome(g.size, False, g) for g in glist] if as_refs else glist)
    import json

    import torch

    backend = dist.get_backend()
    if str(backend).lower() == "nccl" and not str(device).startswith("cuda"):
        # RCCL moves device memory only: a CPU tensor would fail inside the collective with a far less readable message
        raise ValueError("broadcast_model_and_genomes: backend 'nccl' (RCCL) needs device='cuda:<n>', got %r" % (device,))
    rank = dist.get_rank()
    if rank == src:
        fields = [np.ascontiguousarray(getattr(dense, k)) for k in DenseModel.FIELDS]
        parts, gmeta = [], []
        for g in (as_u8(x) for x in genomes):
            words = pack_2bit(g)
            raw = words.view(np.uint8) if words is not None else g
            gmeta.append([int(g.size), 1 if words is not None else 0, int(raw.size)])
            parts.append(raw)
        header = json.dumps({"read_length": dense.read_length, "shapes": [list(f.shape) for f in fields],
                             "genomes": gmeta}).encode()
        segs = [np.frombuffer(header, dtype=np.uint8)] + [f.view(np.uint8).reshape(-1) for f in fields] + parts
        total = 16 + sum(_align16(x.size) for x in segs)
        flat = np.zeros(total, dtype=np.uint8)
        flat[:8] = np.frombuffer(np.int64(len(header)).tobytes(), dtype=np.uint8)
        off = 16
        for x in segs:
            flat[off:off + x.size] = x
            off += _align16(x.size)
        size = torch.tensor([total], dtype=torch.int64, device=device)
    else:
        flat, size = None, torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(size, src=src)
    total = int(size.item())
    t = torch.empty(total, dtype=torch.uint8, device=device)
    if rank == src:
        t.copy_(torch.from_numpy(flat))
    dist.broadcast(t, src=src)  # the one collective of a multi-GPU run
    on_gpu = t.is_cuda
    if on_gpu:
        torch.cuda.synchronize(t.device)  # the engine reads the buffer on its own HIP stream
    # header + model tables: to the host (iss_model_upload takes host tables; < 1 MB); genomes stay where they are
    hlen = int(np.frombuffer(t[:8].cpu().numpy().tobytes(), dtype=np.int64)[0])
    meta = json.loads(t[16:16 + hlen].cpu().numpy().tobytes().decode())
    off = 16 + _align16(hlen)
    out_fields = []
    for k, shape in zip(DenseModel.FIELDS, meta["shapes"]):
        dt = np.uint8 if k in _U8_FIELDS else np.float64
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        out_fields.append(t[off:off + nbytes].cpu().numpy().view(dt).reshape(shape).copy())
        off += _align16(nbytes)
    host = None if on_gpu and as_refs else t.cpu().numpy()
    refs = []
    for length, packed, nbytes in meta["genomes"]:
        if host is not None:
            raw = host[off:off + nbytes]
        else:  # on a GPU only the ASCII records (IUPAC / lower-case letters: validated and packed by iss_genome_upload) come back
            raw = None if packed else t[off:off + nbytes].cpu().numpy()
        refs.append(BroadcastGenome(length, packed, raw, dev_ptr=(t.data_ptr() + off) if on_gpu else None,
                                    dev_bytes=t[off:off + nbytes] if on_gpu else None))
        off += _align16(nbytes)
    if rank != src:
        dense = DenseModel(meta["read_length"], *out_fields)
    if info is not None:
        gbytes = sum(_align16(nb) for _, _, nb in meta["genomes"])
        info.update(payload_bytes=total, genome_bytes=gbytes, model_bytes=total - gbytes, n_genomes=len(meta["genomes"]),
                    packed_genomes=sum(1 for _, pk, _ in meta["genomes"] if pk),
                    collective="one broadcast (%s), %d ranks" % (backend, dist.get_world_size()))
    if as_refs:
        for r in refs:
            r._keep = t  # the device buffer must outlive the uploads
        return dense, refs
    return dense, [r.ascii() for r in refs]


def rank_work(records, readcount_dic, abundance_dic, n_reads, coverage, coverage_file, error_model, output, world,
              rank):
    """The work list of ``rank``: chunk ``rank`` of the reference's divider with ``cpus = world``.
    Returns (work, chunk_size, n_chunks)."""
    n_read_pairs = n_reads // 2
    chunk_size = -((n_read_pairs) // -world)  # ceildiv, iss/app.py:82
    chunks = list(generate_work_divider(records, readcount_dic, abundance_dic, n_reads, coverage, coverage_file,
                                        error_model, output, chunk_size))
    work = chunks[rank] if rank < len(chunks) else None
    return work, chunk_size, len(chunks)


def temp_prefix(output, rank):
    return "%s.iss.tmp.%d" % (output, rank)  # iss/app.py:73


VCF_HEADER = "##fileformat=VCFv4.1\n" + "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"])


def _append_file(src_path, out):
    """Append a file to the open binary file ``out`` (in-kernel copy where the platform has it)."""
    with open(src_path, "rb") as fh:
        size = os.fstat(fh.fileno()).st_size
        out.flush()
        try:
            off = 0
            while off < size:
                n = os.sendfile(out.fileno(), fh.fileno(), off, min(size - off, 1 << 30))
                if n == 0:
                    break
                off += n
            if off == size:
                return
            fh.seek(off)
        except (AttributeError, OSError):
            fh.seek(0)
            out.seek(0, os.SEEK_END)
        shutil.copyfileobj(fh, out, 1 << 22)


def concatenate_rank_files(output, world, suffixes=("_R1.fastq", "_R2.fastq"), cleanup=True, headers=None,
                           out_suffixes=None):
    """util.concatenate over the per-rank temp files, in rank order (iss/app.py:123-133).  Like the
    reference, a missing temp file (fewer chunks than workers) is an error (iss/util.py:233).
    ``headers``: optional {suffix: header text} written first, followed by a newline (util.py:229-230).
    Same bytes as the reference's copy loop; rank 0's file is renamed instead of copied when it is going to be
    removed anyway and nothing precedes it (tens of GB at BASELINE's sizes).  ``out_suffixes``: optional
    {suffix: suffix of the assembled file} (workers that wrote gzip members: "_R1.fastq" -> "_R1.fastq.gz")."""
    out_suffixes = out_suffixes or {}
    for suffix in suffixes:
        for r in range(world):
            if not os.path.exists(temp_prefix(output, r) + suffix):
                raise FileNotFoundError(temp_prefix(output, r) + suffix)

    def assemble(suffix):
        paths = [temp_prefix(output, r) + suffix for r in range(world)]
        header = headers.get(suffix) if headers else None
        target = output + out_suffixes.get(suffix, suffix)
        first = 0
        if cleanup and header is None and world > 0:
            os.replace(paths[0], target)
            first = 1
        with open(target, "ab" if first else "wb") as out:
            if header is not None:
                out.write((header + "\n").encode())
            for path in paths[first:]:
                _append_file(path, out)

    # one thread per assembled file: the copies are in-kernel (sendfile releases the GIL) and serialise on the TARGET's inode,
    # so R1 and R2 go side by side (tens of GB at BASELINE's sizes)
    if len(suffixes) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=len(suffixes)) as pool:
            for f in [pool.submit(assemble, suffix) for suffix in suffixes]:
                f.result()
    else:
        for suffix in suffixes:
            assemble(suffix)
    if cleanup:
        for r in range(world):
            for suffix in tuple(suffixes) + (".vcf",):
                path = temp_prefix(output, r) + suffix
                if os.path.exists(path):
                    os.remove(path)
