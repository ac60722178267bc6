"""Multi-GPU plumbing: one process per GPU, rank r == the reference's worker ``cpu_number`` r.

The path shards with NO data-path collective (pairs are independent given genome, tables and the
RNG address): the only collective is one broadcast of the dense model tables and the genomes from
rank 0 before generation (RCCL over xGMI with backend "nccl"; "gloo" on CPU in the tests), issued
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


def _bcast_bytes(dist, arr_or_none, nbytes, device, src=0):
    import torch

    t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if arr_or_none is not None:
        t.copy_(torch.from_numpy(np.ascontiguousarray(arr_or_none).view(np.uint8).reshape(-1)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_model_and_genomes(dense, genomes, dist=None, device="cpu", src=0):
    """Rank ``src`` passes (DenseModel, list of uint8 arrays / bytes); other ranks pass (None, None).
    Returns (DenseModel, [uint8 arrays]) on every rank.  No-op when ``dist`` is None / world == 1."""
    if dist is None or dist.get_world_size() == 1:
        return dense, [np.frombuffer(g, dtype=np.uint8) if isinstance(g, (bytes, bytearray)) else np.asarray(
            g, dtype=np.uint8) for g in genomes]
    rank = dist.get_rank()
    if rank == src:
        fields = [np.ascontiguousarray(getattr(dense, k)) for k in DenseModel.FIELDS]
        glist = [np.frombuffer(g, dtype=np.uint8) if isinstance(g, (bytes, bytearray)) else np.ascontiguousarray(
            g, dtype=np.uint8) for g in genomes]
        meta = {"read_length": dense.read_length, "shapes": [list(f.shape) for f in fields],
                "genome_lengths": [int(g.size) for g in glist]}
    else:
        fields, glist, meta = None, None, None
    box = [meta]
    dist.broadcast_object_list(box, src=src)
    meta = box[0]
    out_fields = []
    for i, (k, shape) in enumerate(zip(DenseModel.FIELDS, meta["shapes"])):
        dt = np.uint8 if k in _U8_FIELDS else np.float64
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        raw = _bcast_bytes(dist, fields[i] if rank == src else None, nbytes, device, src)
        out_fields.append(raw.view(dt).reshape(shape))
    total = int(sum(meta["genome_lengths"]))
    flat = _bcast_bytes(dist, np.concatenate(glist) if rank == src else None, total, device, src)
    offs = np.concatenate(([0], np.cumsum(meta["genome_lengths"]))).astype(np.int64)
    out_genomes = [flat[offs[i]:offs[i + 1]] for i in range(len(meta["genome_lengths"]))]
    if rank != src:
        dense = DenseModel(meta["read_length"], *out_fields)
    return dense, out_genomes


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
        paths = [temp_prefix(output, r) + suffix for r in range(world)]
        for path in paths:
            if not os.path.exists(path):
                raise FileNotFoundError(path)
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
    if cleanup:
        for r in range(world):
            for suffix in tuple(suffixes) + (".vcf",):
                path = temp_prefix(output, r) + suffix
                if os.path.exists(path):
                    os.remove(path)
