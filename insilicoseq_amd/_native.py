"""ctypes binding of the C ABI in include/iss_mi355x.h (libiss_mi355x.so, built in-tree).

There is NO fallback: if the HIP library is missing or no MI355X is visible, the product
raises (``NativeLibraryError`` / ``EngineError``) instead of computing anything on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libiss_mi355x.so"
LIB_PATH = os.environ.get("ISS_MI355X_LIB") or os.path.join(_HERE, LIB_NAME)  # override: A/B builds (tools/ab_bench.sh)

E_INVALID, E_HIP, E_NOMEM, E_SHORT_RECORD, E_IO = -1, -2, -3, -4, -5
SEQ_TYPES = {"metagenomics": 0, "amplicon": 1}

# every symbol include/iss_mi355x.h declares (tests check the library exports all of them)
EXPORTS = (
    "iss_abi_version", "iss_ctx_create", "iss_ctx_destroy", "iss_last_error", "iss_ctx_set_stream",
    "iss_model_upload", "iss_genome_upload", "iss_genome_upload_packed", "iss_genome_clear", "iss_output_reserve", "iss_output_pitch",
    "iss_output_device_ptrs", "iss_output_row", "iss_generate", "iss_synchronize", "iss_output_download",
    "iss_output_download_coords", "iss_timing_enable", "iss_timing_read", "iss_stats_read", "iss_build_id", "iss_fastq_write",
    "iss_mt_seed", "iss_generate_mt", "iss_mt_peek", "iss_mt_mutations_reserve", "iss_mt_mutations_download",
    "iss_mt_set_fragment", "iss_set_fragment", "iss_mutations_reserve", "iss_mutations_download",
    "iss_mt_path_counts", "iss_fastq_emit", "iss_fastq_flush", "iss_fastq_compress", "iss_deflate_code_build",
    "iss_generate_batch", "iss_fastq_emit_batch", "iss_gen_phred_scores", "iss_mut_sequence", "iss_random_insert_size",
    "iss_introduce_indels", "iss_ev_step", "iss_mt_workers_seed", "iss_generate_mt_workers", "iss_mt_workers_peek",
    "iss_main_kernel", "iss_fastq_emit_scatter",
)


class NativeLibraryError(RuntimeError):
    pass


class EngineError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "%s (iss error %d)" % (message, code))
        self.code = code


class ModelTables(C.Structure):
    _fields_ = [
        ("read_length", C.c_int32), ("n_isize", C.c_int32), ("n_q", C.c_int32),
        ("isize_thr", C.c_void_p), ("bin_thr", C.c_void_p), ("bin_nonempty", C.c_void_p), ("q_thr", C.c_void_p),
        ("subst_thr", C.c_void_p), ("subst_alt", C.c_void_p), ("ins_thr", C.c_void_p), ("ins_letter", C.c_void_p),
        ("del_thr", C.c_void_p), ("mut_thr", C.c_void_p),
        ("quality_mode", C.c_int32), ("basic_insert_size", C.c_int32), ("basic_mean", C.c_double),
        ("basic_sd", C.c_double), ("basic_cap", C.c_double),
    ]


_lib = None


def lib():
    """Load the HIP library (once).  Raises NativeLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    L.iss_abi_version.restype = C.c_int
    L.iss_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.iss_ctx_destroy.argtypes = [vp]
    L.iss_ctx_destroy.restype = None
    L.iss_last_error.argtypes = [vp]
    L.iss_last_error.restype = C.c_char_p
    L.iss_ctx_set_stream.argtypes = [vp, vp]
    L.iss_model_upload.argtypes = [vp, C.POINTER(ModelTables)]
    L.iss_genome_upload.argtypes = [vp, vp, i64, C.POINTER(i32)]
    L.iss_genome_upload_packed.argtypes = [vp, vp, i64, i32, C.POINTER(i32)]
    L.iss_genome_clear.argtypes = [vp]
    L.iss_output_reserve.argtypes = [vp, i64]
    L.iss_output_pitch.argtypes = [vp]
    L.iss_output_row.argtypes = [vp]
    L.iss_output_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.iss_generate.argtypes = [vp, i32, i64, u64, u64, i32, i32, i64]
    L.iss_synchronize.argtypes = [vp]
    L.iss_output_download.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.iss_output_download_coords.argtypes = [vp, i64, i64, vp]
    L.iss_timing_enable.argtypes = [vp, C.c_int]
    L.iss_timing_read.argtypes = [vp, C.POINTER(C.c_double * 4), C.POINTER(i64)]
    L.iss_stats_read.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.iss_build_id.restype = C.c_char_p
    L.iss_build_id.argtypes = []
    L.iss_mt_seed.argtypes = [vp, u64]
    L.iss_generate_mt.argtypes = [vp, i32, i64, i32, i32, i64, C.POINTER(i64)]
    L.iss_mt_peek.argtypes = [vp, vp, vp, i32]
    L.iss_mt_path_counts.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.iss_mt_workers_seed.argtypes = [vp, i32, vp]
    L.iss_generate_mt_workers.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp]
    L.iss_mt_workers_peek.argtypes = [vp, i32, vp, vp, i32]
    L.iss_mt_mutations_reserve.argtypes = [vp, i64]
    L.iss_mt_set_fragment.argtypes = [vp, i32, C.c_double, C.c_double]
    L.iss_set_fragment.argtypes = [vp, i32, C.c_double, C.c_double]
    L.iss_mutations_reserve.argtypes = [vp, i64]
    L.iss_mutations_download.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.iss_mt_mutations_download.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.iss_fastq_emit.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, i64, i32, i64, i64, i32]
    L.iss_fastq_emit_batch.argtypes = [vp, C.c_int, C.c_int, i32, vp, vp, vp, vp, i32]
    L.iss_fastq_emit_scatter.argtypes = [vp, C.c_int, C.c_int, i32, vp, vp, vp, vp, vp, vp, i32]
    L.iss_main_kernel.argtypes = [vp, vp, C.c_int]
    L.iss_fastq_flush.argtypes = [vp]
    L.iss_generate_batch.argtypes = [vp, i32, vp, vp, C.c_uint64, C.c_uint64, i32, i32, i64]
    L.iss_fastq_compress.argtypes = [vp, i32]
    L.iss_deflate_code_build.argtypes = [vp, C.c_uint32, vp, vp, vp, vp]
    L.iss_gen_phred_scores.argtypes = [vp, i32, i64, u64, u64, vp]
    L.iss_mut_sequence.argtypes = [vp, i32, i64, u64, u64, vp, vp, vp]
    L.iss_random_insert_size.argtypes = [vp, i64, u64, u64, vp]
    L.iss_introduce_indels.argtypes = [vp, i32, i64, u64, u64, vp, vp, vp, i64, vp, vp, vp]
    L.iss_ev_step.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp]
    L.iss_fastq_write.argtypes = [C.c_int, C.c_int, C.c_char_p, i64, i32, i64, i32, i32, vp, vp, vp, vp, i32]
    for name in EXPORTS:
        if name not in ("iss_ctx_destroy", "iss_last_error", "iss_build_id"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(ctx, rc):
    if rc < 0:
        msg = lib().iss_last_error(ctx)
        raise EngineError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")
    return rc
