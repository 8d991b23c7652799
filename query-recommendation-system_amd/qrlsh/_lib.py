"""ctypes loader for libqrlsh.so (the C ABI declared in include/qrlsh.h).

There is no CPU fallback: if the HIP library is missing this module raises, loudly.
Build it with  `make -C query-recommendation-system_amd/csrc`  (or __graft_entry__.build()).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QRLSH_LIB", os.path.join(_HERE, "libqrlsh.so"))  # QRLSH_LIB: development override

QRLSH_OK = 0
QRLSH_EINVAL = -1
QRLSH_EHIP = -2
QRLSH_EUNSUPPORTED = -3
QRLSH_EWORKSPACE = -4

PERM_U16 = 0
PERM_I32 = 1
SIG_I32 = 0
SIG_U16 = 1
SORT_MIX = 1
SORT_IOTA = 2
SORT_FOLD = 4
SORT_OWNER = 8
SORT_HOST = 16
SUM_PAIRWISE = 0
SUM_SEQUENTIAL = 1

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_sz = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/qrlsh.h declares
SIGNATURES = {
    "qrlsh_version": (ctypes.c_int, []),
    "qrlsh_last_error": (ctypes.c_char_p, []),
    "qrlsh_mix64_host": (_u64, [_u64]),
    "qrlsh_minhash": (ctypes.c_int, [_vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "qrlsh_check_csr": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "qrlsh_band_keys": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "qrlsh_sort_workspace_bytes": (_sz, [_i64, _i32]),
    "qrlsh_sort_u64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _u32, _u64, _vp, _sz, _vp]),
    "qrlsh_owner_bounds": (ctypes.c_int, [_vp, _i64, _i32, _u64, _i32, _vp, _vp]),
    "qrlsh_pairs_workspace_bytes": (_sz, [_i64, _i32]),
    "qrlsh_pairs_count": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp, _vp]),
    "qrlsh_pairs_fill": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "qrlsh_bucket_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "qrlsh_bucket_pairs_count": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp, _vp]),
    "qrlsh_bucket_pairs_fill": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "qrlsh_bucket_pairs_emit_chunked": (ctypes.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp, _u64, _vp, _vp]),
    "qrlsh_bucket_part_words": (_sz, [_i64, _i32, _i32]),
    "qrlsh_set_big_part_limit": (_i64, [_i64]),
    "qrlsh_set_score_runs": (_i32, [_i32]),
    "qrlsh_bucket_tmp_words": (_sz, [_i64, _i32, _i32]),
    "qrlsh_bucket_pairs_emit": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp, _u64, _vp, _vp]),
    "qrlsh_compact_workspace_bytes": (_sz, [_i64]),
    "qrlsh_row_unique_workspace_bytes": (_sz, [_i64]),
    "qrlsh_row_unique_count": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp, _vp]),
    "qrlsh_row_unique_fill": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "qrlsh_region_unique_workspace_bytes": (_sz, [_i64, _i32]),
    "qrlsh_region_unique_count": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i64, _vp, _vp, _sz, _vp, _vp]),
    "qrlsh_region_unique_fill": (ctypes.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "qrlsh_region_unique_count_regions": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _i64, _vp, _vp, _sz, _vp, _vp]),
    "qrlsh_pair_regions_words": (_sz, [_i64, _i64, _i32, ctypes.c_double]),
    "qrlsh_pair_regions_tmp_words": (_sz, [_i64, _i64, _i32, ctypes.c_double]),
    "qrlsh_pair_regions_cap": (_i64, [_i64, _i64, _i32, ctypes.c_double]),
    "qrlsh_pair_regions_count": (_i64, [_i64, _i64, _i32, ctypes.c_double]),
    "qrlsh_pair_regions_scatter": (ctypes.c_int, [_vp, _i64, _i32, _i64, ctypes.c_double, _vp, _vp, _vp, _vp, _vp]),
    "qrlsh_unique_count": (ctypes.c_int, [_vp, _i64, _vp, _sz, _vp, _vp]),
    "qrlsh_unique_fill": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "qrlsh_row_norms": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "qrlsh_verify_pairs": (ctypes.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "qrlsh_score_pairs": (ctypes.c_int, [_vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp]),
    "qrlsh_remap_pairs": (ctypes.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "qrlsh_pair_edges": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "qrlsh_topk_count": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _sz, _vp, _vp]),
    "qrlsh_topk_fill": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "qrlsh_topk_fill_based": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "qrlsh_score_pairs_rev": (ctypes.c_int, [_vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _vp]),
    "qrlsh_topk_select_workspace_bytes": (_sz, [_i64]),
    "qrlsh_topk_select_count": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _sz, _vp, _vp]),
    "qrlsh_topk_select_fill": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "qrlsh_idset_workspace_bytes": (_sz, [_i64]),
    "qrlsh_idset_build": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _sz, _vp, _vp]),
    "qrlsh_idset_list": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "qrlsh_idset_remap": (ctypes.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "qrlsh_gather_rows": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "qrlsh_score_pairs_split": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "qrlsh_edges_localize": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _i64, _vp, _vp]),
    "qrlsh_answer_sets_count": (ctypes.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp]),
    "qrlsh_answer_sets_fill": (ctypes.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "qrlsh_answer_sets_sweep": (ctypes.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "qrlsh_answer_sets_compact": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "qrlsh_predict_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "qrlsh_predict": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i32, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "qrlsh_center_rows": (ctypes.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "qrlsh_user_gram_workspace_bytes": (_sz, [_i64, _i64]),
    "qrlsh_user_gram": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "qrlsh_gather_sets_workspace_bytes": (_sz, [_i64]),
    "qrlsh_gather_sets_count": (ctypes.c_int, [_vp, _i64, _vp, _i32, _i64, _i64, _vp, _vp, _sz, _vp]),
    "qrlsh_gather_sets_fill": (ctypes.c_int, [_vp, _i64, _vp, _i32, _vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    "qrlsh_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "qrlsh_set_overlap": (ctypes.c_int, [ctypes.c_int]),
    "qrlsh_prof_pause": (ctypes.c_int, [ctypes.c_int]),
    "qrlsh_prof_report": (ctypes.c_int, [ctypes.c_char_p, _sz]),
    "qrlsh_synth_sizes": (ctypes.c_int, [_u64, _i64, _i64, _i64, _i32, _u32, _vp, _i32, _u32, _vp, _vp]),
    "qrlsh_synth_fill": (ctypes.c_int, [_u64, _i64, _i64, _i64, _i32, _u32, _vp, _i32, _u32, _vp, _vp, _vp]),
}


class QrlshError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libqrlsh error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libqrlsh.so once; raise if it is not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libqrlsh.so not found at %s -- the HIP extension is required (there is no CPU fallback). "
            "Build it with: make -C query-recommendation-system_amd/csrc" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Raise on a negative status; pass non-negative values through."""
    if rc < 0:
        msg = load().qrlsh_last_error().decode("utf-8", "replace")
        if rc == QRLSH_EUNSUPPORTED:
            raise NotImplementedError("libqrlsh: " + msg)
        raise QrlshError(rc, msg)
    return rc


def prof_enable(on=True):
    check(load().qrlsh_prof_enable(1 if on else 0))


def prof_pause(paused=True):
    """stop / resume the event bracketing without clearing the records"""
    check(load().qrlsh_prof_pause(1 if paused else 0))


def prof_report():
    """-> {label: (count, total_ms)} of the kernels launched since prof_enable(True)."""
    buf = ctypes.create_string_buffer(1 << 16)
    check(load().qrlsh_prof_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out
