"""ctypes binding of libxmap_hip.so (C ABI declared in include/xmap_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, import of this
module raises (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C x-map_amd/csrc`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XMAP_HIP_LIB") or os.path.normpath(os.path.join(HERE, "..", "..", "libxmap_hip.so"))
# the same sources + the TEST formulations (-DXMAP_CROSSCHECK): loaded on demand by xlib(), by tests and tuning tools only
XLIB_PATH = os.environ.get("XMAP_HIP_XLIB") or os.path.normpath(os.path.join(HERE, "..", "..", "libxmap_hip_xcheck.so"))

COSINE, ADJUST_COSINE = 0, 1
METHODS = {"cosine": COSINE, "adjust_cosine": ADJUST_COSINE}
TOPC = 10
MID_ROWS_SPAN = 36864     # XMAP_MID_ROWS_SPAN: columns of a middle-list row per LDS pass
ERR_HIP, ERR_ARG, ERR_OVERFLOW, ERR_CAPACITY = -1, -2, -3, -4     # XMAP_ERR_* of include/xmap_hip.h


class XmapError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libxmap_hip error %d: %s" % (code, msg))
        self.code = code


class Ratings(C.Structure):
    _fields_ = [("n_users", C.c_int64), ("n_items", C.c_int32), ("nnz", C.c_int64),
                ("user_ptr", C.c_void_p), ("user_item", C.c_void_p), ("user_rating", C.c_void_p),
                ("user_time", C.c_void_p), ("item_ptr", C.c_void_p), ("item_user", C.c_void_p),
                ("item_rating", C.c_void_p), ("prefix_cls", C.c_void_p), ("suffix_cls", C.c_void_p),
                ("contains_mask", C.c_void_p), ("flags", C.c_void_p)]


class Sim(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("row_ptr", C.c_void_p), ("col", C.c_void_p), ("sim", C.c_void_p),
                ("mutu", C.c_void_p), ("nij", C.c_void_p), ("info", C.c_void_p), ("frac", C.c_void_p)]


class ExtTables(C.Structure):
    """xmap_ext_tables: the stage-B tables of one pass (device pointers)"""
    _fields_ = [("n_items", C.c_int32), ("top_k", C.c_int32),
                ("cls", C.c_void_p), ("kcnt", C.c_void_p), ("kcol", C.c_void_p), ("kval", C.c_void_p), ("flags", C.c_void_p),
                ("att_ptr", C.c_void_p), ("att_idx", C.c_void_p), ("att_val", C.c_void_p),
                ("src_ptr", C.c_void_p), ("src_idx", C.c_void_p), ("src_val", C.c_void_p), ("src_flag", C.c_void_p),
                ("rnn_ptr", C.c_void_p), ("rnn_idx", C.c_void_p), ("rnn_val", C.c_void_p),
                ("n_nb", C.c_int32), ("nb_id", C.c_void_p), ("nb_list", C.c_void_p), ("midX", C.c_void_p), ("dir", C.c_void_p),
                ("dir_ptr", C.c_void_p),
                ("n_ends", C.c_int32), ("urank", C.c_void_p), ("uitem", C.c_void_p)]


class PathUnits(C.Structure):
    _fields_ = [("n_units", C.c_int32), ("unit_start", C.c_void_p), ("unit_c", C.c_void_p), ("unit_G", C.c_void_p),
                ("unit_row", C.c_void_p), ("unit_nt", C.c_void_p), ("n_heavy", C.c_int32), ("heavy_unit0", C.c_void_p)]


class PathRows(C.Structure):
    _fields_ = [("n_slots", C.c_int32), ("acc", C.c_void_p), ("touched", C.c_void_p), ("hacc", C.c_void_p),
                ("htouched", C.c_void_p)]


class PathOut(C.Structure):
    _fields_ = [("n_cand", C.c_void_p), ("top_end", C.c_void_p), ("top_val", C.c_void_p), ("xs_cap", C.c_int64),
                ("xs_off", C.c_void_p), ("xs_end", C.c_void_p), ("xs_val", C.c_void_p)]


EXPORTS = [
    "xmap_last_error", "xmap_version", "xmap_trim", "xmap_debug_arena", "xmap_debug_arena_call", "xmap_exclusive_scan_i64", "xmap_exclusive_scan_i32_to_i64",
    "xmap_build_csc", "xmap_user_stats", "xmap_item_stats", 
    "xmap_sim2_layout", "xmap_sim2_plan", "xmap_sim2_units",
    "xmap_sim2_pairs", "xmap_sim2_scatter", "xmap_sim3_layout", "xmap_sim3_plan", "xmap_sim3_mircount", "xmap_sim3_mirror", "xmap_item_partials", "xmap_item_merge", "xmap_sim2_pack_partials",
    "xmap_sim2_sort_partials", "xmap_sim2_merge_partials", "xmap_sim2_pack_pairs", "xmap_sim2_unpack_pairs", "xmap_bridge_flags", "xmap_knn_classify", "xmap_knn_thresholds", "xmap_reverse_count", "xmap_reverse_count_att_rnn",
    "xmap_reverse_fill", "xmap_topc_from_lists", "xmap_path_weights", "xmap_extend_paths", 
    "xmap_mid_rows_count", "xmap_mid_rows_place", "xmap_edge_ranges", "xmap_end_universe", "xmap_extend_cols", "xmap_extend_cols_slots", "xmap_nb_index", "xmap_path_plan", "xmap_end_order", "xmap_dense_normalize", "xmap_dense_layout", "xmap_dense_topk", "xmap_rec_select", "xmap_predict", "xmap_select_map", "xmap_alterego_count", "xmap_alterego_fill",
    "xmap_feed_text", "xmap_feed_texts", "xmap_feed_merge", "xmap_feed_sizes", "xmap_feed_arrays", "xmap_feed_ids", "xmap_feed_free",
    "xmap_ctx_upload_feed", "xmap_feed_format", "xmap_ctx_create", "xmap_ctx_destroy", "xmap_ctx_upload_ratings", "xmap_ctx_item_sim", "xmap_ctx_sim_download", "xmap_ctx_extend",
    "xmap_ctx_ext_download", "xmap_ctx_ext_lists", "xmap_ctx_candidates", "xmap_ctx_generate", "xmap_ctx_gen_download",
]

if not os.path.exists(LIB_PATH):
    raise ImportError("libxmap_hip.so not built (%s); the MI355X engine has no CPU fallback" % LIB_PATH)
lib = C.CDLL(LIB_PATH)
lib.xmap_last_error.restype = C.c_char_p
for _n in EXPORTS:
    getattr(lib, _n)  # every symbol the header declares must be exported
# declared under #ifdef XMAP_CROSSCHECK in the header: the test formulations (stage A by complete rows, the round-1 tile-major
# enumeration, the dense-table form of the middle lists) -- exported by libxmap_hip_xcheck.so only
XCHECK_EXPORTS = ["xmap_sim_plan", "xmap_sim_units", "xmap_sim_count", "xmap_sim_fill", "xmap_sim_row_ptr", "xmap_extend_paths2",
                  "xmap_mid_tally", "xmap_mid_place"]

HEADER_PATH = os.path.normpath(os.path.join(HERE, "..", "..", "..", "include", "xmap_hip.h"))


def header_prototypes(path=HEADER_PATH):
    """{function: [ctypes type per parameter]} parsed from the declarations of include/xmap_hip.h (the single source of
    the C ABI): a pointer of any kind -> c_void_p, int64_t -> c_int64, int / int32_t -> c_int32, float / double."""
    import re
    with open(path) as f:
        text = f.read()
    text = re.sub(r"#\s*ifdef\s+XMAP_CROSSCHECK[^\n]*\n", "\n", text)      # (the guarded prototypes are parsed too: XCHECK_EXPORTS)
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const\s+char\s*\*)\s*(xmap_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        types = []
        for a in ([] if args in ("", "void") else args.split(",")):
            if "*" in a:
                types.append(C.c_void_p)
            elif "int64_t" in a:
                types.append(C.c_int64)
            elif "double" in a:
                types.append(C.c_double)
            elif "float" in a:
                types.append(C.c_float)
            else:
                types.append(C.c_int32)
        out[name] = types
    return out


# argtypes of every export: a mis-ordered or mis-typed argument raises in ctypes instead of corrupting device memory
PROTOTYPES = header_prototypes()


def _bind(L, names):
    for _n in names:
        _f = getattr(L, _n)
        _f.argtypes = PROTOTYPES[_n]
        if _n in ("xmap_ctx_destroy", "xmap_feed_free"):
            _f.restype = None
        elif _n != "xmap_last_error":
            _f.restype = C.c_int


_bind(lib, [n for n in PROTOTYPES if n not in XCHECK_EXPORTS])
_xlib = None


def xlib():
    """libxmap_hip_xcheck.so: every product entry point + the test formulations.  Loaded on first use (tests, tuning tools:
    Engine.item_sim(algo="rows"), extend(algo="mid"), XMAP_MID_TABLE=1); the pipelines never call it."""
    global _xlib
    if _xlib is None:
        if not os.path.exists(XLIB_PATH):
            raise ImportError("libxmap_hip_xcheck.so not built (%s): make -C x-map_amd/csrc" % XLIB_PATH)
        L = C.CDLL(XLIB_PATH)
        L.xmap_last_error.restype = C.c_char_p
        _bind(L, list(PROTOTYPES))
        _xlib = L
    return _xlib


def check(rc):
    if rc != 0:
        raise XmapError(rc, (lib.xmap_last_error() or b"").decode())


def xcheck(rc):
    if rc != 0:
        raise XmapError(rc, (xlib().xmap_last_error() or b"").decode())


def vp(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def i64(v):
    return C.c_int64(int(v))


def i32(v):
    return C.c_int32(int(v))
