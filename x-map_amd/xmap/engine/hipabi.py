"""ctypes binding of libxmap_hip.so (C ABI declared in include/xmap_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, import of this
module raises (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C x-map_amd/csrc`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XMAP_HIP_LIB") or os.path.normpath(os.path.join(HERE, "..", "..", "libxmap_hip.so"))

COSINE, ADJUST_COSINE = 0, 1
METHODS = {"cosine": COSINE, "adjust_cosine": ADJUST_COSINE}
TOPC = 10
MID_ROWS_MAX = 40000      # XMAP_MID_ROWS_MAX
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


EXPORTS = [
    "xmap_last_error", "xmap_version", "xmap_exclusive_scan_i64", "xmap_exclusive_scan_i32_to_i64",
    "xmap_build_csc", "xmap_user_stats", "xmap_item_stats", "xmap_sim_plan", "xmap_sim_units", "xmap_sim_count",
    "xmap_sim_fill", "xmap_sim_row_ptr", "xmap_sim2_layout", "xmap_sim2_plan", "xmap_sim2_units",
    "xmap_sim2_pairs", "xmap_sim2_scatter", "xmap_bridge_flags", "xmap_knn_classify", "xmap_knn_thresholds", "xmap_reverse_count",
    "xmap_reverse_fill", "xmap_topc_from_lists", "xmap_path_weights", "xmap_extend_paths", "xmap_mid_tally", "xmap_mid_place",
    "xmap_mid_rows_count", "xmap_mid_rows_place", "xmap_extend_paths2", "xmap_dense_normalize", "xmap_dense_layout", "xmap_dense_topk", "xmap_rec_select", "xmap_select_map", "xmap_alterego_count", "xmap_alterego_fill",
]

if not os.path.exists(LIB_PATH):
    raise ImportError("libxmap_hip.so not built (%s); the MI355X engine has no CPU fallback" % LIB_PATH)
lib = C.CDLL(LIB_PATH)
lib.xmap_last_error.restype = C.c_char_p
for _n in EXPORTS:
    getattr(lib, _n)  # every symbol the header declares must be exported


def check(rc):
    if rc != 0:
        raise XmapError(rc, (lib.xmap_last_error() or b"").decode())


def vp(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def i64(v):
    return C.c_int64(int(v))


def i32(v):
    return C.c_int32(int(v))
