"""ctypes front-end of the CPU oracle (oracle/xmap_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product path (x-map_amd/).
Parity status: PINNED against tests/golden/*.npz (vectors captured from the
reference's own modules by oracle/ref_harness/make_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libxmap_oracle.so")
METHODS = {"cosine": 0, "adjust_cosine": 1}
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "xmap_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class _XoSim(C.Structure):
    _fields_ = [("I", C.c_int32), ("n_eval", C.c_int64), ("n_contrib", C.c_int64),
                ("row_ptr", C.POINTER(C.c_int64)), ("col", C.POINTER(C.c_int32)),
                ("sim", C.POINTER(C.c_double)), ("mutu", C.POINTER(C.c_int32)),
                ("nij", C.POINTER(C.c_int32)), ("seconds", C.c_double * 3)]


class _XoExt(C.Structure):
    _fields_ = [("I", C.c_int32), ("k", C.c_int32), ("bb", C.POINTER(C.c_uint8)),
                ("cls", C.POINTER(C.c_uint8)), ("cnt", C.POINTER(C.c_int32)),
                ("col", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double)),
                ("n_paths", C.c_int64), ("xs_ptr", C.POINTER(C.c_int64)),
                ("xs_end", C.POINTER(C.c_int32)), ("xs_val", C.POINTER(C.c_double)), ("path_seconds", C.c_double)]


class _XoAlter(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("user", C.POINTER(C.c_int32)), ("item", C.POINTER(C.c_int32)),
                ("rating", C.POINTER(C.c_double)), ("time", C.POINTER(C.c_int64)),
                ("n_target_rows", C.c_int64), ("n_profiles", C.c_int64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.xo_item_sim.restype = C.POINTER(_XoSim)
        L.xo_extend.restype = C.POINTER(_XoExt)
        L.xo_alterego.restype = C.POINTER(_XoAlter)
        for f in (L.xo_user_info, L.xo_item_info, L.xo_sim_free, L.xo_ext_free, L.xo_alter_free, L.xo_select):
            f.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Train(object):
    """Index-space ratings in trainRDD order + per-item string-predicate arrays."""

    def __init__(self, user_ptr, item, rating, time, n_items, prefix_cls, suffix_cls, contains_mask, flags):
        self.ptr = np.ascontiguousarray(user_ptr, np.int64)
        self.item = np.ascontiguousarray(item, np.int32)
        self.rating = np.ascontiguousarray(rating, np.float32)
        self.time = np.ascontiguousarray(time, np.int64)
        self.I = int(n_items)
        self.U = len(self.ptr) - 1
        self.prefix_cls = np.ascontiguousarray(prefix_cls, np.int32)
        self.suffix_cls = np.ascontiguousarray(suffix_cls, np.int32)
        self.contains_mask = np.ascontiguousarray(contains_mask, np.uint32)
        self.flags = np.ascontiguousarray(flags, np.uint8)


def user_info(T):
    avg = np.zeros(T.U, np.float64)
    nrm = np.zeros(T.U, np.float64)
    lib().xo_user_info(C.c_int64(T.U), _p(T.ptr, C.c_int64), _p(T.rating, C.c_float),
                       _p(avg, C.c_double), _p(nrm, C.c_double))
    return avg, nrm


def item_info(T, uavg):
    info = np.zeros((T.I, 4), np.float64)
    lib().xo_item_info(C.c_int64(T.U), C.c_int32(T.I), _p(T.ptr, C.c_int64), _p(T.item, C.c_int32),
                       _p(T.rating, C.c_float), _p(uavg, C.c_double), _p(info, C.c_double))
    return info


class Sim(object):
    pass


def item_sim(T, method, cap, uavg=None, info=None, nthreads=1, rows=None):
    """Stage A.  Returns Sim with CSR arrays (row_ptr, col, sim, mutu, nij) + counters."""
    if uavg is None:
        uavg, _ = user_info(T)
    if info is None:
        info = item_info(T, uavg)
    lo, hi = rows if rows is not None else (0, T.I)
    h = lib().xo_item_sim(C.c_int(METHODS[method]), C.c_int(cap), C.c_int64(T.U), C.c_int32(T.I),
                          _p(T.ptr, C.c_int64), _p(T.item, C.c_int32), _p(T.rating, C.c_float),
                          _p(uavg, C.c_double), _p(info, C.c_double), C.c_int(nthreads),
                          C.c_int32(lo), C.c_int32(hi))
    s = h.contents
    out = Sim()
    out._h = h
    out.I = T.I
    out.n_eval = int(s.n_eval)
    out.n_contrib = int(s.n_contrib)
    out.seconds = tuple(float(x) for x in s.seconds)      # inside the C call: (item-major copy, rows, concatenation)
    out.row_ptr = _arr(s.row_ptr, T.I + 1, np.int64)
    D = int(out.row_ptr[-1])
    out.col = _arr(s.col, D, np.int32)
    out.sim = _arr(s.sim, D, np.float64)
    out.mutu = _arr(s.mutu, D, np.int32)
    out.nij = _arr(s.nij, D, np.int32)
    out.uavg, out.info = uavg, info
    return out


def sim_free(S):
    if getattr(S, "_h", None) is not None:
        lib().xo_sim_free(S._h)
        S._h = None


class Ext(object):
    pass


def extend(T, S, top_k, do_paths=True, s_range=None, max_seconds=0.0, starts=None):
    """Stage B from the oracle's own stage-A handle.  s_range = (lo, hi): paths of the source records in that item
    range only (a bounded sample for timing; the X-Sim lists are then partial).  starts = item indices: only the paths
    that start there (their X-Sim lists are complete; n_paths counts them alone)."""
    lo, hi = (0, T.I) if s_range is None else (int(s_range[0]), int(s_range[1]))
    want = None
    if starts is not None:
        want = np.zeros(max(T.I, 1), np.uint8)
        want[np.asarray(starts, np.int64)] = 1
    h = lib().xo_extend(S._h, C.c_int(top_k), _p(S.info, C.c_double), _p(T.prefix_cls, C.c_int32),
                        _p(T.suffix_cls, C.c_int32), _p(T.contains_mask, C.c_uint32),
                        _p(T.flags, C.c_uint8), C.c_int(1 if do_paths else 0), C.c_int32(lo), C.c_int32(hi),
                        C.c_double(max_seconds), _p(want, C.c_uint8) if want is not None else None)
    x = h.contents
    I, k = T.I, top_k
    out = Ext()
    out._h = h
    out.k = k
    out.bb = _arr(x.bb, I, np.uint8)
    out.cls = _arr(x.cls, I, np.uint8)
    out.cnt = _arr(x.cnt, I * 2, np.int32).reshape(I, 2)
    out.col = _arr(x.col, I * 2 * k, np.int32).reshape(I, 2, k)
    out.val = _arr(x.val, I * 2 * k * 3, np.float64).reshape(I, 2, k, 3)
    out.n_paths = int(x.n_paths)
    out.path_seconds = float(x.path_seconds)
    out.xs_ptr = _arr(x.xs_ptr, I + 1, np.int64)
    n = int(out.xs_ptr[-1])
    out.xs_end = _arr(x.xs_end, n, np.int32)
    out.xs_val = _arr(x.xs_val, n, np.float64)
    return out


def ext_free(X):
    if getattr(X, "_h", None) is not None:
        lib().xo_ext_free(X._h)
        X._h = None


def draw_picks(n_top, seed=None):
    """generator.py:110: one np.random.randint(0, len(top)-1) per start item, ascending start
    order, global NumPy RNG.  Raises ValueError on singleton candidate lists like the reference."""
    if seed is not None:
        np.random.seed(seed)
    picks = np.zeros(len(n_top), np.int32)
    for s in np.nonzero(n_top)[0]:
        picks[s] = np.random.randint(0, int(n_top[s]) - 1)
    return picks


def select(T, X, private, picks=None):
    n_top = np.zeros(T.I, np.int32)
    choice = np.zeros(T.I, np.int32)
    m = np.zeros(T.I, np.int32)
    pk = _p(np.ascontiguousarray(picks, np.int32), C.c_int32) if picks is not None else None
    lib().xo_select(X._h, C.c_int(1 if private else 0), pk, _p(n_top, C.c_int32),
                    _p(choice, C.c_int32), _p(m, C.c_int32))
    return n_top, choice, m


def alterego(T, map_src2tgt):
    m = np.ascontiguousarray(map_src2tgt, np.int32)
    h = lib().xo_alterego(C.c_int64(T.U), _p(T.ptr, C.c_int64), _p(T.item, C.c_int32),
                          _p(T.rating, C.c_float), _p(T.time, C.c_int64), _p(T.flags, C.c_uint8),
                          _p(m, C.c_int32))
    a = h.contents
    n = int(a.n_rows)
    out = dict(user=_arr(a.user, n, np.int32), item=_arr(a.item, n, np.int32),
               rating=_arr(a.rating, n, np.float64), time=_arr(a.time, n, np.int64),
               n_target_rows=int(a.n_target_rows), n_profiles=int(a.n_profiles))
    lib().xo_alter_free(h)
    return out


# ---------------------------------------------------------------------- dense item-factor variant (parity unpinned)
def dense_normalize(F):
    F = np.ascontiguousarray(F, np.float32)
    out = np.empty_like(F)
    lib().xo_dense_normalize(C.c_int32(F.shape[0]), C.c_int32(F.shape[1]), _p(F, C.c_float), _p(out, C.c_float))
    return out


def dense_topk(Fn_t, Fn_s, top_k, nthreads=4):
    """Top-k of the fp32 fmaf-chain dot of the (already normalised) factor rows, by (|v| desc, idx asc)."""
    Fn_t = np.ascontiguousarray(Fn_t, np.float32)
    Fn_s = np.ascontiguousarray(Fn_s, np.float32)
    n_t, K = Fn_t.shape
    idx = np.empty((n_t, top_k), np.int32)
    val = np.empty((n_t, top_k), np.float32)
    lib().xo_dense_topk(C.c_int32(n_t), C.c_int32(Fn_s.shape[0]), C.c_int32(K), _p(Fn_t, C.c_float),
                        _p(Fn_s, C.c_float), C.c_int32(top_k), _p(idx, C.c_int32), _p(val, C.c_float),
                        C.c_int(nthreads))
    return idx, val


# ---------------------------------------------------------------------- RecommenderSim
class _XoRec(C.Structure):
    _fields_ = [("I", C.c_int32), ("row_ptr", C.POINTER(C.c_int64)), ("col", C.POINTER(C.c_int32)),
                ("sim", C.POINTER(C.c_double)), ("ls", C.POINTER(C.c_double)), ("nij", C.POINTER(C.c_int32)),
                ("norm", C.POINTER(C.c_double))]


class Rec(object):
    pass


def rec_sim(user_ptr, item, rating, n_items, cap):
    """RecommenderSim.calculate_sim (cosine branch) in index space: CSR by first item of (col, sim, ls, n_ij) + norms."""
    L = lib()
    L.xo_rec_sim.restype = C.POINTER(_XoRec)
    user_ptr = np.ascontiguousarray(user_ptr, np.int64)
    item = np.ascontiguousarray(item, np.int32)
    rating = np.ascontiguousarray(rating, np.float64)      # np.float64 like the reference's AlterEgo ratings (generator.py:134)
    p = L.xo_rec_sim(C.c_int(int(cap)), C.c_int64(len(user_ptr) - 1), C.c_int32(int(n_items)), _p(user_ptr, C.c_int64),
                     _p(item, C.c_int32), _p(rating, C.c_double))
    c = p.contents
    R = Rec()
    R.row_ptr = _arr(c.row_ptr, n_items + 1, np.int64)
    D = int(R.row_ptr[-1])
    R.col, R.sim, R.ls, R.nij = _arr(c.col, D, np.int32), _arr(c.sim, D, np.float64), _arr(c.ls, D, np.float64), _arr(c.nij, D, np.int32)
    R.norm = _arr(c.norm, n_items, np.float64)
    R._h = p
    return R


def rec_select(R, keep):
    """nonprivate_neighbor_selection on a rec_sim result: (cnt [I], col [I][keep], sim, ls)"""
    I = len(R.row_ptr) - 1
    cnt = np.zeros(I, np.int32)
    col = np.zeros((I, keep), np.int32)
    sim = np.zeros((I, keep), np.float64)
    ls = np.zeros((I, keep), np.float64)
    lib().xo_rec_select(R._h, C.c_int32(keep), _p(cnt, C.c_int32), _p(col, C.c_int32), _p(sim, C.c_double), _p(ls, C.c_double))
    return cnt, col, sim, ls


def rec_free(R):
    if getattr(R, "_h", None) is not None:
        lib().xo_rec_free(R._h)
        R._h = None
