"""The coarse, handle-based C ABI (include/xmap_hip.h: xmap_ctx_*) driven the way a non-Python host would drive it:
plain host arrays (numpy here) through ctypes, no torch, no device pointer on the caller's side -- on the golden
vectors captured from the reference.  Every stage is compared with the reference's own outputs."""
import ctypes as C

import numpy as np
import pytest

from golden_util import METHODS, CAP, Golden

pytestmark = pytest.mark.gpu


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class Ctx(object):
    """a minimal foreign-host binding: what INTEGRATION.md shows for cgo / JNI, in ctypes"""

    def __init__(self):
        from xmap.engine import hipabi          # loads libxmap_hip.so and sets the argtypes from the header
        self.lib, self.abi = hipabi.lib, hipabi
        self.h = C.c_void_p()
        self.abi.check(self.lib.xmap_ctx_create(0, C.byref(self.h)))

    def close(self):
        self.lib.xmap_ctx_destroy(self.h)

    def call(self, name, *args):
        self.abi.check(getattr(self.lib, name)(self.h, *args))


def _draw(n_top):
    starts = np.nonzero(n_top)[0]
    picks = np.zeros(len(n_top), np.int32)
    if len(starts):
        high = n_top[starts].astype(np.int64) - 1
        if (high <= 0).any():
            raise ValueError("low >= high")
        picks[starts] = np.random.randint(0, high)
    return picks


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("case", ["kat7", "small", "multilabel"])
def test_golden_through_the_coarse_abi(case, method):
    gold = Golden(case)
    I, U = gold.I, len(gold.ptr) - 1
    pre, suf, mask, flags = [np.ascontiguousarray(a, t) for a, t in zip(gold.attrs, (np.int32, np.int32, np.uint32, np.uint8))]
    ptr = np.ascontiguousarray(gold.ptr, np.int64)
    item = np.ascontiguousarray(gold.item, np.int32)
    rating = np.ascontiguousarray(gold.rating, np.float32)
    pos = np.ascontiguousarray(gold.time, np.int64)
    ctx = Ctx()
    try:
        ctx.call("xmap_ctx_upload_ratings", U, I, _p(ptr, C.c_int64), _p(item, C.c_int32), _p(rating, C.c_float), _p(pos, C.c_int64),
                 _p(pre, C.c_int32), _p(suf, C.c_int32), _p(mask, C.c_uint32), _p(flags, C.c_uint8))
        # ---- stage A
        n_kept, n_eval = C.c_int64(0), C.c_int64(0)
        ctx.call("xmap_ctx_item_sim", 0 if method == "cosine" else 1, CAP, C.byref(n_kept), C.byref(n_eval))
        n = n_kept.value
        rp, col, sim = np.zeros(I + 1, np.int64), np.zeros(n, np.int32), np.zeros(n, np.float64)
        mutu, nij, info, uavg = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((I, 4)), np.zeros(U)
        ctx.call("xmap_ctx_sim_download", _p(rp, C.c_int64), _p(col, C.c_int32), _p(sim, C.c_double), _p(mutu, C.c_int32),
                 _p(nij, C.c_int32), _p(info, C.c_double), _p(uavg, C.c_double))
        rows = np.repeat(np.arange(I), np.diff(rp))
        o = np.lexsort((col, rows))
        assert np.array_equal(rows[o], gold[method + ".sim_i"]) and np.array_equal(col[o], gold[method + ".sim_j"])
        val = gold[method + ".sim_val"]
        assert np.array_equal(mutu[o].astype(np.float64), val[:, 1])
        if method == "cosine":
            assert np.array_equal(sim[o], val[:, 0])
        else:
            np.testing.assert_allclose(sim[o], val[:, 0], rtol=1e-11, atol=0)
        assert np.array_equal(uavg, gold[method + ".user_info"][:, 0])
        assert np.array_equal(info[:, [0, 1, 3]], gold[method + ".item_info"][:, [0, 1, 3]])
        # ---- stage B / C
        for k in gold.ks(method):
            tag = "%s.k%d" % (method, k)
            n_out, n_paths = C.c_int64(0), C.c_int64(0)
            ctx.call("xmap_ctx_extend", k, C.byref(n_out), C.byref(n_paths))
            n_cand, top_end, top_val = np.zeros(I, np.int32), np.zeros((I, 10), np.int32), np.zeros((I, 10))
            ctx.call("xmap_ctx_ext_download", _p(n_cand, C.c_int32), _p(top_end, C.c_int32), _p(top_val, C.c_double))
            assert int(n_cand.sum()) == n_out.value
            off, xe, xv = np.zeros(I, np.int64), np.zeros(n_out.value, np.int32), np.zeros(n_out.value)
            ctx.call("xmap_ctx_ext_lists", _p(off, C.c_int64), _p(xe, C.c_int32), _p(xv, C.c_double))
            st, en, va = [], [], []
            for s in np.nonzero(n_cand)[0]:
                e, v = xe[off[s]:off[s] + n_cand[s]], xv[off[s]:off[s] + n_cand[s]]
                oo = np.argsort(e)
                st.append(np.full(len(e), s)); en.append(e[oo]); va.append(v[oo])
                best = np.lexsort((e, -np.abs(v)))[:10]
                assert np.array_equal(e[best], top_end[s, :len(best)]) and np.array_equal(v[best], top_val[s, :len(best)])
            xh = gold[tag + ".xsim_head"]
            if st:
                assert np.array_equal(np.concatenate(st), xh[:, 0]) and np.array_equal(np.concatenate(en), xh[:, 1])
                np.testing.assert_allclose(np.concatenate(va), gold[tag + ".xsim_val"], rtol=1e-9, atol=1e-300)
            else:
                assert len(xh) == 0
            for gt in gold.gen_tags(method, k):
                gtag = tag + "." + gt
                private = gt == "priv"
                picks = None
                if not private:
                    n_top = np.zeros(I, np.int32)
                    ctx.call("xmap_ctx_candidates", _p(n_top, C.c_int32))
                    np.random.seed(int(gt[2:]))
                    if gold.has(gtag + ".raises"):
                        with pytest.raises(ValueError):
                            _draw(n_top)
                        continue
                    picks = _draw(n_top)
                choice = np.zeros(I, np.int32)
                n_rows, n_tgt = C.c_int64(0), C.c_int64(0)
                ctx.call("xmap_ctx_generate", 1 if private else 0, _p(picks, C.c_int32), _p(choice, C.c_int32), C.byref(n_rows),
                         C.byref(n_tgt))
                exp = gold[gtag + ".choice"]
                assert np.array_equal(np.nonzero(n_cand)[0], exp[:, 0]) and np.array_equal(choice[exp[:, 0]], exp[:, 1])
                m = n_rows.value
                gu, gi, gr, gp = np.zeros(m, np.int32), np.zeros(m, np.int32), np.zeros(m), np.zeros(m, np.int64)
                ctx.call("xmap_ctx_gen_download", _p(gu, C.c_int32), _p(gi, C.c_int32), _p(gr, C.c_double), _p(gp, C.c_int64))
                eh = gold[gtag + ".ae_head"]
                assert np.array_equal(gu, eh[:, 0]) and np.array_equal(gi, eh[:, 1])
                assert np.array_equal(gr, gold[gtag + ".ae_rating"])
                assert np.array_equal(gp, gold[gtag + ".ae_time"])
    finally:
        ctx.close()


def test_coarse_abi_misuse_is_reported():
    ctx = Ctx()
    try:
        rc = ctx.lib.xmap_ctx_item_sim(ctx.h, 0, 50, None, None)                 # no ratings uploaded
        assert rc == ctx.abi.ERR_ARG and b"have_ratings" in ctx.lib.xmap_last_error()
        rc = ctx.lib.xmap_ctx_extend(ctx.h, 5, None, None)                       # no stage A
        assert rc == ctx.abi.ERR_ARG
    finally:
        ctx.close()
