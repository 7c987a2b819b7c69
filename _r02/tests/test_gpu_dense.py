"""Dense item-factor variant (BASELINE.json configs[4]; no reference counterpart, parity unpinned by construction):
the MFMA contraction + fused row top-k against the oracle's fp32 fmaf chain, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def factors(seed, n, K, dup=0, zero_rows=()):
    rng = np.random.RandomState(seed)
    F = rng.standard_normal((n, K)).astype(np.float32)
    if dup:   # exact duplicates and negated copies: ties in |sim| that the index order has to break
        src = rng.randint(0, n, dup)
        dst = rng.randint(0, n, dup)
        F[dst] = F[src] * np.where(rng.rand(dup) < 0.5, -1.0, 1.0).astype(np.float32)[:, None]
    for z in zero_rows:
        F[z] = 0.0
    return F


def engine():
    import torch
    assert torch.cuda.is_available()
    from xmap.engine import device, synth
    r = synth.make_two_domain(3, 60, 30, 30, overlap=0.5)
    R = device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs(), "cuda:0")
    return device.Engine(R), r


@pytest.mark.parametrize("n_t,n_s,K,k", [
    (1, 1, 128, 1),            # single pair
    (5, 3, 128, 8),            # fewer sources than k: padded lists
    (129, 33, 64, 10),         # one row past a workgroup, one column past a tile
    (1000, 1500, 128, 50),     # ragged in both dimensions, headline k
    (300, 2049, 64, 64),       # maximum list length
    (257, 700, 128, 3),
])
def test_dense_topk_bit_exact(n_t, n_s, K, k):
    from oracle import xmap_oracle as xo
    eng, _ = engine()
    Ft = factors(n_t * 7 + 1, n_t, K, dup=n_t // 8, zero_rows=(0,) if n_t > 4 else ())
    Fs = factors(n_s * 5 + 2, n_s, K, dup=n_s // 4, zero_rows=(n_s // 2,) if n_s > 4 else ())
    Fs[: min(n_s, n_t) // 2] = Ft[: min(n_s, n_t) // 2]     # exact matches: |sim| = 1 candidates
    idx, val = eng.dense_topk(Ft, Fs, k)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    oi, ov = xo.dense_topk(xo.dense_normalize(Ft), xo.dense_normalize(Fs), k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(val.view(np.uint32), ov.view(np.uint32))
    # size-independent properties: sorted by (|v| desc, idx asc), padding only at the tail
    a = np.abs(val)
    assert np.all(a[:, :-1] >= a[:, 1:]) if k > 1 else True
    tie = (a[:, :-1] == a[:, 1:]) & (idx[:, 1:] >= 0) if k > 1 else np.zeros((n_t, 0), bool)
    assert np.all(idx[:, :-1][tie] < idx[:, 1:][tie])
    assert np.all((idx >= 0).sum(1) == min(k, n_s))


@pytest.mark.parametrize("n_t,n_s", [(600, 9000), (70, 40000), (5000, 3000)])
def test_dense_pieces(n_t, n_s):
    """Sizes at which a row block's source tiles fall into several workgroup shares (pieces + merge)."""
    from oracle import xmap_oracle as xo
    eng, _ = engine()
    Ft, Fs = factors(41, n_t, 64, dup=n_t // 10), factors(42, n_s, 64, dup=n_s // 5)
    idx, val = eng.dense_topk(Ft, Fs, 50)
    oi, ov = xo.dense_topk(xo.dense_normalize(Ft), xo.dense_normalize(Fs), 50)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(val.cpu().numpy().view(np.uint32), ov.view(np.uint32))


def test_dense_normalize_matches_oracle_and_fp64():
    import torch
    from oracle import xmap_oracle as xo
    from xmap.engine import hipabi as abi
    F = factors(9, 777, 128, zero_rows=(5,))
    d = torch.from_numpy(F).cuda()
    out = torch.empty_like(d)
    import ctypes as C
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    abi.check(abi.lib.xmap_dense_normalize(st, abi.i32(777), abi.i32(128), abi.vp(d), abi.vp(out)))
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), xo.dense_normalize(F).view(np.uint32))
    ref = F.astype(np.float64) / np.maximum(np.linalg.norm(F.astype(np.float64), axis=1, keepdims=True), 1e-300)
    assert np.allclose(got, ref, rtol=2e-7, atol=0)      # fp32 rounding of an fp64 quotient
    assert np.all(got[5] == 0)


def test_dense_against_fp64_gemm():
    """The defining arithmetic: fp64 normalised GEMM.  The fp32 chain agrees within 1e-5 on the
    values (north_star's fp32 tolerance) and the top-k sets agree wherever the fp64 gap at rank k exceeds it."""
    eng, _ = engine()
    Ft, Fs = factors(21, 400, 128), factors(22, 900, 128)
    k = 20
    idx, val = eng.dense_topk(Ft, Fs, k)
    idx, val = idx.cpu().numpy(), val.cpu().numpy().astype(np.float64)
    nt = Ft.astype(np.float64) / np.linalg.norm(Ft.astype(np.float64), axis=1, keepdims=True)
    ns = Fs.astype(np.float64) / np.linalg.norm(Fs.astype(np.float64), axis=1, keepdims=True)
    G = nt @ ns.T
    assert np.allclose(val, np.take_along_axis(G, idx, 1), rtol=0, atol=1e-5)
    order = np.argsort(-np.abs(G), axis=1, kind="stable")
    srt = np.take_along_axis(np.abs(G), order, 1)
    clear = (srt[:, k - 1] - srt[:, k]) > 1e-5
    assert clear.sum() > 300
    assert all(set(idx[i]) == set(order[i, :k]) for i in np.nonzero(clear)[0])


def test_dense_extend_feeds_stage_c():
    """Dense replacement of stages A+B -> the unchanged stage C: private mapping = the best |sim| source item."""
    import torch
    from oracle import xmap_oracle as xo
    eng, r = engine()
    I = r.n_items
    F = factors(31, I, 64)
    E = eng.dense_extend(F, 10)
    flags = eng.R.flags[:I].cpu().numpy()
    tgt, src = np.nonzero(flags & 2)[0], np.nonzero(flags & 1)[0]
    Fn = xo.dense_normalize(F)
    oi, ov = xo.dense_topk(Fn[tgt], Fn[src], 10)
    te = E.top_end.cpu().numpy()
    assert np.array_equal(te[tgt], src[oi].astype(np.int32))
    assert np.array_equal(E.top_val.cpu().numpy()[tgt], ov.astype(np.float64))
    n_top, choice, mp = eng.select(E, True)
    mp = mp.cpu().numpy()
    # last-writer-wins over starts in ascending order (assist.py:210 map_to_dict): the mapping is source -> target
    exp = {}
    for t, s in zip(tgt, src[oi[:, 0]]):
        exp[int(s)] = int(t)
    assert {int(s): int(mp[s]) for s in range(I) if mp[s] >= 0} == exp
    G = eng.alterego(torch.from_numpy(mp).cuda())
    assert G.n_rows > 0


def test_dense_error_codes():
    """the C ABI reports misuse through its return code and xmap_last_error, it never launches on bad shapes"""
    import ctypes as C
    import torch
    from xmap.engine import hipabi as abi
    lib, vp, i32 = abi.lib, abi.vp, abi.i32
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    F = torch.randn(300, 96, device="cuda")
    out_i = torch.empty((300, 8), dtype=torch.int32, device="cuda")
    out_v = torch.empty((300, 8), dtype=torch.float32, device="cuda")
    rc = lib.xmap_dense_topk(st, i32(300), i32(300), i32(96), vp(F), vp(F), i32(8), i32(1), vp(None), vp(None), vp(out_i), vp(out_v))
    assert rc == abi.ERR_ARG and b"96" in lib.xmap_last_error()
    G = torch.randn(300, 128, device="cuda")
    rc = lib.xmap_dense_topk(st, i32(300), i32(300), i32(128), vp(G), vp(G), i32(65), i32(1), vp(None), vp(None), vp(out_i), vp(out_v))
    assert rc == abi.ERR_ARG
    big = torch.randn(70, 128, device="cuda")
    many = torch.randn(40000, 128, device="cuda")
    npc = C.c_int32(0)
    abi.check(lib.xmap_dense_layout(i32(70), i32(40000), C.byref(npc)))
    assert npc.value > 1
    oi = torch.empty((70, 8), dtype=torch.int32, device="cuda")
    ov = torch.empty((70, 8), dtype=torch.float32, device="cuda")
    rc = lib.xmap_dense_topk(st, i32(70), i32(40000), i32(128), vp(big), vp(many), i32(8), i32(1), vp(None), vp(None), vp(oi), vp(ov))
    assert rc == abi.ERR_CAPACITY and b"pieces" in lib.xmap_last_error()
    with pytest.raises(abi.XmapError):
        abi.check(rc)
