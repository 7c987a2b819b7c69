"""The reference's own large scenario (S1 of TechReport_XMap.pdf) through the default path -- a module of its own: it needs
~130 GB of HBM, so it runs after the BASELINE configs[1] fixture of test_gpu_fullsize.py has been torn down."""
import numpy as np
import pytest

from test_gpu_fullsize import CAP, _candidate_properties, stage_b_against_the_oracle

pytestmark = pytest.mark.gpu


def test_s1_shape_full_size():
    """The reference's own large scenario (TechReport Table 3/5: 128 402 movies -> 403 234 books, 3 % shared users) at the
    reference's own list length (parameters.yaml: extend_among_topk 10) through the default stage-B path: 3.3e5 non-bridge
    items -- eight LDS spans of columns per middle-list row --, column form of the enumeration (2.9e10 paths); against the
    per-path enumeration bit for bit, and the list invariants.  (At k = 50 this shape has 5.8e13 paths: the reference's
    algorithm itself, not a limit of the engine, which refuses it with a clear error.)"""
    import gc
    import torch
    from xmap.engine import synth, device, hipabi
    gc.collect()
    torch.cuda.empty_cache()                      # blocks cached by the tests before
    hipabi.lib.xmap_trim()
    r = synth.config_s1()
    assert r.n_src_items > 120000 and r.n_items - r.n_src_items > 380000
    eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
    I = r.n_items
    S = eng.item_sim("adjust_cosine", CAP)
    d = np.diff(r.user_ptr)
    assert S.n_contrib == int((d * (d - 1)).sum())
    E1 = eng.extend(S, 10)
    assert E1.algo == "cols" and E1.mid is not None
    assert E1.mid.n_nb > 4 * hipabi.MID_ROWS_SPAN            # several column ranges per row
    E2 = eng.extend(S, 10, algo="enum")
    assert E1.n_paths == E2.n_paths == E1.units.total and E1.n_out == E2.n_out and E1.n_paths > 0
    assert np.array_equal(E1.n_cand.cpu().numpy(), E2.n_cand.cpu().numpy())
    assert np.array_equal(E1.top_end.cpu().numpy(), E2.top_end.cpu().numpy())
    assert np.array_equal(E1.top_val.cpu().numpy(), E2.top_val.cpu().numpy())
    del E2
    _candidate_properties(r, E1, I)
    # against the CPU oracle at this shape (round 4; configs[1] has had this since round 3): bridge flags and classified
    # top-k lists of every item, the attach lists, and a sample of starts -- exact path counts, candidate counts, the ten
    # best candidates and their X-Sim values, bit for bit
    stage_b_against_the_oracle(r, S, E1, 10, "adjust_cosine", n_small=40, n_mid=4, at_least=40)
    n_top, choice, mp = eng.select(E1, True)
    G = eng.alterego(mp)
    flags = r.item_attrs()[3]
    assert G.n_target_rows == int((flags[r.item] & 2).astype(bool).sum()) and G.n_rows >= G.n_target_rows
    del E1, G
    with pytest.raises(hipabi.XmapError) as ei:          # k = 50: the algorithm's own path explosion, refused up front
        eng.extend(S, 50)
    assert ei.value.code == hipabi.ERR_CAPACITY and "paths" in str(ei.value)
    eng._scratch.clear()
    torch.cuda.empty_cache()
