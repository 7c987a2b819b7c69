"""Driver entry points: build() compiles every HIP extension for gfx950 (+ the CPU oracle, which is
the checker, not the product); smoke() runs one small invocation of the hot path on cuda:0 and
checks it against the oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "x-map_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def build():
    env = dict(os.environ)
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "ARCH=gfx950"], env=env)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], env=env)
    # the reference is pure Python (no native sources), so there is no oracle/_ref to build.
    from xmap.engine import hipabi  # noqa: F401  loads libxmap_hip.so and checks every declared export
    import xmap.utils.assist  # noqa: F401
    import xmap.core  # noqa: F401


def smoke():
    import numpy as np
    import torch
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    from oracle import xmap_oracle as xo
    from xmap.engine import synth, device
    r = synth.make_two_domain(7, 2000, 500, 500)
    attrs = r.item_attrs()
    eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs, "cuda:0"))
    S = eng.item_sim("cosine", 50)
    E = eng.extend(S, 5, full=True)
    n_top, choice, mp = eng.select(E, True)
    G = eng.alterego(mp)
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *attrs)
    So = xo.item_sim(T, "cosine", 50)
    Xo = xo.extend(T, So, 5)
    _, choice_o, m_o = xo.select(T, Xo, True, None)
    ae = xo.alterego(T, m_o)
    row_ptr = S.row_ptr.cpu().numpy()
    rows = np.repeat(np.arange(r.n_items), np.diff(row_ptr))
    o = np.lexsort((S.col.cpu().numpy(), rows))
    assert np.array_equal(row_ptr, So.row_ptr)
    assert np.array_equal(S.col.cpu().numpy()[o], So.col) and np.array_equal(S.sim.cpu().numpy()[o], So.sim)
    assert E.n_paths == Xo.n_paths
    assert np.array_equal(choice.cpu().numpy()[:r.n_items], choice_o)
    assert np.array_equal(G.user.cpu().numpy(), ae["user"]) and np.array_equal(G.item.cpu().numpy(), ae["item"])
    print("smoke ok: %d kept pairs, %d paths, %d AlterEgo rows" % (S.n_kept, E.n_paths, G.n_rows))


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
