"""Per-unit begin / end of k_paths2 ("What bounds k_paths2").  Build the traced library first:
    cd x-map_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DB_TRACE -c stage_b.hip -o /tmp/bt.o \
      && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libxmap_BTRACE.so _build/util.o _build/stage_a.o _build/stage_a2.o \
         /tmp/bt.o _build/stage_c.o _build/stage_d.o _build/stage_e.o
then on the GPU box, from the repo root: python profiles/tools/trace_b.py"""
import sys, os, ctypes as C, numpy as np, torch
os.environ["XMAP_HIP_LIB"] = "scratch/libxmap_BTRACE.so"
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
from xmap.engine.hipabi import lib
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
S = eng.item_sim("adjust_cosine", 50)
for it in range(2):
    eng.timers = {}
    E = eng.extend(S, 50); torch.cuda.synchronize()
print({k: np.round(v, 1).tolist() for k, v in eng.timer_ms().items() if k in ("paths",)})
U = E.units
n = min(U.n_units, 1 << 20)
buf = np.zeros((n, 2), np.uint64)
lib.xmap_debug_btrace.argtypes = [C.c_void_p, C.c_longlong]
rc = lib.xmap_debug_btrace(buf.ctypes.data, n); assert rc == 0
t0 = buf[:, 0].min()
b = (buf[:, 0] - t0) / 100.0; e = (buf[:, 1] - t0) / 100.0      # us
dur = e - b
print("kernel span us", e.max(), "units", n)
# per-start path counts
tmp = eng._zeros(4 * r.n_items, torch.int64); 
us = U.unit_start.cpu().numpy()[:n]; g = U.unit_G.cpu().numpy()[:n]; row = U.unit_row.cpu().numpy()[:n]
# exact path counts per start: recompute through path_units internals is awkward; use E.n_paths total and per-start from a second call
P = eng._zeros(r.n_items, torch.int64)
from xmap.engine.hipabi import check, vp, i32
st = dev._stream(eng.dev)
check(lib.xmap_path_weights(st, i32(r.n_items), E.k, vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(eng.R.flags), vp(E.att[0]), vp(E.att[1]), vp(E.src[0]), vp(E.src[1]), vp(E.src[3]), vp(E.rnn[0]), vp(E.rnn[1]), vp(tmp), vp(P)))
p = P.cpu().numpy()
pu = p[us] / g
heavy = row >= 0
print("heavy units %d: dur us pct %s ; paths/unit %.3g ; ns/path pct %s" % (heavy.sum(), np.percentile(dur[heavy], [0, 10, 50, 90, 100]).round(0), pu[heavy].mean(), np.percentile(1e3 * dur[heavy] / pu[heavy], [0, 10, 50, 90, 100]).round(2)))
lt = ~heavy
print("light units %d: dur us pct %s ; ns/path pct %s" % (lt.sum(), np.percentile(dur[lt], [0, 10, 50, 90, 99, 100]).round(0), np.percentile(1e3 * dur[lt] / np.maximum(pu[lt], 1), [0, 10, 50, 90, 100]).round(2)))
for lo, hi in ((0, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 4e6), (4e6, 2e7)):
    m = lt & (pu >= lo) & (pu < hi)
    if m.sum(): print("  light paths [%g,%g): n %d, total paths %.3g, total time s %.3f, ns/path %.2f" % (lo, hi, m.sum(), pu[m].sum(), dur[m].sum() / 1e6, 1e3 * dur[m].sum() / pu[m].sum()))
print("  heavy: total paths %.3g total time s %.3f ns/path %.2f" % (pu[heavy].sum(), dur[heavy].sum() / 1e6, 1e3 * dur[heavy].sum() / pu[heavy].sum()))
print("sum of unit time s %.2f -> per slot (5120) ms %.1f" % (dur.sum() / 1e6, dur.sum() / 5120 / 1e3))
# when do units end: time at which X% of the total work is done
order = np.argsort(e)
print("end time pct us", np.percentile(e, [50, 90, 99, 99.9, 100]).round(0))
# last finishing units
last = order[-10:]
for u in last: print("  late unit %d start %d G %d row %d paths/unit %.3g begin %.0f end %.0f" % (u, us[u], g[u], row[u], pu[u], b[u], e[u]))
# concurrency over time
for t in (100e3, 300e3, 500e3, 600e3, 700e3, 800e3, 850e3, 900e3):
    print("  t=%.0f ms running units %d" % (t / 1e3, int(((b <= t) & (e > t)).sum())))
