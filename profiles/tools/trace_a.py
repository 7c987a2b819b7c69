"""Per-unit time stamps of k_pair_tri.  Build the traced library first:
    cd x-map_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DA_TRACE -c stage_a2.hip -o /tmp/a2t.o \
      && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libxmap_ATRACE.so _build/util.o _build/stage_a.o /tmp/a2t.o \
         _build/stage_b.o _build/stage_c.o _build/stage_d.o _build/stage_e.o
then on the GPU box, from the repo root: python profiles/tools/trace_a.py"""
import sys, os, ctypes as C, numpy as np, torch
os.environ["XMAP_HIP_LIB"] = "x-map_amd/_variants/libxmap_ATRACE.so"   # profiles/tools/a_variants.sh build ATRACE... (-DA_TRACE)
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
from xmap.engine.hipabi import lib
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
for it in range(3):
    eng.timers = {}
    S = eng.item_sim("adjust_cosine", 50); torch.cuda.synchronize()
print({k: np.round(v, 2).tolist() for k, v in eng.timer_ms().items()})
L = S.layout
n = L.n_light
buf = np.zeros((n, 2), np.uint64)
lib.xmap_debug_atrace.argtypes = [C.c_void_p, C.c_longlong]
assert lib.xmap_debug_atrace(buf.ctypes.data, n) == 0
t0 = buf[:, 0].min()
b = (buf[:, 0] - t0) / 100.0; e = (buf[:, 1] - t0) / 100.0
dur = e - b
cls = [int(x) for x in L.cls_ptr]
print("class boundaries", cls, "n_light", n)
item = L.uq_item[:n].cpu().numpy()
ni = (eng.R.item_ptr[1:] - eng.R.item_ptr[:-1]).cpu().numpy()
Q = L.Q.cpu().numpy()
names = ["1024x16w", "1024x4w", "512x2w", "256", "128"]
for c in range(5):
    lo, hi = cls[c], cls[c + 1]
    if hi <= lo: continue
    d = dur[lo:hi]; rr = ni[item[lo:hi]]
    print("class %d (%s): units %d, raters/unit mean %.0f max %d, Q mean %.2f; begin %.0f..%.0f us, end max %.0f; dur us pct[50,90,99,100] %s; sum dur ms %.1f" % (
        c, names[c], hi - lo, rr.mean(), rr.max(), Q[item[lo:hi]].mean(), b[lo:hi].min(), b[lo:hi].max(), e[lo:hi].max(), np.percentile(d, [50, 90, 99, 100]).round(1), d.sum() / 1e3))
    # concurrency
    for t in np.linspace(0, e[lo:hi].max(), 9)[1:-1]:
        print("    t=%.0f us running %d" % (t, int(((b[lo:hi] <= t) & (e[lo:hi] > t)).sum())), end="")
    print()
    # longest units
    o = np.argsort(-d)[:5]
    print("    longest:", [(int(item[lo + j]), int(rr[j]), int(Q[item[lo + j]]), round(float(d[j]), 1), round(float(b[lo + j]), 0)) for j in o])
print("span us", e.max())
st = np.zeros((n, 4), np.uint32)
lib.xmap_debug_astamp.argtypes = [C.c_void_p, C.c_longlong]
assert lib.xmap_debug_astamp(st.ctypes.data, n) == 0
for c in range(5):
    lo, hi = cls[c], cls[c + 1]
    if hi <= lo: continue
    x = st[lo:hi].astype(np.float64) / 100.0
    print("class %d stamps us (median): unit read %.1f, first rater step %.1f, walk done %.1f, end %.1f" % (c, np.median(x[:, 0]), np.median(x[:, 1]), np.median(x[:, 2]), np.median(dur[lo:hi])))
    rr = ni[item[lo:hi]].astype(np.float64)
    walk = x[:, 2] - x[:, 1]
    A = np.stack([rr, np.ones_like(rr)], 1)
    coef = np.linalg.lstsq(A, walk, rcond=None)[0]
    print("    walk us ~ %.3f per rater of the row + %.1f; finalise + append (end - walk done) median %.1f us; mean stamps %s, mean dur %.1f" % (
        coef[0], coef[1], np.median(dur[lo:hi] - x[:, 2]), x[:, :3].mean(0).round(1), dur[lo:hi].mean()))
