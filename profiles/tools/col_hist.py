"""Column visits of k_paths4 at BASELINE configs[1] by (heads of the batch, ends of the chunk, prepared records of the round):
what a specialisation of the column loop could apply to (library built with -DQ_HIST: one global atomic per round).
    FILE=stage_b profiles/tools/a_variants.sh build "Q_HIST"          (here)
    python profiles/tools/col_hist.py [lib]                            (GPU box)"""
import sys, os, ctypes as C, numpy as np, torch
os.environ["XMAP_HIP_LIB"] = sys.argv[1] if len(sys.argv) > 1 else "x-map_amd/_variants/libxmap_Q_HIST.so"
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
from xmap.engine.hipabi import lib
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
S = eng.item_sim("adjust_cosine", 50)
lib.xmap_debug_qhist.argtypes = [C.c_void_p, C.c_int]
lib.xmap_debug_qhist(None, 1)
E = eng.extend(S, 50); torch.cuda.synchronize()
buf = np.zeros(128, np.uint64)
assert lib.xmap_debug_qhist(buf.ctypes.data, 0) == 0
h = buf[:105].astype(np.float64).reshape(3, 5, 7)
tot = h.sum()
print("rounds (= column visits but for the < 1 %% with more than one round): %.4g" % tot)
hn = ["the start's only head", "one of several heads has the column", "several heads share the column"]
en = ["<= 4", "5-8", "9-16", "17-32", "33-64"]
rn = ["1", "2", "3-4", "5-8", "9-16", "17-32", "> 32"]
print("\nby heads: " + "; ".join("%s %.1f %%" % (hn[i], 100 * h[i].sum() / tot) for i in range(3)))
print("by ends of the chunk: " + "; ".join("%s: %.1f %%" % (en[i], 100 * h[:, i].sum() / tot) for i in range(5)))
print("by prepared records: " + "; ".join("%s: %.1f %%" % (rn[i], 100 * h[:, :, i].sum() / tot) for i in range(7)))
print("\nends (rows) x records (columns), %% of all rounds")
print("%8s " % "" + " ".join("%7s" % x for x in rn))
for i in range(5):
    print("%8s " % en[i] + " ".join("%7.2f" % (100 * h[:, i, j].sum() / tot) for j in range(7)))
