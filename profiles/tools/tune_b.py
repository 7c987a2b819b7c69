"""k_paths2 against the chunk size of the heavy starts (python profiles/tools/tune_b.py on the GPU box)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
S = eng.item_sim("adjust_cosine", 50)
E = eng.extend(S, 50)
U = E.units
p = None
print("default: units %d heavy %d rows %d chunk %d total %d" % (U.n_units, U.n_heavy, U.n_rows, U.chunk, U.total), flush=True)
g = U.unit_G.cpu().numpy(); us = U.unit_start.cpu().numpy()
hg = g[U.heavy_unit0.cpu().numpy()] if U.n_heavy else np.zeros(0)
print("heavy G: max %s mean %.1f; top10 %s" % (hg.max() if len(hg) else 0, hg.mean() if len(hg) else 0, np.sort(hg)[-10:]), flush=True)
nc = E.n_cand.cpu().numpy()
print("n_cand: mean(nonzero) %.0f max %d" % (nc[nc > 0].mean(), nc.max()), flush=True)
ref = (int(E.top_end.sum().item()), float(E.top_val.abs().sum().item()))
for div in (8192, 2048, 4096, 16384, 32768):
    ch = max(1 << 20, U.total // div)
    for it in range(2):
        eng.timers = {}
        E2 = eng.extend(S, 50, chunk=ch)
        torch.cuda.synchronize()
    tm = {k: np.round(v, 1).tolist() for k, v in eng.timer_ms().items() if k in ("paths", "path_weights")}
    ok = (int(E2.top_end.sum().item()), float(E2.top_val.abs().sum().item())) == ref
    print("div %d chunk %d units %d heavy %d rows %d -> %s same=%s" % (div, ch, E2.units.n_units, E2.units.n_heavy, E2.units.n_rows, tm, ok), flush=True)
