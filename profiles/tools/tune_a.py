"""Stage A at BASELINE configs[1] for a few (slot_target, ch_min) pairs: python profiles/tools/tune_a.py  (GPU box, repo root)"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
for st, ch in ((640, 1024), (512, 1024), (704, 1024), (768, 1024), (832, 1024), (640, 512), (640, 2048), (768, 2048), (768, 512)):
    for it in range(4):
        if it == 1:
            eng.timers = {}
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        S = eng.item_sim_tri("adjust_cosine", 50, slot_target=st, ch_min=ch)
    torch.cuda.synchronize()
    tm = {k: float(np.mean(v)) for k, v in eng.timer_ms().items()}
    L = S.layout
    print("slot_target %d ch_min %d: light units %d heavy units %d | layout3 %.3f plan %.3f pair_tri %.3f scatter %.3f sum %.3f" % (
        st, ch, L.n_light, L.n_heavy_units, tm.get("layout3", 0), tm.get("tri_plan", 0), tm.get("pair_tri", 0), tm.get("scatter", 0),
        sum(tm.get(k, 0) for k in ("layout3", "tri_plan", "pair_tri", "scatter"))), flush=True)
