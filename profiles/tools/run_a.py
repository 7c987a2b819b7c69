"""Stage A of BASELINE configs[1] alone, N passes (for rocprofv3 --pmc runs: profiles/tools/sq_a.sh)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    eng.timers = {}
    S = eng.item_sim("adjust_cosine", 50); torch.cuda.synchronize()
print({k: np.round(v, 3).tolist() for k, v in eng.timer_ms().items()}, S.n_pairs if hasattr(S, "n_pairs") else "")
