"""Soak on the GPU box (python profiles/tools/soak.py [minutes]): (1) the full-size pass repeated -- every run must give the
same bytes (stage A checksums, stage B candidate arrays); (2) random small shapes through every stage against the CPU oracle
(tests/test_gpu_parity._check_all_stages) until the time is up."""
import os, sys, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-map_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from xmap.engine import device, synth
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
t_end = time.time() + 60 * minutes
r = synth.config_c2()
eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs(), "cuda:0"))
ref = None
for it in range(int(os.environ.get("SOAK_FULL", "4"))):
    S = eng.item_sim("adjust_cosine", 50)
    E = eng.extend(S, 50)
    n_top, choice, mp = eng.select(E, True)
    G = eng.alterego(mp)
    bits = S.sim.view(torch.int64)
    key = (int(S.n_kept), int(S.col.long().sum()), int((bits & 0xffffffff).sum()), int((bits >> 32).sum()), int(S.mutu.long().sum()),
           int(E.n_paths), int(E.n_out), int(E.n_cand.long().sum()), int(E.top_end.long().sum()),
           int((E.top_val.view(torch.int64) & 0xffffffff).sum()), int((E.top_val.view(torch.int64) >> 32).sum()),
           int(choice.long().sum()), int(G.n_rows))
    if ref is None: ref = key
    assert key == ref, (it, key, ref)
    del S, E, G
    print("full-size pass", it, "identical", flush=True)
del eng
torch.cuda.empty_cache()
import test_gpu_parity as T
rng = np.random.default_rng(int(time.time()))
n = 0
while time.time() < t_end:
    seed = int(rng.integers(1, 1 << 30))
    U = int(rng.integers(50, 4000)); Is = int(rng.integers(20, 900)); It = int(rng.integers(20, 900))
    k = int(rng.choice([2, 3, 5, 10, 50, 100])); method = str(rng.choice(["cosine", "adjust_cosine"]))
    ov = round(float(rng.uniform(0.1, 0.9)), 6); mu = round(float(rng.uniform(0.3, 3.0)), 6); sg = round(float(rng.uniform(0.5, 1.6)), 6)
    if k >= 50 and (mu > 1.2 or Is + It > 600): k = 10          # (the one-thread oracle enumerates such a shape for minutes: 1e9 paths)
    if rng.random() < 0.3: os.environ["XMAP_REV_LONG"] = "64"
    else: os.environ.pop("XMAP_REV_LONG", None)
    rr = synth.make_two_domain(seed, U, Is, It, overlap=ov, mu=mu, sigma=sg)
    priv = bool(rng.integers(0, 2))
    print("shape", n, dict(seed=seed, U=U, Is=Is, It=It, k=k, method=method, overlap=round(ov, 6), mu=round(mu, 6), sigma=round(sg, 6),
                           private=priv, rev_long=os.environ.get("XMAP_REV_LONG")), flush=True)     # (a silent run is taken to be hung)
    faulthandler.dump_traceback_later(float(os.environ.get("SOAK_SHAPE_S", "200")), exit=True)      # a stuck shape says where
    try:
        T._check_all_stages(device, rr, method, k, private=priv)
        faulthandler.cancel_dump_traceback_later()
    except Exception:
        print("FAILED", dict(seed=seed, U=U, Is=Is, It=It, k=k, method=method, overlap=ov, mu=mu, sigma=sg, rev_long=os.environ.get("XMAP_REV_LONG")), flush=True)
        raise
    n += 1
    if n % 20 == 0: print(n, "random shapes ok", flush=True)
print("soak done:", n, "random shapes, all stages bit-identical to the oracle")
