"""Soak of the user-sharded step on the GPU box (python profiles/tools/soak_users.py [minutes]): random small shapes, one user
share with RCCL as the backend (every collective on device buffers, all-to-all in small pieces) against the plain step --
item statistics, similarity matrix, extension, replacements and AlterEgo rows must be identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-map_amd"))
import numpy as np, torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from xmap.engine import device, synth, sharded
sharded.Comm.A2A_PIECE_BYTES = 1 << 14
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t_end = time.time() + 60 * minutes
rng = np.random.default_rng(int(time.time()))

def summary(res):
    S, E, G = res["S"], res["E"], res["G"]
    rp = S.row_ptr.cpu().numpy()
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp)); col = S.col.cpu().numpy()
    o = np.lexsort((col, rows))
    au, ai, ar = G.user.cpu().numpy().astype(np.int64), G.item.cpu().numpy(), G.rating.cpu().numpy()
    q = np.lexsort((ar, ai, au))
    return dict(n_eval=res["n_eval"], n_kept=res["n_kept"], n_paths=res["n_paths"], n_rows=res["n_rows"], n_profiles=res["n_profiles"],
                row_ptr=rp, col=col[o], sim=S.sim.cpu().numpy()[o], mutu=S.mutu.cpu().numpy()[o], nij=S.nij.cpu().numpy()[o],
                info=S.info.cpu().numpy(), n_cand=E.n_cand.cpu().numpy(), top_end=E.top_end.cpu().numpy(),
                top_val=E.top_val.cpu().numpy(), choice=res["choice"].cpu().numpy(), au=au[q], ai=ai[q], ar=ar[q])

n = 0
while time.time() < t_end:
    seed = int(rng.integers(1, 1 << 30))
    U = int(rng.integers(30, 3000)); Is = int(rng.integers(15, 700)); It = int(rng.integers(15, 700))
    k = int(rng.choice([2, 3, 5, 10])); method = str(rng.choice(["cosine", "adjust_cosine"]))
    ov = float(rng.uniform(0.1, 0.9)); mu = float(rng.uniform(0.3, 3.0)); sg = float(rng.uniform(0.5, 1.6))
    r = synth.make_two_domain(seed, U, Is, It, overlap=ov, mu=mu, sigma=sg)
    print("shape", n, dict(seed=seed, U=U, Is=Is, It=It, k=k, method=method), flush=True)
    mk = lambda: device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs(), "cuda:0"))
    a = summary(sharded.run_step(mk(), method, 50, k, True))
    b = summary(sharded.run_step_users(mk(), 0, method, 50, k, True, dist))
    for key in a:
        assert np.array_equal(a[key], b[key]), (key, dict(seed=seed, U=U, Is=Is, It=It, k=k, method=method, overlap=ov, mu=mu, sigma=sg))
    n += 1
print("soak done:", n, "random shapes, the user-sharded step identical to the plain one")
dist.destroy_process_group()
