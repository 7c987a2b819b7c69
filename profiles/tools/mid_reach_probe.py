"""Item-sharded stage B: which share of the middle-list rows (one per non-bridge x') does a rank need?  A rank enumerates the
paths of a contiguous range of start items (equal path counts); a start y' walks the rows of its heads -- itself (if it is a
non-bridge record) and every x' with y' in NN(x').  BASELINE configs[1], k = 50, world = 2 / 4 / 8.
usage (GPU box, repo root): python profiles/tools/mid_reach_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-map_amd")]
import numpy as np, torch
from xmap.engine import synth, device
from xmap.engine.sharded import balanced_ranges

r = synth.config_c2()
eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
S = eng.item_sim("adjust_cosine", 50)
E = eng.ext_tables(S, 50)
M = eng.mid_lists(E)
U = eng.path_units(E)
I = r.n_items
P = U.P[:I].cpu().numpy()
cls = E.cls[:I].cpu().numpy()
rnn_ptr, rnn_idx = E.rnn[0].cpu().numpy(), E.rnn[1].cpu().numpy()
nb_id = M.nb_id.cpu().numpy()
nrec = (M.dir_ptr[1:] - M.dir_ptr[:-1]).cpu().numpy()          # tiles per row (proxy of the row's build cost)
print("non-bridge rows %d, tiles %d, records %d" % (M.n_nb, M.n_tiles, M.n_records))
for world in (2, 4, 8):
    out = []
    for lo, hi in balanced_ranges(P, world):
        need = np.zeros(M.n_nb, bool)
        s = np.arange(lo, hi)
        own = s[(cls[lo:hi] == 2) & (nb_id[lo:hi] >= 0)]
        need[nb_id[own]] = True
        heads = rnn_idx[rnn_ptr[lo]:rnn_ptr[hi]]
        need[nb_id[heads][nb_id[heads] >= 0]] = True
        out.append((need.mean(), nrec[need].sum() / max(nrec.sum(), 1)))
    print("world %d: rows a rank needs %s ; share of the tiles in them %s" % (
        world, " ".join("%.0f%%" % (100 * a) for a, _ in out), " ".join("%.0f%%" % (100 * b) for _, b in out)))
