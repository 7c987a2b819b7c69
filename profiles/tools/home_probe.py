"""k_paths4: which share of the row updates of a pass goes to an end's HOME column -- the column that is visited first among
the columns listing the end, so that its update needs no load (the entry is known to be zero) -- under two column orders:
item order (home = lowest column, what xmap_end_order does today) and visit-count order (home = the end's most visited
column).  Also the distribution of the number of columns per end, update-weighted.  BASELINE configs[1], k = 50.
usage (GPU box, repo root): python profiles/tools/home_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-map_amd")]
import numpy as np, torch
from xmap.engine import synth, device

r = synth.config_c2()
k = int(os.environ.get("K", "50"))
eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
S = eng.item_sim("adjust_cosine", 50)
E = eng.ext_tables(S, k)
M = eng.mid_lists(E)
I = r.n_items
dev = eng.dev
cls = E.cls[:I].long()
rnn_ptr, rnn_idx = E.rnn[0].long(), E.rnn[1].long()
nb_id = M.nb_id.long()
n_nb = M.n_nb
dir_ptr = M.dir_ptr.long()
d32 = M.dir.view(torch.int32).view(-1, 6)
dx, dne, dcnt, dcol = d32[:, 0].long(), d32[:, 1].long(), d32[:, 2].long(), d32[:, 3].long()
selfh = (cls == 2).long()
nH = selfh + (rnn_ptr[1:I + 1] - rnn_ptr[:I])
ntile = dir_ptr[1:] - dir_ptr[:-1]
tiles_self = torch.where((selfh > 0) & (nb_id[:I] >= 0), ntile[nb_id.clamp(min=0)[:I]], torch.zeros_like(nH))
hx = nb_id[rnn_idx]
head_t = torch.where(hx >= 0, ntile[hx.clamp(min=0)], torch.zeros_like(hx))
cs = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), head_t.cumsum(0)])
tiles = tiles_self + cs[rnn_ptr[1:I + 1]] - cs[rnn_ptr[:I]]
V = torch.zeros(n_nb, dtype=torch.long, device=dev)            # visits (start, column) per column id
R = torch.zeros(n_nb, dtype=torch.long, device=dev)            # merged records per column id
order = torch.argsort(tiles, descending=True)
tiles_c = tiles[order].cpu().numpy()
LIM = 3e7
pos = 0
n_st = int((tiles > 0).sum())
while pos < n_st:
    end = pos; acc = 0
    while end < n_st and (acc + tiles_c[end] <= LIM or end == pos): acc += tiles_c[end]; end += 1
    ss = order[pos:end]
    nh = nH[ss]
    sid = torch.repeat_interleave(torch.arange(len(ss), device=dev), nh)
    first = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), nh.cumsum(0)[:-1]])
    h = torch.arange(int(nh.sum()), device=dev) - first[sid]
    sf = selfh[ss][sid]
    is_self = (h < sf)
    rp = rnn_ptr[ss][sid] + (h - sf)
    xp = torch.where(is_self, ss[sid], rnn_idx[rp.clamp(min=0, max=max(rnn_idx.numel() - 1, 0))])
    xid = nb_id[xp]
    nt = torch.where(xid >= 0, ntile[xid.clamp(min=0)], torch.zeros_like(xid))
    xid = xid.clamp(min=0)
    hid = torch.repeat_interleave(torch.arange(len(xid), device=dev), nt)
    f2 = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), nt.cumsum(0)[:-1]])
    t = dir_ptr[xid][hid] + (torch.arange(int(nt.sum()), device=dev) - f2[hid])
    col = dcol[t]                                               # column id (index of x in nb_list)
    key = (sid[hid] << 20) | col
    R += torch.bincount(col, weights=dcnt[t].double(), minlength=n_nb).long()
    uk = torch.unique(key)
    V += torch.bincount(uk & ((1 << 20) - 1), minlength=n_nb)
    pos = end
# ends of every column: x itself + NB_NN(x)
nb_list = M.nb_list.long()
kcnt = E.kcnt.view(-1, 2)[:, 1].long()[nb_list]                 # |NN(x)|
kcol = E.kcol.view(-1, 2, k)[:, 1, :].long()[nb_list]           # [n_nb][k]
colid = torch.arange(n_nb, device=dev)
ar = torch.arange(k, device=dev)[None, :]
valid = ar < kcnt[:, None]
e_all = torch.cat([nb_list, kcol[valid]])                       # end item of every (column, end) slot
c_all = torch.cat([colid, colid[:, None].expand(-1, k)[valid]])
n_slots = int(e_all.numel())
upd = V[c_all]                                                  # updates this slot receives per pass
tot = int(upd.sum())
print("columns %d, visits %.4g, merged records %.4g, slots %d, ends %d, row updates %.4g (%.1f ends per visit)" % (
    n_nb, float(V.sum()), float(R.sum()), n_slots, int(torch.unique(e_all).numel()), tot, tot / float(V.sum())))
# home = lowest column (item order == column id order)
lo = torch.full((I,), 1 << 40, dtype=torch.long, device=dev).scatter_reduce(0, e_all, c_all, "amin")
home_lo = int(upd[c_all == lo[e_all]].sum())
mx = torch.zeros(I, dtype=torch.long, device=dev).scatter_reduce(0, e_all, upd, "amax")
home_pop = int(mx.sum())
print("home = lowest column:        %.4g updates need no load (%.1f %%)" % (home_lo, 100.0 * home_lo / tot))
print("home = most visited column:  %.4g updates need no load (%.1f %%)" % (home_pop, 100.0 * home_pop / tot))
m = torch.bincount(e_all, minlength=I)                          # columns per end
for lim in (1, 2, 4, 8, 16, 64, 256, 1024, 1 << 30):
    sel = m[e_all] <= lim
    print("ends listed by <= %-10d columns: %7d ends, %5.1f %% of the slots, %5.1f %% of the updates" % (
        lim, int((m[(m > 0)] <= lim).sum()), 100.0 * float(sel.sum()) / n_slots, 100.0 * float(upd[sel].sum()) / tot))
# records x ends per visit, by ends-per-column class (the lanes = ends x slices question)
ne = kcnt + 1
for a, b in ((1, 8), (9, 16), (17, 21), (22, 32), (33, 50), (51, 51)):
    sel = (ne >= a) & (ne <= b)
    v = float(V[sel].sum())
    print("columns with %2d..%2d ends: %5d columns, %5.1f %% of the visits, %5.1f %% of the updates, %.1f records per visit" % (
        a, b, int(sel.sum()), 100.0 * v / float(V.sum()), 100.0 * float((V * ne)[sel].sum()) / tot, float(R[sel].sum()) / max(v, 1)))
