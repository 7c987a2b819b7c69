"""k_paths4: how many column visits (= row updates / ends per column) a pass makes with the heads of a start merged in batches
of 64 (one wave: what the kernel does), of 256, or all at once.  BASELINE configs[1], k = 50.
usage (GPU box, repo root): python profiles/tools/batch_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-map_amd")]
import numpy as np, torch
from xmap.engine import synth, device

r = synth.config_c2()
eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
S = eng.item_sim("adjust_cosine", 50)
E = eng.ext_tables(S, 50)
M = eng.mid_lists(E)
I = r.n_items
dev = eng.dev
cls = E.cls[:I].long()
rnn_ptr, rnn_idx = E.rnn[0].long(), E.rnn[1].long()
nb_id = M.nb_id.long()
dir_ptr = M.dir_ptr.long()
d32 = M.dir.view(torch.int32).view(-1, 6)
dx, dne, dcnt = d32[:, 0].long(), d32[:, 1].long(), d32[:, 2].long()
selfh = (cls == 2).long()
nH = selfh + (rnn_ptr[1:I + 1] - rnn_ptr[:I])
ntile = dir_ptr[1:] - dir_ptr[:-1]                       # tiles of a non-bridge x'
# tiles walked per start = sum over its heads of ntile[head]
tiles_self = torch.where((selfh > 0) & (nb_id[:I] >= 0), ntile[nb_id.clamp(min=0)[:I]], torch.zeros_like(nH))
hx = nb_id[rnn_idx]
head_t = torch.where(hx >= 0, ntile[hx.clamp(min=0)], torch.zeros_like(hx))     # per rnn entry
cs = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), head_t.cumsum(0)])
tiles_rnn = cs[rnn_ptr[1:I + 1]] - cs[rnn_ptr[:I]]
tiles = tiles_self + tiles_rnn
print("starts with heads %d, heads %d, tile visits %.4g; starts with > 64 heads: %d (%.1f %% of the tile visits), > 256: %d (%.1f %%)" % (
    int((nH > 0).sum()), int(nH.sum()), float(tiles.sum()), int((nH > 64).sum()), 100.0 * float(tiles[nH > 64].sum()) / float(tiles.sum()),
    int((nH > 256).sum()), 100.0 * float(tiles[nH > 256].sum()) / float(tiles.sum())))
tot = {b: [0, 0, 0] for b in (64, 256, 1 << 30)}         # visits, updates (ends), records
order = torch.argsort(tiles, descending=True)
tiles_c = tiles[order].cpu().numpy()
LIM = 3e7
pos = 0
n_st = int((tiles > 0).sum())
while pos < n_st:
    end = pos; acc = 0
    while end < n_st and (acc + tiles_c[end] <= LIM or end == pos): acc += tiles_c[end]; end += 1
    ss = order[pos:end]
    # heads of these starts: (start index within chunk, h, xp)
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
    x, ne, cnt = dx[t], dne[t], dcnt[t]
    s_of, h_of = sid[hid], h[hid]
    for b in tot:
        key = (s_of << 44) | ((h_of // b) << 20) | x
        uk, inv = torch.unique(key, return_inverse=True)
        ne_u = torch.zeros(len(uk), dtype=torch.long, device=dev).scatter_(0, inv, ne)     # same x -> same ne
        tot[b][0] += len(uk); tot[b][1] += int(ne_u.sum()); tot[b][2] += int(cnt.sum())
    pos = end
for b, (v, u, rec) in tot.items():
    print("batch %s heads: column visits %.4g, row updates (ends) %.4g, merged records %.4g, records per visit %.1f" % (
        "all" if b > 1 << 20 else b, v, u, rec, rec / max(v, 1)))
