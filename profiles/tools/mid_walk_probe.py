"""What the walk of k_mid_rows has to go through per row x' at BASELINE configs[1], k = 50: neighbours t with role T, source-list
entries scanned, 64-entry chunks, joint (t, s), records.
usage (GPU box, repo root): python profiles/tools/mid_walk_probe.py"""
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
I, k = r.n_items, 50
flags = eng.R.flags[:I].cpu().numpy()
kcnt = E.kcnt.cpu().numpy().reshape(-1, 2)[:I]
kcol = E.kcol.cpu().numpy().reshape(-1, 2, k)[:I]
src_ptr, src_idx, src_flag = E.src[0].cpu().numpy(), E.src[1].cpu().numpy(), E.src[3].cpu().numpy()
att_ptr = E.att[0].cpu().numpy()
nb_list = M.nb_list.cpu().numpy()[:M.n_nb]
srclen = src_ptr[1:] - src_ptr[:-1]
attlen = att_ptr[1:] - att_ptr[:-1]
jcount = np.add.reduceat(np.append(src_flag & 1, 0).astype(np.int64), src_ptr[:-1])      # joint entries per t
jcount[srclen == 0] = 0
jrec = np.add.reduceat(np.append((src_flag & 1) * attlen[src_idx], 0).astype(np.int64), src_ptr[:-1])
jrec[srclen == 0] = 0
tot = dict(t=0, scanned=0, chunks=0, joints=0, records=0)
per_row = []
for xp in nb_list:
    ts = kcol[xp, 0, :kcnt[xp, 0]]
    ts = ts[(flags[ts] & 2) != 0]
    sc, ch, jo, re = srclen[ts].sum(), ((srclen[ts] + 63) // 64).sum(), jcount[ts].sum(), jrec[ts].sum()
    per_row.append((len(ts), sc, ch, jo, re))
a = np.array(per_row, np.float64)
print("rows %d (records by this count %d, by the lists %d)" % (len(a), a[:, 4].sum(), M.n_records))
for i, n in enumerate(("neighbours t with role T", "source entries scanned", "64-entry chunks", "joint (t, s)", "records")):
    c = a[:, i]
    print("%-28s per row: mean %10.1f  median %10.1f  p90 %10.1f  max %10.0f   total %.4g" % (n, c.mean(), np.median(c), np.percentile(c, 90), c.max(), c.sum()))
print("records per joint %.1f, joints per chunk %.2f, records per chunk %.1f" % (a[:, 4].sum() / a[:, 3].sum(), a[:, 3].sum() / a[:, 2].sum(), a[:, 4].sum() / a[:, 2].sum()))
