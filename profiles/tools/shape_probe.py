"""Stage-B size of a workload without running the enumeration: bridge / non-bridge items, exact path count, middle-list
records and tiles (count pass only).  usage: python profiles/tools/shape_probe.py {c2|s1} [k] [overlap]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "x-map_amd")]
import numpy as np
import torch
from xmap.engine import synth, device
from xmap.engine.device import lib, check, vp, i32, i64, _stream

name = sys.argv[1] if len(sys.argv) > 1 else "s1"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
if name == "s1" and len(sys.argv) > 3:
    ov = float(sys.argv[3])
    r = synth.make_two_domain(3, 1164670, 128402, 403234, overlap=ov, src_share=0.39)
else:
    r = getattr(synth, "config_" + name)()
eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
S = eng.item_sim("adjust_cosine", 50)
E = eng.ext_tables(S, k)
I = r.n_items
cls = E.cls.cpu().numpy()[:I]
U = eng.path_units(E)
print("workload %s k=%d: users %d items %d nnz %d kept %d" % (name, k, r.n_users, I, r.nnz, S.n_kept))
print("bridge %d  non-bridge records %d  dropped %d" % ((cls == 1).sum(), (cls == 2).sum(), (cls == 0).sum()))
print("paths %.4g  units %d heavy starts %d rows %d" % (U.total, U.n_units, U.n_heavy, U.n_rows))
att = np.diff(E.att[0].cpu().numpy())
print("attach lists: max %d mean(nonzero) %.1f" % (att.max(), att[att > 0].mean() if (att > 0).any() else 0))
st = _stream(eng.dev)
nb_list = eng._empty(max(I, 1), torch.int32); nb_id = eng._empty(max(I, 1), torch.int32)
nn = C.c_int64(0)
check(lib.xmap_nb_index(st, i32(I), vp(E.cls), vp(nb_list), vp(nb_id), C.byref(nn)))
n_nb = int(nn.value)
ng = eng._empty(n_nb, torch.int32); nrec = eng._empty(n_nb, torch.int64)
R = eng.R
common = (i32(I), E.k, vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(E.kval), vp(R.flags), vp(E.att[0]), vp(E.att[1]), vp(E.att[2]),
          vp(E.src[0]), vp(E.src[1]), vp(E.src[2]), vp(E.src[3]), i32(n_nb), vp(nb_list[:n_nb]), vp(nb_id))
check(lib.xmap_mid_rows_count(st, *common, vp(ng), vp(nrec)))
print("middle lists: records %.4g (%.1f GB) tiles %.4g" % (float(nrec.sum().item()), float(nrec.sum().item()) * 64 / 1e9, float(ng.sum().item())))
