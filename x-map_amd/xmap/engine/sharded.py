"""One pass of the hot path, optionally item-sharded over the ranks of one node.

Sharding: inputs (CSR + CSC) are replicated in every GPU's HBM; outputs are sharded by
item.  Stage A: each rank computes the complete similarity rows of its contiguous item range (no
cross-GPU partials), then the kept rows are exchanged with one all-gather per array (S4/S6 of SURVEY
2.3) so that every rank holds the full CSR.  Stage B: the knn tables are derived from the full CSR on
every rank (one HBM pass, cheaper than exchanging them), the path enumeration is sharded by start
item, and the fixed-size per-start candidate arrays are combined with an all-reduce (S7/S10).  Stage C
is a few HBM passes over nnz and is replicated.

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on the MI355X node; "gloo"
(host staging) is used by the CPU-side tests and for rehearsals with several ranks on one GPU.
"""
import numpy as np
import torch


# ----------------------------------------------------------------------------- host-side helpers
def balanced_ranges(weights, world):
    """Split [0, n) into `world` contiguous ranges of near-equal total weight.
    weights: 1-D array-like of non-negative numbers.  Returns [(lo, hi)] * world."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    c = np.concatenate([[0.0], np.cumsum(w)])
    total = c[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(min(max(np.searchsorted(c, target, side="left"), cuts[-1]), n)))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class Comm(object):
    """Thin wrapper over torch.distributed that stages through the host for gloo."""

    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.host = dist.get_backend() == "gloo"

    def all_reduce(self, t, op="sum"):
        ops = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX}
        if self.host and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=ops[op])
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=ops[op])
        return t

    def all_gather_var(self, t):
        """Concatenate 1-D tensors of different lengths from all ranks, in rank order."""
        dev = t.device
        n = torch.tensor([t.numel()], dtype=torch.int64, device="cpu" if self.host else dev)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        m = max(max(sizes), 1)
        src = t.cpu() if self.host else t
        pad = torch.zeros(m, dtype=t.dtype, device=src.device)
        pad[:t.numel()] = src
        parts = [torch.empty(m, dtype=t.dtype, device=src.device) for _ in range(self.world)]
        self.dist.all_gather(parts, pad)
        out = torch.cat([p[:s] for p, s in zip(parts, sizes)])
        return out.to(dev)


# ----------------------------------------------------------------------------- the step
def run_step(eng, method, cap, k, private, dist=None, rank=0, world=1, full=False):
    """stage A -> B -> C once.  Returns the counters the bench reports."""
    I = eng.R.n_items
    if dist is None or world == 1:
        with eng.timed("stage_a"):
            S = eng.item_sim(method, cap)
        with eng.timed("stage_b"):
            E = eng.extend(S, k, full=full)
        with eng.timed("stage_c"):
            n_top, choice, mp = eng.select(E, private)
            G = eng.alterego(mp)
            n_prof = eng.n_profiles(G)
        return dict(n_eval=S.n_eval, n_kept=S.n_kept, n_contrib=S.n_contrib, n_contrib_local=S.n_contrib,
                    n_kept_local=S.n_kept, n_paths=E.n_paths, n_rows=G.n_rows, n_profiles=n_prof,
                    S=S, E=E, G=G, choice=choice, map=mp)

    comm = Comm(dist)
    dev = eng.dev
    # ---- stage A: complete rows of this rank's item range, then all-gather the kept rows
    with eng.timed("stage_a"):
        stats = eng.stats()
        plan = eng.plan()
        wts = eng.item_weights(plan).cpu().numpy()
        lo, hi = balanced_ranges(wts, world)[rank]
        Sl = eng.item_sim(method, cap, item_range=(lo, hi), stats=stats, plan=plan)
        counts = (Sl.row_ptr[1:] - Sl.row_ptr[:-1]).contiguous()
        comm.all_reduce(counts)                       # rows outside [lo,hi) are empty locally
        row_ptr = torch.zeros(I + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=row_ptr[1:])
        col = comm.all_gather_var(Sl.col)
        sim = comm.all_gather_var(Sl.sim)
        mutu = comm.all_gather_var(Sl.mutu)
        nij = comm.all_gather_var(Sl.nij)
        tot = torch.tensor([Sl.n_eval, Sl.n_kept], dtype=torch.int64, device=dev)
        comm.all_reduce(tot)
        S = eng.sim_from_device(row_ptr, col, sim, mutu, nij, Sl.info)
        local_contrib = int(plan.W[lo:hi].sum().item())
    # ---- stage B: knn tables everywhere, paths sharded by start item
    with eng.timed("stage_b"):
        slo, shi = balanced_ranges(np.ones(I), world)[rank]
        E = eng.extend(S, k, full=False, start_range=(slo, shi))
        comm.all_reduce(E.n_cand)
        comm.all_reduce(E.top_end, "max")             # -1 outside the local range
        comm.all_reduce(E.top_val)                    # 0.0 outside the local range
        pt = torch.tensor([E.n_paths], dtype=torch.int64, device=dev)
        comm.all_reduce(pt)
    # ---- stage C: replicated
    with eng.timed("stage_c"):
        n_top, choice, mp = eng.select(E, private)
        G = eng.alterego(mp)
        n_prof = eng.n_profiles(G)
    return dict(n_eval=int(tot[0].item()), n_kept=int(tot[1].item()), n_contrib=plan.contrib,
                n_contrib_local=local_contrib, n_kept_local=Sl.n_kept, n_paths=int(pt.item()),
                n_rows=G.n_rows, n_profiles=n_prof, S=S, E=E, G=G, choice=choice, map=mp)
