"""One pass of the hot path, optionally item-sharded over the ranks of one node.

Sharding: the ratings (CSR by user) are replicated in every GPU's HBM; the work is sharded by
item.  Stage A: every rank lays the ratings out (one transposition: 1.6 ms, replicated), the per-item statistics are
computed for a share of the items and all-gathered (32 B x I: the all-gather of item norms) before the mutuality flags are
set from them; the work units (item, partition) of the pair kernel are split into contiguous ranges of
equal rater-steps and the heavy rows dealt round-robin; each rank appends the kept pairs of its units to a half COO (every
unordered pair is owned by exactly one unit, so there are no cross-GPU partials) and mirrors them into a CSR of its own:
that is the rank's partition of item2item_simRDD (both directions of its pairs), and where stage A ends.  Stage B starts
with the exchange its input needs (the reference broadcasts the knn tables, utils/assist.py:88-101): the per-item row
counts are all-reduced, the kept pairs travel as packed 24-byte records (S4/S6 of SURVEY 2.3) and every rank mirrors only
ITS SHARE of the rows into the CSR (contiguous row shares of equal entries); it classifies the top-k lists of those rows
and the knn tables are all-gathered (S7), the three reverse adjacencies are built in the same row shares and all-gathered
(S8/S9), the middle lists are derived on every rank, the path enumeration is sharded by start item (ranges of equal path
counts), and the fixed-size per-start candidate arrays are combined with an all-reduce (S7/S10).  Stage C is a few HBM
passes over nnz and is replicated.

run_step_users is the other split (inputs sharded by USER, the partial similarities of a pair exchanged and added up at
the pair's owner): see its docstring.

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on the MI355X node; "gloo"
(host staging) is used by the CPU-side tests and for rehearsals with several ranks on one GPU.
"""
import os

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


def unit_cuts(c, w_heavy, world):
    """Cut points of the light work units over `world` ranks.  c: inclusive prefix sum (float64 tensor) of the units' rater
    steps; w_heavy: rater visits of the heavy rows (0-d tensor or None), which rank 0 computes as well and which therefore
    count against its share.  Returns world + 1 non-decreasing unit indices from 0 to len(c)."""
    n = int(c.numel())
    if n == 0:
        return np.zeros(world + 1, np.int64)
    wh = w_heavy if w_heavy is not None else torch.zeros((), dtype=torch.float64, device=c.device)
    first = torch.minimum(((c[-1] + wh) / world - wh).clamp(min=0.0), c[-1])
    tgt = first + (c[-1] - first) / max(world - 1, 1) * torch.arange(0, world - 1, dtype=torch.float64, device=c.device)
    cuts = [0] + torch.searchsorted(c, tgt).clamp(max=n).tolist() + [n]
    return np.maximum.accumulate(np.asarray(cuts, np.int64))


class Comm(object):
    """Thin wrapper over torch.distributed that stages through the host for gloo."""
    A2A_PIECE_BYTES = 1 << 28

    def __init__(self, dist, group=None):
        """group: a torch.distributed process group (None = all ranks); rank / world are relative to it"""
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.host = dist.get_backend(group) == "gloo"
        # gloo has an all-to-all for host tensors too: XMAP_A2A_NATIVE=1 sends all_to_all_rows down the RCCL branch on it (the
        # piece / offset arithmetic of that branch under several ranks, which a one-GPU box cannot run on RCCL itself)
        self.a2a_native = (not self.host) or os.environ.get("XMAP_A2A_NATIVE") == "1"

    def all_reduce(self, t, op="sum"):
        ops = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}
        if self.host and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=ops[op], group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t

    def agree(self, err, what):
        """Collective error check: every rank calls it with its own exception (or None); if ANY rank failed, EVERY rank
        raises, so that nobody is left waiting in the next collective."""
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int64)
        if not self.host:
            flag = flag.cuda()
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            if err is not None:
                raise err
            raise RuntimeError("%s failed on another rank (this is rank %d of %d)" % (what, self.rank, self.world))

    def all_gather_var(self, t):
        """Concatenate 1-D tensors of different lengths from all ranks, in rank order."""
        dev = t.device
        n = torch.tensor([t.numel()], dtype=torch.int64, device="cpu" if self.host else dev)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        m = max(max(sizes), 1)
        src = t.cpu() if self.host else t
        pad = torch.zeros(m, dtype=t.dtype, device=src.device)
        pad[:t.numel()] = src
        parts = [torch.empty(m, dtype=t.dtype, device=src.device) for _ in range(self.world)]
        self.dist.all_gather(parts, pad, group=self.group)
        out = torch.cat([p[:s] for p, s in zip(parts, sizes)])
        return out.to(dev)


    def all_gather_multi(self, ts):
        """Several 1-D tensors per rank (any dtypes, any lengths) -> for each of them the concatenation over the ranks, in
        rank order -- TWO collectives whatever the number of tensors (the table of byte counts, then one padded byte buffer
        per rank), where all_gather_var takes two per tensor."""
        dev = ts[0].device
        parts = []
        for t in ts:
            b = t.contiguous().view(-1).view(torch.uint8)
            if b.numel() % 8:                                   # (every part starts 8-byte aligned inside the buffer)
                b = torch.cat([b, torch.zeros(8 - b.numel() % 8, dtype=torch.uint8, device=b.device)])
            parts.append(b)
        table = self.all_gather_fixed(torch.tensor([t.numel() * t.element_size() for t in ts], dtype=torch.int64, device=dev)).cpu()
        padded = (table + 7) // 8 * 8                           # [world, len(ts)] bytes incl. alignment
        mine = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)
        m = max(int(padded.sum(dim=1).max()), 8)
        src = mine.cpu() if self.host else mine
        pad = torch.zeros(m, dtype=torch.uint8, device=src.device)
        pad[:src.numel()] = src
        bufs = [torch.empty(m, dtype=torch.uint8, device=src.device) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad, group=self.group)
        out = []
        for j, t in enumerate(ts):
            pieces = []
            for r in range(self.world):
                at = int(padded[r, :j].sum())
                pieces.append(bufs[r][at:at + int(table[r, j])])
            cat = torch.cat(pieces)
            if cat.numel() % 8 == 0 and t.element_size() > 1:
                cat = cat.clone()                               # (fresh storage: offset 0, viewable as any dtype)
            out.append(cat.view(t.dtype).to(dev))
        return out

    def all_gather_fixed(self, t):
        """[world, *t.shape]: the same-shaped tensor of every rank, in rank order"""
        src = t.contiguous().cpu() if self.host else t.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(parts, src, group=self.group)
        return torch.stack(parts).to(t.device)

    def all_to_all_rows(self, rows, cuts):
        """rows [n, w] int64 cut into `world` consecutive row ranges by cuts (world + 1 indices): range r goes to rank r.
        Returns the ranges received, concatenated in rank order.  RCCL: one all_to_all_single with the split sizes; gloo
        (no all-to-all): every rank all-gathers everything and keeps its ranges."""
        w = int(rows.shape[1])
        send = [int(cuts[r + 1] - cuts[r]) for r in range(self.world)]
        table = torch.tensor(send, dtype=torch.int64)
        if not self.host:
            table = table.to(rows.device)
        tabs = [torch.zeros_like(table) for _ in range(self.world)]
        self.dist.all_gather(tabs, table, group=self.group)
        tabs = torch.stack(tabs).cpu()                     # tabs[s][r]: rows rank s sends to rank r
        if not self.a2a_native:
            allrows = self.all_gather_var(rows.reshape(-1)).view(-1, w)
            start = torch.cumsum(tabs.sum(dim=1), 0) - tabs.sum(dim=1)      # first row of rank s in allrows
            inner = torch.cumsum(tabs, 1) - tabs                            # offset of the range for r inside s's rows
            parts = [allrows[int(start[s] + inner[s][self.rank]):int(start[s] + inner[s][self.rank] + tabs[s][self.rank])]
                     for s in range(self.world)]
            return torch.cat(parts) if parts else allrows[:0]
        recv = [int(tabs[s][self.rank]) for s in range(self.world)]
        dev0 = rows.device
        if self.host:
            rows = rows.cpu()
        out = torch.empty((sum(recv), w), dtype=rows.dtype, device=rows.device)
        # In pieces of at most 256 MiB per peer: RCCL 2.26's all_to_all_single returned wrong data for pieces above 1 GiB
        # (measured with one rank: 1.28 GB of int64 rows came back wrong from byte 640 M on; all_gather / all_reduce of
        # 2 GiB were fine).  Piece c of sender s lands behind its pieces 0 .. c-1, so the result is sender-major as above.
        piece = max(1, self.A2A_PIECE_BYTES // (rows.element_size() * w))
        rounds = (int(tabs.max()) + piece - 1) // piece
        first = [sum(recv[:s_]) for s_ in range(self.world)]
        for c in range(rounds):
            send_c = [min(max(n_ - c * piece, 0), piece) for n_ in send]
            recv_c = [min(max(n_ - c * piece, 0), piece) for n_ in recv]
            buf = torch.cat([rows[int(cuts[r]) + c * piece:int(cuts[r]) + c * piece + send_c[r]] for r in range(self.world)])
            tmp = torch.empty((sum(recv_c), w), dtype=rows.dtype, device=rows.device)
            self.dist.all_to_all_single(tmp, buf.contiguous(), output_split_sizes=recv_c, input_split_sizes=send_c, group=self.group)
            at = 0
            for s_ in range(self.world):
                out[first[s_] + c * piece:first[s_] + c * piece + recv_c[s_]] = tmp[at:at + recv_c[s_]]
                at += recv_c[s_]
        return out.to(dev0)


# ----------------------------------------------------------------------------- the step
def run_step(eng, method, cap, k, private, dist=None, rank=0, world=1, full=False, group=None):
    """stage A -> B -> C once.  Returns the counters the bench reports.  group: the process group that shares this
    problem (None: all ranks); rank / world are then taken from it."""
    I = eng.R.n_items
    if dist is not None and group is not None:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    # (XMAP_FORCE_DIST=1: the sharded path also with ONE rank -- every collective of the step runs on the backend's device
    #  buffers, which is how a one-GPU box rehearses the RCCL code path)
    if dist is None or (world == 1 and os.environ.get("XMAP_FORCE_DIST") != "1"):
        with eng.timed("stage_a"):
            S = eng.item_sim(method, cap)
        with eng.timed("stage_b"):
            E = eng.extend(S, k, full=full)
        with eng.timed("stage_c"):
            n_top, choice, mp = eng.select(E, private)
            G = eng.alterego(mp)
            n_prof = eng.n_profiles(G)
        L = S.layout
        return dict(n_eval=S.n_eval, n_kept=S.n_kept, n_contrib=S.n_contrib,
                    n_contrib_light=None, n_kept_local=S.n_kept,       # (bench: 2 * (half_contrib - eng.heavy_half(L)), on demand)
                    n_paths=E.n_paths, n_out=E.n_out, n_rows=G.n_rows, n_profiles=n_prof,
                    knn_entries=int(E.kcnt.sum().item()), S=S, E=E, G=G, choice=choice, map=mp)

    comm = Comm(dist, group)
    dev = eng.dev
    # ---- stage A: every rank lays out the (replicated) ratings -- one transposition, rater records through the tile sort --,
    # the item statistics are computed for a share of the items and all-gathered; every rank computes the pairs of its share
    # of the work units (+ its share of the heavy rows) into a half COO and mirrors them into a CSR of its own
    with eng.timed("stage_a"):
        ilo, ihi = I * rank // world, I * (rank + 1) // world
        nI = max(I, 1)

        def gather(info, norms):
            # ONE exchange of the 32-byte item records and the two dense norm columns (S3 of SURVEY 2.3: the reference
            # collects and broadcasts item_info, utils/assist.py:71-73) -- the all-gather of per-item norms
            with eng.timed("stats_gather"):
                gi, g0, g1 = comm.all_gather_multi([info[ilo:ihi].reshape(-1), norms[ilo:ihi], norms[nI + ilo:nI + ihi]])
                info[:I] = gi.view(I, 4)
                norms[:I] = g0
                norms[nI:nI + I] = g1
        # (no collective between a possible raise and the agree() that follows it)
        err, stats, L = None, None, None
        try:
            stats, L = eng.layout3(item_range=(ilo, ihi))
        except Exception as e:
            err = e
        comm.agree(err, "stage A (layout)")
        gather(stats[2], eng.norms)
        err = None
        try:
            L.finish()
        except Exception as e:
            err = e
        comm.agree(err, "stage A (flags + plan)")
        while True:
            # contiguous unit ranges of equal rater-steps: prefix sum and cut points on the device (the units are listed
            # by table class, every rank gets a slice of every class boundary it spans); the heavy rows are dealt
            # round-robin by item index, so every rank carries about 1 / world of their rater visits
            lo, hi = 0, 0
            if L.n_light:
                n_i = (eng.R.item_ptr[1:] - eng.R.item_ptr[:-1])
                c = torch.cumsum(n_i[L.uq_item[:L.n_light].long()].double(), 0)
                cuts = unit_cuts(c, None, world)
                lo, hi = int(cuts[rank]), int(cuts[rank + 1])
            err, ovf = None, 0
            try:
                coo, own, n, n_unordered, mir, shards, ovf = eng.tri_pairs(method, cap, stats, L, unit_range=(lo, hi), retry=False,
                                                                           split=True, heavy_deal=(rank, world))
            except Exception as e:          # e.g. half-COO overflow on this rank only
                err = e
            comm.agree(err, "stage A (pair kernels)")
            flag = torch.tensor([ovf], dtype=torch.int64, device=dev)
            comm.all_reduce(flag, "max")
            if int(flag.item()) == 0:
                break
            L.replan(L, L.slot_target // 2)      # every rank re-plans identically
        # the rank's partition of item2item_simRDD: both directions of ITS kept pairs, CSR by first item (the same
        # mirror step a single GPU does for all pairs)
        S_part = eng.tri_mirror(coo, own, mir, stats[2], n, shards)
        tot = torch.tensor([n_unordered, n], dtype=torch.int64, device=dev)
        comm.all_reduce(tot)
        it = L.uq_item[lo:hi].long()           # a partitioned item's contributions are split over its Q units
        light_local = 2 * int((L.Wp[it].double() / L.Q[it].clamp(min=1).double()).sum().item()) if hi > lo else 0
    S, E, pt = _stage_b(eng, comm, coo, (own, mir), stats[2], L, k, rank, world, n)
    # ---- stage C: the replacement map on every rank (one pass over the candidate arrays), the AlterEgo rows of a share of
    # the USERS (contiguous shares of equal ratings), gathered kind by kind -- pass-through rows, then merged rows -- so that
    # the result is row for row the one-rank output
    with eng.timed("stage_c"):
        n_top, choice, mp = eng.select(E, private)
        nU = eng.R.n_users
        ucut = torch.searchsorted(eng.R.user_ptr, (eng.R.user_ptr[nU] * torch.arange(1, world, device=dev)) // world).tolist()
        ucut = [0] + [min(max(int(x), 0), nU) for x in ucut] + [nU]
        ucut = np.maximum.accumulate(np.asarray(ucut))
        Gl = eng.alterego(mp, users=(int(ucut[rank]), int(ucut[rank + 1])))
        with eng.timed("rows_gather"):
            nt = Gl.n_target_rows
            rec = torch.stack([Gl.user.long() | (Gl.item.long() << 32), Gl.rating.view(torch.int64), Gl.time], dim=1)
            parts = [g.view(-1, 3) for g in comm.all_gather_multi([rec[:nt].reshape(-1), rec[nt:].reshape(-1)])]
            allr = torch.cat(parts)
            G = _Rows()
            G.user, G.item = (allr[:, 0] & 0xffffffff).int(), (allr[:, 0] >> 32).int()
            G.rating, G.time = allr[:, 1].contiguous().view(torch.float64), allr[:, 2].contiguous()
            G.n_rows, G.n_target_rows = int(allr.shape[0]), int(parts[0].shape[0])
            prof = torch.tensor([eng.n_profiles(Gl)], dtype=torch.int64, device=dev)
            comm.all_reduce(prof)
        n_prof = int(prof.item())
    return dict(n_eval=2 * int(tot[0].item()), n_kept=2 * int(tot[1].item()), n_contrib=2 * L.half_contrib,
                n_contrib_light=light_local, n_kept_local=2 * n, n_paths=int(pt[0].item()), n_out=int(pt[1].item()),
                n_rows=G.n_rows, n_profiles=n_prof, knn_entries=int(E.kcnt.sum().item()),
                S=S, S_part=S_part, E=E, G=G, choice=choice, map=mp)


class _Rows(object):
    """AlterEgo rows gathered from the ranks (same attributes as the engine's GenResult)"""
    pass


def run_step_users(eng, user_lo, method, cap, k, private, dist, group=None, slot_target=768):
    """stage A -> B -> C once over USER-sharded input (BASELINE configs[2]: "reduce-scatter of cross-shard
    partial similarities"): eng.R holds the complete profiles of this rank's users -- users [user_lo, user_lo + n_users) of
    the whole data set, items indexed globally -- instead of a replica of all ratings.

    Stage A: every rank sums its users' contributions.  Item statistics: the shares [I][5] are all-gathered (the all-gather
    of per-item norms) and added up in rank order, the adjusted norm exactly.  Pairs: the pair kernel runs over the rank's
    users in "raw" mode and emits, per pair two of its users co-rated, the partial dot product (an exact (value, error)
    pair), n_ij and the mutuality; the 32-byte records are sorted by pair key and sent to the rank that owns the pair's
    lower item (all-to-all of sparse partials = the reduce-scatter: every rank receives only the shares of its own pairs,
    added up on arrival -- the dot product exactly, so the result does not depend on the number of ranks); the owner
    finishes the pair (cosine, significance weighting, zero filter).  From there the step is the item-sharded one: the kept
    pairs are exchanged for stage B, the path enumeration is sharded by start item.  Stage C runs over the rank's own users
    and the AlterEgo rows are concatenated in rank (= user) order."""
    comm = Comm(dist, group)
    rank, world = comm.rank, comm.world
    I = eng.R.n_items
    dev = eng.dev
    with eng.timed("stage_a"):
        u_avg, u_norm, partial = eng.stats_partial()
        with eng.timed("stats_gather"):
            parts = comm.all_gather_fixed(partial)
        info = eng.stats_merge(parts)
        # the layout of the rank's pair kernel orders items by their LOCAL rater counts (its partner bounds are local);
        # the item averages behind the mutuality flags are the global ones
        info_loc = info.clone()
        if I:
            info_loc[:I, 3] = (eng.R.item_ptr[1:] - eng.R.item_ptr[:-1]).double()
        stats_loc = (u_avg, u_norm, info_loc, None, None)
        err, out = None, None
        try:
            L = eng.tri_layout(stats_loc, slot_target, ch_min=max(64, eng.R.n_users + 2))       # no heavy set
            out = eng.tri_pairs(method, cap, stats_loc, L, do_heavy=False, raw=True)
        except Exception as e:
            err = e
        comm.agree(err, "stage A (pair kernels)")
        coo_raw, _, n_raw, _ = out
        rec = eng.partial_records(coo_raw, n_raw, world)         # grouped by owner of the lower item (key = lower << 32 | higher)
        del coo_raw, out
        with eng.timed("exchange_partials"):
            thr = torch.tensor([(I * r // world) << 32 for r in range(1, world)], dtype=torch.int64, device=dev)
            # (the keys are ordered group-wise only: "key < first key of rank r" is still monotone along the array)
            inner = torch.searchsorted(rec[:, 0].contiguous(), thr).tolist() if n_raw else [0] * (world - 1)
            cuts = [0] + [int(x) for x in inner] + [n_raw]
            got = comm.all_to_all_rows(rec, cuts)
        err = None
        try:
            got = eng.sort_records(got)                          # stable: the shares of a pair stay in rank order
            coo, rowcnt, n, n_unordered = eng.merge_records(got, method, cap)
            S_part = eng.tri_scatter(coo, rowcnt, info, n, L)
        except Exception as e:
            err = e
        comm.agree(err, "stage A (merge of the partial similarities)")
        tot = torch.tensor([n_unordered, n, L.half_contrib], dtype=torch.int64, device=dev)
        comm.all_reduce(tot)
    S, E, pt = _stage_b(eng, comm, coo, rowcnt, info, L, k, rank, world, n)
    S.u_avg = None                         # user info stays with the rank that holds the users
    with eng.timed("stage_c"):
        n_top, choice, mp = eng.select(E, private)
        Gl = eng.alterego(mp)
        G = _Rows()
        G.user, G.item, G.rating, G.time = comm.all_gather_multi([Gl.user.long() + int(user_lo), Gl.item, Gl.rating, Gl.time])
        G.n_rows = int(G.user.numel())
        prof = torch.tensor([eng.n_profiles(Gl)], dtype=torch.int64, device=dev)
        comm.all_reduce(prof)
    return dict(n_eval=2 * int(tot[0].item()), n_kept=2 * int(tot[1].item()), n_contrib=2 * int(tot[2].item()),
                n_contrib_light=2 * L.half_contrib, n_kept_local=2 * n, n_paths=int(pt[0].item()), n_out=int(pt[1].item()),
                n_rows=G.n_rows, n_profiles=int(prof.item()), knn_entries=int(E.kcnt.sum().item()),
                S=S, S_part=S_part, E=E, G=G, G_local=Gl, choice=choice, map=mp, info=info)


def _stage_b(eng, comm, coo, rowcnt, info, L, k, rank, world, n_local):
    """stage B of a sharded step: coo / rowcnt = the n_local kept pairs this rank holds (any orientation, every unordered pair
    on exactly one rank).  Returns (S: this rank's row share of the similarity matrix (the full one on the round-2 path), E: the
    extension with the candidate arrays of ALL starts, [paths, candidates] over all ranks)."""
    dev = eng.dev
    # ---- stage B: knn tables + reverse lists everywhere (one HBM pass), paths sharded by start item with
    # ranges balanced by the exact per-start path counts
    with eng.timed("stage_b"):
        # The extension needs the whole similarity matrix on every rank (the reference broadcasts its knn tables,
        # utils/assist.py:88-101): the ranks' COO parts are exchanged here -- per-item row counts all-reduced, the
        # compacted parts (one index list for the five columns) sent as ONE variable-length all-gather of 24-byte
        # records (i | j << 32, sim bits, mutu | n_ij << 32) -- and every rank mirrors the full COO into the CSR.
        split = isinstance(rowcnt, tuple)          # (own, mirrored) counts of the round-3 mirror, or one combined array
        I = eng.R.n_items
        share = None
        with eng.timed("exchange"):
            rec = eng.pack_pairs(coo, n_local)                                  # 24-byte records (i | j << 32, sim bits, mutu | n_ij << 32)
            if split and I > 0:
                # Round 4: every kept pair goes to the (at most two) ranks that own its rows, not to everybody.  Row counts
                # first -- own entries (the pair kernels counted them) and mirrored entries (one count over the rank's own
                # records) all-reduced: 2 x 4 B x I --, contiguous row shares of equal entries cut from them on every rank
                # alike, then ONE all-to-all of the records grouped by destination (a record whose two rows have the same
                # owner travels once).  A rank receives 2 D' / N records instead of the D' of the all-gather it replaces
                # (0.64 GB per rank at configs[1]); S4 / S6 of SURVEY 2.3, the shuffle behind get_item_sim
                # (core/baselinerSim.py:218-233).
                ri, rj = rec[:, 0] & 0xffffffff, rec[:, 0] >> 32
                mirc = rowcnt[1]
                mirc.zero_()
                if n_local:
                    mirc.add_(torch.bincount(rj, minlength=mirc.numel())[:mirc.numel()].to(mirc.dtype))
                comm.all_reduce(rowcnt[0])
                comm.all_reduce(rowcnt[1])
                cuts = eng.row_shares(rowcnt[0] + rowcnt[1], world)
                share = (int(cuts[rank]), int(cuts[rank + 1]))
                inner = torch.as_tensor(cuts[1:-1], dtype=torch.int64, device=dev)
                oi, oj = torch.bucketize(ri, inner, right=True), torch.bucketize(rj, inner, right=True)
                second = torch.nonzero(oj != oi).reshape(-1)
                dest = torch.cat([oi, oj[second]])
                src = torch.cat([torch.arange(n_local, dtype=torch.int64, device=dev), second])
                order = torch.argsort(dest, stable=True)
                send_cuts = np.concatenate([[0], np.cumsum(torch.bincount(dest, minlength=world).cpu().numpy())])
                rec = comm.all_to_all_rows(rec[src[order]].contiguous(), send_cuts)
            else:
                comm.all_reduce(rowcnt[0] if split else rowcnt)
                rec = comm.all_gather_var(rec.reshape(-1)).view(-1, 3)
            n_all = int(rec.shape[0])
            coo = eng.unpack_pairs(rec)
        # No collective sits between a possible raise and the agree() that follows it: the local phases run in try blocks,
        # the all-gathers of the knn tables (ext_gather) run outside any of them.
        err, S, E, bb = None, None, None, None
        try:
            # the rows of the matrix this rank builds: its knn and reverse-list shares read nothing else
            S = eng.tri_mirror(coo, rowcnt[0], rowcnt[1], info, n_all, rows=share, counted=True) if split else eng.tri_scatter(coo, rowcnt, info, None, L)
            if share is not None:
                bb = eng.bridge_flags(S).to(torch.int32)      # of this rank's rows; a list entry is classified by its
        except Exception as e:                                # NEIGHBOUR's flag, so the flags are completed first
            err = e
        comm.agree(err, "stage B (mirror)")
        if share is not None:
            comm.all_reduce(bb, "max")
        err = None
        try:
            E = eng.ext_knn(S, k, comm, rows=share, bb=bb)
        except Exception as e:
            err = e
        comm.agree(err, "stage B (knn tables)")
        eng.ext_gather(E, comm)
        # the three reverse adjacencies (attach / src / rnn) in row shares: the list of a row comes from that row alone, so
        # the ranks' pieces are contiguous -- counts all-gathered, entries all-gathered (S8 / S9 of SURVEY 2.3)
        rows = getattr(E, "row_share", None)
        with eng.timed("reverse"):
            err = None
            try:
                eng.ext_thresholds(E)
            except Exception as e:
                err = e
            comm.agree(err, "stage B (list thresholds)")
            # attach and rnn lists together (one count phase, one agreement, one exchange each way), then the src lists, whose
            # predicate reads the attach offsets
            for group in (((0, "att"), (2, "rnn")), ((1, "src"),)):
                err, sts = None, []
                try:
                    if len(group) == 2:
                        sts = eng.reverse_count_pair(S, E, rows)          # (attach + rnn: one pass over the rows for both)
                    else:
                        for mode, _ in group:
                            sts.append(eng.reverse_count(S, E, mode, E.att[0] if mode == 1 else None, rows))
                except Exception as e:
                    err = e
                comm.agree(err, "stage B (reverse lists: count)")
                if rows is not None:
                    eng.reverse_gather_counts(sts, comm)
                err = None
                try:
                    for st8 in sts:
                        eng.reverse_fill(st8)
                except Exception as e:
                    err = e
                comm.agree(err, "stage B (reverse lists: fill)")
                outs = eng.reverse_gather(sts, comm) if rows is not None else [st8.out for st8 in sts]
                for (_, name), out in zip(group, outs):
                    setattr(E, name, out)
        err = None
        try:
            E = eng.extend_tables(E, False, None, start_split=(rank, world))
        except Exception as e:
            err = e
        comm.agree(err, "stage B (extension)")
        comm.all_reduce(E.n_cand)
        comm.all_reduce(E.top_end, "max")             # -1 outside the local range
        comm.all_reduce(E.top_val)                    # 0.0 outside the local range
        pt = torch.tensor([E.n_paths, E.n_out], dtype=torch.int64, device=dev)
        comm.all_reduce(pt)
    return S, E, pt

