"""Multi-domain AlterEgo generation (BASELINE configs[3]): N source domains -> one target.

Reference shape (code/multidomain_demo.py:101-128): every source domain is an INDEPENDENT two-domain problem against
the same target -- its own item-item similarities, extension and generator -- and the AlterEgo profiles are united at
the end (`alterEgo_profile1.union(alterEgo_profile2)`, :128).  That is domain-parallel by construction:
the source domains are dealt to rank groups, a group shards the items of its domain(s) as xmap.engine.sharded does,
and the only exchange between groups is the final union (one variable-length all-gather of the AlterEgo rows).

Index spaces: every domain problem indexes its items [its source items ..., the target items ...] (lexicographic id
order), so the target item of a row is `item - n_src_items` in a numbering common to all domains; users are common.
"""
import numpy as np
import torch

from . import sharded


def domain_groups(n_domains, world):
    """[(ranks of the group, domains of the group)]: world >= n_domains: contiguous rank groups, one domain each (the
    first world % n_domains groups one rank larger); world < n_domains: one rank per group, domains dealt round-robin."""
    if world >= n_domains:
        out, lo = [], 0
        for d in range(n_domains):
            n = world // n_domains + (1 if d < world % n_domains else 0)
            out.append((list(range(lo, lo + n)), [d]))
            lo += n
        return out
    return [([r], list(range(r, n_domains, world))) for r in range(world)]


def run_multidomain(make_engine, n_domains, method, cap, k, private, dist=None):
    """make_engine(d) -> (Engine over domain d's ratings, n_src_items of that domain); called only on the ranks that own
    domain d.  Returns the union of the AlterEgo rows of all domains on EVERY rank, domain by domain:
    dict(user, item (common target numbering), rating, time, domain, n_paths, n_rows per domain)."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    plan = domain_groups(n_domains, world)
    groups = []
    for ranks, _ in plan:           # every rank creates every group (torch.distributed requires it), in the same order
        groups.append(dist.new_group(ranks) if (dist is not None and world > 1) else None)
    mine = [(g, ranks, doms) for g, (ranks, doms) in zip(groups, plan) if rank in ranks]
    (group, ranks, doms), = mine
    parts = {"user": [], "item": [], "rating": [], "time": [], "domain": []}
    stats = torch.zeros((n_domains, 2), dtype=torch.int64)
    for d in doms:
        eng, n_src = make_engine(d)
        res = sharded.run_step(eng, method, cap, k, private, dist if len(ranks) > 1 else None, group=group)
        G = res["G"]
        if rank == ranks[0]:        # the group's first rank contributes the domain's rows (stage C is replicated in a group)
            parts["user"].append(G.user.cpu())
            parts["item"].append(G.item.cpu() - int(n_src))
            parts["rating"].append(G.rating.cpu())
            parts["time"].append(G.time.cpu())
            parts["domain"].append(torch.full((G.n_rows,), d, dtype=torch.int32))
            stats[d, 0], stats[d, 1] = int(res["n_paths"]), int(G.n_rows)
        del eng, res, G
    cat = lambda k_, dt: (torch.cat(parts[k_]) if parts[k_] else torch.zeros(0, dtype=dt))
    out = dict(user=cat("user", torch.int32), item=cat("item", torch.int32), rating=cat("rating", torch.float64),
               time=cat("time", torch.int64), domain=cat("domain", torch.int32))
    if dist is not None and world > 1:
        comm = sharded.Comm(dist)
        dev = torch.device("cpu") if comm.host else torch.device("cuda", torch.cuda.current_device())
        out = {k_: comm.all_gather_var(v.to(dev)).cpu() for k_, v in out.items()}
        st = stats.to(dev)
        comm.all_reduce(st)
        stats = st.cpu()
    # rows arrive group by group; order them by domain (a stable order independent of the rank layout)
    o = torch.sort(out["domain"], stable=True).indices
    out = {k_: v[o].numpy() for k_, v in out.items()}
    out["n_paths"], out["n_rows"] = stats[:, 0].numpy(), stats[:, 1].numpy()
    return out
