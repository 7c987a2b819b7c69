"""The item-sharded step (xmap.engine.sharded.run_step with world_size 2, gloo collectives, both ranks on the
one GPU of the test box) must give exactly the single-rank result."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _collect(q, procs, n, limit=600):
    """n results from the workers' queue; fails at once when a worker has died (instead of waiting out the limit)"""
    import queue
    import time
    got, t0 = [], time.time()
    while len(got) < n:
        try:
            got.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, "a worker exited with %s before delivering its result" % dead
            assert time.time() - t0 < limit, "workers silent for %d s" % limit
    return got


def _summary(res):
    S, E, G = res["S"], res["E"], res["G"]
    rp = S.row_ptr.cpu().numpy()
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    col = S.col.cpu().numpy()
    o = np.lexsort((col, rows))
    part = {}
    if "S_part" in res:      # the rank's own partition of the similarity matrix (where its stage A ends)
        P = res["S_part"]
        prp = P.row_ptr.cpu().numpy()
        prow = np.repeat(np.arange(len(prp) - 1), np.diff(prp))
        part = dict(part_row=prow, part_col=P.col.cpu().numpy(), part_sim=P.sim.cpu().numpy())
    return dict(part, n_eval=res["n_eval"], n_kept=res["n_kept"], n_paths=res["n_paths"], n_rows=res["n_rows"],
                n_profiles=res["n_profiles"], row_ptr=rp, col=col[o], sim=S.sim.cpu().numpy()[o],
                n_cand=E.n_cand.cpu().numpy(), top_end=E.top_end.cpu().numpy(), top_val=E.top_val.cpu().numpy(),
                choice=res["choice"].cpu().numpy(), map=res["map"].cpu().numpy(),
                ae_user=G.user.cpu().numpy(), ae_item=G.item.cpu().numpy(), ae_rating=G.rating.cpu().numpy())


def _engine(dev="cuda:0"):
    from xmap.engine import synth, device
    r = synth.make_two_domain(21, 4000, 700, 700)
    return device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs(), dev))


def _init(rank, world, port, backend):
    """gloo: every rank on cuda:0 (several ranks on the one GPU of the test box); nccl: rank r on cuda:r over RCCL"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        return dist, "cuda:%d" % rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, "cuda:0"


def _rccl_world():
    """ranks of the RCCL tests: one per visible GPU, at most 8; the tests are skipped on a one-GPU box"""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("RCCL with more than one rank needs >= 2 visible GPUs (%d here)" % n)
    return n


def _worker(rank, world, port, q, backend="gloo"):
    dist, dev = _init(rank, world, port, backend)
    from xmap.engine import sharded
    res = sharded.run_step(_engine(dev), "adjust_cosine", 50, 5, True, dist, rank, world)
    q.put((rank, _summary(res)))
    dist.barrier()
    dist.destroy_process_group()


def _rccl_item_worker(port, q):
    os.environ["XMAP_FORCE_DIST"] = "1"          # the sharded path with one rank
    dist, dev = _init(0, 1, port, "nccl")
    from xmap.engine import sharded
    res = sharded.run_step(_engine(dev), "adjust_cosine", 50, 5, True, dist, 0, 1)
    q.put((0, _summary(res)))
    dist.barrier()
    dist.destroy_process_group()


def test_item_sharded_collectives_through_rccl():
    """the item-sharded step with ONE share and RCCL as the backend (XMAP_FORCE_DIST=1): the exchange of the kept pairs to their
    row owners (all_to_all_single), the packed all-gathers of the item statistics, knn tables, reverse lists and AlterEgo rows,
    the agreements and all-reduces all run on device buffers; the result is the plain step's"""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    from xmap.engine import sharded
    ref = _summary(sharded.run_step(_engine(), "adjust_cosine", 50, 5, True))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_item_worker, args=(_free_port(), q))
    p.start()
    (rank, out), = _collect(q, [p], 1)
    p.join(120)
    assert p.exitcode == 0
    assert ref["n_kept"] > 0 and ref["n_paths"] > 0 and ref["n_rows"] > 0
    for key, v in ref.items():
        assert np.array_equal(out[key], v), key


def test_item_sharded_over_rccl_equals_world1():
    """run_step on the nccl backend, one rank per visible GPU (BASELINE configs[2]); bit-equal to one rank"""
    _check_world_equals_world1(_rccl_world(), "nccl")


@pytest.mark.parametrize("world", [2, 3])
def test_world2_equals_world1(world):
    _check_world_equals_world1(world, "gloo")


def _check_world_equals_world1(world, backend):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    from xmap.engine import sharded
    ref = _summary(sharded.run_step(_engine(), "adjust_cosine", 50, 5, True))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    got = _collect(q, procs, world)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ref["n_kept"] > 0 and ref["n_paths"] > 0 and ref["n_rows"] > 0
    for rank, out in got:
        for key, v in ref.items():
            if key in ("row_ptr", "col", "sim"):
                continue
            assert np.array_equal(out[key], v), (rank, key)
    # stage B builds the matrix in row shares (a rank mirrors its rows only): contiguous, disjoint, together the whole matrix
    I = len(ref["row_ptr"]) - 1
    lens = np.stack([np.diff(out["row_ptr"]) for _, out in sorted(got)])
    assert np.array_equal(lens.sum(axis=0), np.diff(ref["row_ptr"])) and np.all((lens > 0).sum(axis=0) <= 1)
    owner = np.where((lens > 0).any(axis=0), lens.argmax(axis=0), -1)
    assert np.all(np.diff(owner[owner >= 0]) >= 0)
    assert world > 4 or all((owner == r).any() for r in range(world))      # (eight shares of this small matrix may leave one empty)
    s_rows = np.concatenate([np.repeat(np.arange(I), np.diff(out["row_ptr"])) for _, out in sorted(got)])
    s_cols = np.concatenate([out["col"] for _, out in sorted(got)])
    s_sims = np.concatenate([out["sim"] for _, out in sorted(got)])
    o = np.lexsort((s_cols, s_rows))
    assert np.array_equal(s_rows[o], np.repeat(np.arange(I), np.diff(ref["row_ptr"])))
    assert np.array_equal(s_cols[o], ref["col"]) and np.array_equal(s_sims[o], ref["sim"])
    # the two stage-A partitions are disjoint and together are the whole matrix
    rows = np.concatenate([out["part_row"] for _, out in got])
    cols = np.concatenate([out["part_col"] for _, out in got])
    sims = np.concatenate([out["part_sim"] for _, out in got])
    o = np.lexsort((cols, rows))
    full_rows = np.repeat(np.arange(I), np.diff(ref["row_ptr"]))
    assert np.array_equal(rows[o], full_rows) and np.array_equal(cols[o], ref["col"])
    assert np.array_equal(sims[o], ref["sim"])
    assert world > 4 or all(len(out["part_row"]) > 0 for _, out in got)


def _md_engine(d, dev="cuda:0"):
    from xmap.engine import synth, device
    r = synth.make_multi_domain(33, 3000, 500, 600, 3)[d]
    return device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs(), dev)), r.n_src_items


def _md_worker(rank, world, port, q, backend="gloo"):
    dist, dev = _init(rank, world, port, backend)
    from xmap.engine import multidomain
    out = multidomain.run_multidomain(lambda d: _md_engine(d, dev), 3, "adjust_cosine", 50, 5, True, dist)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_multidomain_over_rccl_equals_one_rank():
    """run_multidomain on the nccl backend, one rank per visible GPU (rank groups with group-local RCCL communicators)"""
    _check_multidomain(_rccl_world(), "nccl")


@pytest.mark.parametrize("world", [2, 4])
def test_multidomain_rank_groups_equal_one_rank(world):
    """BASELINE configs[3] shape (3 source domains -> one target): the domains dealt to rank groups (world 2: one rank
    takes two domains; world 4: one domain is item-sharded over a group of two ranks) give exactly the union one rank
    computes domain after domain."""
    _check_multidomain(world, "gloo")


def _check_multidomain(world, backend):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    from xmap.engine import multidomain
    ref = multidomain.run_multidomain(_md_engine, 3, "adjust_cosine", 50, 5, True)
    assert (ref["n_paths"] > 0).all() and (ref["n_rows"] > 0).all() and len(ref["user"]) == ref["n_rows"].sum()
    assert sorted(set(ref["domain"].tolist())) == [0, 1, 2]
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_md_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    got = _collect(q, procs, world)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, out in got:
        for key, v in ref.items():
            assert np.array_equal(out[key], v), (rank, key)


def _user_share(rank, world, dev="cuda:0"):
    """the complete profiles of a contiguous share of the users (items indexed globally)"""
    from xmap.engine import synth, device
    r = synth.make_two_domain(21, 4000, 700, 700)
    lo, hi = r.n_users * rank // world, r.n_users * (rank + 1) // world
    e0, e1 = int(r.user_ptr[lo]), int(r.user_ptr[hi])
    ptr = (r.user_ptr[lo:hi + 1] - r.user_ptr[lo]).astype(np.int64)
    R = device.DeviceRatings(ptr, r.item[e0:e1].copy(), r.rating[e0:e1].copy(), r.time[e0:e1].copy(), r.n_items, r.item_attrs(), dev)
    return device.Engine(R), lo


def _users_worker(rank, world, port, method, q, backend="gloo"):
    dist, dev = _init(rank, world, port, backend)
    from xmap.engine import sharded
    eng, lo = _user_share(rank, world, dev)
    res = sharded.run_step_users(eng, lo, method, 50, 5, True, dist)
    out = _summary(res)
    out["info"] = res["info"].cpu().numpy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_user_sharded_over_rccl_equals_world1():
    """run_step_users on the nccl backend, one rank per visible GPU: the all-to-all of the partial similarities between
    real peers (the one-rank RCCL test below only sends to itself)"""
    _check_user_sharded(_rccl_world(), "adjust_cosine", "nccl")


@pytest.mark.parametrize("world,method", [(2, "adjust_cosine"), (3, "cosine"), (4, "adjust_cosine")])
def test_user_sharded_equals_world1(world, method):
    """BASELINE configs[2]'s other split: every rank holds a share of the USERS, the partial similarities of a pair are
    sent to the rank that owns it and added up there (sharded.run_step_users).  Item statistics, similarity matrix, extension,
    replacements and AlterEgo rows must be those of one rank over all ratings, bit for bit."""
    _check_user_sharded(world, method, "gloo")


def _check_user_sharded(world, method, backend):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    from xmap.engine import sharded
    one = sharded.run_step(_engine(), method, 50, 5, True)
    ref = _summary(one)
    ref["info"] = one["S"].info.cpu().numpy()
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_users_worker, args=(r, world, port, method, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    got = _collect(q, procs, world)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ref["n_kept"] > 0 and ref["n_paths"] > 0 and ref["n_rows"] > 0
    I = len(ref["row_ptr"]) - 1

    def canon(d):       # the AlterEgo rows as a set: one rank lists them kind by kind, the shares are concatenated user range by user range
        o = np.lexsort((d["ae_rating"], d["ae_item"], d["ae_user"]))
        for key in ("ae_user", "ae_item", "ae_rating"):
            d[key] = np.asarray(d[key])[o]
    canon(ref)
    for rank, out in got:
        canon(out)
        for key, v in ref.items():
            assert np.array_equal(out[key], v), (rank, key)
    # the ranks' partitions (pairs owned through their lower item) are disjoint and together are the whole matrix
    rows = np.concatenate([out["part_row"] for _, out in got])
    cols = np.concatenate([out["part_col"] for _, out in got])
    sims = np.concatenate([out["part_sim"] for _, out in got])
    o = np.lexsort((cols, rows))
    full_rows = np.repeat(np.arange(I), np.diff(ref["row_ptr"]))
    assert np.array_equal(rows[o], full_rows) and np.array_equal(cols[o], ref["col"])
    assert np.array_equal(sims[o], ref["sim"])


def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from xmap.engine import sharded
    sharded.Comm.A2A_PIECE_BYTES = 1 << 12        # 128 records per piece: the all-to-all takes hundreds of rounds
    eng, lo = _user_share(0, 1)
    res = sharded.run_step_users(eng, lo, "adjust_cosine", 50, 5, True, dist)
    out = _summary(res)
    out["info"] = res["info"].cpu().numpy()
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_user_sharded_collectives_through_rccl():
    """the user-sharded step with ONE share and RCCL as the backend: every collective of the path (fixed and variable
    all-gathers, the all-to-all of the partial records in many small pieces, all-reduces) runs on device buffers; the
    result is the plain step's"""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    from xmap.engine import sharded
    one = sharded.run_step(_engine(), "adjust_cosine", 50, 5, True)
    ref = _summary(one)
    ref["info"] = one["S"].info.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    for d in (ref, out):
        o = np.lexsort((d["ae_rating"], d["ae_item"], d["ae_user"]))
        for key in ("ae_user", "ae_item", "ae_rating"):
            d[key] = np.asarray(d[key])[o]
    for key, v in ref.items():
        if not key.startswith("part_"):
            assert np.array_equal(out[key], v), key
