"""world_size-2 gloo test (CPU) of the collectives wrapper the sharded step uses (xmap.engine.sharded.Comm):
variable-length all-gather in rank order, all-reduce sum/max, and the shard planning being identical on all ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmap.engine.sharded import Comm, balanced_ranges
    comm = Comm(dist)
    # the half-COO parts of two ranks: different lengths, including an empty one
    part = torch.arange(3 * rank, dtype=torch.int32) + 100 * rank
    allp = comm.all_gather_var(part)
    val = comm.all_gather_var(torch.full((2 - rank,), float(rank), dtype=torch.float64))
    cnt = torch.tensor([1, 2, 3], dtype=torch.int32) * (rank + 1)
    comm.all_reduce(cnt)
    top = torch.full((4,), -1, dtype=torch.int32)
    top[rank * 2:(rank + 1) * 2] = rank + 7
    comm.all_reduce(top, "max")
    w = np.array([4, 1, 1, 1, 1, 4, 2, 2])
    # several tensors of different types and lengths in ONE exchange (odd byte counts, an empty part, nothing at all from rank 0)
    multi = comm.all_gather_multi([torch.arange(5 * rank, dtype=torch.uint8) + rank, torch.full((3 - rank,), 0.5 + rank, dtype=torch.float64),
                                   torch.arange(7 * rank, dtype=torch.int32) * (rank + 1), torch.zeros(0, dtype=torch.int64)])
    q.put((rank, allp.tolist(), val.tolist(), cnt.tolist(), top.tolist(), balanced_ranges(w, world),
           [(str(m.dtype), m.tolist()) for m in multi]))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, allp, val, cnt, top, ranges, multi in got:
        assert multi == [("torch.uint8", [1, 2, 3, 4, 5]), ("torch.float64", [0.5, 0.5, 0.5, 1.5, 1.5]),
                         ("torch.int32", [0, 2, 4, 6, 8, 10, 12]), ("torch.int64", [])]
        assert allp == [100, 101, 102]            # rank 0 contributed nothing, rank 1 three entries
        assert val == [0.0, 0.0, 1.0]
        assert cnt == [3, 6, 9]
        assert top == [7, 7, 8, 8]
        assert ranges == [(0, 5), (5, 8)]


def _worker_groups(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmap.engine.sharded import Comm
    from xmap.engine.multidomain import domain_groups
    plan = domain_groups(2, world)                       # two domains on four ranks: groups {0,1} and {2,3}
    groups = [dist.new_group(ranks) for ranks, _ in plan]
    g = [i for i, (ranks, _) in enumerate(plan) if rank in ranks][0]
    comm = Comm(dist, groups[g])
    t = torch.tensor([rank + 1], dtype=torch.int64)
    comm.all_reduce(t)                                   # sum inside the group only
    parts = comm.all_gather_var(torch.arange(comm.rank + 1, dtype=torch.int32) + 10 * rank)
    # collective error agreement: the second rank of group 1 fails, both ranks of THAT group raise, group 0 goes on
    outcome = "ok"
    try:
        comm.agree(ValueError("boom") if rank == 3 else None, "stage X")
    except ValueError:
        outcome = "own"
    except RuntimeError as e:
        outcome = "peer" if "another rank" in str(e) else "other"
    q.put((rank, g, comm.rank, comm.world, int(t.item()), parts.tolist(), outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_groups_and_error_agreement_world4_gloo():
    from xmap.engine.multidomain import domain_groups
    assert domain_groups(4, 8) == [([0, 1], [0]), ([2, 3], [1]), ([4, 5], [2]), ([6, 7], [3])]
    assert domain_groups(4, 2) == [([0], [0, 2]), ([1], [1, 3])]
    assert domain_groups(3, 4) == [([0, 1], [0]), ([2], [1]), ([3], [2])]
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r, g, gr, gw) for r, g, gr, gw, _, _, _ in got] == [(0, 0, 0, 2), (1, 0, 1, 2), (2, 1, 0, 2), (3, 1, 1, 2)]
    assert [t for *_, t, _, _ in got] == [3, 3, 7, 7]
    assert got[0][5] == [0, 10, 11] and got[2][5] == [20, 30, 31]
    assert [o for *_, o in got] == ["ok", "ok", "peer", "own"]


def _worker_a2a(rank, world, port, q, native=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if native:
        os.environ["XMAP_A2A_NATIVE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xmap.engine.sharded import Comm
    comm = Comm(dist)
    assert comm.a2a_native == native
    if native:
        comm.A2A_PIECE_BYTES = 16 * 3             # three rows of two int64 per piece: several rounds, the last ones ragged
    # rank s holds rows (key, s) with keys 0 .. 5+s, sorted; keys [2 r, 2 r + 2) belong to rank r, the rest to the last
    n = 6 + rank
    rows = torch.stack([torch.arange(n, dtype=torch.int64), torch.full((n,), rank, dtype=torch.int64)], dim=1)
    cuts = [0] + [min(2 * r, n) for r in range(1, world)] + [n]
    got = comm.all_to_all_rows(rows, cuts)
    fixed = comm.all_gather_fixed(torch.tensor([[rank, 10 * rank]], dtype=torch.float64))
    q.put((rank, got.tolist(), fixed.tolist()))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("native", [False, True])
def test_all_to_all_rows_world3_gloo(native):
    """the exchange of the sharded steps' records (Comm.all_to_all_rows) and the fixed-size all-gather of the item sums: every
    rank receives exactly its key ranges, sender after sender.  native=False: the all-gather form the gloo rehearsals on one
    GPU use; native=True: the branch RCCL takes -- all_to_all_single with split sizes, in pieces (3 rows each here: several
    rounds, ragged last ones) -- run on gloo's own all-to-all with three ranks, which no one-GPU box can do on RCCL itself"""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_a2a, args=(r, world, port, q, native)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, rows, fixed in got:
        lo = 2 * rank
        want = []
        for s in range(world):
            hi = 2 * rank + 2 if rank < world - 1 else 6 + s
            want += [[key, s] for key in range(lo, hi)]
        assert rows == want, (rank, rows, want)
        assert fixed == [[[float(s), 10.0 * s]] for s in range(world)]
