"""The N > 1 path on CPU: prompt sharding + the stats all-gather, world_size 2 over gloo."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from specdec_hip.dist_stats import FIELDS, gather_stats, shard_indices


def test_shard_indices_cover_and_balance():
    for n, world in ((32, 8), (10, 4), (3, 8), (0, 2)):
        parts = [shard_indices(n, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_indices(32, 3, 8) == [3, 11, 19, 27]
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def test_gather_is_identity_without_process_group():
    st = gather_stats({f: i + 1 for i, f in enumerate(FIELDS)}, torch.device("cpu"))
    assert st.per_rank.shape == (1, len(FIELDS)) and st.total("tokens") == 1


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(7, rank, world)
    local = {"tokens": 100 * (rank + 1), "proposed": 40, "accepted": 30 + rank, "accepted_strict": 20 + rank,
             "wall_ns": int((1.0 + rank) * 1e9), "steps": len(mine)}
    dist.barrier()
    st = gather_stats(local, torch.device("cpu"))
    q.put((rank, st.per_rank.tolist(), st.tokens_per_s(), st.acceptance(), st.acceptance(True), mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_all_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, per_rank, tps, acc, acc_s, mine in got:
        # every rank sees both rows; whole-job rate = all tokens / slowest rank
        assert per_rank[0][0] == 100 and per_rank[1][0] == 200
        assert tps == pytest.approx(300 / 2.0)
        assert acc == pytest.approx(61 / 80) and acc_s == pytest.approx(41 / 80)
    assert got[0][5] == [0, 2, 4, 6] and got[1][5] == [1, 3, 5]
