"""
The N > 1 path on CPU: two processes, gloo backend, world_size 2.  The per-star
evaluation itself needs the GPU, so it is replaced here by a deterministic
function of the star index; what is tested is the sharding and the all-gather of
per-star values (ragged shards included) -- the only inter-rank step of the
path.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_lnlike(s):
    return -0.5 * (s + 1) ** 1.5 + np.cos(s)


def _worker(rank, world, port, S, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from starry_process_amd.ensemble import all_gather_values, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(S, rank, world)
    local = torch.tensor([_fake_lnlike(s) for s in range(lo, hi)], dtype=torch.float64)
    full = all_gather_values(local, S)
    q.put((rank, lo, hi, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [8, 9, 1, 64])
def test_two_rank_allgather(S):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.array([_fake_lnlike(s) for s in range(S)])
    covered = np.zeros(S, dtype=int)
    for rank, lo, hi, full in res:
        assert np.array_equal(full, ref)       # every rank has every star
        covered[lo:hi] += 1
    assert np.all(covered == 1)                # each star evaluated exactly once


def test_shard_bounds_partition():
    from starry_process_amd.ensemble import shard_bounds

    for S in (0, 1, 7, 64, 511, 512):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(S, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == S
            for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
                assert b == c and b - a >= d - c >= 0
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
