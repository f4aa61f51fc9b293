"""
Dry run of bench.py's multi-rank control flow WITHOUT GPUs: two processes,
gloo backend, the benchmark's own `timed_steps` / `Harness` loop with a stub evaluator in place
of the device step.  What is exercised: warm-up, the stated pre-warm, the barrier / max-over-
ranks bracket of exactly K steps, three steps in flight each issuing its all-gather on the one
communicator (the order of the collectives must be the same on every rank: a mismatch deadlocks
or mixes shards), the weak-scaling shard of every rank and the whole-job value.  Also: the ragged
sharded path with a rank that owns no star (world 3, S 2).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_lnlike(s, step_id):
    return -0.5 * (s + 1) ** 1.5 + np.cos(s) + 1e-3 * step_id


class StubSlot(object):
    """A slot of the benchmark whose "device step" is a deterministic function of (star, step):
    per-star values of this rank's shard, then the all-gather bench.py's Slot.step issues."""

    def __init__(self, dist, torch, rank, world, S, slot_id):
        self.dist, self.torch, self.rank, self.world, self.S = dist, torch, rank, world, S
        self.slot_id = slot_id
        self.calls = 0
        self.out = torch.zeros(S, dtype=torch.float64)
        self.gathered = torch.zeros(world * S, dtype=torch.float64)
        self.history = []

    def run(self):
        first = self.rank * self.S
        # (the step id is a function of the call count only: the same on every rank)
        step_id = self.calls * 16 + self.slot_id
        self.out[:] = self.torch.tensor([_fake_lnlike(s, step_id) for s in range(first, first + self.S)])
        self.dist.all_gather_into_tensor(self.gathered, self.out)
        self.history.append((step_id, self.gathered.clone()))
        self.calls += 1


def _bench_worker(rank, world, port, S, steps, warmup, F, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slots = [StubSlot(dist, torch, rank, world, S, i) for i in range(F)]

    def max_over_ranks(x):
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    h = bench.Harness(slots, lambda: None, dist.barrier, max_over_ranks)
    armed = []
    elapsed, enq, pre = bench.timed_steps(h, steps, warmup, prewarm_ms=5.0, before_timed=lambda: armed.append(1))
    hist = [(sid, g.numpy().copy()) for sl in slots for sid, g in sl.history]
    q.put((rank, elapsed, pre, len(armed), [sl.calls for sl in slots], hist))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F", [1, 3])
def test_bench_control_flow_two_ranks(F):
    world, S, steps, warmup = 2, 4, 7, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, S, steps, warmup, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, pre0, armed0, calls0, hist0), (r1, e1, pre1, armed1, calls1, hist1) = res
    assert e0 == e1 > 0.0                       # max over ranks: the same number on every rank
    assert armed0 == armed1 == 1
    assert pre0 >= 3 * F and pre0 % F == 0      # every slot at least 3 times
    # exactly `steps` timed steps after max(warmup, F) + pre-warm steps, dealt round-robin
    nwarm = max(warmup, F)
    total = nwarm + pre0 + steps
    assert sum(calls0) == total
    # every gathered vector holds rank 0's stars then rank 1's, of the SAME step on both ranks
    assert len(hist0) == len(hist1)
    for (sid0, g0), (sid1, g1) in zip(hist0, hist1):
        assert sid0 == sid1
        ref = np.array([_fake_lnlike(s, sid0) for s in range(world * S)])
        assert np.array_equal(g0, ref) and np.array_equal(g1, ref)
    # whole-job value as bench.py computes it
    value = world * S * steps / e0
    assert np.isfinite(value) and value > 0


def _ragged_worker(rank, world, port, S, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from starry_process_amd import ensemble

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeEngine(object):
        device = torch.device("cpu")

    class FakeProcess(object):
        _engine = FakeEngine()

        def log_likelihood_ensemble(self, t, flux, data_cov, **kw):
            assert len(flux) > 0, "an empty shard must not reach the device call"
            return np.array([float(np.sum(f)) for f in flux])

    t = [np.linspace(0, 1, 5 + s) for s in range(S)]
    flux = [np.full(5 + s, 1.0 + s) for s in range(S)]
    full = ensemble.sharded_log_likelihood(FakeProcess(), t, flux, [np.ones(5 + s) for s in range(S)])
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_shards_with_an_empty_rank():
    world, S = 3, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.array([(5 + s) * (1.0 + s) for s in range(S)])
    for rank, full in res:
        assert np.array_equal(full, ref)
