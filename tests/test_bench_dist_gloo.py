"""
Dry run of bench.py's multi-rank control flow WITHOUT GPUs: two processes,
gloo backend, the benchmark's own `timed_steps` / `Harness` loop with a stub evaluator in place
of the device step.  What is exercised: warm-up, the stated pre-warm, the barrier / max-over-
ranks bracket of exactly K steps, three steps in flight each issuing its all-gather on the one
communicator (the order of the collectives must be the same on every rank: a mismatch deadlocks
or mixes shards), the weak-scaling shard of every rank and the whole-job value.  Also: the ragged
sharded path with a rank that owns no star (world 3, S 2).

Round 3: `python bench.py --gpus 2` END TO END -- the script starts its two
ranks itself (launch_ranks), they rendezvous on 127.0.0.1, shard, all-gather and rank 0 prints
the job's line; with SP_BENCH_BACKEND=gloo the evaluator is bench.StubSlot.  A WORLD_SIZE that
contradicts --gpus, or a rank that dies, must end the job with a non-zero exit code.
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
    sys.path.insert(0, ROOT)
    import bench

    return bench._stub_lnlike(s, step_id)


def _bench_worker(rank, world, port, S, steps, warmup, F, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slots = [bench.StubSlot(dist, torch, rank, world, S, i) for i in range(F)]

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


def _run_bench(argv, env_extra, timeout=240):
    import subprocess

    env = dict(os.environ, SP_BENCH_BACKEND="gloo", SP_BENCH_STARS="64")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)


def test_bench_py_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two ranks, ONE line, n_gpus 2, 128 gathered values."""
    import json

    r = _run_bench(["--gpus", "2", "--steps", "7", "--warmup", "2"], {})
    assert r.returncode == 0, r.stderr
    # (gloo itself prints a "[Gloo] Rank 0 is connected ..." notice on stdout; RCCL does not)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 7 and rec["warmup"] == 2
    assert rec["gathered_values"] == 128 and rec["parity_ok"] is True
    assert rec["backend"] == "gloo-stub" and rec["scaling"] == "weak"
    assert np.isfinite(rec["value"]) and rec["value"] > 0


def test_bench_py_refuses_a_world_that_is_not_gpus():
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and "{" not in r.stdout


def test_bench_py_reports_a_dead_rank():
    """A rank that fails takes the job down with a non-zero code (no hang on the survivor's collective)."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"SP_BENCH_STUB_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0 and "rank 1 exited" in r.stderr


def test_bench_py_eight_ranks_gathers_512_values_in_star_order():
    """cfg4's shape on CPU: `python bench.py --gpus 8` (gloo stub) -- eight ranks, 64 stars each, and rank 0's
    line reports 512 gathered values that equal the per-star reference IN STAR ORDER (contiguous shards:
    rank r owns stars 64 r .. 64 r + 63), every rank's own ms_per_step, and says where the CPU baseline is."""
    import json
    import subprocess

    env = dict(os.environ, SP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2",
                        "--in-flight", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["gathered_values"] == 512 and line["parity_ok"] is True
    assert line["scaling"] == "weak" and line["config"]["stars_per_gpu"] == 64
    pr = line["per_rank_ms_per_step"]
    assert len(pr["all"]) == 8 and pr["min"] <= pr["max"] <= line["ms_per_step"] * 1.5
    assert "N = 1" in line["cpu_baseline"]


def test_bench_py_parent_killed_takes_its_ranks_down():
    """SIGTERM on the parent of `bench.py --gpus 2` (a `timeout`, the driver) must end both ranks."""
    import signal
    import subprocess
    import time

    env = dict(os.environ, SP_BENCH_BACKEND="gloo", SP_BENCH_STUB_SLEEP="60")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(6.0)                       # the ranks are up (and asleep: SP_BENCH_STUB_SLEEP)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 2, kids
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode != 0 and "interrupted" in err
    time.sleep(0.5)
    for k in kids:
        assert not os.path.exists("/proc/%s" % k) or open("/proc/%s/stat" % k).read().split()[2] == "Z"
