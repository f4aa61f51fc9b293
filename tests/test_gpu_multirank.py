"""
The N > 1 path on real GPUs (VERDICT r02 item 2 ii): two processes, one per GPU, backend `nccl`
(= RCCL over xGMI).  Skipped when fewer than two GPUs are visible (the round's one-GPU box).

  * `ensemble.sharded_log_likelihood`: contiguous shards of an 11-star ensemble (uneven: 6 + 5),
    each rank evaluates its own stars on its own GPU, one all-gather -- every rank must hold the
    values a single GPU computes for the whole ensemble, bit for bit (the per-star results do not
    depend on the batch);
  * `sp_allgather_lnlike` through a raw RCCL communicator of two ranks created
    from a unique id that rank 0 broadcasts with torch.distributed.

Sharding as SURVEY 8(e); the caller it serves: calibrate/log_prob.py:53-85.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ensemble(S, K):
    from starry_process_amd.synthetic import synthetic_star

    sts = [synthetic_star(s, K) for s in range(S)]
    return (np.array([st["t"] for st in sts]), np.array([st["flux"] for st in sts]),
            np.array([st["p"] for st in sts]))


def _worker(rank, world, port, S, K, q):
    sys.path.insert(0, ROOT)
    import ctypes

    import torch.distributed as dist

    from conftest import golden
    from starry_process_amd import StarryProcess, _lib
    from starry_process_amd.ensemble import shard_bounds, sharded_log_likelihood

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    mom = golden("moments_L5")
    sp = StarryProcess(ydeg=5, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"], device=rank)
    t, flux, p = _ensemble(S, K)
    full = sharded_log_likelihood(sp, t, flux, 1e-6, p=p)

    # sp_allgather_lnlike on a raw communicator: rank 0's unique id reaches the others through torch
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)

    class UID(ctypes.Structure):
        _fields_ = [("b", ctypes.c_char * 128)]

    u = UID()
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(u)) == 0
    ut = torch.tensor(list(bytes(u)), dtype=torch.uint8, device="cuda")
    dist.broadcast(ut, 0)
    ctypes.memmove(ctypes.byref(u), bytes(ut.cpu().numpy().tolist()), 128)
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, u, rank) == 0
    n = 7
    x = torch.arange(n, dtype=torch.float64, device="cuda") * 1.5 - 2.0 + 100.0 * rank
    y = torch.full((world * n,), float("nan"), dtype=torch.float64, device="cuda")
    e = sp._engine
    _lib.check(_lib.lib().sp_allgather_lnlike(e._h, comm, x.data_ptr(), n, y.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)
    q.put((rank, shard_bounds(S, rank, world), np.asarray(full), y.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_rccl_sharded_lnlike_and_allgather():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    from conftest import golden
    from starry_process_amd import StarryProcess

    world, S, K = 2, 11, 100
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, K, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    # one GPU, the whole ensemble
    mom = golden("moments_L5")
    sp = StarryProcess(ydeg=5, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"], device=0)
    t, flux, p = _ensemble(S, K)
    ref = np.asarray(sp.log_likelihood_ensemble(t, flux, 1e-6, p=p))
    assert np.all(np.isfinite(ref))
    covered = np.zeros(S, dtype=int)
    n = 7
    gathered = np.concatenate([np.arange(n) * 1.5 - 2.0 + 100.0 * r for r in range(world)])
    for rank, (lo, hi), full, y in res:
        assert np.array_equal(full, ref)
        assert np.array_equal(y, gathered)
        covered[lo:hi] += 1
    assert np.all(covered == 1)
