"""
Multi-GPU evaluation of an ensemble of independent light curves.

Given one hyperparameter sample the stars are independent: star ``s`` has its own
``t, flux, p, i, u, noise`` -> its own K x K covariance -> its own scalar.  The
ensemble is therefore sharded over ranks (one process per GPU, launched with
``torch.distributed.run``) with NO collective on the data path; the only
exchange is one all-gather of the per-star log-likelihoods (``S`` doubles, a few
KB: latency-bound on xGMI, bucket size irrelevant) so that every rank -- i.e.
every walker of an MCMC / nested sampler -- holds the full vector and its sum.

On GPUs the process group backend is ``nccl`` (= RCCL on ROCm) and the gather
runs on device tensors; the same code runs on ``gloo`` with CPU tensors, which is
what the CPU tests use.
"""
import numpy as np

__all__ = ["shard_bounds", "all_gather_values", "sharded_log_likelihood"]


def shard_bounds(S, rank, world):
    """Contiguous, balanced partition: rank r owns stars [lo, hi)."""
    base, rem = divmod(int(S), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_gather_values(local, S, group=None):
    """Gather per-star values (torch tensor, this rank's shard in star order)
    into the full length-S vector on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        assert local.shape[0] == S
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(S, rank, world)
    assert local.shape[0] == hi - lo, "local shard has the wrong length"
    width = -(-S // world)  # ceil: every rank contributes an equal-size slot
    slot = torch.full((width,), float("nan"), dtype=local.dtype, device=local.device)
    slot[: hi - lo] = local
    buf = torch.empty(world * width, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, slot, group=group)
    out = torch.empty(S, dtype=local.dtype, device=local.device)
    for r in range(world):
        a, b = shard_bounds(S, r, world)
        out[a:b] = buf[r * width : r * width + (b - a)]
    return out


def sharded_log_likelihood(sp, t, flux, data_cov, p=None, i=None, u=None,
                           baseline_mean=0.0, baseline_var=0.0, group=None):
    """Per-star log-likelihoods of the whole ensemble on every rank.

    ``sp`` is a ``StarryProcess`` bound to this rank's GPU; the arguments
    describe ALL S stars (as for ``StarryProcess.log_likelihood_ensemble``);
    each rank evaluates only its shard."""
    import torch
    import torch.distributed as dist

    ragged = isinstance(flux, (list, tuple))   # light curves of different lengths
    if not ragged:
        flux = np.asarray(flux, dtype=np.float64)
    S = len(flux)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(S, rank, world)

    def cut(x):
        if isinstance(x, (list, tuple)):
            return x[lo:hi] if len(x) == S else x
        x = np.asarray(x)
        return x[lo:hi] if x.ndim >= 1 and x.shape[0] == S else x

    if hi == lo:
        # more ranks than stars (or an uneven split that leaves this rank empty): nothing to
        # evaluate, but the collective below is entered by every rank
        local_t = torch.empty(0, dtype=torch.float64, device=sp._engine.device)
        return all_gather_values(local_t, S, group).cpu().numpy()
    if ragged:
        tl = cut(t)
    else:
        t = np.asarray(t, dtype=np.float64)
        tl = cut(t) if t.ndim == 2 else t
    local = sp.log_likelihood_ensemble(
        tl, flux[lo:hi], cut(data_cov),
        i=None if i is None else cut(i), p=None if p is None else cut(p),
        u=None if u is None else (cut(u) if np.ndim(u) == 2 else u),
        baseline_mean=cut(baseline_mean), baseline_var=cut(baseline_var))
    local_t = torch.from_numpy(np.asarray(local, dtype=np.float64)).to(sp._engine.device)
    return all_gather_values(local_t, S, group).cpu().numpy()
