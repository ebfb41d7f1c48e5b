"""Frame sharding across the GPUs of one node.

Every (frame, person, keypoint) unit is independent, so rank r of G takes the contiguous frame
block [r*F/G, (r+1)*F/G); the only exchange step is ONE all-gather of the packed per-unit results
(32 + 1 bytes per unit) -- RCCL over xGMI when the process group is 'nccl', gloo on CPU in tests.
The sequential post-processing (tracking, interpolation, .trc) then runs on every rank's copy and
only rank 0 writes files.
"""
import os

import numpy as np


def dist_info():
    """(rank, world) of the default process group, (0, 1) when torch.distributed is not in use."""
    import sys
    if 'torch' not in sys.modules and 'WORLD_SIZE' not in os.environ:
        return 0, 1                  # single process that never touched torch: do not pay its import (~2-5 s)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def shard_bounds(n_frames, rank, world):
    """Contiguous frame block of `rank`: sizes differ by at most one, earlier ranks get the extra."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def pack_results(Q, err, nex, mask):
    """[n][K] results -> one contiguous uint8 buffer [Q f64 x3 | err f32 | mask u32 | n_excl u8]."""
    parts = [np.ascontiguousarray(Q, dtype=np.float64).view(np.uint8).ravel(),
             np.ascontiguousarray(err, dtype=np.float32).view(np.uint8).ravel(),
             np.ascontiguousarray(mask, dtype=np.uint32).view(np.uint8).ravel(),
             np.ascontiguousarray(nex, dtype=np.uint8).ravel()]
    return np.concatenate(parts)


def unpack_results(buf, n_blocks, K):
    n = n_blocks * K
    o = 0
    Q = buf[o:o + n * 24].view(np.float64).reshape(n_blocks, K, 3); o += n * 24
    err = buf[o:o + n * 4].view(np.float32).reshape(n_blocks, K); o += n * 4
    mask = buf[o:o + n * 4].view(np.uint32).reshape(n_blocks, K); o += n * 4
    nex = buf[o:o + n].reshape(n_blocks, K)
    return Q, err, nex, mask


def sharded_triangulate(compute, xyl):
    """Run ``compute(xyl_shard) -> (Q, err, n_excl, mask)`` on this rank's frames and all-gather.

    xyl: [F][Pn][C][K][3] (every rank holds, or can load, the same tensor; only its shard is read).
    Returns the full-size (Q [F][Pn][K][3], err, n_excl, mask) on every rank.
    """
    rank, world = dist_info()
    F = xyl.shape[0]
    if world == 1 and not os.environ.get('P2S_FORCE_COLLECTIVE'):   # the variable: tests of the RCCL path on one GPU
        return compute(xyl)
    lo, hi = shard_bounds(F, rank, world)
    return gather_results(compute(xyl[lo:hi]), F, xyl.shape[1], xyl.shape[3])


def gather_results(local, F, Pn, K):
    """The single collective of the path: every rank contributes the results of its contiguous frame block
    (``shard_bounds``) and receives the whole trajectory.  local = (Q [n][Pn][K][3], err, n_excl, mask)."""
    rank, world = dist_info()
    Q, err, nex, mask = local
    if world == 1 and not os.environ.get('P2S_FORCE_COLLECTIVE'):
        return Q, err, nex, mask
    import torch
    import torch.distributed as dist
    lo, hi = shard_bounds(F, rank, world)
    nb_max = (shard_bounds(F, 0, world)[1] - shard_bounds(F, 0, world)[0]) * Pn
    per_unit = 24 + 4 + 4 + 1
    local_buf = np.zeros(nb_max * K * per_unit, dtype=np.uint8)
    nb = (hi - lo) * Pn
    # sections are sized for the largest block: pad the local block to it
    local_buf[:] = pack_results(_pad(np.asarray(Q).reshape(nb, K, 3), nb_max), _pad(np.asarray(err).reshape(nb, K), nb_max),
                                _pad(np.asarray(nex).reshape(nb, K), nb_max), _pad(np.asarray(mask).reshape(nb, K), nb_max))
    backend = dist.get_backend()
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    t_local = torch.from_numpy(local_buf).to(dev)
    t_all = torch.empty(world * t_local.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(t_all, t_local)          # the single collective of the path
    allbuf = t_all.cpu().numpy().reshape(world, -1)
    outs = []
    for r in range(world):
        rlo, rhi = shard_bounds(F, r, world)
        q, e, n, m = unpack_results(allbuf[r], nb_max, K)
        k = (rhi - rlo) * Pn
        outs.append((q[:k], e[:k], n[:k], m[:k]))
    Qf = np.concatenate([o[0] for o in outs]).reshape(F, Pn, K, 3)
    ef = np.concatenate([o[1] for o in outs]).reshape(F, Pn, K)
    nf = np.concatenate([o[2] for o in outs]).reshape(F, Pn, K)
    mf = np.concatenate([o[3] for o in outs]).reshape(F, Pn, K)
    return Qf, ef, nf, mf


def agree_max(value, error=None):
    """max of an integer over the ranks, and every rank raises when any of them hit an error (so that a rank
    that cannot read its share of the files does not leave the others waiting in a collective)."""
    rank, world = dist_info()
    if world == 1:
        if error is not None:
            raise error
        return value
    import torch
    import torch.distributed as dist
    backend = dist.get_backend()
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    t = torch.tensor([int(value), 1 if error is not None else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[1].item()):
        if error is not None:
            raise error
        raise RuntimeError('another rank failed while reading its share of the pose files')
    return int(t[0].item())


def gather_lists(*lists):
    """Concatenate per-rank Python lists in rank order on every rank (statistics for the recap lines)."""
    rank, world = dist_info()
    if world == 1:
        return lists if len(lists) > 1 else lists[0]
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, [list(x) for x in lists])
    out = tuple([v for part in parts for v in part[i]] for i in range(len(lists)))
    return out if len(lists) > 1 else out[0]


def _pad(a, n):
    if a.shape[0] == n:
        return a
    out = np.zeros((n,) + a.shape[1:], dtype=a.dtype)
    out[:a.shape[0]] = a
    return out
