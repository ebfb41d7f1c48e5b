"""Frame sharding across the GPUs of one node.

Every (frame, person, keypoint) unit is independent, so rank r of G takes the contiguous frame
block [r*F/G, (r+1)*F/G); the only large exchange is ONE all-gather of the points (24 bytes per unit,
gather_trajectory) -- RCCL over xGMI when the process group is 'nccl', gloo on CPU in tests -- beside three small
ones: the per-frame means that pick the kept frames, rank 0's verdict on them, and the column sums of the report.
The sequential post-processing (tracking, interpolation, .trc) runs on rank 0, which writes the files.
(gather_results, the all-gather of all 33 bytes per unit, remains for callers that want every table on every rank.)
"""
import os

import numpy as np


def dist_info():
    """(rank, world) of the default process group, (0, 1) when torch.distributed is not in use.
    A process launched as one of several ranks (WORLD_SIZE > 1) that has not initialised the process group would
    take every frame and write the same files as its siblings: that is refused."""
    import sys
    if 'torch' not in sys.modules and 'WORLD_SIZE' not in os.environ:
        return 0, 1                  # single process that never touched torch: do not pay its import (~2-5 s)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    if int(os.environ.get('WORLD_SIZE', '1') or 1) > 1:
        raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: call "
                           "torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', LOCAL_RANK)) "
                           "before the Pose2Sim stage, or unset WORLD_SIZE for a single-process run")
    return 0, 1


def collective_device():
    """Where this rank's collective operands live: its own GPU for RCCL (backend 'nccl') -- cuda:LOCAL_RANK, the same
    source the HIP engine takes its device from (triangulation._make_engine), NOT torch's current device, which is
    cuda:0 in every rank unless somebody called set_device -- and the host for gloo."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    return torch.device('cpu')


def shard_bounds(n_frames, rank, world):
    """Contiguous frame block of `rank`: sizes differ by at most one, earlier ranks get the extra."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def section_offsets(n_units):
    """Byte offsets of the four sections of a packed result buffer [Q f64 x3 | err f32 | mask u32 | n_excl u8] for
    n_units units, each section starting on 16 bytes (the kernels store 16 bytes per lane), and the total size."""
    a16 = lambda v: (v + 15) // 16 * 16                                # noqa: E731
    off_e = a16(n_units * 24)
    off_m = a16(off_e + n_units * 4)
    off_n = a16(off_m + n_units * 4)
    return off_e, off_m, off_n, a16(off_n + n_units)


def pack_results(Q, err, nex, mask):
    """[n][K] results -> one contiguous uint8 buffer (layout: section_offsets)."""
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    n = Q.size // 3
    off_e, off_m, off_n, total = section_offsets(n)
    buf = np.zeros(total, dtype=np.uint8)
    buf[:n * 24] = Q.view(np.uint8).ravel()
    buf[off_e:off_e + n * 4] = np.ascontiguousarray(err, dtype=np.float32).view(np.uint8).ravel()
    buf[off_m:off_m + n * 4] = np.ascontiguousarray(mask, dtype=np.uint32).view(np.uint8).ravel()
    buf[off_n:off_n + n] = np.ascontiguousarray(nex, dtype=np.uint8).ravel()
    return buf


def unpack_results(buf, n_blocks, K):
    n = n_blocks * K
    off_e, off_m, off_n, _ = section_offsets(n)
    Q = buf[:n * 24].view(np.float64).reshape(n_blocks, K, 3)
    err = buf[off_e:off_e + n * 4].view(np.float32).reshape(n_blocks, K)
    mask = buf[off_m:off_m + n * 4].view(np.uint32).reshape(n_blocks, K)
    nex = buf[off_n:off_n + n].reshape(n_blocks, K)
    return Q, err, nex, mask


class PackedDeviceResults:
    """Results of one rank's frame block left on its GPU: `buf` is a uint8 torch tensor in the packed layout
    (section_offsets) for `n_blocks_padded` blocks of K units, of which the first `n_blocks` are this rank's."""

    def __init__(self, buf, n_blocks, n_blocks_padded, K):
        self.buf, self.n_blocks, self.n_blocks_padded, self.K = buf, n_blocks, n_blocks_padded, K


def largest_shard(F, world):
    lo, hi = shard_bounds(F, 0, world)
    return hi - lo


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


def gather_results(local, F, Pn, K, host_copy_on=None):
    """The single collective of the path: every rank contributes the results of its contiguous frame block
    (``shard_bounds``) and receives the whole trajectory.

    local: (Q [n][Pn][K][3], err, n_excl, mask) host arrays, or a PackedDeviceResults (Engine.triangulate_packed):
    then the all-gather runs on the device buffers as they are -- no copy to the host, NumPy pack and copy back --
    and the gathered buffer comes to the host once.  host_copy_on: None = every rank returns the trajectory; a rank
    number = only that rank copies it to the host and the others return None (the sequential post-processing and
    the file writes are that rank's alone)."""
    rank, world = dist_info()
    device_form = isinstance(local, PackedDeviceResults)
    if world == 1 and not os.environ.get('P2S_FORCE_COLLECTIVE') and not device_form:
        return local
    import torch
    import torch.distributed as dist
    nb_max = largest_shard(F, world) * Pn
    if device_form:
        assert local.n_blocks_padded == nb_max and local.K == K
        t_local = local.buf
    else:
        Q, err, nex, mask = local
        lo, hi = shard_bounds(F, rank, world)
        nb = (hi - lo) * Pn
        # sections are sized for the largest block: pad the local block to it
        local_buf = pack_results(_pad(np.asarray(Q).reshape(nb, K, 3), nb_max), _pad(np.asarray(err).reshape(nb, K), nb_max),
                                 _pad(np.asarray(nex).reshape(nb, K), nb_max), _pad(np.asarray(mask).reshape(nb, K), nb_max))
        t_local = torch.from_numpy(local_buf).to(collective_device())
    if dist.is_available() and dist.is_initialized():
        t_all = torch.empty(world * t_local.numel(), dtype=torch.uint8, device=t_local.device)
        dist.all_gather_into_tensor(t_all, t_local)      # the single collective of the path
    else:
        t_all = t_local                                  # one process, device form: nothing to exchange
    if host_copy_on is not None and rank != host_copy_on:
        return None
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


class LocalTables:
    """What a rank keeps of its own frame block after the trajectory exchange: err f32 / n_excl u8 / mask u32, each
    [frames][Pn][K], and the first frame of the block."""

    def __init__(self, lo, err, n_excl, mask):
        self.lo, self.err, self.n_excl, self.mask = lo, err, n_excl, mask


def gather_trajectory(local, F, Pn, K, skipna, host_copy_on=None):
    """The path's exchange in its lean form: what the sequential steps need of every frame is the points (24 bytes per
    unit) and, per (frame, person), the mean error and mean number of excluded cameras that decide which frames a
    trial keeps (triangulation.py:897-901) -- 16 bytes per K units.  The per-unit errors, exclusion counts and camera
    masks (9 bytes per unit) stay on the rank that computed them; the report's column means over the kept frames are
    summed per rank and reduced (reduce_report_sums).

    local: (Q, err, n_excl, mask) host arrays of this rank's block or a PackedDeviceResults.  Returns
    (Q [F][Pn][K][3] -- None on the ranks other than host_copy_on --, row_means [F][Pn][2] on every rank, LocalTables).
    """
    from . import postproc
    rank, world = dist_info()
    import torch
    import torch.distributed as dist
    nb_max = largest_shard(F, world) * Pn
    lo, hi = shard_bounds(F, rank, world)
    nb = (hi - lo) * Pn
    if isinstance(local, PackedDeviceResults):
        assert local.n_blocks_padded == nb_max and local.K == K and local.n_blocks == nb
        off_e, off_m, off_n, total = section_offsets(nb_max * K)
        t_q = local.buf[:nb_max * K * 24]                                   # the points' section, as the kernels wrote it
        tail = local.buf[off_e:total].cpu().numpy()                         # 9 of the 33 bytes per unit: to this rank's host only
        n = nb_max * K
        err = tail[:n * 4].view(np.float32).reshape(nb_max, K)[:nb]
        mask = tail[off_m - off_e:off_m - off_e + n * 4].view(np.uint32).reshape(nb_max, K)[:nb]
        nex = tail[off_n - off_e:off_n - off_e + n].reshape(nb_max, K)[:nb]
    else:
        Q, err, nex, mask = local
        err = np.asarray(err, dtype=np.float32).reshape(nb, K)
        nex = np.asarray(nex, dtype=np.uint8).reshape(nb, K)
        mask = np.asarray(mask, dtype=np.uint32).reshape(nb, K)
        q = _pad(np.ascontiguousarray(Q, dtype=np.float64).reshape(nb, K * 3), nb_max)
        t_q = torch.from_numpy(q.view(np.uint8).reshape(-1)).to(collective_device())
    means = np.full((nb_max, 2), np.nan)
    if nb:
        means[:nb, 0] = postproc.frame_means(err.astype(np.float64), skipna=skipna)
        means[:nb, 1] = postproc.frame_means(nex.astype(np.float64))
    t_m = torch.from_numpy(means.reshape(-1)).to(t_q.device)
    t_q_all = torch.empty(world * t_q.numel(), dtype=torch.uint8, device=t_q.device)
    t_m_all = torch.empty(world * t_m.numel(), dtype=torch.float64, device=t_q.device)
    dist.all_gather_into_tensor(t_q_all, t_q)                               # the one large collective of the path
    dist.all_gather_into_tensor(t_m_all, t_m)
    m_all = t_m_all.cpu().numpy().reshape(world, nb_max, 2)
    sizes = [(shard_bounds(F, r, world)[1] - shard_bounds(F, r, world)[0]) * Pn for r in range(world)]
    row_means = np.concatenate([m_all[r, :sizes[r]] for r in range(world)]).reshape(F, Pn, 2)
    tables = LocalTables(lo, err.reshape(hi - lo, Pn, K), nex.reshape(hi - lo, Pn, K), mask.reshape(hi - lo, Pn, K))
    if host_copy_on is not None and rank != host_copy_on:
        return None, row_means, tables
    q_all = t_q_all.cpu().numpy().reshape(world, nb_max, K * 24)
    Qf = np.concatenate([q_all[r, :sizes[r]] for r in range(world)]).reshape(-1).view(np.float64).reshape(F, Pn, K, 3)
    return Qf, row_means, tables


def broadcast_ints(values, n, src=0):
    """An int64 vector of n entries from rank src to every rank (the sections kept and the person order per frame)."""
    import torch
    import torch.distributed as dist
    t = torch.zeros(n, dtype=torch.int64, device=collective_device())
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(values, dtype=np.int64).reshape(-1)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def reduce_report_sums(tables, ids, sections, n_cams):
    """Column sums of the report (triangulation.py:312-360, 934-943) over the kept frames of every person, from the
    rank's own tables and summed over the ranks: per person and keypoint the sum and count of the errors that are
    numbers and the sum of the exclusion counts, per person the number of frames and, per camera, of units that
    excluded it.  ids [F][Pn]: the detection that person slot n takes in frame f (-1: none, whose row reads error NaN,
    every camera excluded); sections [Pn][2]: first and one past the last kept frame.  Returns a dict of arrays."""
    import torch
    import torch.distributed as dist
    nF, Pn, K = tables.err.shape
    allmask = np.uint32((1 << n_cams) - 1) if n_cams < 32 else np.uint32(0xFFFFFFFF)
    out = np.zeros((Pn, 3 * K + 1 + n_cams))
    for n in range(Pn):
        a, b = max(int(sections[n][0]), tables.lo), min(int(sections[n][1]), tables.lo + nF)
        if b <= a:
            continue
        rows = np.arange(a - tables.lo, b - tables.lo)
        d = ids[a:b, n]
        took = d >= 0
        dd = np.where(took, d, 0)
        e = np.where(took[:, None], tables.err[rows, dd].astype(np.float64), np.nan)
        x = np.where(took[:, None], tables.n_excl[rows, dd].astype(np.float64), float(n_cams))
        m = np.where(took[:, None], tables.mask[rows, dd], allmask)
        ok = ~np.isnan(e)
        out[n, :K] = np.where(ok, e, 0.0).sum(axis=0)
        out[n, K:2 * K] = ok.sum(axis=0)
        out[n, 2 * K:3 * K] = x.sum(axis=0)
        out[n, 3 * K] = b - a
        flat = m.reshape(-1)
        out[n, 3 * K + 1:] = [np.count_nonzero((flat >> np.uint32(c)) & np.uint32(1)) for c in range(n_cams)]
    t = torch.from_numpy(out).to(collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.cpu().numpy()
    return {'err_sum': out[:, :K], 'err_count': out[:, K:2 * K], 'excl_sum': out[:, 2 * K:3 * K], 'frames': out[:, 3 * K],
            'cam_count': out[:, 3 * K + 1:]}


def agree_ok(error=None):
    """Every rank calls this before a collective with the exception its own stage raised (or None): if any rank
    failed, all of them raise -- the failing one its own exception -- instead of one leaving and the others waiting
    in the collective until the backend's timeout."""
    agree_max(0, error)


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
    dev = collective_device()
    t = torch.tensor([int(value), 1 if error is not None else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[1].item()):
        if error is not None:
            raise error
        raise RuntimeError('another rank failed in its share of the stage (its own log has the exception)')
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
