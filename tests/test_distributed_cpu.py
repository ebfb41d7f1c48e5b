"""N > 1 path on CPU: two gloo ranks shard the frames, each computes its block (oracle-backed
test double standing in for the HIP engine) and ONE all-gather reassembles the trajectory.
The result on every rank must equal the single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, F, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import tri_oracle
    from pose2sim_amd import parallel, skeletons, synth
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _, _, swap = skeletons.keypoints('HALPE_26')
    wl = synth.make_config(F, 4, 26, 2, seed=31, p_lowlik=0.08, p_outlier=0.06)
    calls = []

    def compute(x):
        calls.append(x.shape[0])
        Q, e, n, m = tri_oracle.triangulate_batch(x, wl['P'], None, swap, 0.3, 15.0, 2)
        return Q, e.astype(np.float32), n.astype(np.uint8), m
    Q, e, n, m = parallel.sharded_triangulate(compute, wl['xyl'])
    lo, hi = parallel.shard_bounds(F, rank, world)
    assert calls == [hi - lo]                        # each rank computed only its own frames
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), Q=Q, e=e, n=n, m=m)
    dist.destroy_process_group()


@pytest.mark.parametrize('F', [21, 8])
def test_two_ranks_reassemble_the_trajectory(tmp_path, F):
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), F, str(tmp_path)), nprocs=world, join=True)
    _, _, swap = skeletons.keypoints('HALPE_26')
    wl = synth.make_config(F, 4, 26, 2, seed=31, p_lowlik=0.08, p_outlier=0.06)
    Q, e, n, m = tri_oracle.triangulate_batch(wl['xyl'], wl['P'], None, swap, 0.3, 15.0, 2)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))
        assert z['Q'].shape == Q.shape
        assert np.array_equal(z['Q'], Q, equal_nan=True)
        assert np.array_equal(z['e'], e.astype(np.float32), equal_nan=True)
        assert np.array_equal(z['n'], n.astype(np.uint8)) and np.array_equal(z['m'], m)


def test_shard_bounds_cover_every_frame_once():
    from pose2sim_amd import parallel
    for F in (0, 1, 7, 8, 100_001):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(F, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == F
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
