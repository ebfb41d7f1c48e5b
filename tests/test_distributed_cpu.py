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


def _trial_worker(rank, world, port, root, trial, multi):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import torch.distributed as dist
    import e2e_common as ec
    from pose2sim_amd import poseio, triangulation
    from test_e2e_trc import OracleEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    os.chdir(root)
    parsed = []
    orig = poseio.load_observations

    def spy(root_, dirs, maps, f_range, *a, **k):
        parsed.append(tuple(f_range))
        return orig(root_, dirs, maps, f_range, *a, **k)
    poseio.load_observations = spy
    triangulation._make_engine = lambda: OracleEngine()
    cfg = ec.base_config(trial, multi)
    cfg['project']['project_dir'] = trial
    import logging
    logging.basicConfig(filename=os.path.join(root, f'log{rank}.txt'), level=logging.INFO, format='%(message)s', force=True)
    paths = triangulation.triangulate_all(cfg)
    logging.shutdown()
    with open(os.path.join(root, f'rank{rank}.txt'), 'w') as fh:
        fh.write(repr(parsed) + '\n' + repr([os.path.basename(p) for p in paths if p]))
    dist.destroy_process_group()


@pytest.mark.parametrize('multi', [False, True])
def test_two_ranks_read_only_their_frames_and_write_the_same_trc(tmp_path, multi):
    """triangulate_all under two gloo ranks: each rank parses only its block of frames (the ingest is sharded, not
    just the kernels), the person count of multi-person mode is agreed with one all-reduce, and the .trc written
    by rank 0 equals the single-process file byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import e2e_common as ec
    from pose2sim_amd import skeletons, synth, triangulation
    from test_e2e_trc import OracleEngine
    ids, names, swap = skeletons.keypoints('HALPE_26')
    F = 23
    wl = synth.make_config(F, 4, len(ids), 2 if multi else 1, seed=41, p_missing_cam=0.0)
    sub = 'pose-associated' if multi else 'pose'
    outs = {}
    for mode in ('single', 'two'):
        root = str(tmp_path / mode)
        trial = ec.write_trial(root, 'trial', wl['cams'], ec.people_from_xyl(wl['xyl'], ids, 26), json_subdir=sub)
        if mode == 'single':
            cwd = os.getcwd()
            os.chdir(root)
            import logging
            handler = logging.FileHandler(os.path.join(root, 'log0.txt'))
            handler.setFormatter(logging.Formatter('%(message)s'))
            level = logging.getLogger().level
            logging.getLogger().addHandler(handler)
            logging.getLogger().setLevel(logging.INFO)
            try:
                orig = triangulation._make_engine
                triangulation._make_engine = lambda: OracleEngine()
                triangulation.triangulate_all(ec.base_config(trial, multi))
            finally:
                triangulation._make_engine = orig
                os.chdir(cwd)
                logging.getLogger().removeHandler(handler)
                logging.getLogger().setLevel(level)
                handler.close()
        else:
            mp.spawn(_trial_worker, args=(2, _free_port(), root, trial, multi), nprocs=2, join=True)
        d = os.path.join(trial, 'pose-3d')
        outs[mode] = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f.endswith('.trc')}
    assert outs['single'] and outs['single'] == outs['two']
    # the report (triangulation.py:255-360): rank 0 of the two prints what the single process prints -- its per-keypoint
    # means come from column sums reduced over the ranks instead of the whole tables -- and rank 1 prints none of it
    report = {mode: open(os.path.join(str(tmp_path / mode), 'log0.txt')).read().replace(str(tmp_path / mode), '') for mode in outs}
    assert 'Mean reprojection error for' in report['single'] and 'excluded' in report['single']
    assert report['single'] == report['two']
    assert 'Mean reprojection error' not in open(os.path.join(str(tmp_path / 'two'), 'log1.txt')).read()
    r0 = open(os.path.join(str(tmp_path / 'two'), 'rank0.txt')).read().split('\n')
    r1 = open(os.path.join(str(tmp_path / 'two'), 'rank1.txt')).read().split('\n')
    assert r0[0] == repr([(0, 12)]) and r1[0] == repr([(12, 23)])          # each rank parsed only its own block


def _assoc_worker(rank, world, port, root, trial, multi, thr, min_cams):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import torch.distributed as dist
    import e2e_common as ec
    from pose2sim_amd import personAssociation as pa
    from test_e2e_assoc import OracleAssocEngine, OracleSingleEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    os.chdir(root)
    pa._make_engine = (lambda: OracleAssocEngine()) if multi else (lambda: OracleSingleEngine())
    cfg = ec.base_config(trial, multi, min_cameras_for_triangulation=min_cams)
    cfg['personAssociation']['single_person']['reproj_error_threshold_association'] = thr
    pa.associate_all(cfg)
    dist.destroy_process_group()


@pytest.mark.parametrize('multi', [True, False])
def test_two_ranks_associate_their_own_frames(golden_dir, tmp_path, multi):
    """associate_all under two gloo ranks: every rank handles its block of frames and writes its own files; together
    they are exactly the files the reference wrote (goldens of test_e2e_assoc.py)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import e2e_common as ec
    if multi:
        z = np.load(os.path.join(golden_dir, 'e2e_assoc.npz'))
        pre, thr, min_cams = '', 20.0, 2
    else:
        z = np.load(os.path.join(golden_dir, 'e2e_single.npz'))
        pre = 's4_'
        thr, min_cams = float(z['s4_thr']), int(z['s4_min_cams'])
    cams = ec.cams_from_arrays(z, prefix=pre)
    n_persons = z[pre + 'n_persons']
    F, C = n_persons.shape
    frames, row = [], 0
    for f in range(F):
        per_cam = []
        for c in range(C):
            if multi and z['missing'][f, c]:
                per_cam.append(None)
                continue
            n = int(n_persons[f, c])
            per_cam.append([z[pre + 'kpts'][row + i].ravel() for i in range(n)])
            row += n
        frames.append(per_cam)
    root = str(tmp_path / 'two')
    trial = ec.write_trial(root, 'trial_assoc' if multi else 'trial_s4', cams, frames, json_subdir='pose')
    mp.spawn(_assoc_worker, args=(2, _free_port(), root, trial, multi, thr, min_cams), nprocs=2, join=True)
    got = {}
    d = os.path.join(trial, 'pose-associated')
    for cam in sorted(os.listdir(d)):
        for fn in sorted(os.listdir(os.path.join(d, cam))):
            got[f'{cam}/{fn}'] = open(os.path.join(d, cam, fn)).read()
    want = {str(n): str(t) for n, t in zip(z[pre + 'names'], z[pre + 'texts'])}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k


def test_collectives_use_the_local_rank_gpu(monkeypatch):
    """With RCCL the operands of every collective live on cuda:LOCAL_RANK -- the device the engine takes -- and not on
    torch's current device, which is cuda:0 in every rank unless somebody called set_device."""
    import torch
    import torch.distributed as dist
    from pose2sim_amd import parallel
    monkeypatch.setattr(dist, 'get_backend', lambda *a, **k: 'nccl')
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert parallel.collective_device() == torch.device('cuda', 3)
    monkeypatch.setattr(dist, 'get_backend', lambda *a, **k: 'gloo')
    assert parallel.collective_device() == torch.device('cpu')


def test_ranks_without_a_process_group_are_refused(monkeypatch):
    """WORLD_SIZE > 1 with no initialised process group would make every rank take every frame and write the same
    files: dist_info refuses."""
    from pose2sim_amd import parallel
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(RuntimeError, match='not initialised'):
        parallel.dist_info()
    monkeypatch.setenv('WORLD_SIZE', '1')
    assert parallel.dist_info() == (0, 1)


def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pose2sim_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        parallel.agree_ok(ValueError('rank 1 cannot read its files') if rank == 1 else None)
        outcome = 'no error'
    except ValueError as e:
        outcome = f'own: {e}'
    except RuntimeError as e:
        outcome = f'other: {e}'
    with open(os.path.join(out_dir, f'agree{rank}.txt'), 'w') as fh:
        fh.write(outcome)
    dist.destroy_process_group()


def test_a_failing_rank_makes_every_rank_raise(tmp_path):
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / 'agree1.txt').read().startswith('own: rank 1 cannot read')
    assert open(tmp_path / 'agree0.txt').read().startswith('other: another rank failed')


def _sums_worker(rank, world, port, out_dir, F, Pn, K, C):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from pose2sim_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    Q, err, nex, mask, ids, sections = _sums_case(F, Pn, K, C)
    lo, hi = parallel.shard_bounds(F, rank, world)
    Qf, means, tables = parallel.gather_trajectory((Q[lo:hi], err[lo:hi], nex[lo:hi], mask[lo:hi]), F, Pn, K, skipna=True, host_copy_on=0)
    sums = parallel.reduce_report_sums(tables, ids, sections, C)
    np.savez(os.path.join(out_dir, f'sums{rank}.npz'), means=means, has_q=Qf is not None, Q=Qf if Qf is not None else np.zeros(0), **sums)
    dist.destroy_process_group()


def _sums_case(F, Pn, K, C):
    rng = np.random.default_rng(5)
    Q = rng.normal(size=(F, Pn, K, 3))
    Q[rng.random((F, Pn, K)) < 0.1] = np.nan
    err = rng.uniform(0, 20, size=(F, Pn, K)).astype(np.float32)
    err[np.isnan(Q[..., 0])] = np.nan
    err[3] = np.nan                                                     # a frame without any error
    nex = rng.integers(0, C, size=(F, Pn, K)).astype(np.uint8)
    mask = rng.integers(0, 1 << C, size=(F, Pn, K)).astype(np.uint32)
    ids = np.stack([rng.permutation(Pn) for _ in range(F)])
    ids[rng.random((F, Pn)) < 0.15] = -1                                # slots that took no detection
    sections = np.array([[2, F - 3], [0, F], [5, 5]][:Pn])
    return Q, err, nex, mask, ids, sections


def test_report_sums_match_the_whole_tables(tmp_path):
    """gather_trajectory + reduce_report_sums under two gloo ranks against the one-process arithmetic on the whole
    tables: the points reassembled bit for bit on rank 0 only, the per-frame means on every rank, and the column sums
    over each person's kept frames in tracked order (empty slots: error NaN, every camera excluded)."""
    from pose2sim_amd import postproc, triangulation
    F, Pn, K, C = 29, 3, 7, 5
    mp.spawn(_sums_worker, args=(2, _free_port(), str(tmp_path), F, Pn, K, C), nprocs=2, join=True)
    Q, err, nex, mask, ids, sections = _sums_case(F, Pn, K, C)
    z0, z1 = np.load(tmp_path / 'sums0.npz'), np.load(tmp_path / 'sums1.npz')
    assert bool(z0['has_q']) and not bool(z1['has_q'])
    assert np.array_equal(z0['Q'], Q, equal_nan=True)
    want_means = np.stack([postproc.frame_means(err.reshape(-1, K).astype(np.float64)).reshape(F, Pn),
                           postproc.frame_means(nex.reshape(-1, K).astype(np.float64)).reshape(F, Pn)], axis=-1)
    for z in (z0, z1):
        assert np.array_equal(z['means'], want_means, equal_nan=True)
    e_rows = triangulation.in_tracked_order(err.astype(np.float64), ids, np.nan)
    n_rows = triangulation.in_tracked_order(nex.astype(np.float64), ids, float(C))
    m_rows = triangulation.in_tracked_order(mask, ids, np.uint32((1 << C) - 1))
    for n in range(Pn):
        a, b = sections[n]
        for z in (z0, z1):
            assert z['frames'][n] == b - a
            if b > a:
                cnt = z['err_count'][n]
                assert np.array_equal(cnt, (~np.isnan(e_rows[a:b, n])).sum(axis=0))
                np.testing.assert_allclose(z['err_sum'][n] / np.where(cnt > 0, cnt, np.nan), postproc.column_means(e_rows[a:b, n]), rtol=1e-13)
                np.testing.assert_allclose(z['excl_sum'][n] / (b - a), postproc.column_means(n_rows[a:b, n]), rtol=1e-13)
                frac = postproc.camera_exclusion_fractions(m_rows[a:b, n], C)
                assert {c: int(v) / ((b - a) * K) for c, v in enumerate(z['cam_count'][n])} == frac
            else:
                assert not z['err_sum'][n].any() and not z['cam_count'][n].any()
