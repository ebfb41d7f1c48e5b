#!/usr/bin/env python
"""Headline benchmark: keypoint-triangulations/s of the fused HIP triangulation path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg4|cfg5|cfg2_clean]

A "step" is one pass of the hot path (p2s_triangulate_device: stage + undistort/mask + weighted
DLT + camera-subset search) over one synthetic batch that is already resident in HBM.  With
N > 1 ranks (torchrun, one per GPU) every rank holds its own frame shard of the same size (weak
scaling) and a step ends with the all-gather of the per-unit points (24 B per unit; and of the
per-frame means, 16 B per frame and person) over RCCL/xGMI that BASELINE.json's north_star names.  Rank 0 prints ONE JSON line.

The roofline leg times the kernel alone with HIP events on the launch stream; the cpu_baseline
leg times the CPU oracle (oracle/, a loop-faithful port of the reference) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    'cfg2': dict(workload='synthetic 8-cam x HALPE_26 x 100k frames, single person (BASELINE configs[1])',
                 F=100_000, C=8, model='HALPE_26', Pn=1, thr=15.0, lik=0.3, min_cams=2,
                 undistort=False, lr_swap=False, seed=2),
    'cfg2_clean': dict(workload='cfg2 without outliers / low-confidence / missing observations',
                       F=100_000, C=8, model='HALPE_26', Pn=1, thr=15.0, lik=0.3, min_cams=2,
                       undistort=False, lr_swap=False, seed=2,
                       gen=dict(p_lowlik=0.0, p_outlier=0.0, p_missing_cam=0.0)),
    # BASELINE configs[2]: multi-person association (its own metric: frames/s)
    'cfg3': dict(workload='Demo_MultiPerson-style: 8 cams x 4 persons x HALPE_26 x 50k frames, epipolar association (BASELINE configs[2])',
                 F=50_000, C=8, model='HALPE_26', Pn=4, thr=15.0, lik=0.3, min_cams=2,
                 undistort=False, lr_swap=False, seed=3, assoc=True),
    # single-person association: the person of interest among 3 detections per camera
    'single': dict(workload='8 cams x 3 detections per camera x 50k frames, single-person association on the Neck keypoint (Demo_SinglePerson mode)',
                   F=50_000, C=8, model='HALPE_26', Pn=3, thr=20.0, lik=0.3, min_cams=2,
                   undistort=False, lr_swap=False, seed=6, single=True,
                   gen=dict(p_lowlik=0.02, p_outlier=0.02, p_missing_cam=0.0)),
    # per-GPU shard of BASELINE configs[3] (1M frames over 8 GPUs)
    'cfg4': dict(workload='synthetic 16-cam x COCO_133 (131 kpts) x 125k frames/GPU, min_cameras=3 (BASELINE configs[3] shard)',
                 F=125_000, C=16, model='COCO_133', Pn=1, thr=15.0, lik=0.3, min_cams=3,
                 undistort=False, lr_swap=False, seed=4),
    # per-GPU shard of BASELINE configs[4] (10M frames over 8 GPUs): 1.25 M frames = 12.5 GB of observations per GPU
    # the same at 1/10 length (quick runs of the search-bound 32-camera path)
    'cfg5_tenth': dict(workload='synthetic 32-cam x HALPE_26 x 125k frames/GPU, undistort + LR swap (BASELINE configs[4] shard, 1/10 length)',
                       F=125_000, C=32, model='HALPE_26', Pn=1, thr=15.0, lik=0.3, min_cams=2,
                       undistort=True, lr_swap=True, seed=5),
    'cfg5': dict(workload='synthetic 32-cam x HALPE_26 x 1.25M frames/GPU, undistort + LR swap (BASELINE configs[4] shard, full length)',
                 F=1_250_000, C=32, model='HALPE_26', Pn=1, thr=15.0, lik=0.3, min_cams=2,
                 undistort=True, lr_swap=True, seed=5),
}


def native_library_record():
    """Which C-ABI library this run loaded: its fingerprint, and the translation units this process compiled itself
    (none when the library travelled with the snapshot and was newer than its sources)."""
    import hashlib
    import __graft_entry__ as entry
    from pose2sim_amd import _lib
    path = os.environ.get('P2S_LIB') or entry.LIB
    rec = entry.BUILD_RECORD.get(entry.LIB, {})
    try:
        sha = hashlib.sha1(open(path, 'rb').read()).hexdigest()[:16]
    except OSError:
        sha = None
    return {'file': os.path.relpath(path, ROOT), 'sha1_16': sha, 'compiled_by_this_run': rec.get('recompiled', []) if rec else None,
            'abi_version': int(_lib.load().p2s_version())}


def make_workload(cfg, rank):
    from pose2sim_amd import skeletons, synth
    ids, names, swap = skeletons.keypoints(cfg['model'])
    K = len(ids)
    gen = dict(cfg.get('gen', {}))
    cams = synth.make_cameras(cfg['C'], seed=cfg['seed'], distort=cfg['undistort'])
    # frames are generated in chunks to bound host memory; every rank gets different frames
    F = cfg['F']
    chunks = []
    step = 25_000
    for f0 in range(0, F, step):
        n = min(step, F - f0)
        Q3d = synth.make_points3d(n, cfg['Pn'], K, seed=cfg['seed'] + 977 * rank + f0)
        chunks.append(synth.make_observations(Q3d, cams, seed=cfg['seed'] + 977 * rank + f0,
                                              distort=cfg['undistort'],
                                              p_lr_swap=0.02 if cfg['lr_swap'] else 0.0, swap_idx=swap, **gen))
    xyl = np.concatenate(chunks, axis=0)
    P = synth.projection_matrices(cams, cfg['undistort'])
    return xyl, cams, P, np.asarray(swap, dtype=np.int32), K


def make_association_inputs(xyl, seed, drop_missing=False):
    """Per (frame, camera): persons in random order, 2 % of them undetected.
    -> n_persons i32 [F][C], kpts f32 [rows][K][3] (camera-major, then person).
    drop_missing: a person a camera does not see at all (NaN everywhere) is left out of that camera's list instead of
    becoming an all-zero detection (which has affinity 1 with everybody: ties that rounding decides)."""
    rng = np.random.default_rng(seed + 5000)
    F, Pn, C, K, _ = xyl.shape
    keys = rng.random((F, C, Pn))
    keys[rng.random((F, C, Pn)) < 0.02] = 2.0                      # undetected -> sorted last, dropped
    if drop_missing:
        keys[np.isnan(xyl[..., 0]).all(axis=3).transpose(0, 2, 1)] = 2.0
    order = np.argsort(keys, axis=2)
    kept = np.take_along_axis(keys, order, axis=2) < 1.5
    per_cam = xyl.transpose(0, 2, 1, 3, 4)                         # [F][C][Pn][K][3]
    gathered = np.take_along_axis(per_cam, order[..., None, None], axis=2)
    gathered = np.nan_to_num(gathered, nan=0.0)                    # a pose estimator writes zeros, not NaN
    return kept.sum(axis=2).astype(np.int32), np.ascontiguousarray(gathered[kept])


def bench_association(args, cfg, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    from pose2sim_amd.engine import Engine, P2S_F32
    xyl, cams, P, swap, K = make_workload(cfg, rank)
    n_persons, kpts = make_association_inputs(xyl, cfg['seed'] + rank)
    F, C = n_persons.shape
    per_frame = n_persons.sum(axis=1, dtype=np.int64)
    offsets = np.zeros(F + 1, dtype=np.int64)
    np.cumsum(per_frame, out=offsets[1:])
    n_max = max(2, (int(per_frame.max()) + 1) & ~1)
    dev = torch.device('cuda', local_rank)
    eng = Engine(local_rank)
    eng.set_calibration(P, cams)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    prm = Engine.assoc_params(0.1, 0.2, cfg['min_cams'])
    d_np = torch.from_numpy(n_persons).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_kp = torch.from_numpy(kpts).to(dev)
    d_aff = torch.empty((F, n_max, n_max), dtype=torch.float64, device=dev)

    def step():
        eng.associate_device(F, K, n_max, P2S_F32, d_np, d_off, d_kp, prm, d_aff)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.assoc_stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    st = eng.assoc_stats()
    tflops = st['fp64_flops'] / dt / 1e12             # this rank's kernel; every rank runs the same workload shape
    out = {'metric': 'association-frames/sec', 'value': F * world * args.steps / dt, 'unit': 'frames/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': cfg['workload'], 'frames_per_gpu': F, 'cams': C, 'kpts_json': K,
                      'detections_per_frame_max': n_max, 'detections_per_frame_mean': float(per_frame.mean()),
                      'parallelism': f'frame shards x{world}'},
           'roofline': {'bound': 'fp64_valu', 'achieved': tflops, 'peak': FP64_PEAK / 1e12, 'unit': 'TFLOP/s',
                        'frac': tflops * 1e12 / FP64_PEAK, 'traffic': None, 'fp64_frac': tflops * 1e12 / FP64_PEAK,
                        'admm_passes_per_frame': st['admm_passes'] / max(st['frames'], 1),
                        'jacobi_sweeps_per_pass': st['jacobi_sweeps'] / max(st['admm_passes'], 1),
                        'note': 'LDS-resident fp64 Jacobi iteration, bound by vector instruction issue: achieved = fp64 operations '
                                'counted by the kernel (p2s_get_assoc_stats) / time, peak = fp64 vector peak; HBM traffic is ~10 KB per '
                                '~1e7 flop and not the bound (SURVEY 8d)'}}
    # ---- the rest of BASELINE configs[2] ("association then triangulation"): the triangulation of the 4 persons'
    # 5.2 M units (triangulation.py:831-865 over P = 4), and the stage with its host half -- affinity matrices back to the
    # host and proposal extraction (personAssociation.py:512-549: the per-detection argmax natively, the order-deciding
    # NumPy calls per frame) -- on a bounded sample of the frames
    from pose2sim_amd import personAssociation as pa_mod, skeletons, synth_device
    ids, names, swap_list = skeletons.keypoints(cfg['model'])
    Pn = cfg['Pn']
    n_blocks, n_units = F * Pn, F * Pn * K
    d_xyl = synth_device.make_observations_device(cams, F, Pn, K, seed=cfg['seed'] + 977 * rank, device=dev)
    tprm = Engine.tri_params(cfg['thr'], cfg['lik'], cfg['min_cams'], False, False)
    d_Q = torch.empty((n_units, 3), dtype=torch.float64, device=dev)
    d_e = torch.empty(n_units, dtype=torch.float32, device=dev)
    d_m = torch.empty(n_units, dtype=torch.int32, device=dev)
    d_x = torch.empty(n_units, dtype=torch.uint8, device=dev)
    d_swap = torch.from_numpy(np.asarray(swap_list, dtype=np.int32)).to(dev)

    def tri_step():
        eng.triangulate_device(n_blocks, K, P2S_F32, d_xyl, d_swap, tprm, d_Q.data_ptr(), d_e.data_ptr(), d_x.data_ptr(), d_m.data_ptr())
    for _ in range(3):
        tri_step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_tri = max(5, args.steps)
    ev0.record()
    for _ in range(n_tri):
        tri_step()
    ev1.record()
    torch.cuda.synchronize()
    tri_ms = ev0.elapsed_time(ev1) / n_tri
    tri_bytes = n_units * (12 * C + 32)
    out['triangulation_leg'] = {'units': n_units, 'kernel_ms': tri_ms, 'units_per_s': n_units / (tri_ms * 1e-3),
                                'roofline': {'bound': 'hbm', 'achieved': tri_bytes / (tri_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                             'frac': tri_bytes / (tri_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_unit': 12 * C + 32},
                                'note': 'p2s_triangulate_device on the 4 persons of every frame (device-generated observations of the same shape)'}
    assoc_ms = dt / args.steps * 1e3
    out['association_plus_triangulation'] = {'frames_per_s': F / ((assoc_ms + tri_ms) * 1e-3), 'ms_per_50k_frames': assoc_ms + tri_ms,
                                             'note': 'both kernels, inputs resident in HBM'}
    n_e2e = min(F, args.e2e_frames)
    t0 = time.perf_counter()
    eng.associate_device(n_e2e, K, n_max, P2S_F32, d_np, d_off, d_kp, prm, d_aff)
    torch.cuda.synchronize()
    aff_host = d_aff[:n_e2e].cpu().numpy()
    t1 = time.perf_counter()
    props = pa_mod.proposals_batch(aff_host, n_persons[:n_e2e], cfg['min_cams'])
    t2 = time.perf_counter()
    out['stage_with_host_half'] = {'frames': n_e2e, 'frames_per_s': n_e2e / (t2 - t0), 'kernel_and_copy_s': t1 - t0, 'proposals_s': t2 - t1,
                                   'mean_proposals_per_frame': float(np.mean([len(p) for p in props])),
                                   'note': 'association kernel + affinity matrices to the host + proposal extraction (argmax rows, distinct rows and '
                                           'filters native for all frames, np.argsort of the multiplicities as the reference calls it); JSON reading '
                                           'and rewriting not included (profiles/e2e_assoc_bench.py times the stage on files)'}
    if not args.no_cpu_baseline:
        from oracle import association_ref as ar
        cal = {'inv_K': cams['inv_K'], 'R_mat': cams['R_mat'], 'T': cams['T']}
        nfr = args.cpu_frames or 300
        t0 = time.perf_counter()
        row = 0
        for f in range(nfr):
            per_cam = []
            for c in range(C):
                per_cam.append([kpts[row + i].astype(np.float64).ravel() for i in range(n_persons[f, c])])
                row += n_persons[f, c]
            ar.associate_frame(per_cam, cal, 0.1, 0.2, cfg['min_cams'])
        cdt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': nfr / cdt, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                               'sample': f'first {nfr} frames of the same workload ({cdt:.1f} s), NumPy oracle incl. proposal extraction'}
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)


def bench_single(args, cfg, rank, world, local_rank):
    """Single-person association: every frame searches the 3^8 person combinations in the reference's order."""
    import torch
    import torch.distributed as dist
    from pose2sim_amd import skeletons
    from pose2sim_amd.engine import Engine, P2S_F32
    xyl, cams, P, swap, K = make_workload(cfg, rank)
    n_persons, kpts = make_association_inputs(xyl, cfg['seed'] + rank)
    ids, names, _ = skeletons.keypoints(cfg['model'])
    kid = names.index('Neck')                              # column of the tracked keypoint in the packed array
    tracked = np.ascontiguousarray(kpts[:, kid, :])
    F, C = n_persons.shape
    offsets = np.zeros(F + 1, dtype=np.int64)
    np.cumsum(n_persons.sum(axis=1, dtype=np.int64), out=offsets[1:])
    dev = torch.device('cuda', local_rank)
    eng = Engine(local_rank)
    eng.set_calibration(P, cams)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    d_np = torch.from_numpy(n_persons).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_tk = torch.from_numpy(tracked).to(dev)
    d_comb = torch.empty((F, C), dtype=torch.int32, device=dev)
    d_err = torch.empty((F,), dtype=torch.float64, device=dev)
    d_Q = torch.empty((F, 3), dtype=torch.float64, device=dev)

    def step():
        eng.associate_single_device(F, P2S_F32, d_np, d_off, d_tk, cfg['thr'], cfg['lik'], cfg['min_cams'], d_comb, d_err, d_Q)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    found = int(torch.isfinite(d_err).sum().item())
    out = {'metric': 'association-frames/sec', 'value': F * world * args.steps / dt, 'unit': 'frames/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': cfg['workload'], 'frames_per_gpu': F, 'cams': C,
                      'combinations_per_frame_max': int(np.prod(np.maximum(n_persons, 1).astype(np.float64), axis=1).max()),
                      'frames_with_a_solution': found, 'parallelism': f'frame shards x{world}'},
           'roofline': {'bound': 'mfma', 'achieved': None, 'peak': None, 'unit': 'TFLOP/s', 'frac': None, 'traffic': None,
                        'note': 'fp64-VALU-bound search on 24 B per detection; HBM and MFMA fractions are not meaningful'}}
    if not args.no_cpu_baseline:
        from oracle import association_single_ref as sr
        nfr = args.cpu_frames or 10
        t0 = time.perf_counter()
        row = 0
        for f in range(nfr):
            per_cam = []
            for c in range(C):
                per_cam.append([list(tracked[row + i].astype(np.float64)) for i in range(n_persons[f, c])])
                row += n_persons[f, c]
            sr.best_persons_and_cameras(per_cam, sr.persons_combinations(n_persons[f]), [np.asarray(p) for p in P], 0,
                                        cfg['thr'], cfg['min_cams'], cfg['lik'])
        cdt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': nfr / cdt, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                               'sample': f'first {nfr} frames of the same workload ({cdt:.1f} s), NumPy oracle'}
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)


def host_cores():
    """CPUs this process may really use: its affinity mask, cut down to the cgroup's CPU quota where there is one (a
    one-GPU box shares its host: 256 hardware threads in the mask, 16 CPUs of quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def _oracle_slice(args):
    """Worker of the process-parallel CPU baseline: the NumPy oracle on one block of frames."""
    xyl, P, cams, swap, lik, thr, min_cams, lr_swap, undistort = args
    from oracle import triangulation_ref as tr
    try:                                   # one BLAS thread per process: the pool is the parallel axis
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            tr.triangulate_batch(xyl, P, cams, swap, lik, thr, min_cams, lr_swap, undistort)
    except ImportError:
        tr.triangulate_batch(xyl, P, cams, swap, lik, thr, min_cams, lr_swap, undistort)
    return xyl.shape[0]


def host_sample(cfg, n_frames):
    """A bounded slice of the same workload from the host generator (the oracle needs host arrays)."""
    from pose2sim_amd import skeletons, synth
    ids, names, swap = skeletons.keypoints(cfg['model'])
    gen = dict(cfg.get('gen', {}))
    cams = synth.make_cameras(cfg['C'], seed=cfg['seed'], distort=cfg['undistort'])
    Q3d = synth.make_points3d(n_frames, cfg['Pn'], len(ids), seed=cfg['seed'])
    xyl = synth.make_observations(Q3d, cams, seed=cfg['seed'], distort=cfg['undistort'],
                                  p_lr_swap=0.02 if cfg['lr_swap'] else 0.0, swap_idx=swap, **gen)
    return xyl, cams, synth.projection_matrices(cams, cfg['undistort']), list(swap)


def cpu_baselines(cfg, frames_1core, frames_all):
    """the NumPy oracle (loop-faithful port of the reference, parity-checked against the goldens) on a
    bounded slice of the workload, arithmetic only: single process on one core, and process-parallel over frames on
    all the host cores this process may use.  Plus the same algorithm in C + OpenMP (oracle/tri_oracle.c) on all
    cores: what a compiled CPU implementation would reach.  Runs BEFORE anything touches the GPU (fork)."""
    import multiprocessing as mp
    from oracle import triangulation_ref as tr
    cores = host_cores()
    xyl, cams, P, swap = host_sample(cfg, max(frames_1core, frames_all))
    per_frame = xyl.shape[1] * xyl.shape[3]
    out = {}
    t0 = time.perf_counter()
    tr.triangulate_batch(xyl[:frames_1core], P, cams, swap, cfg['lik'], cfg['thr'], cfg['min_cams'], cfg['lr_swap'], cfg['undistort'])
    dt = time.perf_counter() - t0
    out['cpu_baseline_1core'] = {'value': frames_1core * per_frame / dt, 'unit': 'keypoint-triangulations/s', 'cores': 1, 'kind': 'port',
                                 'sample': f'first {frames_1core} frames of the same workload ({frames_1core * per_frame} units, {dt:.1f} s), NumPy oracle, arithmetic only'}
    workers = max(1, min(cores, frames_all, 64))              # 64 processes: beyond that the fork / pickle overhead of a ~10 s sample dominates
    bounds = np.linspace(0, frames_all, workers + 1).astype(int)
    jobs = [(xyl[a:b], P, cams, swap, cfg['lik'], cfg['thr'], cfg['min_cams'], cfg['lr_swap'], cfg['undistort'])
            for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    with mp.get_context('fork').Pool(workers) as pool:
        pool.map(_oracle_slice, [(j[0][:1],) + j[1:] for j in jobs])           # start-up and imports outside the timing
        t0 = time.perf_counter()
        pool.map(_oracle_slice, jobs, chunksize=1)
        dt = time.perf_counter() - t0
    out['cpu_baseline'] = {'value': frames_all * per_frame / dt, 'unit': 'keypoint-triangulations/s', 'cores': workers, 'kind': 'port',
                           'sample': f'first {frames_all} frames of the same workload ({frames_all * per_frame} units, {dt:.1f} s), NumPy oracle, '
                                     f'{workers} processes over frames, arithmetic only'}
    try:
        from oracle import tri_oracle
        n = frames_all * (50 if cfg['C'] <= 8 else 5)
        x, cams2, P2, swap2 = host_sample(cfg, n) if n > xyl.shape[0] else (xyl, cams, P, swap)
        x = np.ascontiguousarray(x[:n], dtype=np.float64)
        tri_oracle.triangulate_batch(x[:64], P2, cams2, swap2, cfg['lik'], cfg['thr'], cfg['min_cams'], cfg['lr_swap'], cfg['undistort'], threads=cores)
        t0 = time.perf_counter()
        tri_oracle.triangulate_batch(x, P2, cams2, swap2, cfg['lik'], cfg['thr'], cfg['min_cams'], cfg['lr_swap'], cfg['undistort'], threads=cores)
        dt = time.perf_counter() - t0
        out['cpu_baseline_native'] = {'value': n * per_frame / dt, 'unit': 'keypoint-triangulations/s', 'cores': cores, 'kind': 'port',
                                      'sample': f'first {n} frames of the same workload ({n * per_frame} units, {dt:.1f} s), C + OpenMP oracle, arithmetic only'}
    except Exception as e:                         # the C oracle is optional test infrastructure
        out['cpu_baseline_native'] = {'error': str(e)}
    return out


# fp64 operations of the path, counted from the kernels' arithmetic (an FMA = 2): the normal-matrix contribution of one
# camera (18 multiply / FMA for the two weighted rows, 20 FMA for the rank-2 update of the 10 entries), the eigen-solve
# (first pass ~95, later passes ~115 at 1.1 later passes on average), the reprojection distance of one camera
# (9 FMA projection, 4 FMA / multiply residuals, 2 for s z^2, reciprocal square root with one Newton step, sum).
FLOP_ACC_PER_CAM, FLOP_EIGEN, FLOP_ERR_PER_CAM = 66, 220, 36
FP64_PEAK = 78.6e12                                # MI355X fp64 vector (= matrix) peak: AMD's published spec (the guide has no fp64 row; a bare v_fma_f64 loop reaches 66.6, exp/mfma_valu_overlap.hip)


def fp64_flops_per_step(n_units, C, stats_per_step):
    level0 = n_units * (C * (FLOP_ACC_PER_CAM + FLOP_ERR_PER_CAM) + FLOP_EIGEN)
    # a subset evaluation: downdate of the removed cameras (1.2 on average over the levels that occur), eigen-solve,
    # error over all C cameras (masked)
    # the long levels drop hopeless candidates after a few cameras: their error work is counted by the kernels
    deep = stats_per_step.get('pruned_subsets', 0.0)
    evals = (stats_per_step['subsets_evaluated'] - deep) * (1.2 * FLOP_ACC_PER_CAM + FLOP_EIGEN + C * FLOP_ERR_PER_CAM)
    evals += deep * (1.2 * FLOP_ACC_PER_CAM + FLOP_EIGEN) + stats_per_step.get('pruned_camera_errors', 0.0) * FLOP_ERR_PER_CAM
    return level0 + evals


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run N ranks as children through torch.distributed.run (one per GPU,
    rendezvous on 127.0.0.1 at a free port) and return their exit code.  The parent imports neither torch nor the
    engine: a process that has initialised the GPU must never be replaced or forked into ranks."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """P2S_BENCH_DRY=1 (tests/test_bench_launch.py, no GPU): every rank joins a gloo group, the ranks are counted with
    one all-reduce and rank 0 prints a JSON line -- the launch path of `--gpus N` without the engine."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        dist.init_process_group('gloo')
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1
    if int(os.environ.get('RANK', '0')) == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'ranks_seen': seen, 'steps': args.steps, 'config': args.config}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 200 for the triangulation configs whose step is < 10 ms, 5 otherwise)')
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', default='cfg2', choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pool-singles', type=int, default=-1, help='kernel experiments: P2S_TUNE_POOL_SINGLES_PCT')
    ap.add_argument('--tri-path', default='auto', choices=['auto', 'worklist', 'onetile', 'twotiles', 'pooled'], help='kernel experiments (p2s_set_tuning)')
    ap.add_argument('--no-screen', action='store_true', help='kernel experiments: pooled kernel without its fp32 screen (P2S_TUNE_SCREEN 0)')
    ap.add_argument('--pool-tiles', type=int, default=0, help='kernel experiments: P2S_TUNE_POOL_TILES')
    ap.add_argument('--deep-min', type=int, default=-1, help='kernel experiments: P2S_TUNE_DEEP_MIN_SUBSETS')
    ap.add_argument('--preroll-ms', type=float, default=100.0,
                    help='run the step untimed for this long before the W warmup steps: an idle MI355X needs tens of ms of load '
                         'before it runs at its sustained clocks (a 20-step run read 13 %% slower per kernel without; 0 = off)')
    ap.add_argument('--cpu-frames', type=int, default=0, help='frames for the cpu_baseline sample (0 = auto)')
    ap.add_argument('--e2e-frames', type=int, default=10_000, help='cfg3: frames of the association stage timed with its host half')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as a plain `python bench.py --gpus N`: this process becomes the launcher (it has not touched the GPU
        # and never will) and the N ranks are fresh child processes; rank 0's JSON line goes to our stdout
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world_env}: start one rank per GPU '
                         f'(python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 '
                         f'bench.py --gpus {args.gpus} ...) or run plain `python bench.py --gpus {args.gpus}`, which does that itself')
    if os.environ.get('P2S_BENCH_DRY') == '1':
        return dry_run(args)
    cfg0 = CONFIGS[args.config]
    cpu = None
    if not args.no_cpu_baseline and world_env == 1 and not (cfg0.get('assoc') or cfg0.get('single')):
        # the CPU legs fork worker processes: before anything initialises the GPU in this process
        import __graft_entry__ as entry0
        entry0.build_oracle()
        # ~2.5 s of single-core work per process: 400 frames at 8 cameras (4.3e3 units/s/core), fewer where a unit is dearer
        f1 = args.cpu_frames or {8: 400, 16: 12, 32: 6}.get(cfg0['C'], 100)
        fall = args.cpu_frames or max(2000 if cfg0['C'] <= 8 else 0, f1 * min(host_cores(), 64))
        cpu = cpu_baselines(cfg0, f1, max(fall, f1))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = world_env
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
    # rehearsal on a one-GPU box: P2S_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device); the driver's multi-GPU runs never set it
    rehearsal = os.environ.get('P2S_BENCH_REHEARSAL') == '1'
    # P2S_BENCH_FORCE_COLLECTIVE=1 (tests): run the multi-rank code path -- process group, packed asynchronous
    # all-gather, max over ranks -- even with a single rank, so that the RCCL calls are exercised on one GPU
    multi = world > 1 or os.environ.get('P2S_BENCH_FORCE_COLLECTIVE') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    import __graft_entry__ as entry
    if rank == 0:
        entry.build_hip()
    if multi:
        dist.barrier()
    from pose2sim_amd.engine import Engine, P2S_F32

    cfg = CONFIGS[args.config]
    fast = args.config in ('cfg2', 'cfg2_clean', 'cfg4', 'single')
    if args.steps is None:
        args.steps = 200 if fast else 5
    if args.warmup is None:
        args.warmup = 10 if fast else 2
    if cfg.get('assoc') or cfg.get('single'):
        (bench_single if cfg.get('single') else bench_association)(args, cfg, rank, world, local_rank)
        if multi:
            dist.destroy_process_group()
        return
    # ---- triangulation configs -------------------------------------------------------------------------------------------
    from pose2sim_amd import skeletons, synth, synth_device
    ids, names, swap_list = skeletons.keypoints(cfg['model'])
    K, C, Pn, F = len(ids), cfg['C'], cfg['Pn'], cfg['F']
    swap = np.asarray(swap_list, dtype=np.int32)
    cams = synth.make_cameras(C, seed=cfg['seed'], distort=cfg['undistort'])
    P = synth.projection_matrices(cams, cfg['undistort'])
    n_blocks = F * Pn
    n_units = n_blocks * K

    dev = torch.device('cuda', local_rank)
    eng = Engine(local_rank)
    eng.set_calibration(P, cams)
    if args.pool_singles >= 0:
        eng.set_tuning(Engine.TUNE_POOL_SINGLES_PCT, args.pool_singles)
    if args.tri_path != 'auto':
        eng.set_tuning(Engine.TUNE_TRI_PATH, {'worklist': Engine.TRI_PATH_WORKLIST, 'onetile': Engine.TRI_PATH_ONE_TILE,
                                              'twotiles': Engine.TRI_PATH_TWO_TILES, 'pooled': Engine.TRI_PATH_POOLED}[args.tri_path])
    if args.no_screen:
        eng.set_tuning(Engine.TUNE_SCREEN, 0)
    if args.deep_min >= 0:
        eng.set_tuning(Engine.TUNE_DEEP_MIN_SUBSETS, args.deep_min)
    if args.pool_tiles:
        eng.set_tuning(Engine.TUNE_POOL_TILES, args.pool_tiles)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    prm = Engine.tri_params(cfg['thr'], cfg['lik'], cfg['min_cams'], cfg['undistort'], cfg['lr_swap'])

    # Inputs: generated ON THE DEVICE per rank from the seed, in NB distinct buffers that the steps rotate
    # over, so that no step can be served from the 256 MiB Infinity Cache by the previous one: NB x input >= 1 GiB and
    # NB >= 4 (one buffer when a single input already exceeds 4 GiB).
    in_bytes = n_blocks * C * K * 12
    n_buf = 1 if in_bytes >= (4 << 30) else max(4, -(-(1 << 30) // in_bytes))
    gen = dict(cfg.get('gen', {}))
    d_xyls = [synth_device.make_observations_device(cams, F, Pn, K, seed=cfg['seed'] + 977 * rank + 101 * i, device=dev,
                                                    distort=cfg['undistort'], p_lr_swap=0.02 if cfg['lr_swap'] else 0.0,
                                                    swap_idx=swap_list, **gen) for i in range(n_buf)]
    d_swap = torch.from_numpy(swap).to(dev)
    # packed result buffer [Q f64 x3 | err f32 | mask u32 | n_excl u8], 16-byte aligned sections.  What travels is what
    # parallel.gather_trajectory sends: the points' section (24 B per unit) and, in a second small all-gather, the
    # per-(frame, person) mean error and mean exclusion count (16 B per K units; the product computes them on the host
    # from the rank's own tables, which like every result at N = 1 stay where the kernel wrote them during the bench)
    from pose2sim_amd import parallel
    off_e, off_m, off_n, nbytes = parallel.section_offsets(n_units)
    q_bytes, mean_bytes = n_units * 24, n_blocks * 16
    # two result buffers: the all-gather of step i runs on RCCL's stream while step i+1 computes
    d_outs = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2 if multi else 1)]
    d_alls = [torch.empty(q_bytes * world, dtype=torch.uint8, device=dev) for _ in range(2)] if multi else []
    d_means = [torch.zeros(mean_bytes, dtype=torch.uint8, device=dev) for _ in range(2)] if multi else []
    d_mean_alls = [torch.empty(mean_bytes * world, dtype=torch.uint8, device=dev) for _ in range(2)] if multi else []
    d_out = d_outs[0]
    base = d_out.data_ptr()
    pending = [None, None]
    counter = [0]

    def step():
        i = counter[0] % len(d_outs)
        x = d_xyls[counter[0] % n_buf]
        counter[0] += 1
        if multi and pending[i] is not None:
            for w in pending[i]:
                w.wait()                           # buffer i is free again (its all-gathers have finished)
        b = d_outs[i].data_ptr()
        eng.triangulate_device(n_blocks, K, P2S_F32, x, d_swap, prm, b, b + off_e, b + off_n, b + off_m)
        if multi:
            pending[i] = (dist.all_gather_into_tensor(d_alls[i], d_outs[i][:q_bytes], async_op=True),
                          dist.all_gather_into_tensor(d_mean_alls[i], d_means[i], async_op=True))

    def drain():
        for ws in pending:
            for w in (ws or ()):
                w.wait()
        pending[0] = pending[1] = None

    kcount = [0]

    def kernel_only():
        x = d_xyls[kcount[0] % n_buf]
        kcount[0] += 1
        eng.triangulate_device(n_blocks, K, P2S_F32, x, d_swap, prm, base, base + off_e, base + off_n, base + off_m)

    # Pre-roll: the path is a steady stream of batches, and a GPU that has just been idle (the process start, the synthetic
    # data) runs its first ~50 ms below its sustained clocks: with --steps 20 --warmup 5 the kernel read 0.205 ms against
    # 0.181 ms after 50 ms, 200 ms or 1 s of the same steps (profiles/r03/preroll.log).  Untimed, reported in the line.
    if args.preroll_ms > 0:
        t_pre = time.perf_counter()
        more, batch = True, 1
        while more:
            t_b = time.perf_counter()
            for _ in range(batch):
                step()
            drain()
            torch.cuda.synchronize()
            if batch == 1 and not multi and time.perf_counter() - t_b < 2e-3:
                batch = 20                             # short steps: fewer synchronisations (a 1.4 s step stays alone; with
                                                       # several ranks every rank must run the same count: one at a time)
            more = (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms
            if multi:                                  # every rank runs the same number of steps (they hold collectives)
                flag = torch.tensor([1 if more else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = bool(int(flag.item()))
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                        # every all-gather of the K steps is inside the timed region
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline leg: the kernels alone over the same rotation of inputs, HIP events on the launch stream; the
    # context's counters give the subset evaluations of exactly these steps
    eng.tri_stats(reset=True)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(args.steps):
        kernel_only()
    ev1.record()
    torch.cuda.synchronize()
    k_ms = ev0.elapsed_time(ev1) / args.steps
    stats = {k: v / args.steps for k, v in eng.tri_stats(reset=True).items()}

    # sanity: the timed work produced results
    err = d_out[off_e:off_e + n_units * 4].view(torch.float32)
    ok_frac = float((~torch.isnan(err)).float().mean().item())

    if rank == 0:
        alg_bytes = n_units * (12 * C + 32)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        flops = fp64_flops_per_step(n_units, C, stats)
        # HBM bytes per step as counted by rocprofv3 (profiles/collect.sh): reported only while the kernel sources are the
        # ones the counters were taken on (profiles/summarize.py keeps their fingerprint beside the figure)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                import hashlib
                entry_t = json.load(open(tpath)).get(args.config)
                h = hashlib.sha1()
                for name in ('p2s_tri_pool.hip', 'p2s_tri_fused.hip', 'p2s_tri.hip', 'p2s_tri_deep.hip', 'p2s_tri_dev.h', 'p2s_internal.h', 'p2s_api.hip'):
                    h.update(open(os.path.join(ROOT, 'pose2sim_amd', 'csrc', name), 'rb').read())
                if isinstance(entry_t, dict) and entry_t.get('sources_sha1') == h.hexdigest() and args.tri_path == 'auto':
                    traffic, traffic_source = entry_t['bytes_per_step'], entry_t.get('source')
            except Exception:
                traffic = None
        fused = (not cfg['undistort']) and (not cfg['lr_swap']) and C <= 16
        out = {
            'metric': 'keypoint-triangulations/sec',
            'value': n_units * world * args.steps / dt,
            'unit': 'keypoint-triangulations/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'frames_per_gpu': F, 'cams': C, 'kpts': K, 'persons': Pn,
                       'units_per_gpu': n_units, 'input_dtype': 'f32',
                       'input_buffers_rotated': n_buf, 'input_bytes_per_buffer': in_bytes, 'generated': 'on device, per rank, from the seed',
                       'preroll_ms': args.preroll_ms,
                       'native_library': native_library_record(),
                       'params': {'thr_px': cfg['thr'], 'lik_thr': cfg['lik'], 'min_cams': cfg['min_cams'],
                                  'undistort': cfg['undistort'], 'lr_swap': cfg['lr_swap']},
                       'accepted_fraction': ok_frac,
                       'search': {'units_entering_search_per_step': stats['search_units'], 'subsets_evaluated_per_step': stats['subsets_evaluated'],
                                  'evaluation_passes_per_step': stats['passes'],
                                  'screened_subsets_per_step': stats['screened_subsets'], 'screen_passes_per_step': stats['screen_passes'],
                                  'capped_units_per_step': stats['capped_units']},
                       'parallelism': f'frame shards x{world}' + (' + all-gather of the points (24 B per unit) per step (async, overlapped with the next step)' if multi else '')},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'fp64_frac': flops / (k_ms * 1e-3) / FP64_PEAK, 'fp64_tflops': flops / (k_ms * 1e-3) / 1e12,
                         'fp64_flop_per_step': flops,
                         # the screen's single-precision work (not in fp64_frac): per subset looked at, the downdate of ~1.2
                         # cameras (84 each), three 3x3 solves and two Rayleigh quotients (~290), C reprojection distances (~40)
                         'screen_fp32_flop_per_step': stats.get('screened_subsets', 0.0) * (1.2 * 84 + 290 + 40 * C),
                         'kernel': ('p2s_tri_pool_kernel (streaming pass, failures of up to 5 tiles pooled, fp32 screen + fp64 evaluation of the surviving camera subsets, one launch)'
                                    if (fused and args.tri_path in ('auto', 'pooled')) else
                                    'p2s_tri_fused_kernel (round 2: streaming pass + in-wave fp64 subset search, one launch)'
                                    if (fused and args.tri_path != 'worklist') else
                                    'p2s_tri_level0_kernel + p2s_tri_search_kernel (one pass of the path)'),
                         'kernel_ms': k_ms, 'algorithmic_bytes_per_unit': 12 * C + 32, 'units_per_step': n_units,
                         # the one-launch kernels take the shard in chunks of < 2^31 bytes of observations (32-bit offsets)
                         'launches_per_step': (-(-n_blocks // max(16, ((1 << 31) // (C * K * 12)) // 16 * 16))) if fused else None},
        }
        if multi:
            # the split a reader needs to judge the scaling: the all-gather moves 24 B per unit (the points) and 16 B per
            # (frame, person) into every rank over xGMI (one link per peer), the kernels alone run at kernel_ms per step
            out['collective'] = {'kind': 'all_gather_into_tensor (RCCL) x2 (points, per-frame means), async, double-buffered',
                                 'bytes_per_rank_per_step': int(q_bytes + mean_bytes),
                                 'value_kernels_only': n_units * world / (k_ms * 1e-3)}
        out.update(cpu if cpu else {'cpu_baseline': None})
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
