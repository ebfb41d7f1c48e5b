"""Diagnostics: time of the 32-camera shard (BASELINE configs[4], 1/10 length) against the deep-level threshold
(P2S_TUNE_DEEP_MIN_SUBSETS) on device-generated data, one process, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pose2sim_amd import skeletons, synth, synth_device, parallel
from pose2sim_amd.engine import Engine, P2S_F32

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg5_tenth']
ids, names, swap_list = skeletons.keypoints(cfg['model'])
K, C, F = len(ids), cfg['C'], int(sys.argv[2]) if len(sys.argv) > 2 else cfg['F']
dev = torch.device('cuda', 0)
cams = synth.make_cameras(C, seed=cfg['seed'], distort=cfg['undistort'])
P = synth.projection_matrices(cams, cfg['undistort'])
x = synth_device.make_observations_device(cams, F, 1, K, seed=cfg['seed'], device=dev, distort=cfg['undistort'],
                                          p_lr_swap=0.02 if cfg['lr_swap'] else 0.0, swap_idx=swap_list)
eng = Engine(0)
eng.set_calibration(P, cams)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
prm = Engine.tri_params(cfg['thr'], cfg['lik'], cfg['min_cams'], cfg['undistort'], cfg['lr_swap'])
n = F * K
off_e, off_m, off_n, nbytes = parallel.section_offsets(n)
out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
d_swap = torch.from_numpy(np.asarray(swap_list, dtype=np.int32)).to(dev)
b = out.data_ptr()
ref = None
for thr in [int(v) for v in (sys.argv[3:] or ['16384', '4096', '1024', '256', '0'])] * 2:
    eng.set_tuning(Engine.TUNE_DEEP_MIN_SUBSETS, thr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.triangulate_device(F, K, P2S_F32, x, d_swap, prm, b, b + off_e, b + off_n, b + off_m)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sig = (float(torch.nan_to_num(out[:n * 24].view(torch.float64)).sum().item()), int(out[off_m:off_m + n * 4].view(torch.int32).sum().item()))
    ref = ref or sig
    print(f'deep_min_subsets {thr:6d}: {dt * 1e3:9.1f} ms  {"same results" if sig == ref else "RESULTS DIFFER"}', flush=True)
