"""Pruning statistics of the 32-camera work-list path on a cfg5-like shard: camera errors computed per pruned subset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pose2sim_amd import skeletons, synth, synth_device
from pose2sim_amd.engine import Engine
ids, names, swap = skeletons.keypoints('HALPE_26')
F = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
wl = synth.make_config(8, 32, 26, 1, seed=5, undistort=True, lr_swap=True, swap_idx=swap)
dev = torch.device('cuda', 0)
x = synth_device.make_observations_device(wl['cams'], F, 1, 26, seed=5, device=dev, distort=True, p_lr_swap=0.02, swap_idx=list(swap))
eng = Engine(0); eng.set_calibration(wl['P'], wl['cams'])
prm = eng.tri_params(15.0, 0.3, 2, True, True)
xh = x.cpu().numpy()
for rep in range(2):
    eng.tri_stats(reset=True)
    t = time.time(); out = eng.triangulate(xh, prm, swap); dt = time.time() - t
st = eng.tri_stats(reset=True)
print(st)
print('camera errors per pruned subset %.2f of 32; passes per 64 subsets %.3f' % (st['pruned_camera_errors'] / max(1, st['pruned_subsets']), st['passes'] * 64 / max(1, st['subsets_evaluated'])))
