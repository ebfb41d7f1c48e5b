"""Diagnostics: per-wave timeline of the work-list search kernel (kernel diagnostics mode 5 of a -DP2S_DIAG build of the library:
built here into pose2sim_amd/csrc/libp2s_hip_diag.so and selected through P2S_LIB; the shipped library refuses the mode)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
os.environ['P2S_LIB'] = entry.build_hip(defines=['P2S_DIAG'], lib=os.path.join(ROOT, 'pose2sim_amd', 'csrc', 'libp2s_hip_diag.so'))
import numpy as np, torch
import bench
from pose2sim_amd.engine import Engine, P2S_F32
cfg = bench.CONFIGS['cfg2']
xyl, cams, P, swap, K = bench.make_workload(cfg, 0)
F, Pn, C = xyl.shape[:3]
dev = torch.device('cuda', 0)
eng = Engine(0); eng.set_calibration(P, cams); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_WORKLIST); eng.set_tuning(Engine.TUNE_DIAG_MODE, 5)
prm = eng.tri_params(cfg['thr'], cfg['lik'], cfg['min_cams'], False, False)
d = torch.from_numpy(np.ascontiguousarray(xyl.reshape(F * Pn, C, K, 3))).to(dev)
n = F * Pn * K
Q = torch.zeros((n, 3), dtype=torch.float64, device=dev); e = torch.empty(n, dtype=torch.float32, device=dev)
ne = torch.empty(n, dtype=torch.uint8, device=dev); m = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(3):
    eng.triangulate_device(F * Pn, K, P2S_F32, d, None, prm, Q, e, ne, m)
torch.cuda.synchronize()
t = Q.cpu().numpy().reshape(-1)[:3072 * 8].reshape(3072, 8)
t0 = t[:, 0].min()
beg, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # s_memtime ticks at 100 MHz -> us
print('waves', len(t), 'span us', end.max(), 'start us pct', np.percentile(beg, [0, 50, 90, 99, 100]))
print('end us pct', np.percentile(end, [0, 10, 50, 90, 100]))
print('jobs per wave', np.bincount(t[:, 2].astype(int)))
live = end - beg
print('busy fraction of wave slots', live.sum() / (len(t) * end.max()))
print('live us mean', live.mean(), 'fetch us mean', t[:, 3].mean() / 100, 'N-phase', t[:, 4].mean() / 100, 'level1', t[:, 5].mean() / 100, 'search', t[:, 6].mean() / 100)
print('per job: fetch', t[:, 3].sum() / t[:, 2].sum() / 100, 'N', t[:, 4].sum() / t[:, 2].sum() / 100, 'lvl1', t[:, 5].sum() / t[:, 2].sum() / 100, 'search', t[:, 6].sum() / t[:, 2].sum() / 100, 'total', live.sum() / t[:, 2].sum())
sh = np.arange(3072) % 128
jps = np.bincount(sh, weights=t[:, 2], minlength=128)
print('jobs per shard: min %d max %d mean %.1f' % (jps.min(), jps.max(), jps.mean()))
endps = np.array([end[sh == s].max() for s in range(128)])
print('shard end us: min %.1f median %.1f max %.1f' % (endps.min(), np.median(endps), endps.max()))
print('corr(jobs per shard, shard end)', np.corrcoef(jps, endps)[0, 1])
# waves of the slowest shard
s = int(np.argmax(endps)); w = np.flatnonzero(sh == s)
print('slowest shard', s, 'jobs', t[w, 2].astype(int), 'end', np.round(end[w], 0))
cu = (np.arange(3072) // 4)   # block id
