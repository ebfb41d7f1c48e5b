"""Diagnostics: per-frame phase timeline of the association kernel (kernel diagnostics mode 7 of a -DP2S_DIAG build of the library:
built here into pose2sim_amd/csrc/libp2s_hip_diag.so and selected through P2S_LIB; the shipped library refuses the mode)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
os.environ['P2S_LIB'] = entry.build_hip(defines=['P2S_DIAG'], lib=os.path.join(ROOT, 'pose2sim_amd', 'csrc', 'libp2s_hip_diag.so'))
import bench
from pose2sim_amd.engine import Engine
cfg = dict(bench.CONFIGS['cfg3']); cfg['F'] = 4000
xyl, cams, P, swap, K = bench.make_workload(cfg, 0)
n_persons, kpts = bench.make_association_inputs(xyl, cfg['seed'])
eng = Engine(0); eng.set_calibration(P, cams); eng.set_tuning(Engine.TUNE_DIAG_MODE, 7)
aff = eng.associate(n_persons, kpts, Engine.assoc_params(0.1, 0.2, 2))
t = aff.reshape(aff.shape[0], -1)[:, :8]
tot = t[:, 0].mean()
print('frames', len(t), 'N mean', t[:, 7].mean(), 'iterations mean', t[:, 6].mean(), 'sweeps per iteration', (t[:, 5] / t[:, 6]).mean())
print('ticks per frame %.0f: affinity %.1f%%, product %.1f%%, svd %.1f%%, update %.1f%%' % (tot, 100 * t[:, 1].mean() / tot, 100 * t[:, 2].mean() / tot, 100 * t[:, 3].mean() / tot, 100 * t[:, 4].mean() / tot))
steps = t[:, 5] * (t[:, 7] - 1)
print('ticks per Jacobi step %.0f' % (t[:, 3].sum() / steps.sum()))
