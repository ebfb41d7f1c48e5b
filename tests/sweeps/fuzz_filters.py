"""Randomised sweep of the column filters: random lengths (1 .. 600 frames), column counts, gap patterns and parameters,
every column of every case against oracle/filtering_ref.py (the reference's own SciPy / NumPy calls; Kalman: the
restatement, parity unpinned).     python tests/sweeps/fuzz_filters.py [n_cases] [seed]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import filtering_ref as fr
from pose2sim_amd import filtering
from pose2sim_amd.engine import Engine

def run(n_cases, seed, verbose=True):
    """-> (mismatches, worst relative deviation per filter).  Also called by tests/test_filter_gpu.py with a small n_cases."""
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    state = {'bad': 0}
    worst = {}

    def check(name, got, want, exact=False):
        if got.shape != want.shape or not np.array_equal(np.isnan(got), np.isnan(want)):
            state['bad'] += 1
            print('MISMATCH', name, 'shape / NaN pattern')
            return
        ok = ~np.isnan(want)
        d = float((np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))).max()) if ok.any() else 0.0
        worst[name.split()[0]] = max(worst.get(name.split()[0], 0.0), d)
        if d > (0.0 if exact else 1e-9):
            state['bad'] += 1
            print('MISMATCH', name, f'{d:.3e}')

    for case in range(n_cases):
        F = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 31, 100, 333, 600]))
        ncol = int(rng.choice([1, 3, 64, 65, 78]))
        fps = float(rng.choice([30, 60, 120]))
        t = np.arange(F)[:, None] / fps
        data = 1.0 + 0.5 * np.sin(2 * np.pi * (0.3 + rng.random(ncol)) * t) + rng.normal(0, 0.01, (F, ncol))
        clean = data.copy()
        data[rng.random((F, ncol)) < float(rng.choice([0.0, 0.01, 0.1]))] = np.nan
        data[rng.random((F, ncol)) < float(rng.choice([0.0, 0.01]))] = 0.0
        if ncol > 2:
            data[:, 1] = np.nan
        tag = f'case {case} F={F} ncol={ncol} fps={fps}'
        with np.errstate(all='ignore'), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            cols = range(ncol)
            order, cutoff = int(rng.choice([2, 4, 6])), float(rng.choice([3, 6, 10]))
            check(f'butterworth {tag} order {order} cutoff {cutoff}', filtering.butterworth_filter(data, order, cutoff, fps, eng),
                  np.stack([fr.butterworth_filter_1d(data[:, c], order, cutoff, fps) for c in cols], 1))
            if F > 1:                                                  # (one frame: the reference's col_diff[1] raises IndexError)
                check(f'speed {tag}', filtering.butterworth_on_speed_filter(data, order, cutoff, fps, eng),
                      np.stack([fr.butterworth_on_speed_filter_1d(data[:, c], order, cutoff, fps) for c in cols], 1))
            ns = float(rng.choice([1.0, 2.0, 3.5]))
            check(f'hampel {tag} n_sigma {ns}', filtering.hampel_filter(data, eng, ns), np.stack([fr.hampel_filter(data[:, c], 7, ns) for c in cols], 1), exact=True)
            sg = int(rng.choice([1, 2, 5]))
            check(f'gaussian {tag} sigma {sg}', filtering.gaussian_filter(data, sg, eng), np.stack([fr.gaussian_filter_1d(data[:, c], sg) for c in cols], 1))
            k = int(rng.choice([1, 3, 5, 9]))
            check(f'median {tag} k {k}', filtering.median_filter(clean, k, eng), np.stack([fr.median_filter_1d(clean[:, c], k) for c in cols], 1), exact=True)
            mc, beta, dc = float(rng.choice([0.5, 2.5, 10.0])), float(rng.choice([0.0, 0.9, 3.0])), float(rng.choice([0.5, 1.0]))
            check(f'one_euro {tag} {mc} {beta} {dc}', filtering.one_euro_filter(data, fps, mc, beta, dc, eng),
                  np.stack([fr.one_euro_filter_1d(data[:, c], fps, mc, beta, dc) for c in cols], 1))
            trust, smooth = int(rng.choice([5, 500])), bool(rng.random() < 0.7)
            check(f'kalman {tag} trust {trust} smooth {smooth}', filtering.kalman_filter(data, fps, trust, smooth, eng),
                  np.stack([fr.kalman_filter_1d(data[:, c], fps, trust, smooth) for c in cols], 1))
    eng.close()
    if verbose:
        print(f'{n_cases} cases x 7 filters: {state["bad"]} mismatches; worst relative deviation per filter:', {k: f'{v:.1e}' for k, v in worst.items()})
    return state['bad'], worst


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
