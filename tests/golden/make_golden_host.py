"""Unit-level goldens of the sequential host steps, recorded from the reference (SURVEY 8c fixtures 3, 4):
sort_people_sports2d (common.py:1037-1136), interpolate_zeros_nans (common.py:669-712),
indices_of_first_last_non_nan_chunks (triangulation.py:93-148), skeleton order / swap map
(triangulation.py:716-749).  -> host_units.npz"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402


def gen():
    common, tri, pa, sk = ref_shim.load()
    rng = np.random.default_rng(77)
    out = {}

    # ---- person tracking: persons entering / leaving / all-NaN / swapped order ---------------------
    n = 0
    for case in range(60):
        K = int(rng.integers(3, 9))
        n_prev, n_curr = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        base = rng.uniform(-2, 2, (6, K, 3))
        prev = base[:n_prev] + rng.normal(0, 0.02, (n_prev, K, 3))
        perm = rng.permutation(6)[:n_curr]
        curr = base[perm] + rng.normal(0, 0.03, (n_curr, K, 3))
        if n_prev and rng.random() < 0.3:
            prev[rng.integers(0, n_prev)] = np.nan                      # a person never seen so far
        if n_curr and rng.random() < 0.3:
            curr[rng.integers(0, n_curr), rng.integers(0, K)] = np.nan  # a missing keypoint
        if n_curr and rng.random() < 0.2:
            curr[rng.integers(0, n_curr)] = np.nan
        max_dist = [None, 0.1, 1.0][case % 3]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = common.sort_people_sports2d(prev.copy(), curr.copy(), max_dist=max_dist)
        out[f'sort{n}_prev'] = prev; out[f'sort{n}_curr'] = curr
        out[f'sort{n}_max'] = np.array(np.nan if max_dist is None else max_dist)
        out[f'sort{n}_n_out'] = np.array(len(res))
        for i, r in enumerate(res):
            out[f'sort{n}_out{i}'] = np.asarray(r, dtype=float)
        n += 1
    out['n_sort'] = np.array(n)

    # ---- interpolation: kinds, gap limits, zeros and NaNs, short columns, offset indices ---------------
    n = 0
    for case in range(48):
        L = int(rng.integers(4, 60))
        start = int(rng.integers(0, 50))
        t = np.arange(L)
        col = np.sin(t / 7.0) * 3 + rng.normal(0, 0.05, L) + 5
        bad = rng.random(L) < rng.choice([0.0, 0.1, 0.3, 0.6])
        if L > 20 and rng.random() < 0.5:
            g0 = int(rng.integers(0, L - 12)); bad[g0:g0 + int(rng.integers(3, 12))] = True
        col[bad] = np.where(rng.random(bad.sum()) < 0.5, np.nan, 0.0)
        if rng.random() < 0.2:
            col[:int(rng.integers(1, 4))] = np.nan                       # extrapolation at the start
        if rng.random() < 0.2:
            col[-int(rng.integers(1, 4)):] = np.nan                      # ... and at the end
        kind = ['linear', 'slinear', 'quadratic', 'cubic'][case % 4]
        N = [3, 10, 20, 100][(case // 4) % 4]
        s = pd.Series(col, index=range(start, start + L))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = common.interpolate_zeros_nans(s.copy(), N, kind)
        out[f'interp{n}_col'] = col; out[f'interp{n}_start'] = np.array(start)
        out[f'interp{n}_N'] = np.array(N); out[f'interp{n}_kind'] = np.array(kind)
        out[f'interp{n}_out'] = np.asarray(res, dtype=float)
        n += 1
    out['n_interp'] = np.array(n)

    # ---- valid sections -----------------------------------------------------------------------------
    n = 0
    for case in range(40):
        L = int(rng.integers(5, 120))
        v = rng.normal(0, 1, L)
        k = 0
        while k < L:
            run = int(rng.integers(1, 25))
            if rng.random() < 0.4:
                v[k:k + run] = np.nan
            k += run
        method = ['largest', 'all', 'first', 'last', 'bogus'][case % 5]
        mcs = [None, 3, 10, 15][(case // 5) % 4]
        res = tri.indices_of_first_last_non_nan_chunks(pd.Series(v), min_chunk_size=mcs, chunk_choice_method=method)
        out[f'chunk{n}_v'] = v; out[f'chunk{n}_method'] = np.array(method)
        out[f'chunk{n}_mcs'] = np.array(-1 if mcs is None else mcs); out[f'chunk{n}_out'] = np.array(res)
        n += 1
    out['n_chunk'] = np.array(n)

    # ---- skeleton order and swap map of every built-in model (triangulation.py:734-749) --------------
    from anytree import RenderTree
    models = ['HALPE_26', 'COCO_133_WRIST', 'COCO_133', 'COCO_17', 'HAND_21', 'FACE_106', 'ANIMAL2D_17', 'BODY_25B', 'BODY_25',
              'BODY_135', 'BLAZEPOSE', 'HALPE_68', 'HALPE_136', 'COCO', 'MPII']
    done = []
    for m in models:
        model = getattr(sk, m, None)
        if model is None:
            continue
        ids = [node.id for _, _, node in RenderTree(model) if node.id is not None]
        names = [node.name for _, _, node in RenderTree(model) if node.id is not None]
        keypoints_names_swapped = ['L' + nm[1:] if nm.startswith('R') else 'R' + nm[1:] if nm.startswith('L') else nm for nm in names]
        keypoints_names_swapped = [nm.replace('right', 'left') if nm.startswith('right') else nm.replace('left', 'right')
                                   if nm.startswith('left') else nm for nm in keypoints_names_swapped]
        try:
            idx = [names.index(nm) for nm in keypoints_names_swapped]
        except ValueError:
            idx = list(range(len(names)))
        out[f'skel_{m}_ids'] = np.array(ids); out[f'skel_{m}_names'] = np.array(names, dtype='U40'); out[f'skel_{m}_swap'] = np.array(idx)
        done.append(m)
    out['skel_models'] = np.array(done, dtype='U20')
    np.savez_compressed(os.path.join(HERE, 'host_units.npz'), **out)
    print('wrote host_units.npz:', int(out['n_sort']), 'tracking,', int(out['n_interp']), 'interpolation,', int(out['n_chunk']), 'section cases,', len(done), 'skeletons')


if __name__ == '__main__':
    gen()
