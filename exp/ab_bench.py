"""Diagnostics: interleaved A/B timing of several builds of the library in ONE process on one GPU (cdna guide rule 24:
never rank builds by timings from different devices or processes).
    python exp/ab_bench.py name=path/to/libA.so name2=path/to/libB.so [--config cfg2] [--rounds 12] [--steps 20]
Each round times `steps` back-to-back p2s_triangulate_device calls per variant with HIP events; prints median / min."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--rounds', type=int, default=12)
    ap.add_argument('--steps', type=int, default=20)
    args = ap.parse_args()
    import torch
    import bench
    from pose2sim_amd import _lib
    cfg = bench.CONFIGS[args.config]
    xyl, cams, P, swap, K = bench.make_workload(cfg, 0)
    n_blocks = xyl.shape[0] * xyl.shape[1]
    n_units = n_blocks * K
    dev = torch.device('cuda', 0)
    d_xyl = torch.from_numpy(xyl).to(dev)
    d_swap = torch.from_numpy(swap).to(dev)
    d_Q = torch.empty(n_units * 3, dtype=torch.float64, device=dev)
    d_e = torch.empty(n_units, dtype=torch.float32, device=dev)
    d_m = torch.empty(n_units, dtype=torch.int32, device=dev)
    d_n = torch.empty(n_units, dtype=torch.uint8, device=dev)
    prm = _lib.TriParams(float(cfg['thr']), float(cfg['lik']), int(cfg['min_cams']), int(cfg['undistort']), int(cfg['lr_swap']), 0)
    Pm = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 12))
    stream = torch.cuda.current_stream().cuda_stream
    variants = []
    keep = []
    for spec in args.libs:
        name, path = spec.split('=', 1)
        lib = C.CDLL(os.path.abspath(path))
        for fn, (res, argt) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):
                getattr(lib, fn).restype = res
                getattr(lib, fn).argtypes = argt
        h = C.c_void_p()
        assert lib.p2s_create(0, C.byref(h)) == 0
        n = Pm.shape[0]
        Kc = np.ascontiguousarray(np.asarray(cams['K'], dtype=np.float64).reshape(n, 9))
        dc = np.zeros((n, 5)); dc[:, :4] = np.asarray(cams['dist'], dtype=np.float64).reshape(n, -1)[:, :4]
        Rc = np.ascontiguousarray(np.asarray(cams['R_mat'], dtype=np.float64).reshape(n, 9))
        Tc = np.ascontiguousarray(np.asarray(cams['T'], dtype=np.float64).reshape(n, 3))
        nk = np.ascontiguousarray(np.asarray(cams['optim_K'], dtype=np.float64).reshape(n, 9))
        keep.append((Kc, dc, Rc, Tc, nk))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)                      # noqa: E731
        assert lib.p2s_set_calibration(h, n, ptr(Pm), ptr(Kc), ptr(dc), ptr(Rc), ptr(Tc), ptr(nk)) == 0
        assert lib.p2s_set_stream(h, C.c_void_p(stream)) == 0

        def run(lib=lib, h=h):
            rc = lib.p2s_triangulate_device(h, n_blocks, K, 0, C.c_void_p(d_xyl.data_ptr()), C.c_void_p(d_swap.data_ptr()), C.byref(prm),
                                            C.c_void_p(d_Q.data_ptr()), C.c_void_p(d_e.data_ptr()), C.c_void_p(d_n.data_ptr()), C.c_void_p(d_m.data_ptr()))
            assert rc == 0, lib.p2s_last_error()
        variants.append((name, run, []))
    ref = None
    for name, run, _ in variants:                         # warm-up + agreement of the results
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        sig = (float(torch.nan_to_num(d_Q).sum().item()), int(d_m.sum().item()), int(d_n.sum().item()))
        if ref is None:
            ref = sig
        print(f'{name}: checksum {sig} {"== first" if sig == ref else "DIFFERS from first"}', flush=True)
    for r in range(args.rounds):
        for name, run, times in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                run()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / args.steps * 1e3)
    for name, _, times in variants:
        t = np.array(times)
        print(f'{args.config} {name:12s} median {np.median(t):8.1f} us  min {t.min():8.1f} us  max {t.max():8.1f} us  '
              f'-> {n_units / np.median(t) * 1e6:.3e} units/s', flush=True)


if __name__ == '__main__':
    main()
