"""The synthetic generator of synth.py on the GPU ("generated on-device per rank from the seed"),
for bench.py's full-length shards of BASELINE configs[3] / configs[4]: 12.5 GB of observations per GPU cannot come
from a host generator in bench time.  Same cameras (synth.make_cameras on the host), same point model, noise,
likelihood, outlier, missing-camera and L/R-swap rates as synth.make_observations; the random stream is torch's, so the
numbers differ from the host generator's (the host one stays the source of every parity fixture).  Bench
infrastructure: torch allocates and fills, the product path never imports this module.
"""
import numpy as np


def make_observations_device(cams, F, Pn, K, seed, device, noise_px=1.5, p_lowlik=0.05, p_outlier=0.03, p_missing_cam=0.01,
                             p_lr_swap=0.0, swap_idx=None, distort=False, chunk_frames=65536, out=None):
    """float32 CUDA tensor [F][Pn][C][K][3] (x px, y px, likelihood; NaN = missing)."""
    import torch
    C = len(cams['K'])
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + 7919)
    f64 = dict(dtype=torch.float64, device=device)
    xyl = out if out is not None else torch.empty((F, Pn, C, K, 3), dtype=torch.float32, device=device)
    # per-person model: root random walk folded into a 3 x 3 m area + per-joint offsets and sinusoids (synth.make_points3d)
    start = torch.rand((Pn, 2), generator=g, **f64) * 2.0 - 1.0
    off = torch.empty((Pn, K, 3), **f64)
    off[..., 0] = torch.rand((Pn, K), generator=g, **f64) * 0.6 - 0.3
    off[..., 1] = torch.rand((Pn, K), generator=g, **f64) * 0.6 - 0.3
    off[..., 2] = torch.rand((Pn, K), generator=g, **f64) * 1.8
    amp = torch.rand((Pn, K, 3), generator=g, **f64) * 0.15
    freq = torch.rand((Pn, K, 3), generator=g, **f64) * 0.09 + 0.01
    ph = torch.rand((Pn, K, 3), generator=g, **f64) * (2 * np.pi)
    Rm = torch.tensor(np.stack(cams['R_mat']), **f64)
    Tv = torch.tensor(np.stack(cams['T']), **f64)
    Km = torch.tensor(np.stack(cams['K']), **f64)
    dist = torch.tensor(np.stack([np.concatenate([np.asarray(d, dtype=np.float64).ravel(), np.zeros(5)])[:5] for d in cams['dist']]), **f64)
    sw_idx = torch.tensor(np.asarray(swap_idx, dtype=np.int64), device=device) if (p_lr_swap > 0 and swap_idx is not None) else None
    walk_end = start.clone()
    for f0 in range(0, F, chunk_frames):
        n = min(chunk_frames, F - f0)
        steps = torch.randn((n, Pn, 2), generator=g, **f64) * 0.02
        walk = torch.cumsum(steps, dim=0) + walk_end[None]
        walk_end = walk[-1].clone()
        walk = torch.abs((walk + 1.5) % 6.0 - 3.0) - 1.5
        t = torch.arange(f0, f0 + n, **f64)[:, None, None, None]
        Q = off[None] + amp[None] * torch.sin(freq[None] * t + ph[None])                      # [n][Pn][K][3]
        Q[..., 0] += walk[:, :, None, 0]
        Q[..., 1] += walk[:, :, None, 1]
        for c in range(C):
            X = Q @ Rm[c].T + Tv[c]
            z = X[..., 2]
            z = torch.where(z == 0, torch.ones_like(z), z)
            x, y = X[..., 0] / z, X[..., 1] / z
            if distort:
                k = dist[c]
                r2 = x * x + y * y
                cd = 1 + k[0] * r2 + k[1] * r2 * r2 + k[4] * r2 * r2 * r2
                xd = x * cd + k[2] * (2 * x * y) + k[3] * (r2 + 2 * x * x)
                yd = y * cd + k[2] * (r2 + 2 * y * y) + k[3] * (2 * x * y)
            else:
                xd, yd = x, y
            uv = torch.stack([xd * Km[c, 0, 0] + Km[c, 0, 2], yd * Km[c, 1, 1] + Km[c, 1, 2]], dim=-1)
            uv = uv + torch.randn(uv.shape, generator=g, **f64) * noise_px
            outl = torch.rand((n, Pn, K), generator=g, device=device) < p_outlier
            uv = uv + outl[..., None] * torch.randn(uv.shape, generator=g, **f64) * 60.0
            lik = torch.rand((n, Pn, K), generator=g, **f64) * 0.7 + 0.3
            low = torch.rand((n, Pn, K), generator=g, device=device) < p_lowlik
            lik = torch.where(low, torch.rand((n, Pn, K), generator=g, **f64) * 0.3, lik)
            if sw_idx is not None:
                sw = torch.rand((n, Pn), generator=g, device=device) < p_lr_swap
                uv = torch.where(sw[..., None, None], uv[:, :, sw_idx, :], uv)
                lik = torch.where(sw[..., None], lik[:, :, sw_idx], lik)
            blk = xyl[f0:f0 + n, :, c]
            blk[..., 0] = uv[..., 0].to(torch.float32)
            blk[..., 1] = uv[..., 1].to(torch.float32)
            blk[..., 2] = lik.to(torch.float32)
            miss = torch.rand((n, Pn), generator=g, device=device) < p_missing_cam
            blk[miss] = float('nan')
    return xyl
