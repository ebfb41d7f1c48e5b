// Internal structures shared by the C-ABI layer (p2s_api.hip) and the kernels.
#ifndef P2S_INTERNAL_H
#define P2S_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// One camera, device resident (computeP + retrieve_calib_params, common.py:254-324).
struct P2sCam {
    double P[12];          // projection matrix (optim_K based when undistorting)
    double fx, fy, cx, cy; // original intrinsics
    double ifx, ify;       // 1/fx, 1/fy (OpenCV multiplies by reciprocals)
    double k[5];           // k1 k2 p1 p2 k3
    double R[9];           // rotation matrix, world -> camera
    double T[3];
    double nk[9];          // optim_K
    double iK[9];          // inverse of the original K        (association rays)
    double center[3];      // -R^T T                            (association rays)
};

struct P2sTriArgs {
    const void *xyl;
    const int32_t *swap_idx;
    double *Q;
    float *err;
    uint8_t *n_excl;
    uint32_t *mask;
    const P2sCam *cams;
    const uint32_t *binom;   // [33][33] binomial coefficients
    int64_t n_blocks;
    int32_t K, C, FB, G;
    int32_t lds_P_off, lds_binom_off;
    int32_t min_cams, undistort, lr_swap;
    double thr, lik_thr;
};

struct P2sAssocArgs {
    const int32_t *n_persons;   // [F][C]
    const int64_t *offsets;     // [F+1]
    const void *kpts;           // [rows][Kj][3]
    double *affinity;           // [F][Nmax][Nmax]
    const P2sCam *cams;
    int64_t n_frames;
    int32_t C, Kj, Nmax, max_iter;
    double recon_thr, min_affinity, w_rank, tol, w_sparse;
};

hipError_t p2s_launch_tri(const P2sTriArgs &a, int dtype, int grid, int threads, size_t lds, hipStream_t s);
hipError_t p2s_launch_assoc(const P2sAssocArgs &a, int dtype, hipStream_t s);

#endif
