/*
 * p2s.h -- C-ABI of the MI355X-native multi-view triangulation / person-association engine.
 *
 * Drop-in boundary for the hot path of Pose2Sim.triangulation() / Pose2Sim.personAssociation().
 * The reference is pure Python with no FFI of its own; each entry point below replaces the inner
 * loops of one reference function (citations relative to /root/reference/Pose2Sim/) and is what a
 * reference-side ctypes binding would call (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions cross the ABI; every call returns a status
 *     (0 = ok, <0 = error) and p2s_last_error() returns the message of the calling thread's
 *     last failure.
 *   - one context per GPU per process; a context is not re-entrant.
 *   - "_device" entry points take DEVICE pointers and enqueue on the context's stream without
 *     synchronising; "_host" entry points take HOST pointers, copy in/out and block.  One exception:
 *     p2s_triangulate_device on the work-list path (undistortion, L/R swap, or more than 16 cameras) with a camera
 *     count whose subset levels can exceed 4 096 subsets (15 cameras or more, min_cameras permitting) reads a 4-byte
 *     count after each chunk's search to drive the deep-level rounds, i.e. it synchronises the stream.
 *   - observation tensor layout: xyl[n_blocks][C][K][3] (x px, y px, likelihood), one block
 *     per (frame, person); NaN = missing.  dtype float32 or float64 (P2S_F32 / P2S_F64).  A detection whose
 *     coordinates are not finite is taken as missing whatever its likelihood (the reference reaches the same result one
 *     search level later); one whose likelihood is exactly 0 must carry finite coordinates (pose estimators write
 *     (0, 0, 0)) or be NaN throughout.
 *   - camera count C <= P2S_MAX_CAMS (the excluded-camera set is returned as a 32-bit mask).
 */
#ifndef P2S_H
#define P2S_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2S_MAX_CAMS 32
#define P2S_MAX_PERSONS_TOTAL 48   /* association: sum over cameras of detected persons per frame */

#define P2S_OK 0
#define P2S_ERR_INVALID_ARG (-1)
#define P2S_ERR_HIP (-2)
#define P2S_ERR_NO_DEVICE (-3)
#define P2S_ERR_NO_CALIB (-4)
#define P2S_ERR_OOM (-5)

#define P2S_F32 0
#define P2S_F64 1

typedef struct p2s_ctx p2s_ctx;

/* Parameters of triangulation_from_best_cameras (triangulation.py:389-393) plus the likelihood
 * mask applied by its caller (triangulation.py:686, 817-821). */
typedef struct p2s_tri_params {
    double reproj_error_threshold;  /* [triangulation] reproj_error_threshold_triangulation, px */
    double likelihood_threshold;    /* [triangulation] likelihood_threshold_triangulation       */
    int32_t min_cameras;            /* [triangulation] min_cameras_for_triangulation (>= 1)     */
    int32_t undistort_points;       /* [triangulation] undistort_points  (0/1)                  */
    int32_t handle_lr_swap;         /* [triangulation] handle_LR_swap    (0/1)                  */
    int32_t reserved;
} p2s_tri_params;

/* Parameters of the multi-person association (personAssociation.py:670-673, 799-801). */
typedef struct p2s_assoc_params {
    double reconstruction_error_threshold; /* [personAssociation.multi_person], metres */
    double min_affinity;                   /* [personAssociation.multi_person]         */
    int32_t min_cameras;                   /* [triangulation] min_cameras_for_triangulation */
    int32_t max_iter;                      /* matchSVT max_iter (reference: 20)  */
    double w_rank;                         /* matchSVT w_rank   (reference: 50)  */
    double tol;                            /* matchSVT tol      (reference: 1e-4)*/
    double w_sparse;                       /* matchSVT w_sparse (reference: 0.1) */
} p2s_assoc_params;

int p2s_version(void);
const char *p2s_last_error(void);

/* Number of visible HIP devices (0 without a GPU; never fails the process). */
int p2s_device_count(int *count);

/* Create / destroy the per-GPU context (device buffers, stream, calibration). */
int p2s_create(int device_id, p2s_ctx **out);
int p2s_destroy(p2s_ctx *ctx);

/* Enqueue on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream; NULL = HIP's
 * default stream).  A fresh context uses a private non-blocking stream. */
int p2s_set_stream(p2s_ctx *ctx, void *hip_stream);
int p2s_synchronize(p2s_ctx *ctx);

/* Calibration of C cameras, the outputs of computeP (common.py:291-324) and
 * retrieve_calib_params (common.py:254-288), all HOST pointers, float64, row-major:
 *   P[C][12]      projection matrices (built from optim_K when undistorting)
 *   Kmat[C][9]    original intrinsics               (may be NULL if never undistorting/associating)
 *   dist[C][5]    k1,k2,p1,p2,k3                    (may be NULL)
 *   Rmat[C][9]    rotation matrices                 (may be NULL)
 *   T[C][3]       translations                      (may be NULL)
 *   newK[C][9]    optim_K                           (may be NULL)
 */
int p2s_set_calibration(p2s_ctx *ctx, int32_t n_cams, const double *P, const double *Kmat,
                        const double *dist, const double *Rmat, const double *T, const double *newK);

/* Robust triangulation of every (block, keypoint) unit: replaces the frame / person / keypoint
 * loops of triangulate_all (triangulation.py:796-845) around triangulation_from_best_cameras
 * (triangulation.py:363-604), including undistortion (:808-813) and the likelihood mask
 * (:817-821).
 *   xyl       [n_blocks][C][K][3], dtype P2S_F32 or P2S_F64
 *   swap_idx  [K] keypoints_idx_swapped (triangulation.py:745); may be NULL when !handle_lr_swap
 * outputs, one per unit u = block*K + k:
 *   Q        [n][3] float64  3D point (NaN when rejected)
 *   err      [n]    float32  mean reprojection error px (NaN when rejected)
 *   n_excl   [n]    uint8    nb_cams_excluded
 *   excl_mask[n]    uint32   bit c set <=> camera c in id_excluded_cams
 */
int p2s_triangulate_device(p2s_ctx *ctx, int64_t n_blocks, int32_t n_kpts, int32_t dtype,
                           const void *d_xyl, const int32_t *d_swap_idx, const p2s_tri_params *params,
                           double *d_Q, float *d_err, uint8_t *d_n_excl, uint32_t *d_excl_mask);
int p2s_triangulate_host(p2s_ctx *ctx, int64_t n_blocks, int32_t n_kpts, int32_t dtype,
                         const void *xyl, const int32_t *swap_idx, const p2s_tri_params *params,
                         double *Q, float *err, uint8_t *n_excl, uint32_t *excl_mask);

/* Multi-person association of every frame: replaces the per-frame body of associate_all
 * (personAssociation.py:783-801): compute_rays (:277-316), compute_affinity (:347-408),
 * circular_constraint (:411-428), matchSVT (:450-509) and the min_affinity cut (:800).
 *   n_persons [F][C]  int32   persons detected per camera (read_json, :260-274)
 *   offsets   [F+1]   int64   start of each frame's rows in kpts (rows = sum_c n_persons[f][c])
 *   kpts      [rows][Kj][3]   dtype P2S_F32/P2S_F64: camera-major, then person, JSON keypoint order
 * output:
 *   affinity  [F][Nmax][Nmax] float64, the thresholded matchSVT result, top-left N_f x N_f valid
 *             (N_f = sum_c n_persons[f][c] <= Nmax <= P2S_MAX_PERSONS_TOTAL)
 * The order-sensitive proposal extraction (person_index_per_cam, :512-549) stays on the host.
 */
int p2s_associate_device(p2s_ctx *ctx, int64_t n_frames, int32_t n_kpts_json, int32_t n_max, int32_t dtype,
                         const int32_t *d_n_persons, const int64_t *d_offsets, const void *d_kpts,
                         const p2s_assoc_params *params, double *d_affinity);
int p2s_associate_host(p2s_ctx *ctx, int64_t n_frames, int32_t n_kpts_json, int32_t n_max, int32_t dtype,
                       const int32_t *n_persons, const int64_t *offsets, const void *kpts,
                       const p2s_assoc_params *params, double *affinity);

/* Parameters of the single-person association (personAssociation.py:177-180). */
typedef struct p2s_single_params {
    double reproj_error_threshold;  /* [personAssociation.single_person] reproj_error_threshold_association, px */
    double likelihood_threshold;    /* [personAssociation] likelihood_threshold_association               */
    int32_t min_cameras;            /* [triangulation] min_cameras_for_triangulation (>= 1)                */
    int32_t reserved;
} p2s_single_params;

/* Single-person association of every frame: replaces persons_combinations (personAssociation.py:67-99)
 * and best_persons_and_cameras_combination (:154-257, pinhole branch) with triangulate_comb (:102-151):
 * every combination of one person per camera x every subset of cameras switched off, in the
 * reference's visiting order and with its carry-over rules.
 *   n_persons [F][C]  int32   persons per camera (read_json order), each <= P2S_MAX_PERSONS_PER_CAM
 *   offsets   [F+1]   int64   start of each frame's rows in tracked
 *   tracked   [rows][3]       (x, y, likelihood) of the tracked keypoint of every person, camera-major,
 *                             dtype P2S_F32 / P2S_F64
 * outputs per frame:
 *   comb [F][C] int32  chosen person per camera, -1 = camera off / nothing detected (NaN in the reference)
 *   err  [F]    f64    best reprojection error (inf when no combination could be triangulated)
 *   Q    [F][3] f64    the tracked keypoint in 3D for that combination
 * The product of the per-camera person counts of a frame must not exceed P2S_MAX_COMBINATIONS. */
#define P2S_MAX_PERSONS_PER_CAM 16
#define P2S_MAX_COMBINATIONS (1 << 20)
#define P2S_MAX_SINGLE_SEARCH 2147483648.0  /* worst-case (combination, camera subset) evaluations per frame */
int p2s_associate_single_device(p2s_ctx *ctx, int64_t n_frames, int32_t dtype, const int32_t *d_n_persons,
                                const int64_t *d_offsets, const void *d_tracked, const p2s_single_params *params,
                                int32_t *d_comb, double *d_err, double *d_Q);
int p2s_associate_single_host(p2s_ctx *ctx, int64_t n_frames, int32_t dtype, const int32_t *n_persons,
                              const int64_t *offsets, const void *tracked, const p2s_single_params *params,
                              int32_t *comb, double *err, double *Q);

/* ---- downstream of the .trc: filtering and quality metrics -----------------------------------
 * Zero-phase Butterworth filter of every column of a row-major [n_frames][n_cols] float64 matrix: replaces
 * Q_coords.apply(butterworth_filter_1d) in filter_all (filtering.py:437-471, 804): per column, every run of valid
 * samples (not NaN, not 0) longer than padlen = 3 * n_coef goes through scipy.signal.filtfilt(b, a, run) (odd
 * padding, steady-state initial conditions); other samples are copied.  b, a [n_coef] from
 * scipy.signal.butter(order / 2, cutoff / (frame_rate / 2)) with a[0] = 1, zi [n_coef - 1] from
 * scipy.signal.lfilter_zi(b, a); 2 <= n_coef <= 9.  All pointers are HOST pointers; the call blocks. */
int p2s_butterworth_host(p2s_ctx *ctx, int64_t n_frames, int32_t n_cols, const double *data, int32_t n_coef,
                         const double *b, const double *a, const double *zi, double *out);

/* The other column filters of the reference's filtering stage, every column of a row-major [n_frames][n_cols] float64
 * matrix at once (replaces Q_coords.apply(hampel_filter) and Q_coords.apply(filter1d), filtering.py:794-798):
 *   P2S_FILTER_HAMPEL    hampel_filter filtering.py:63-85, 7-sample window; params = {n_sigma}
 *   P2S_FILTER_GAUSSIAN  gaussian_filter_1d :513-529 = scipy.ndimage.gaussian_filter1d(col, sigma), mode 'reflect';
 *                        params = the 2 radius + 1 kernel weights (scipy.ndimage._filters._gaussian_kernel1d)
 *   P2S_FILTER_MEDIAN    median_filter_1d :561-577 = scipy.signal.medfilt(col, kernel_size), zero padding;
 *                        params = {kernel_size} (odd); NaN samples are refused (scipy's result for them depends on its
 *                        selection algorithm)
 *   P2S_FILTER_ONE_EURO  one_euro_filter_1d :87-160, forward and backward pass over every run of >= 2 samples that are
 *                        not NaN; params = {1 / frame_rate, min_cutoff, beta, d_cutoff}
 *   P2S_FILTER_KALMAN    kalman_filter_1d :316-434: constant-acceleration Kalman filter (predict, Joseph-form update per
 *                        sample) and, with smooth != 0, the Rauch-Tung-Striebel smoother over every run of >= 4 samples
 *                        that are neither NaN nor 0; params = {1 / frame_rate, measurement_noise, process_noise, smooth}.
 *                        PARITY UNPINNED: the reference takes both from filterpy (not importable where this was built);
 *                        restated from filterpy's published algorithm
 * All pointers are HOST pointers; the call blocks. */
#define P2S_FILTER_HAMPEL 1
#define P2S_FILTER_GAUSSIAN 2
#define P2S_FILTER_MEDIAN 3
#define P2S_FILTER_ONE_EURO 4
#define P2S_FILTER_KALMAN 5
int p2s_filter_columns_host(p2s_ctx *ctx, int32_t kind, int64_t n_frames, int32_t n_cols, const double *data,
                            const double *params, int32_t n_params, double *out);
/* trc_evaluate's per-frame quantities and sums (Utilities/trc_evaluate.py:114-238) for xyz [n_frames][n_markers][3]:
 *   bones      [n_bones][2] int32  (parent, child) marker indices
 *   bone_len   [n_bones][n_frames]      |child - parent|, 0 -> NaN             (compute_bone_lengths :135-139)
 *   bone_stats [n_bones][3]             nanmean, nanstd (population), n_valid   (:141-150)
 *   accel      [n_markers][n_frames-2]  |p[f+2] - 2 p[f+1] + p[f]|              (compute_smoothness :185-186)
 *   missing    [n_markers] int64        frames with a NaN coordinate            (compute_missing_data :226-227)
 * Medians / percentiles of accel stay with the caller (np.median, np.percentile).  HOST pointers; blocks. */
int p2s_trc_metrics_host(p2s_ctx *ctx, int64_t n_frames, int32_t n_markers, const double *xyz, int32_t n_bones,
                         const int32_t *bones, double *bone_len, double *bone_stats, double *accel, int64_t *missing);

/* Counters of the triangulation calls of this context since creation (or the last reset), after synchronising its
 * streams; out holds 8 values: out[0] units that entered the camera-subset search (triangulation.py:408 beyond the
 * first pass), out[1] camera subsets evaluated, out[2] 64-lane evaluation passes, out[3] units whose search stopped at
 * the safety valve (a level with more than 2^26 subsets, i.e. C(32, 11) and beyond, is not entered: such a unit comes
 * back as not triangulated where the reference would have gone on for hours -- callers report the count), out[4]
 * per-camera reprojection errors computed for the candidates of the pruned passes (levels of hundreds of subsets and
 * more: hopeless candidates are dropped after a few cameras, so fewer than cameras x subsets), out[5] those candidates
 * (a part of out[1]), out[6] camera subsets looked at by the fp32 screen of the pooled kernel (only its survivors are
 * evaluated in fp64 and counted in out[1]), out[7] 64-lane screen passes. */
int p2s_get_tri_stats(p2s_ctx *ctx, uint64_t *out, int32_t reset);

/* Counters of the multi-person association calls of this context since creation (or the last reset), after
 * synchronising its stream: out[0] frames with at least one detection, out[1] ADMM passes of matchSVT
 * (personAssociation.py:477-505), out[2] Jacobi sweeps of their singular value decompositions, out[3] fp64 operations
 * by the kernels' own count (a multiplication or addition is 1, a fused multiply-add 2; rotations, products and
 * updates as the kernel that ran does them -- the figure bench.py divides by the fp64 vector peak). */
int p2s_get_assoc_stats(p2s_ctx *ctx, uint64_t *out, int32_t reset);

/* Experiments and tests only -- apart from P2S_TUNE_MAX_SUBSETS nothing here changes a result, and the library never
 * reads the environment.
 *   P2S_TUNE_TRI_PATH     P2S_TRI_PATH_AUTO (default): the pooled one-launch kernel (failures of three tiles pooled, fp32
 *                         screen + fp64 evaluation of the surviving camera subsets) where it applies (pinhole, no L/R swap,
 *                         float32 observations, up to 16 cameras), else round 2's one-launch kernel (float64 observations,
 *                         up to 8 cameras), else the streaming + work-list search pair;
 *                         P2S_TRI_PATH_POOLED: the same choice (named, for tests); P2S_TRI_PATH_TWO_TILES / _ONE_TILE: round
 *                         2's one-launch kernel with two tiles per wave where it pools / one tile everywhere;
 *                         P2S_TRI_PATH_WORKLIST: always the pair
 *   P2S_TUNE_SCREEN       0: the pooled kernel sends every camera subset to the fp64 evaluation (default 1: only those
 *                         its fp32 screen cannot rule out; same results bit for bit)
 *   P2S_TUNE_POOL_TILES   tiles of 64 units a wave of the pooled kernel streams before it searches their failures (2..6,
 *                         default 5; 9-16 cameras: always 2; up to 4 cameras: always 5)
 *   P2S_TUNE_POOL_SINGLES_PCT  share (%) of the tiles that the last workgroups of every XCD take one at a time instead
 *                         of several (default 8)
 *   P2S_TUNE_FORCE_TILED  1: the LDS-tiled streaming kernel even where observations fit in registers
 *   P2S_TUNE_NO_OVERLAP   1: search kernels on the main stream instead of beside the next chunk's streaming pass
 *   P2S_TUNE_SEARCH_JOB   work-list records a search wave takes at a time (8..64; 0 = automatic)
 *   P2S_TUNE_MAX_SUBSETS  the work-list search does not enter a level with more camera subsets than this (default
 *                         2^26; this one DOES change results -- tests of the valve only)
 *   P2S_TUNE_DEEP_MIN_SUBSETS  levels of the work-list search with more camera subsets than this (default 4 096) are
 *                         cut into chunks and spread over the whole GPU instead of being walked by one wave; 0 = never
 *   P2S_TUNE_DEEP_PRUNE   0: the long levels evaluate every camera of every candidate (default 1: exact pruning, same
 *                         results)
 *   P2S_TUNE_ASSOC_FORM   P2S_ASSOC_FORM_AUTO (default): up to 32 detections per frame take the symmetric one-wave kernel;
 *                         P2S_ASSOC_FORM_GENERAL: the general kernel (no symmetry assumed) at every size
 *   P2S_TUNE_DIAG_MODE    kernel diagnostics of a -DP2S_DIAG build (exp/README.md); refused by the shipped library */
#define P2S_TUNE_TRI_PATH 1
#define P2S_TUNE_FORCE_TILED 2
#define P2S_TUNE_NO_OVERLAP 3
#define P2S_TUNE_SEARCH_JOB 4
#define P2S_TUNE_DIAG_MODE 5
#define P2S_TUNE_MAX_SUBSETS 6
#define P2S_TUNE_DEEP_MIN_SUBSETS 7
#define P2S_TUNE_ASSOC_FORM 8
#define P2S_TUNE_POOL_SINGLES_PCT 9
#define P2S_TUNE_DEEP_PRUNE 10
#define P2S_TUNE_SCREEN 11
#define P2S_TUNE_POOL_TILES 12
#define P2S_ASSOC_FORM_AUTO 0
#define P2S_ASSOC_FORM_GENERAL 1
#define P2S_TRI_PATH_AUTO 0
#define P2S_TRI_PATH_WORKLIST 1
#define P2S_TRI_PATH_ONE_TILE 2
#define P2S_TRI_PATH_POOLED 3
#define P2S_TRI_PATH_TWO_TILES 4
int p2s_set_tuning(p2s_ctx *ctx, int32_t key, int32_t value);

/* Kernel timing on the context's stream with HIP events: begin, enqueue work, end (blocks). */
int p2s_timing_begin(p2s_ctx *ctx);
int p2s_timing_end(p2s_ctx *ctx, float *elapsed_ms);

/* Tile geometry the triangulation kernel will use for (C, K, dtype): diagnostics for DESIGN.md /
 * bench.py (blocks per tile, threads per workgroup, LDS bytes). */
int p2s_tri_geometry(int32_t n_cams, int32_t n_kpts, int32_t dtype, int32_t *blocks_per_tile,
                     int32_t *threads, int32_t *lds_bytes);

/* ---- OpenPose-JSON ingest (host threads; no GPU involved) ------------------------------------------------
 * Replaces the reference's per-frame, per-person file parsing -- extract_files_frame_f
 * (triangulation.py:607-653: json.load of every camera file once PER PERSON), count_persons_in_json
 * (:77-90) and read_json (personAssociation.py:260-274) -- by one parse of every file into a batch and
 * gather calls that lay the numbers out for p2s_triangulate_* / p2s_associate_*.  A file counts as
 * unreadable exactly when Python's open(path, 'r') + json.load would raise (missing, not UTF-8, not JSON
 * in json.load's dialect: NaN / Infinity literals accepted, control characters in strings rejected,
 * trailing data rejected); repeated object keys take the last value. */
typedef struct p2s_json_batch p2s_json_batch;

#define P2S_JSON_UNREADABLE (-1)      /* people count of a file json.load would raise on (or an empty path) */
#define P2S_JSON_NO_PEOPLE_LIST (-2)  /* valid JSON without a top-level object holding a "people" array      */
#define P2S_JSON_PERSON_NO_LIST (-1)      /* person length: not an object / no "pose_keypoints_2d" array    */
#define P2S_JSON_PERSON_NOT_NUMERIC (-2)  /* person length: the array holds strings / arrays / objects      */

/* paths: the file names back to back (no separators needed), path_offsets [n_files+1] byte offsets into it;
 * an empty name = no file for that slot.  n_threads <= 0: one per hardware thread. */
int p2s_json_parse(const char *paths, const int64_t *path_offsets, int64_t n_files, int32_t n_threads,
                   p2s_json_batch **out);
int p2s_json_free(p2s_json_batch *batch);
/* counts [n_files]: len(js['people']) or a P2S_JSON_* code; person_base [n_files+1]: prefix sums of
 * max(count, 0) = row of each file's first person in p2s_json_person_lengths.  Either may be NULL. */
int p2s_json_people_counts(const p2s_json_batch *batch, int32_t *counts, int64_t *person_base);
/* lengths [person_base[n_files]]: len(person['pose_keypoints_2d']) or a P2S_JSON_PERSON_* code. */
int p2s_json_person_lengths(const p2s_json_batch *batch, int32_t *lengths);
/* extract_files_frame_f for every file at once: for file i and person n < max_persons writes
 * out[file_offsets[i] + n*person_stride + 3*k + {0,1,2}] = values[3*keypoint_ids[k] + {0,1,2}], NaN when
 * the file, the person or the triplet does not exist (triangulation.py:629-644); file_offsets[i] < 0 skips
 * the file.  Offsets and strides are in elements of dtype.  n_inexact (may be NULL) counts the values a
 * P2S_F32 output could not hold exactly, so that the caller can ask again in P2S_F64. */
int p2s_json_gather_keypoints(const p2s_json_batch *batch, const int32_t *keypoint_ids, int32_t n_ids,
                              int32_t max_persons, const int64_t *file_offsets, int64_t person_stride,
                              int32_t dtype, void *out, int64_t *n_inexact);
/* read_json layout: row r = the first n_values numbers of person person_index[r] of file file_index[r]
 * (NaN-padded), out [n_rows][n_values]. */
int p2s_json_gather_people(const p2s_json_batch *batch, const int64_t *file_index, const int32_t *person_index,
                           int64_t n_rows, int32_t n_values, int32_t dtype, void *out, int64_t *n_inexact);

/* ---- .trc data rows (host threads) ---------------------------------------------------------------------------
 * Replaces DataFrame.to_csv in make_trc (triangulation.py:214): appends n_rows lines
 * `frames[r] \t repr(time[r]) \t repr(data[r][0]) \t ...\n` to the file (the 5 header lines are written by the
 * caller), floats exactly as Python's repr() prints them, NaN as an empty field. */
int p2s_trc_append_rows(const char *path, int64_t n_rows, int32_t n_cols, const int64_t *frames, const double *time,
                        const double *data, int32_t n_threads);
/* repr(float) of one value into out (>= 32 bytes, NUL-terminated); returns the length.  For tests. */
int p2s_format_float_repr(double value, char *out, int32_t capacity);

/* ---- associated-pose JSON files (host threads) -----------------------------------------------------------------
 * rewrite_json_files (personAssociation.py:552-580) for n_files (source, destination) pairs at once: destination =
 * json.dumps of the source document with 'people' replaced by the selected persons -- sel[sel_offsets[i] ..
 * sel_offsets[i+1]) holds, per proposal, the index of the person in the source's 'people' list or -1 for {} --
 * in Python's text (', ' / ': ' separators, ensure_ascii escapes, repr() floats, first-position / last-value for
 * repeated keys).  Whenever the reference would raise (unreadable or invalid source, no 'people' list, index out
 * of range) the destination is removed, as there.  Paths as in p2s_json_parse.  written [n_files] (may be NULL):
 * 1 = file written, 0 = removed. */
int p2s_json_rewrite_people(const char *src_paths, const int64_t *src_offsets, const char *dst_paths,
                            const int64_t *dst_offsets, int64_t n_files, const int64_t *sel_offsets, const int32_t *sel,
                            int32_t n_threads, int8_t *written);

/* ---- proposals from the association result, first half (host threads) -----------------------------------------
 * The per-detection rows of person_index_per_cam (personAssociation.py:512-527) for every frame: rows[f][r][c] =
 * np.argmax of detection r's affinities to camera c's detections, -1 when camera c has none or none is positive.
 * rows [F][n_max][C] (the first sum(n_persons[f]) rows of a frame are written).  The order-sensitive second half
 * (np.unique / np.argsort / first-come filter, :528-549) stays with the caller in NumPy: np.argsort's order among
 * equal counts is unspecified and build-dependent, and it decides the person order in the output files. */
int p2s_assoc_argmax_rows(int64_t n_frames, int32_t n_cams, int32_t n_max, const double *affinity, const int32_t *n_persons,
                          int32_t n_threads, int32_t *rows);

/* Proposals, second half (personAssociation.py:528-549), every frame at once, around the one call whose result is not
 * specified -- np.argsort of the multiplicities, which the caller makes itself on the array the reference would pass:
 *   p2s_assoc_unique_rows: np.unique(rows, axis=0, return_counts=True) per frame: uniq [F][n_max][C] (the distinct rows of
 *     rows[f][0 .. n_rows[f]) in lexicographic order), counts [F][n_max] int64, n_uniq [F];
 *   p2s_assoc_filter_rows: with rank[f][i] = index of the i-th ranked distinct row (np.argsort(counts)[::-1]): the
 *     first-come filter (a row that names, for some camera, a person named by ANY row ranked before it is dropped) and the
 *     minimum number of cameras; props [F][n_max][C] (-1 = the reference's NaN), n_props [F]. */
int p2s_assoc_unique_rows(int64_t n_frames, int32_t n_cams, int32_t n_max, const int32_t *rows, const int32_t *n_rows,
                          int32_t n_threads, int32_t *uniq, int64_t *counts, int32_t *n_uniq);
int p2s_assoc_filter_rows(int64_t n_frames, int32_t n_cams, int32_t n_max, const int32_t *uniq, const int32_t *n_uniq,
                          const int32_t *rank, int32_t min_cams, int32_t n_threads, int32_t *props, int32_t *n_props);

#ifdef __cplusplus
}
#endif
#endif /* P2S_H */
