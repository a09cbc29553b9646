/*
 * city2ba_hip_experimental.h -- entry points of libcity2ba_hip.so that are NOT part of the stable boundary (city2ba_hip.h):
 * measurement aids the benchmark uses in the same process as the timed run, diagnostics that name the kernel instance a
 * launch takes, and the f32 extension of BASELINE configs[4] (the reference has no f32 path).  They may change between
 * rounds; a drop-in host needs none of them.
 */
#ifndef CITY2BA_HIP_EXPERIMENTAL_H
#define CITY2BA_HIP_EXPERIMENTAL_H
#include "city2ba_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostics ---- */
/* Diagnostic: synchronises `stream`, then counts the workspace's non-zero arrival counters.  Zero whenever no launch
 * using it is in flight; anything else means a fold did not complete.  -1: the workspace was never initialised. */
int c2b_workspace_selfcheck(const void *workspace, void *stream, int64_t *nonzero_words);
/* "RCCL 2.x.y (path)" or "unavailable: ..." */
const char *c2b_comm_backend(void);
/* tiles of 64 observations a wave of the residual + Jacobian launch takes: 1 below ~6 M observations, else 2
 * (diagnostic: names the kernel instance a launch of this size runs; results do not depend on it, the rounding of
 * the folded error sum does, like on any other change of the grid) */
int c2b_jacobian_tiles_per_wave(int64_t n_obs);
/* ... and the full shape -- waves of 64 per workgroup, tiles per wave -- of a launch of n_obs observations into an output
 * set that takes streaming stores at store_GBs (GB/s; 0 = unknown): 16 x 1 below ~6 M observations; above, 8 x 2 -- or
 * 4 x 1 when the set is one of the slow-store kind (< 6.3 TB/s) and 16 x 1 when it lies between the classes (< 6.85 TB/s),
 * which only c2b_residual_jacobian_rows_placed knows.  Diagnostic, like the two above: results do not depend on the shape.
 * The shapes named are those of a launch WITH an error sum (workspace != NULL, the bench step).  One exception: a launch
 * without a sum (workspace == NULL) that would take 8 x 2 runs 16 x 1 instead -- the 8 x 2 instance without the sum's
 * fold does not fit its 128 registers without scratch (capi.hip: launch_jac_l). */
int c2b_jacobian_launch_shape(int64_t n_obs, double store_GBs, int *waves_per_workgroup, int *tiles_per_wave);

/* ---- placed output sets: the search's log, overriding the recorded rate ---- */
/* store rate (GB/s) measured for every attempt, how many there were, and which one was kept */
int c2b_jacobian_outputs_log(const c2b_jacobian_outputs *h, double *store_GBs_per_attempt, int capacity, int *attempts,
                             int *chosen);
/* replace it: for a caller that timed the set itself, or wants one particular launch shape (<= 0: "unknown") */
int c2b_jacobian_outputs_set_store_rate(c2b_jacobian_outputs *h, double store_GBs);

/* Calibration (measurement aids, no reference counterpart; used by bench.py in the same process as the timed run so
 * that a slow device can be told from a slow kernel).  _store_pattern writes a fill pattern over r [n][2], Jc [n][18],
 * Jp [n][6] in exactly the residual+Jacobian kernel's store geometry with no loads and no arithmetic -- the time its
 * stores alone take; _copy is a 16-bytes-per-lane streaming copy (bytes % 16 == 0). */
int c2b_calib_store_pattern(int64_t n_obs, double *r, double *Jc, double *Jp, void *stream);
int c2b_calib_copy(const void *src, void *dst, int64_t bytes, void *stream);

/* ---- f32 extension (BASELINE.json configs[4]).  The reference has NO f32 compute path:
 * these run the same kernels over a float state -- cam15 / pts4 stored as float -- with the draws and
 * the statistics kept in f64; results track the f64 path to f32 accuracy (tested at an f32 tolerance). */
int c2b_convert_f64_to_f32(const double *src, int64_t n, float *dst, void *stream);
int c2b_convert_f32_to_f64(const float *src, int64_t n, double *dst, void *stream);
int c2b_stats_f32(const float *cam15, int64_t n_cam, const float *pts4, int64_t n_pts, void *workspace,
                  double *stats, void *stream);
int c2b_add_drift_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *origin,
                      double strength, double angle_strength, double std, double dir_x, double dir_y,
                      double dir_z, uint64_t seed, void *stream);
int c2b_add_drift_normalized_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts,
                                 const double *stats, double strength, double angle_strength,
                                 double std, uint64_t seed, void *stream);
int c2b_add_noise_entities_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts,
                               const double *stats, double translation_std, double rotation_std,
                               double point_std, uint64_t seed, void *stream);
int c2b_add_sin_noise_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats,
                          double dir_x, double dir_y, double dir_z, double ndir_x, double ndir_y,
                          double ndir_z, double strength, double frequency, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CITY2BA_HIP_EXPERIMENTAL_H */
