/*
 * city2ba_hip.h -- C ABI of the MI355X (gfx950) hot path of city2ba.
 *
 * The reference (tkonolige/city2ba, Rust) has no FFI seam for this path; its extension
 * points are `trait Camera` (src/baproblem.rs:107-125), the inherent methods of
 * `BAProblem<C>` (src/baproblem.rs:262-390) and the free functions of src/noise.rs.
 * Per-point trait calls are far too fine-grained for a device, so this ABI is batched at
 * BAProblem-method / noise-function granularity.  Each entry point names the reference
 * item it replaces.  A Rust host binds these with a plain `extern "C"` block
 * (INTEGRATION.md shows it); nothing here mentions torch, C++ or HIP types.
 *
 * Data conventions (all little-endian host order, f64 unless noted)
 *   cam15   SnavelyCamera in memory (src/baproblem.rs:130-138):
 *           dir as 9 doubles COLUMN-major (cgmath Matrix3 `as_ref::<[f64;9]>()`),
 *           loc (t) 3, intrin (f,k1,k2) 3.
 *   bal9    SnavelyCamera::to_vec order (src/baproblem.rs:189-202): w0 w1 w2 t0 t1 t2 f k1 k2.
 *   pts3    packed Point3<f64> [n][3];   pts4 = device copy padded to [n][4] (32-B rows,
 *           one aligned 2x16-B gather per observation).
 *   CSR     vis_graph: Vec<Vec<(usize,(f64,f64))>> (src/baproblem.rs:256-260) flattened
 *           camera-major: row_ptr[n_cam+1] (u64), pt_idx[n_obs] (u64 host / u32 device),
 *           uv[n_obs][2].  cam_idx[n_obs] (u32) is the COO expansion of row_ptr, i.e. the
 *           camera column of a .bal observation line (src/baproblem.rs:718-722).
 *   camblk  derived per-camera record, C2B_CAMBLK_DOUBLES doubles (256 bytes):
 *           R row-major[9] | t[3] | f,k1,k2 | J_l(w) row-major[9] | center[3] | pad[5] -- two 128-byte lines, the first of which
 *           is exactly what projection needs.  The TABLE is blocked in groups of C2B_CAMBLK_GROUP = 8 cameras (r05): a group's
 *           eight first lines are contiguous (1 KB), then its eight J_l tails (64 bytes each), its eight centres (32 bytes
 *           each) and the pad -- the projection-only passes then touch whole kilobytes instead of every other line (and the
 *           Jacobian 1.5 KB, the visibility predicate 1.25 KB of every 2), which the 256-MB Infinity Cache needs to hold their
 *           inputs.  ALLOCATE IT FOR WHOLE GROUPS: C2B_CAMBLK_DOUBLES * (n_cam rounded up to a multiple of 8)
 *           doubles (= c2b_camblk_doubles(n_cam)), 256-byte aligned; a table is written by c2b_camblk_from_* for its own n_cam cameras (camera c of the
 *           call = row c of the table) and is never sliced or copied by camera range -- a shard prepares its own.
 *   cen4    the cameras' centres alone, [n_cam][4] doubles (x y z 0: 32-byte rows like pts4; 16-byte aligned), written
 *           by the same launch that derives camblk.  What the statistics read: 32 bytes per camera instead of a
 *           128-byte line of the 256-byte record (84.5 MB instead of 148.7 MB per pass at --blocks 128).  Optional
 *           everywhere: NULL = "not kept" (the statistics then read the centre field of camblk).
 *   Jacobian (NOT in the reference; build-defined): residual r = project(project_world(X))
 *           - uv_obs (sign of src/baproblem.rs:273).  Jc[n_obs][2][9] row-major, columns in
 *           to_vec order (w0 w1 w2 t0 t1 t2 f k1 k2); Jp[n_obs][2][3] (d/dX).
 *
 * The boundary comes in three headers: THIS one -- Level 0 (stateless launchers over device pointers), the collectives, Level 1
 * (a resident BAProblem): the stable device boundary; city2ba_hip_host.h -- the host-side rows (CPU C++ of the generators, cull,
 * file IO: they never touch the GPU); city2ba_hip_experimental.h -- measurement aids, launch diagnostics and the f32 extension,
 * which may change between rounds.
 *
 * ---- index of entry points by level (tools/abi_index.py) ----
 *   library (4):
 *     version, abi_version, last_error, device_count
 *   Level 0: workspace, camera records, points, rows (14):
 *     workspace_bytes, workspace_init, cameras_from_bal, cameras_to_bal, camblk_doubles, camblk_from_state,
 *     camblk_from_bal, cameras_from_position_direction, project_world, to_world, cameras_transform, points_pad,
 *     points_unpad, expand_rows
 *   Level 0: per-observation passes (cam_idx and row-structure forms) (15):
 *     project, reprojection_error_sum, rows_tiles_bytes, rows_pack, project_rows, reprojection_error_sum_rows,
 *     visibility_rows, visibility_rows_bits, reprojection_error_sums2_rows, add_noise_observations_error_sums2_rows,
 *     jacobian_stream_policy, residual_jacobian_rows, residual_jacobian, error_sum_finish, residual_jacobian_sum
 *   Level 0: Jacobian output sets and calibration (5):
 *     jacobian_outputs_alloc, jacobian_outputs_pointers, jacobian_outputs_store_rate, jacobian_outputs_free,
 *     residual_jacobian_rows_placed
 *   Level 0: visibility sweeps and occlusion (6):
 *     visibility_pairs, visibility_dense_tiles, visibility_dense_count, visibility_dense_fill, occlusion_filter,
 *     occlusion_filter_bvh
 *   Level 0: statistics (5):
 *     stats, stats_partial_pass1, stats_partial_pass2, stats_combine_shares, stats_finish_shares
 *   collectives (RCCL) and sharded Level-0 forms (12):
 *     comm_unique_id, comm_init_rank, comm_init_all, comm_group_start, comm_group_end, comm_info,
 *     comm_all_reduce_sum_f64, comm_all_gather_f64, comm_destroy, stats_sharded, add_drift_sharded,
 *     add_noise_entities_sharded
 *   Level 0: noise (5):
 *     add_drift, add_drift_normalized, add_noise_entities, add_noise_observations, add_sin_noise
 *   the shard map (1):
 *     partition_cameras
 *   Level 1: a resident BAProblem (46):
 *     problem_create, problem_destroy, problem_options_init, problem_set_options, problem_get_options,
 *     host_set_io_threads, problem_upload, problem_upload_bal, problem_synthetic_grid_layout,
 *     problem_synthetic_line_layout, problem_sizes, problem_download, problem_download_bal, problem_write,
 *     problem_read, problem_from_position_direction, problem_centers, problem_project,
 *     problem_total_reprojection_error, problem_total_reprojection_error_sharded,
 *     problem_total_reprojection_errors_l1_l2, problem_total_reprojection_errors_l1_l2_sharded,
 *     problem_residual_jacobian, problem_residual_jacobian_device, host_alloc, host_free, problem_stats, problem_cull,
 *     problem_largest_connected_component, problem_remove_singletons, problem_adopt_visibility,
 *     problem_download_graph, problem_export_device, problem_visibility_pairs, problem_visibility_pairs_compact,
 *     problem_visibility_within_distance, problem_generate_world_points, problem_visibility_dense,
 *     problem_visibility_dense_fetch, problem_visibility_dense_occlude, problem_visibility_dense_occlude_bvh,
 *     problem_add_drift, problem_add_drift_normalized, problem_add_noise, problem_add_sin_noise,
 *     problem_add_noise_errors_l1_l2
 *   Level 1: one shard of a larger problem (6):
 *     problem_set_shard, problem_stats_sharded, problem_add_drift_sharded, problem_add_noise_sharded,
 *     problem_add_sin_noise_sharded, problem_add_noise_errors_l1_l2_sharded
 *   -- city2ba_hip.h: 119 entry points --
 *   city2ba_hip_host.h: host-side rows (CPU; never touch the GPU) (42):
 *     synthetic_grid_sizes, synthetic_grid_layout, synthetic_line_layout, candidate_pairs, pairs_count, pairs_cam_idx,
 *     pairs_pt_idx, pairs_free, obj_load, obj_model_count, obj_model_name, obj_model_sizes, obj_model_copy,
 *     obj_move_to_origin, obj_triangles, obj_free, generate_cameras_path, generate_cameras_poisson,
 *     generate_cameras_poisson_bvh, modify_intrinsics, generate_world_points, cull, largest_connected_component,
 *     remove_singletons, add_incorrect_correspondences, drop_features, split_landmarks, join_landmarks, bal_read,
 *     bal_sizes, bal_copy, bal_close, bal_write, bal_read_as, bal_write_as, format_f64, parse_f64, ply_write,
 *     bvh_build, bvh_sizes, bvh_copy, bvh_free
 *   city2ba_hip_experimental.h: diagnostics, calibration, f32 extension (15):
 *     workspace_selfcheck, comm_backend, jacobian_tiles_per_wave, jacobian_launch_shape, jacobian_outputs_log,
 *     jacobian_outputs_set_store_rate, calib_store_pattern, calib_copy, convert_f64_to_f32, convert_f32_to_f64,
 *     stats_f32, add_drift_f32, add_drift_normalized_f32, add_noise_entities_f32, add_sin_noise_f32
 *   (176 entry points in all; names above without their c2b_ prefix)
 * ---- end of index ----
 *
 * Every function returns C2B_OK or a negative status; c2b_last_error() gives the text.
 * No exception or abort crosses this boundary.  A context/problem is not thread-safe.
 */
#ifndef CITY2BA_HIP_H
#define CITY2BA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2B_OK                       0
#define C2B_ERR_INVALID_ARGUMENT    -1
#define C2B_ERR_INDEX_OUT_OF_RANGE  -2  /* the asserts of src/baproblem.rs:345-346, 365-369 */
#define C2B_ERR_HIP                 -3
#define C2B_ERR_OOM                 -4
#define C2B_ERR_NO_DEVICE           -5
#define C2B_ERR_RCCL                -6  /* a collective failed, or RCCL could not be loaded */

#define C2B_CAMBLK_DOUBLES 32
#define C2B_CAMBLK_GROUP 8          /* cameras per group of the blocked table: allocate C2B_CAMBLK_DOUBLES * ((n_cam + 7) / 8 * 8) doubles */
#define C2B_STATS_DOUBLES  20  /* mean[3] std[3] min[3] max[3] dim[3] origin[3] origin_idx |std| */

const char *c2b_version(void);
/* The ABI's number: bumped whenever an existing entry's argument list or a buffer's layout changes (r05 did both without a number:
 * camblk became a blocked table, c2b_stats* and the camera-table writers gained a pointer in the middle of their lists).  A host
 * compares c2b_abi_version() with the C2B_ABI_VERSION it was compiled against, once, before its first call. */
#define C2B_ABI_VERSION 6
int c2b_abi_version(void);
const char *c2b_last_error(void);
int c2b_device_count(int *count);

/* ===================================================================================== *
 * Level 0 -- stateless launchers.  Every pointer is a DEVICE pointer, every call is
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream).
 * Indices are NOT validated here (they live on the device): cam_idx[i] < n_cam and
 * pt_idx[i] < n_pts are the caller's contract; Level 1 validates like the reference's asserts.
 * Observations may come in any order; camera-major (CSR) order is the fast path.
 * ===================================================================================== */

/* Bytes of scratch the reductions below need for a problem with n_obs observations: allocate that many bytes of
 * device memory, 128-byte aligned, and call c2b_workspace_init ONCE before the first use.  The workspace holds the
 * per-workgroup partials of the in-kernel folds AND the arrival counters that count them (so two launches can only
 * ever meet on the same counters by sharing a workspace, which the partials already forbid): one workspace serves one
 * launch at a time -- launches that may run concurrently (other streams, other threads, a replayed graph next to eager
 * launches) need a workspace each.  Every fold leaves the counters zero, so a workspace is initialised once, not per
 * launch, and a captured launch replays cleanly.  A workspace that was never initialised makes every sum NaN (the
 * kernels check a magic word), never a stale or partial value. */
int64_t c2b_workspace_bytes(int64_t n_obs);
/* zero the arrival counters, write the magic: one tiny launch on `stream` (capture-safe) */
int c2b_workspace_init(void *workspace, void *stream);

/* SnavelyCamera::from_vec / from_rodrigues (src/baproblem.rs:78-90, 180-186) */
int c2b_cameras_from_bal(const double *bal9, int64_t n_cam, double *cam15, void *stream);
/* SnavelyCamera::to_vec / to_rodrigues (src/baproblem.rs:93-102, 189-202) */
int c2b_cameras_to_bal(const double *cam15, int64_t n_cam, double *bal9, void *stream);
/* derive camblk (and, when cen4 != NULL, the compact centre table) from the in-memory state; Jacobian columns refer
 * to w = to_rodrigues(R).  Run it again after anything moved the cameras (c2b_add_drift*, c2b_add_noise_entities,
 * c2b_add_sin_noise, c2b_cameras_transform mutate cam15, the truth state): both tables are derived data. */
/* camblk_doubles = the CAPACITY of the buffer behind camblk, in doubles: rejected (C2B_ERR_INVALID_ARGUMENT, nothing written) when
 * it is less than c2b_camblk_doubles(n_cam) -- the table holds WHOLE groups of 8 cameras, so a buffer sized 32 * n_cam is too
 * small whenever n_cam is not a multiple of 8.  (r06: these two replace c2b_camblk_from_state / _bal, whose argument list
 * had changed in place in r05 -- a binding written against the older list must fail to link, not write out of bounds.) */
int64_t c2b_camblk_doubles(int64_t n_cam);
int c2b_camblk_from_state(const double *cam15, int64_t n_cam, double *camblk, int64_t camblk_doubles, double *cen4, void *stream);
/* derive camblk (and cen4) from 9-vectors; R = from_rodrigues(w), Jacobian columns refer to that w */
int c2b_camblk_from_bal(const double *bal9, int64_t n_cam, double *camblk, int64_t camblk_doubles, double *cen4, void *stream);
/* Camera::from_position_direction (src/baproblem.rs:153-159): pos [n][3], dir [n][9] col-major */
int c2b_cameras_from_position_direction(const double *pos3, const double *dir9, int64_t n_cam,
                                        double *cam15, void *stream);
/* Camera::project_world (src/baproblem.rs:141-143) and Camera::to_world (:173-175), batched over pairs:
 * pair i = (camera cam_idx[i], point p3[i]) -> out3[i] */
int c2b_project_world(const double *cam15, const uint32_t *cam_idx, const double *p3, int64_t n,
                      double *out3, void *stream);
int c2b_to_world(const double *cam15, const uint32_t *cam_idx, const double *p3, int64_t n,
                 double *out3, void *stream);
/* Camera::transform (src/baproblem.rs:165-171) of every camera, in place: dir' = dir * delta_dir[i],
 * loc' = -dir * (center + delta_loc[i]) (the OLD dir, like the reference) */
int c2b_cameras_transform(double *cam15, const double *delta_dir9, const double *delta_loc3, int64_t n_cam,
                          void *stream);
/* pts3 [n][3] -> pts4 [n][4] and back */
int c2b_points_pad(const double *pts3, int64_t n_pts, double *pts4, void *stream);
int c2b_points_unpad(const double *pts4, int64_t n_pts, double *pts3, void *stream);
/* COO expansion of row_ptr: cam_idx[o] = c  iff  row_ptr[c] <= o + obs_base < row_ptr[c+1] */
int c2b_expand_rows(const uint64_t *row_ptr, int64_t n_cam, int64_t obs_base, int64_t n_obs,
                    uint32_t *cam_idx, void *stream);

/* Camera::project(Camera::project_world(p)) per observation (src/baproblem.rs:141-151, :272) */
int c2b_project(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                const uint32_t *pt_idx, int64_t n_obs, double *uv_out, void *stream);

/* Sum over observations of |du|^norm + |dv|^norm, i.e. BAProblem::total_reprojection_error
 * (src/baproblem.rs:265-279) WITHOUT the final powf(1/norm): the quantity that is
 * all-reduced across GPUs.  ONE launch: every workgroup leaves a partial, the last one to arrive folds them
 * in index order (no float atomics; the same problem on the same device gives the same bits every run). */
int c2b_reprojection_error_sum(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                               const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                               double norm, void *workspace, double *out_sum, void *stream);

/* The same three per-observation passes for a camera-major list addressed through its ROW STRUCTURE -- the
 * reference's own representation, one list per camera (vis_graph, src/baproblem.rs:256-260) -- instead of a 4-byte
 * camera index per observation.  row_ptr [n_cam + 1] (row_ptr[0] = 0, row_ptr[n_cam] = n_obs, non-decreasing; lists
 * may be empty) and `tiles`, c2b_rows_tiles_bytes(n_obs) bytes (16-byte aligned) filled once per list by
 * c2b_rows_pack: per 64 observations the camera of the first one and a mask of the lanes that open a new list.
 * Results are bit-identical to c2b_project / c2b_reprojection_error_sum / c2b_visibility_pairs on the expanded list;
 * the kernels read 0.25 instead of 4 bytes of camera addressing per observation (SURVEY section 8(d) counts the
 * algorithmic bytes of these passes this way: 4 B of point index per observation, the row structure once). */
int64_t c2b_rows_tiles_bytes(int64_t n_obs);
int c2b_rows_pack(const uint64_t *row_ptr, int64_t n_cam, int64_t n_obs, void *tiles, void *stream);
int c2b_project_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                     const void *tiles, const uint32_t *pt_idx, int64_t n_obs, double *uv_out, void *stream);
int c2b_reprojection_error_sum_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                    const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                    double norm, void *workspace, double *out_sum, void *stream);
int c2b_visibility_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                        const void *tiles, const uint32_t *pt_idx, int64_t n_pairs, double max_dist,
                        double *uv_out, uint8_t *keep, void *stream);
/* the same pass with the keep mask as one 64-bit word per 64 pairs -- bit l of keep_bits[t] = pair 64 t + l; the bits
 * past n_pairs in the last word are 0 -- written as ONE ballot per wave and tile instead of a byte per lane:
 * keep_bits (device, 8-byte aligned) holds ceil(n_pairs / 64) words.  Same predicate, same uv_out. */
int c2b_visibility_rows_bits(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                             const void *tiles, const uint32_t *pt_idx, int64_t n_pairs, double max_dist,
                             double *uv_out, uint64_t *keep_bits, void *stream);
/* total_reprojection_error's numerators for norm 1 AND norm 2 from ONE pass -- run_noise evaluates exactly this pair,
 * back to back on the same data, before and after the noise (src/bin/city2ba.rs:283-287, 350-354):
 * out_sums[0] = sum |du| + |dv|, out_sums[1] = sum du^2 + dv^2 (device pointer, 2 doubles).  Each is bit-identical to
 * c2b_reprojection_error_sum_rows with that norm (same grid, same fold order per sum, one arrival count). */
int c2b_reprojection_error_sums2_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                      const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                      void *workspace, double *out_sums, void *stream);
/* noise::add_noise's observation pass (src/noise.rs:152-170) fused with the two error sums run_noise evaluates right
 * after it (src/bin/city2ba.rs:350-354): uv [n_obs][2] is perturbed in place exactly as c2b_add_noise_observations
 * would (same draws, counter = obs_base + i; the stored bits are identical) and out_sums = the L1 / L2 sums of the
 * PERTURBED observations against camblk / pts4 as passed (run c2b_add_noise_entities and re-derive camblk first). */
int c2b_add_noise_observations_error_sums2_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr,
                                                int64_t n_cam, const void *tiles, const uint32_t *pt_idx, double *uv,
                                                int64_t n_obs, int64_t obs_base, double observations_std, uint64_t seed,
                                                void *workspace, double *out_sums, void *stream);
/* c2b_residual_jacobian / _sum in the row-structure form, for the whole list (obs_base = 0) or a slice of it: tiles,
 * pt_idx, uv_obs, r, Jc, Jp all point at observation obs_base (a multiple of 64) of the list row_ptr describes and
 * n_obs observations are processed.  workspace == NULL: no error sum.  Otherwise sum |r|^norm goes to out_sum[0]
 * (device pointer), or into the workspace (c2b_error_sum_finish) when out_sum is NULL.
 * n_pts = rows of pts4, or 0 if unknown: together with n_cam and n_obs it sizes the launch's working set against the
 * 256 MB Infinity Cache, which decides whether the once-read streams (observed uv, point index) bypass the caches so
 * that the camera and point tables stay in them (c2b_jacobian_stream_policy returns that decision: 0 none, 2 uv,
 * 3 uv and index; results are identical under every policy). */
int c2b_jacobian_stream_policy(int64_t n_obs, int64_t n_cam, int64_t n_pts);
int c2b_residual_jacobian_rows(const double *camblk, const double *pts4, int64_t n_pts, const uint64_t *row_ptr, int64_t n_cam,
                               const void *tiles, int64_t obs_base, const uint32_t *pt_idx, const double *uv_obs,
                               int64_t n_obs, double *r, double *Jc, double *Jp, double norm, void *workspace,
                               double *out_sum, void *stream);

/* residual + 2x9 camera block + 2x3 point block per observation (no reference equivalent).
 * With workspace != NULL the same launch also folds sum |du|^norm + |dv|^norm (fused error reduce, same fold
 * as c2b_reprojection_error_sum) into the workspace; c2b_error_sum_finish copies it out. */
int c2b_residual_jacobian(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                          const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                          double *r, double *Jc, double *Jp,
                          double norm, void *workspace, void *stream);
/* the sum left in the workspace by the last c2b_residual_jacobian on this stream -> out_sum[0] (8-byte copy) */
int c2b_error_sum_finish(const void *workspace, int64_t n_obs, double *out_sum, void *stream);
/* c2b_residual_jacobian + the folded error sum straight into out_sum[0] (device pointer): the whole
 * total_reprojection_error numerator and the Jacobian in ONE launch.  workspace must not be NULL. */
int c2b_residual_jacobian_sum(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                              const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                              double *r, double *Jc, double *Jp,
                              double norm, void *workspace, double *out_sum, void *stream);

/* Output arrays of the residual + Jacobian launch -- r [n][2], Jc [n][2][9], Jp [n][2][3], 208 bytes per observation --
 * as device allocations CHOSEN FOR STREAMING-STORE SPEED.  On MI355X the very same launch runs ~20 % faster or slower
 * depending only on which allocation its outputs live in; nothing visible from user space
 * predicts it, so this entry allocates a set, times the kernel's own store pattern into it (~4 ms), keeps it if it
 * streams at fast_store_GBs (<= 0: 7000) or better and otherwise holds it and tries again, at most max_attempts
 * (clamped to 1..64, and to what fits three quarters of the free device memory) times; the best set wins, the others are
 * freed before it returns.  (r05: fast sets exist on every device mapped -- 8-12 of 60 consecutive 4-GB sets -- but not always among
 * the first eight: a depth of 32-48 finds one.)  max_attempts = 1 takes the
 * first set (its rate is still measured: c2b_residual_jacobian_rows_placed chooses its workgroup shape by it); n_obs
 * < 10^6 allocates without measuring.  An attempt that runs out of memory ends the search with the best set so far.
 * TRANSIENT FOOTPRINT: every rejected set (and, for sets below 2 GiB, a 2-GiB filler per reject) is held until the call
 * returns -- up to max_attempts x max(208 B x n_obs, 2 GiB), never more than three quarters of the memory that was free
 * at the call; the free memory is asked for again before every further attempt and the search stops once less than a
 * quarter of that (plus one set) is left, so a process, rank or allocator sharing the device is not driven out of memory.
 * Ask for depth deliberately: 8 is what the bindings default to, 32-48 what a long-lived solver or a benchmark asks for.
 * Synchronises `stream`.  The handle owns the memory until c2b_jacobian_outputs_free.
 * (No reference counterpart: the Jacobian itself is build-defined; a Rust host holds the handle next to its
 * device mirror, INTEGRATION.md.) */
typedef struct c2b_jacobian_outputs c2b_jacobian_outputs;
int c2b_jacobian_outputs_alloc(int64_t n_obs, int max_attempts, double fast_store_GBs, void *stream,
                               c2b_jacobian_outputs **out);
int c2b_jacobian_outputs_pointers(const c2b_jacobian_outputs *h, double **r, double **Jc, double **Jp);
/* the kept set's measured store rate in GB/s (0: not measured) */
int c2b_jacobian_outputs_store_rate(const c2b_jacobian_outputs *h, double *store_GBs);
void c2b_jacobian_outputs_free(c2b_jacobian_outputs *h);
/* c2b_residual_jacobian_rows over the WHOLE list (obs_base = 0, n_obs = the set's observation count) INTO a placed set:
 * r / Jc / Jp are the handle's arrays, and the launch takes the workgroup shape that is fastest for stores of the speed
 * measured for that set (c2b_jacobian_launch_shape: 256-thread workgroups with one tile per wave into a slow-store
 * set, 2.3 % faster there; 512 threads x two tiles otherwise).  Same results, bit for bit, under either shape; the
 * folded sum's rounding follows the grid, as always.  The current device must be the set's. */
int c2b_residual_jacobian_rows_placed(const double *camblk, const double *pts4, int64_t n_pts, const uint64_t *row_ptr,
                                      int64_t n_cam, const void *tiles, const uint32_t *pt_idx, const double *uv_obs,
                                      int64_t n_obs, const c2b_jacobian_outputs *outputs, double norm, void *workspace,
                                      double *out_sum, void *stream);


/* visibility predicate of the generators (src/synthetic.rs:285-291, 368-375;
 * src/generate.rs:448-454): keep = |center - p| < max_dist && q.z <= 0 && |u|,|v| <= 1.
 * uv_out is NaN where project() was not reached. */
int c2b_visibility_pairs(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                         const uint32_t *pt_idx, int64_t n_pairs, double max_dist,
                         double *uv_out, uint8_t *keep, void *stream);

/* The O(cameras x points) sweep of the mesh generator (visibility_graph, src/generate.rs:446-469)
 * WITHOUT its Embree occlusion stream: every camera against every point with the predicate above,
 * survivors per camera in ascending point order (the reference's push order).  Two passes:
 *   _count: survivors per (camera, 256-point tile) -> tile_counts[n_cam * tiles] (turned into per-camera
 *           exclusive offsets in place) and row_ptr[n_cam + 1]; cam_total[n_cam] is scratch;
 *   _fill:  after the caller has read row_ptr[n_cam] and allocated pt_idx / uv of that size. */
int64_t c2b_visibility_dense_tiles(int64_t n_pts);
int c2b_visibility_dense_count(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts,
                               double max_dist, uint32_t *tile_counts, uint64_t *cam_total,
                               uint64_t *row_ptr, void *stream);
int c2b_visibility_dense_fill(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts,
                              double max_dist, const uint32_t *tile_offsets, const uint64_t *row_ptr,
                              uint32_t *pt_idx, double *uv, void *stream);

/* Occlusion test of the mesh generator (src/generate.rs:455-476) with a brute-force stand-in for Embree's
 * occluded stream: per observation a f32 ray from the camera centre towards the point, tfar = |dir| - 1e-6;
 * keep[i] = 0 iff some triangle of tri9 [n_tri][9] (device, f32) is hit with 0 < t <= tfar. */
int c2b_occlusion_filter(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                         const uint32_t *pt_idx, int64_t n_obs, const float *tri9, int64_t n_tri,
                         uint8_t *keep, void *stream);
/* The same filter through a bounding-volume hierarchy, for meshes too large for the all-triangles loop (the
 * reference commits its meshes to an Embree scene, src/bin/city2ba.rs:515-521).  c2b_bvh_build runs on the HOST
 * over host triangles; c2b_bvh_copy fills host buffers of n_nodes * C2B_BVH_NODE_BYTES and n_slots *
 * C2B_BVH_TRI_BYTES bytes (and, optionally, order[n_slots] = input triangle of each slot) for the caller to place
 * in device memory (16-byte aligned); c2b_occlusion_filter_bvh traverses them.  Leaves run the triangle test of
 * c2b_occlusion_filter, so both filters return the same mask.  overflow (device, one word, zeroed by the caller) is
 * set to 1 if a ray's traversal needed more than the 64-entry stack -- impossible for a hierarchy from c2b_bvh_build
 * (which refuses deeper ones), possible for caller-made node arrays; the mask is then invalid.  Level 1 turns a set
 * flag into C2B_ERR_INVALID_ARGUMENT. */
#define C2B_BVH_NODE_BYTES 64
#define C2B_BVH_TRI_BYTES 48
typedef struct c2b_bvh c2b_bvh;
/* (c2b_bvh_build / _sizes / _copy / _free: host-side rows, city2ba_hip_host.h) */
int c2b_occlusion_filter_bvh(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                             const uint32_t *pt_idx, int64_t n_obs, const void *nodes, int64_t n_nodes,
                             const void *tris, int64_t n_slots, uint8_t *keep, uint32_t *overflow, void *stream);

/* BAProblem::mean/std/extent/dimensions (src/baproblem.rs:282-337) + add_drift's origin
 * (src/noise.rs:75-87) over camera centers ++ points, into stats[C2B_STATS_DOUBLES].  ONE launch: mean, min, max and
 * the origin as the reference folds them, the standard deviation from per-thread (count, mean, M2) triples merged by
 * Chan's pairwise update -- equal to the reference's second pass around the finished mean up to rounding. */
int c2b_stats(const double *camblk, const double *cen4 /* may be NULL */, int64_t n_cam, const double *pts4, int64_t n_pts,
              void *workspace, double *stats, void *stream);

/* The same statistics when cameras are sharded over GPUs: this rank holds cameras
 * [cam_base, cam_base + n_cam) of n_cam_global and reduces points [pt_base, pt_base + n_pts) of the replicated table
 * (pts4 points at the first of them).  Pass 1 leaves this shard's share in part[C2B_STATS_DOUBLES]:
 * [0..2] sum of x / n_entities_global, [6..8] min, [9..11] max, [15..17] its entity closest to the world origin,
 * [18] that entity's GLOBAL index (cameras first, then points; -1 if the shard is empty), [19] its distance.
 * The host sums [0..2] over ranks, takes min / max, picks the smallest [19] (ties: the larger [18], like fold1 at
 * src/noise.rs:80-86) and hands the global mean to pass 2, which leaves the shard's three sums of squared
 * deviations in sumsq3; std = sqrt(sum over ranks / n_entities_global).  city2ba_amd/dist.py: stats_sharded. */
int c2b_stats_partial_pass1(const double *camblk, const double *cen4, int64_t n_cam, int64_t cam_base, int64_t n_cam_global,
                            const double *pts4, int64_t n_pts, int64_t pt_base, int64_t n_entities_global,
                            void *workspace, double *part, void *stream);
int c2b_stats_partial_pass2(const double *camblk, const double *cen4, int64_t n_cam, const double *pts4, int64_t n_pts,
                            const double *mean3, void *workspace, double *sumsq3, void *stream);
/* the host halves of the two steps above (plain CPU arithmetic; every rank computes the same bits from the same
 * gathered rows): shares [world][20] in rank order -> stats[0..2] mean, [6..8] min, [9..11] max, [12..14] extent,
 * [15..17] origin, [18] its global index, [19] its distance;  sumsq [world][3] in rank order -> stats[3..5] std,
 * stats[19] |std| (overwriting the distance, like c2b_stats). */
int c2b_stats_combine_shares(const double *shares, int world, double *stats);
int c2b_stats_finish_shares(const double *sumsq, int world, int64_t n_entities_global, double *stats);

/* ===================================================================================== *
 * Collectives of the sharded path -- RCCL over xGMI behind this ABI.
 * One process (or thread) per GPU holds one c2b_comm; rank k owns a contiguous camera range
 * (c2b_partition_cameras), every point is replicated, outputs stay sharded.  What the ranks
 * exchange: the 8-byte sum behind BAProblem::total_reprojection_error (src/baproblem.rs:265-279;
 * the reference itself is single-process, rayon only: src/synthetic.rs:268-269) and the 20- /
 * 3-double statistics shares above.  RCCL is loaded on first use (dlopen of librccl.so.1; the
 * environment variable C2B_RCCL_LIB names another file), so single-GPU hosts never load it; a
 * failure to load or any RCCL error is C2B_ERR_RCCL with the text in c2b_last_error().
 * All collectives are asynchronous on `stream` and may be captured in a HIP graph.
 * ===================================================================================== */
#define C2B_COMM_ID_BYTES 128
typedef struct c2b_comm c2b_comm;
/* Rank 0 makes the id and hands its C2B_COMM_ID_BYTES bytes to every rank through whatever channel the host has
 * (a file, a pipe, MPI, the torch store); then EVERY rank calls c2b_comm_init_rank with it (collective; makes `device`
 * the calling thread's current device). */
int c2b_comm_unique_id(void *id128);
int c2b_comm_init_rank(const void *id128, int rank, int world, int device, c2b_comm **out);
/* One process driving n_dev GPUs (SURVEY 8(b)'s ctx_create(n_dev, dev_ids)): out[n_dev] communicators, rank i on
 * device dev_ids[i] (NULL: devices 0..n_dev-1).  Collectives issued from ONE thread for several of them must sit
 * between c2b_comm_group_start / _end. */
int c2b_comm_init_all(int n_dev, const int *dev_ids, c2b_comm **out);
int c2b_comm_group_start(void);
int c2b_comm_group_end(void);
int c2b_comm_info(const c2b_comm *c, int *rank, int *world, int *device);
/* in place: buf[0..n) <- sum over ranks (device pointer) */
int c2b_comm_all_reduce_sum_f64(c2b_comm *c, double *buf, int64_t n, void *stream);
/* recv[world][n_per_rank] <- every rank's send[n_per_rank], rank order (device pointers) */
int c2b_comm_all_gather_f64(c2b_comm *c, const double *send, int64_t n_per_rank, double *recv, void *stream);
void c2b_comm_destroy(c2b_comm *c);
/* BAProblem::mean/std/extent/dimensions + add_drift's origin over SHARDED cameras in one call: pass 1, all-gather,
 * host combine, pass 2, all-gather, host finish.  camblk = this rank's cameras [cam_base, cam_base + n_cam) of
 * n_cam_global, pts4 = the whole replicated table (rank r reduces its r-th slice).  Collective and synchronous;
 * stats (DEVICE, C2B_STATS_DOUBLES) holds the same bits on every rank when it returns. */
int c2b_stats_sharded(c2b_comm *c, const double *camblk, const double *cen4, int64_t n_cam, int64_t cam_base, int64_t n_cam_global,
                      const double *pts4, int64_t n_pts, void *workspace, double *stats, void *stream);

/* add_drift[_normalized] / add_noise on a shard: cam15 holds cameras [cam_base, cam_base + n_cam) and every draw
 * is keyed by the GLOBAL camera index, so the sharded result equals the unsharded one row for row; pts4 is the
 * whole replicated table (every rank perturbs it identically: same counters, no traffic).  stats = the GLOBAL
 * statistics (device).  normalized != 0: direction and scale from stats (src/noise.rs:47-56), dir_* ignored. */
int c2b_add_drift_sharded(double *cam15, int64_t n_cam, int64_t cam_base, double *pts4, int64_t n_pts,
                          const double *stats, int normalized, double strength, double angle_strength, double std,
                          double dir_x, double dir_y, double dir_z, uint64_t seed, void *stream);
int c2b_add_noise_entities_sharded(double *cam15, int64_t n_cam, int64_t cam_base, double *pts4, int64_t n_pts,
                                   const double *stats, double translation_std, double rotation_std,
                                   double point_std, uint64_t seed, void *stream);

/* noise::add_drift (src/noise.rs:68-116), in place.  `origin` = device pointer to 3 doubles
 * (stats + 15).  Draws: Philox4x32-10 keyed by (seed; stream, entity, slot) -- the reference
 * is unseeded, see DESIGN.md. */
int c2b_add_drift(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *origin,
                  double strength, double angle_strength, double std, double dir_x, double dir_y,
                  double dir_z, uint64_t seed, void *stream);
/* noise::add_drift_normalized (src/noise.rs:47-56): dir and scale come from stats (device) */
int c2b_add_drift_normalized(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts,
                             const double *stats, double strength, double angle_strength,
                             double std, uint64_t seed, void *stream);
/* noise::add_noise (src/noise.rs:119-177): cameras + points (bal_std = stats[19], device) ... */
int c2b_add_noise_entities(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts,
                           const double *stats, double translation_std, double rotation_std,
                           double point_std, uint64_t seed, void *stream);
/* ... and every observation (src/noise.rs:152-170); obs_base = global index of uv[0] */
int c2b_add_noise_observations(double *uv, int64_t n_obs, int64_t obs_base,
                               double observations_std, uint64_t seed, void *stream);
/* noise::add_sin_noise (src/noise.rs:388-416); dimensions read from stats (device) */
int c2b_add_sin_noise(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts,
                      const double *stats, double dir_x, double dir_y, double dir_z,
                      double ndir_x, double ndir_y, double ndir_z, double strength,
                      double frequency, void *stream);


/* contiguous camera ranges with ~equal observation counts (host pointers): the multi-GPU
 * shard map.  bounds[n_parts+1]; part k owns cameras [bounds[k], bounds[k+1]). */
int c2b_partition_cameras(const uint64_t *row_ptr, int64_t n_cam, int n_parts, int64_t *bounds);

/* ===================================================================================== *
 * Level 1 -- a BAProblem resident on one device.  Pointers are HOST pointers; calls are
 * synchronous (results are valid on return).  Mirrors BAProblem<SnavelyCamera>.
 * ===================================================================================== */
typedef struct c2b_problem c2b_problem;

int c2b_problem_create(int device, c2b_problem **out);
void c2b_problem_destroy(c2b_problem *p);

/* Behaviour switches of a problem's calls -- which of two equivalent routes a call takes, and how many host threads it
 * may use.  They are ARGUMENTS, set through the ABI per problem: the library reads no environment variable for them
 * (rounds 1-4 did: invisible to, and not settable per call by, a host that binds this header).  Every route gives the
 * same files and the same resident state; the switches exist for tests, comparisons and odd inputs.
 * c2b_problem_options_init writes the defaults; set_options replaces all of them (get, change, set to change one). */
typedef struct c2b_problem_options {
    int32_t host_text;              /* != 0: .bal text formatted / parsed by the host code over a download / an upload
                                       (c2b_problem_write / _read; default 0 = on the device) */
    int32_t text_device_strict;     /* != 0: a text file the device parser declines is an ERROR instead of going to the host
                                       parser (tests: proves which parser ran; default 0) */
    int32_t read_threads;           /* reader threads of c2b_problem_read's pinned ring (0 = default: 3) */
    int32_t io_threads;             /* threads of the host text formatter / parser when this problem's calls use it
                                       (0 = default: c2b_host_set_io_threads, else the usable cores, at most 16) */
    int32_t rank_sort_max_row;      /* c2b_problem_visibility_within_distance: rows longer than this are sorted on the host
                                       (0 = default: 2048) */
    int32_t reserved;               /* 0 */
    int64_t text_device_min_bytes;  /* text files smaller than this go to the host parser (< 0 = default: 65536) */
} c2b_problem_options;
void c2b_problem_options_init(c2b_problem_options *o);
int c2b_problem_set_options(c2b_problem *p, const c2b_problem_options *o);
int c2b_problem_get_options(const c2b_problem *p, c2b_problem_options *o);
/* process-wide: threads of the host text formatter / parser for the handle-less entries (c2b_bal_read / _write) and for
 * problems that do not say (n < 1 = default: the usable cores, at most 16) */
void c2b_host_set_io_threads(int n);

/* BAProblem::from_visibility (src/baproblem.rs:360-376): validates like its asserts */
int c2b_problem_upload(c2b_problem *p, int64_t n_cam, const double *cams15, int64_t n_pts,
                       const double *pts3, const uint64_t *row_ptr, const uint64_t *pt_idx,
                       const double *uv);
/* same, cameras as 9-vectors (what from_file_* parses, src/baproblem.rs:605-608) */
int c2b_problem_upload_bal(c2b_problem *p, int64_t n_cam, const double *bal9, int64_t n_pts,
                           const double *pts3, const uint64_t *row_ptr, const uint64_t *pt_idx,
                           const double *uv);
/* The layout loops of synthetic_grid / synthetic_line (src/synthetic.rs:178-258, :323-344) evaluated on the device,
 * straight into the resident problem: cameras through Camera::from_position_direction (src/baproblem.rs:153-159),
 * points, an empty vis_graph (c2b_problem_visibility_within_distance fills it).  Entity for entity and bit for bit what
 * c2b_synthetic_*_layout + c2b_problem_from_position_direction + c2b_problem_upload produce; nothing crosses PCIe. */
int c2b_problem_synthetic_grid_layout(c2b_problem *p, int64_t cameras_per_block, int64_t points_per_block, int64_t blocks,
                                      double block_length, double block_inset, double camera_height, double point_height);
int c2b_problem_synthetic_line_layout(c2b_problem *p, int64_t n_cam, int64_t n_pts, double length, double point_offset,
                                      double camera_height, double point_height);
int c2b_problem_sizes(const c2b_problem *p, int64_t *n_cam, int64_t *n_pts, int64_t *n_obs);
/* any of the outputs may be NULL */
int c2b_problem_download(c2b_problem *p, double *cams15, double *pts3, double *uv);
int c2b_problem_download_bal(c2b_problem *p, double *bal9);
/* BAProblem::write / write_text / write_binary (src/baproblem.rs:709-785) of the RESIDENT problem: format -1 = by
 * extension (.bal text, .bbal binary, anything else an error, like the reference), 0 = text, 1 = binary.  The binary
 * image -- to_vec of every camera, per-camera counts, big-endian words -- is assembled on the device and leaves through
 * a ring of pinned chunks; so is the text image (csrc/text_kernels.hpp: every decimal is formatted on the device).  The
 * bytes equal c2b_bal_write's on the downloaded arrays. */
int c2b_problem_write(c2b_problem *p, const char *path, int format);
/* BAProblem::from_file / from_file_text / from_file_binary (src/baproblem.rs:580-706) INTO the resident problem (what
 * it held is dropped): format as above.  A `.bbal` is streamed to the device through pinned chunks while the host walks
 * the per-camera counts; byte order, the index / uv split, the point-index range check (C2B_ERR_INDEX_OUT_OF_RANGE) and
 * SnavelyCamera::from_vec run on the device.  Same resident state as c2b_bal_read + c2b_problem_upload_bal. */
int c2b_problem_read(c2b_problem *p, const char *path, int format);

/* Camera::from_position_direction (src/baproblem.rs:153-159) for n cameras on p's device:
 * pos3 [n][3], dir9 [n][9] col-major -> cams15 [n][15].  Does not change the problem. */
int c2b_problem_from_position_direction(c2b_problem *p, int64_t n_cam, const double *pos3,
                                        const double *dir9, double *cams15);
/* Camera::center (src/baproblem.rs:161-163) of every camera of the problem -> centers3 [n_cam][3] */
int c2b_problem_centers(c2b_problem *p, double *centers3);

int c2b_problem_project(c2b_problem *p, double *uv_out);
int c2b_problem_total_reprojection_error(c2b_problem *p, double norm, double *out);
/* ... when `p` is one shard (a contiguous camera range) of a larger problem: local sum, one 8-byte all-reduce through
 * `comm` on the problem's stream, then powf(1/norm); every rank returns the global error.  Collective. */
int c2b_problem_total_reprojection_error_sharded(c2b_problem *p, c2b_comm *comm, double norm, double *out);
/* Both norms run_noise prints (src/bin/city2ba.rs:283-287, 350-354) from one pass over the observations:
 * *l1 = total_reprojection_error(1.), *l2 = total_reprojection_error(2.).  _sharded: ONE 2-element all-reduce. */
int c2b_problem_total_reprojection_errors_l1_l2(c2b_problem *p, double *l1, double *l2);
int c2b_problem_total_reprojection_errors_l1_l2_sharded(c2b_problem *p, c2b_comm *comm, double *l1, double *l2);
/* Jacobian columns refer to the uploaded 9-vector's w while cameras are unmodified since
 * c2b_problem_upload_bal, otherwise to w = to_rodrigues(R) (what to_vec would write).
 * r [n_obs][2], Jc [n_obs][18], Jp [n_obs][6] are HOST buffers: the results leave the device in chunks whose copies
 * overlap the kernel of the next chunk.  Buffers from c2b_host_alloc (pinned) receive them at link speed (PCIe);
 * pageable memory works at the runtime's staged rate (about a third of that). */
int c2b_problem_residual_jacobian(c2b_problem *p, double *r, double *Jc, double *Jp);
/* The same launch with the results left ON THE DEVICE, in output arrays placed for streaming stores
 * (c2b_jacobian_outputs, above): the whole list in ONE launch -- r, Jc, Jp and the folded sum of squared residuals
 * (*sum_sq, may be NULL; total_reprojection_error(2.) is its square root) -- i.e. the Level-0 headline rate for a
 * BAProblem-level caller that consumes the Jacobian on the GPU.  *outputs == NULL: a set is allocated
 * (max_attempts placements tried, see c2b_jacobian_outputs_alloc) and handed over; pass it back on later calls to reuse
 * it (the observation count must still match), read it through c2b_jacobian_outputs_pointers, release it with
 * c2b_jacobian_outputs_free.  Bits equal to c2b_problem_residual_jacobian's. */
int c2b_problem_residual_jacobian_device(c2b_problem *p, int max_attempts, c2b_jacobian_outputs **outputs, double *sum_sq);
/* page-locked host memory for buffers that cross PCIe often (hipHostMalloc / hipHostFree) */
int c2b_host_alloc(void **ptr, int64_t bytes);
void c2b_host_free(void *ptr);
int c2b_problem_stats(c2b_problem *p, double *stats /* C2B_STATS_DOUBLES */);
/* BAProblem::cull (src/baproblem.rs:538-549) of the resident problem, on the device and in place: union-find
 * components, singleton counts and order-preserving renumbering by scans; same result as c2b_cull (same `faithful`
 * switch, same tie-break).  Sizes change: read them with c2b_problem_sizes, the new graph with
 * c2b_problem_download_graph (row_ptr[n_cam + 1], pt_idx[n_obs]) and the payloads with c2b_problem_download. */
int c2b_problem_cull(c2b_problem *p, int faithful);
int c2b_problem_largest_connected_component(c2b_problem *p, int faithful);   /* one application, on the device */
int c2b_problem_remove_singletons(c2b_problem *p);                            /* one application, on the device */
/* BAProblem::from_visibility (src/baproblem.rs:360-376) without leaving the device: the pending result of
 * c2b_problem_visibility_pairs_compact / _dense (+ _dense_occlude) becomes the problem's vis_graph. */
int c2b_problem_adopt_visibility(c2b_problem *p);
int c2b_problem_download_graph(c2b_problem *p, uint64_t *row_ptr, uint64_t *pt_idx);
/* Level 1 -> Level 0 without PCIe: copy the resident problem -- or the shard that belongs to the camera range
 * [cam_lo, cam_hi) (c2b_partition_cameras) -- into DEVICE buffers the caller owns (on the problem's device; any may be
 * NULL): cam15 [cam_hi - cam_lo][15]; pts4 [n_pts][4] (every point: the table is replicated across shards); row_ptr
 * [cam_hi - cam_lo + 1] rebased so that row_ptr[0] = 0; pt_idx (u32) and uv of the range's observations.  *obs_lo = the
 * index of the range's first observation in the whole list (the obs_base of noise draws), *n_obs_slice = how many it
 * has: call once with NULL buffers to size them.  What a problem born on the device (c2b_problem_synthetic_*_layout +
 * c2b_problem_visibility_within_distance, c2b_problem_read, c2b_problem_cull) hands to the stateless launchers; the
 * BAProblem side of it is the move of vis_graph / cameras / points out of the struct (src/baproblem.rs:256-260). */
int c2b_problem_export_device(c2b_problem *p, int64_t cam_lo, int64_t cam_hi, double *cam15, double *pts4, uint64_t *row_ptr,
                              uint32_t *pt_idx, double *uv, int64_t *obs_lo, int64_t *n_obs_slice);
int c2b_problem_visibility_pairs(c2b_problem *p, int64_t n_pairs, const uint32_t *cam_idx,
                                 const uint32_t *pt_idx, double max_dist, double *uv_out,
                                 uint8_t *keep);
/* The same predicate over candidate pairs whose cam_idx is non-decreasing (the generators' loops,
 * src/synthetic.rs:275-296, :360-378), with the kept pairs compacted on the device in their order: writes
 * row_ptr[n_cam + 1]; the kept (point index, uv) lists come back through c2b_problem_visibility_dense_fetch, so only
 * survivors cross PCIe. */
int c2b_problem_visibility_pairs_compact(c2b_problem *p, int64_t n_pairs, const uint32_t *cam_idx,
                                         const uint32_t *pt_idx, double max_dist, uint64_t *row_ptr);
/* The synthetic generators' WHOLE visibility loop on the device (src/synthetic.rs:268-297 grid, :353-378 line), the
 * candidate search included: for every camera the points within max_dist of its centre (rstar's
 * locate_within_distance(center, max_dist^2): squared distance <= max_dist^2, here a cell list), minus those whose
 * sight line hits a building when `occlusion` != 0 (hits_building, :52-124, with block_length / block_inset), filtered
 * by the predicate ((center - p).magnitude() < max_dist, z <= 0, |u|, |v| <= 1), kept per camera in ascending point
 * index.  Same result, index for index and bit for bit, as c2b_candidate_pairs + c2b_problem_visibility_pairs_compact;
 * nothing but row_ptr [n_cam + 1] (may be NULL) crosses PCIe.  The kept lists become the pending visibility result
 * (c2b_problem_adopt_visibility / c2b_problem_visibility_dense_fetch). */
int c2b_problem_visibility_within_distance(c2b_problem *p, double max_dist, int occlusion, double block_length,
                                           double block_inset, uint64_t *row_ptr);
/* dense sweep (src/generate.rs:446-469, no occlusion) over the problem's cameras and points: writes
 * row_ptr[n_cam + 1] and keeps the survivors on the device; _fetch copies pt_idx[row_ptr[n_cam]] and
 * uv[row_ptr[n_cam]][2] out (either may be NULL). */
/* generate_world_points_uniform (src/generate.rs:356-420) for the cameras of the resident problem, on the device: the
 * problem's points are REPLACED by num_points points sampled on the triangles (HOST array tri9 [n_tri][9] f32) by area,
 * each within max_dist of some camera; the problem must hold no observations.  The same points, bit for bit, as
 * c2b_generate_world_points with the same seed (candidate k draws from its own counter-based stream in both).  The
 * reference's errors -- no cameras; 10 * num_points failures -- come back as C2B_ERR_INVALID_ARGUMENT with its text. */
int c2b_problem_generate_world_points(c2b_problem *p, const float *tri9, int64_t n_tri, int64_t num_points, double max_dist,
                                      uint64_t seed, int64_t *n_out);
int c2b_problem_visibility_dense(c2b_problem *p, double max_dist, uint64_t *row_ptr);
int c2b_problem_visibility_dense_fetch(c2b_problem *p, uint64_t *pt_idx, double *uv);
/* filter the survivors of the last dense sweep through c2b_occlusion_filter against host triangles
 * tri9 [n_tri][9] (f32); rewrites row_ptr[n_cam + 1]; a following _fetch returns the filtered lists */
int c2b_problem_visibility_dense_occlude(c2b_problem *p, const float *tri9, int64_t n_tri, uint64_t *row_ptr);
/* the same with a hierarchy the caller built (c2b_bvh_build runs on the host and needs only the triangles, so a caller
 * can build it on another thread while cameras are placed, points sampled and the sweep runs -- cli/main.cpp does) */
int c2b_problem_visibility_dense_occlude_bvh(c2b_problem *p, const c2b_bvh *bvh, uint64_t *row_ptr);
int c2b_problem_add_drift(c2b_problem *p, double strength, double angle_strength, double std,
                          const double dir[3], uint64_t seed);
int c2b_problem_add_drift_normalized(c2b_problem *p, double strength, double angle_strength,
                                     double std, uint64_t seed);
int c2b_problem_add_noise(c2b_problem *p, double translation_std, double rotation_std,
                          double point_std, double observations_std, uint64_t seed);
int c2b_problem_add_sin_noise(c2b_problem *p, const double dir[3], const double noise_dir[3],
                              double strength, double frequency);
/* c2b_problem_add_noise followed by c2b_problem_total_reprojection_errors_l1_l2 -- the tail of run_noise
 * (src/bin/city2ba.rs:334-354) -- with the observation pass and both error sums in one launch: same resident state
 * afterwards (bit for bit), same two numbers. */
int c2b_problem_add_noise_errors_l1_l2(c2b_problem *p, double translation_std, double rotation_std, double point_std,
                                       double observations_std, uint64_t seed, double *l1, double *l2);

/* ---- a problem that is ONE SHARD of a larger one (multi-GPU at Level 1) ----
 * One c2b_problem per GPU: a contiguous camera range [cam_base, cam_base + n_cam) of n_cam_global cameras
 * (c2b_partition_cameras), that range's observations (row_ptr rebased to 0; obs_base = the global index of its first
 * observation) and the WHOLE point table with global point indices.  c2b_problem_set_shard records the three numbers;
 * the *_sharded entries then do, shard by shard, exactly what the unsharded calls do on the whole problem (the noise
 * functions of src/noise.rs:47-177, 388-416 and run_noise, src/bin/city2ba.rs:280-357): noise draws are keyed by global
 * indices, the statistics go through `comm`, every rank perturbs the replicated points identically.  All are
 * collective -- every rank calls them, in the same order -- and synchronous.  dir == NULL in _add_drift_sharded means
 * add_drift_normalized.  (c2b_problem_total_reprojection_error_sharded, above, completes the set.) */
int c2b_problem_set_shard(c2b_problem *p, int64_t cam_base, int64_t n_cam_global, int64_t obs_base);
int c2b_problem_stats_sharded(c2b_problem *p, c2b_comm *comm, double *stats);
int c2b_problem_add_drift_sharded(c2b_problem *p, c2b_comm *comm, double strength, double angle_strength, double std,
                                  const double *dir, uint64_t seed);
int c2b_problem_add_noise_sharded(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std,
                                  double point_std, double observations_std, uint64_t seed);
int c2b_problem_add_sin_noise_sharded(c2b_problem *p, c2b_comm *comm, const double dir[3], const double noise_dir[3],
                                      double strength, double frequency);
int c2b_problem_add_noise_errors_l1_l2_sharded(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std,
                                               double point_std, double observations_std, uint64_t seed, double *l1,
                                               double *l2);

#ifdef __cplusplus
}
#endif
#endif /* CITY2BA_HIP_H */
