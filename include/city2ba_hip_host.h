/*
 * city2ba_hip_host.h -- the HOST-SIDE rows of the city2ba hot path's callers: the generators' layout
 * and samplers, candidate search, cull(), .bal / .bbal IO, .obj loading, PLY export.  Plain CPU C++ inside
 * libcity2ba_hip.so behind a C ABI: HOST pointers, synchronous, they never touch the GPU.  The device boundary is
 * city2ba_hip.h (status codes, c2b_last_error and the Level-1 forms that run these rows' device twins live there).
 */
#ifndef CITY2BA_HIP_HOST_H
#define CITY2BA_HIP_HOST_H
#include "city2ba_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================================== *
 * Host-side generator pieces around the device predicate (HOST pointers, CPU, synchronous).
 * These are the callers either side of the hot path, in C++ because the
 * reference's generator is compiled code; they never touch the GPU.
 * ===================================================================================== */

/* synthetic_grid's camera / point counts: 4*cpb*B*(B+1), 12*ppb*B*(B+1) (src/synthetic.rs:179-258) */
int c2b_synthetic_grid_sizes(int64_t cameras_per_block, int64_t points_per_block, int64_t blocks,
                             int64_t *n_cam, int64_t *n_pts);
/* layout loops of synthetic_grid (src/synthetic.rs:178-258) in the reference's push and arithmetic
 * order: camera positions [n_cam][3], directions [n_cam][9] (col-major Basis3), points [n_pts][3] */
int c2b_synthetic_grid_layout(int64_t cameras_per_block, int64_t points_per_block, int64_t blocks,
                              double block_length, double block_inset, double camera_height,
                              double point_height, double *cam_pos3, double *cam_dir9, double *pts3);
/* layout of synthetic_line (src/synthetic.rs:323-344) */
int c2b_synthetic_line_layout(int64_t n_cam, int64_t n_pts, double length, double point_offset,
                              double camera_height, double point_height, double *cam_pos3,
                              double *cam_dir9, double *pts3);

/* Candidate (camera, point) pairs for cameras [cam_lo, cam_hi): squared distance from the camera
 * centre <= max_dist^2 (rstar locate_within_distance, src/synthetic.rs:277-280), camera-major,
 * ascending point index per camera; with occlusion != 0, pairs whose sight line crosses a building
 * are dropped (hits_building, src/synthetic.rs:52-124, incl. its end-point quirk at :93). */
typedef struct c2b_pairs c2b_pairs;
int c2b_candidate_pairs(const double *centers3, int64_t n_cam, const double *pts3, int64_t n_pts,
                        double max_dist, int64_t cam_lo, int64_t cam_hi, int occlusion,
                        double block_length, double block_inset, int n_threads, c2b_pairs **out);
int64_t c2b_pairs_count(const c2b_pairs *p);
const uint32_t *c2b_pairs_cam_idx(const c2b_pairs *p);
const uint32_t *c2b_pairs_pt_idx(const c2b_pairs *p);
void c2b_pairs_free(c2b_pairs *p);

/* ---- mesh generator (src/generate.rs), host side.  Embree is replaced by brute-force f32 ray / triangle
 * tests; every sampler takes a seed where the reference draws from an unseeded thread_rng(). ---- */

/* tobj::load_obj conventions (tobj 0.1.12): one model per `o`/`g` that owns faces or lines, f32 positions
 * re-indexed per model, polygons fan-triangulated, `l` polylines as index pairs. */
typedef struct c2b_obj c2b_obj;
int c2b_obj_load(const char *path, c2b_obj **out);
int64_t c2b_obj_model_count(const c2b_obj *o);
const char *c2b_obj_model_name(const c2b_obj *o, int64_t model);
/* is_lines: 1 when the model is a polyline (indices are segment pairs), 0 for triangles */
int c2b_obj_model_sizes(const c2b_obj *o, int64_t model, int64_t *n_positions, int64_t *n_indices, int *is_lines);
int c2b_obj_model_copy(const c2b_obj *o, int64_t model, float *positions3, uint32_t *indices);
/* move_to_origin (src/generate.rs:484-527), in place, over every model except `skip_model` (-1: none):
 * run_generate (src/bin/city2ba.rs:493-513) takes the --path model out of the list before the move */
int c2b_obj_move_to_origin(c2b_obj *o, int64_t skip_model);
/* triangles of every non-polyline model except `skip_model` (-1: none) as packed f32 [n_tri][9];
 * call with tri9 == NULL to get the count */
int c2b_obj_triangles(const c2b_obj *o, int64_t skip_model, float *tri9, int64_t *n_tri);
void c2b_obj_free(c2b_obj *o);

/* generate_cameras_path (step_size <= 0, src/generate.rs:109-148) / generate_cameras_path_step (:152-213) along
 * polyline model `path_model`: positions [num_cameras][3], directions [num_cameras][9] (col-major Basis3) */
int c2b_generate_cameras_path(const c2b_obj *o, int64_t path_model, int64_t num_cameras, double step_size,
                              uint64_t seed, double *cam_pos3, double *cam_dir9);
/* generate_cameras_poisson (:217-280) over triangles tri9: Poisson-disk x-z samples, downward ray casts, the
 * `pt[2] < lower_y + ground` filter of :264, random yaw.  *n_out = the number of cameras generated; the first
 * min(*n_out, capacity) are written (same seed => same cameras, so call with capacity 0 to size the buffers). */
int c2b_generate_cameras_poisson(const float *tri9, int64_t n_tri, int64_t num_points, double height, double ground,
                                 uint64_t seed, int64_t capacity, double *cam_pos3, double *cam_dir9, int64_t *n_out);
/* the same with a hierarchy the caller already built over tri9 (c2b_bvh_build) for the downward rays; without one the
 * function builds its own while it throws the darts.  The rays are cast on all host threads, gathered in sample order. */
int c2b_generate_cameras_poisson_bvh(const float *tri9, int64_t n_tri, const c2b_bvh *bvh, int64_t num_points, double height,
                                     double ground, uint64_t seed, int64_t capacity, double *cam_pos3, double *cam_dir9,
                                     int64_t *n_out);
/* modify_intrinsics (:530-544) */
int c2b_modify_intrinsics(double *cams15, int64_t n_cam, const double start[3], const double end[3], uint64_t seed);
/* generate_world_points_uniform (:356-420): area-weighted samples on the triangles that lie within max_dist of
 * some camera centre; fails like the reference's panics (no cameras / too many rejections) */
int c2b_generate_world_points(const float *tri9, int64_t n_tri, const double *centers3, int64_t n_cam,
                              int64_t num_points, double max_dist, uint64_t seed, double *pts3, int64_t *n_out);

/* BAProblem::cull (src/baproblem.rs:538-549) = largest_connected_component + remove_singletons to a
 * fixed point, IN PLACE on host arrays (outputs are subsets, so they fit).  Camera rows are opaque
 * `cam_stride` doubles.  On return *n_cam / *n_pts hold the new counts and row_ptr[*n_cam] the new
 * observation count.  faithful != 0 keeps the reference's observation filter at :523 (it indexes the
 * camera-first union-find array with a point index); ties between equally large components go to the one
 * with the smallest member (the reference: HashMap order). */
int c2b_cull(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3,
             uint64_t *row_ptr, uint64_t *pt_idx, double *uv, int faithful);
/* its two halves as the reference also exposes them: BAProblem::largest_connected_component (:456-534) and
 * BAProblem::remove_singletons (:426-453), one application each, same calling convention */
int c2b_largest_connected_component(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3,
                                    uint64_t *row_ptr, uint64_t *pt_idx, double *uv, int faithful);
int c2b_remove_singletons(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3,
                          uint64_t *row_ptr, uint64_t *pt_idx, double *uv);

/* noise.rs' index-corruption functions: sequential random reshuffles of the visibility graph, on the host over
 * the flat CSR arrays, IN PLACE, seeded (the reference: thread_rng()).
 * add_incorrect_correspondences (src/noise.rs:180-226): per observation, with probability mismatch_chance, swap
 *   its point index with a distance-weighted partner of the same camera (weights as at :198-206). */
int c2b_add_incorrect_correspondences(int64_t n_cam, const uint64_t *row_ptr, uint64_t *pt_idx, const double *uv,
                                      double mismatch_chance, uint64_t seed);
/* drop_features (:229-251): per camera keep floor(len * keep_fraction) observations of a random shuffle;
 *   rewrites row_ptr and compacts pt_idx / uv. */
int c2b_drop_features(int64_t n_cam, uint64_t *row_ptr, uint64_t *pt_idx, double *uv, double keep_fraction,
                      uint64_t seed);
/* split_landmarks (:255-291): floor(split_fraction * *n_pts) landmarks are duplicated at the end of pts3 (which must
 *   hold pts_capacity >= *n_pts + that many rows) and their observations move to the copy with probability 1/2. */
int c2b_split_landmarks(int64_t *n_pts, double *pts3, int64_t pts_capacity, int64_t n_obs, uint64_t *pt_idx,
                        double split_fraction, uint64_t seed);
/* join_landmarks (:326-378): floor(join_fraction * n_pts) random observations are re-pointed at one of the 10
 *   nearest other landmarks of their landmark. */
int c2b_join_landmarks(int64_t n_pts, const double *pts3, int64_t n_obs, uint64_t *pt_idx, double join_fraction,
                       uint64_t seed);

/* BAProblem::from_file (src/baproblem.rs:697-706): ".bal" text / ".bbal" big-endian binary by extension.
 * Cameras come back as 9-vectors (upload them with c2b_problem_upload_bal = from_vec). */
typedef struct c2b_balfile c2b_balfile;
int c2b_bal_read(const char *path, c2b_balfile **out);
int c2b_bal_sizes(const c2b_balfile *f, int64_t *n_cam, int64_t *n_pts, int64_t *n_obs);
int c2b_bal_copy(const c2b_balfile *f, double *bal9, double *pts3, uint64_t *row_ptr, uint64_t *pt_idx,
                 double *uv);
void c2b_bal_close(c2b_balfile *f);
/* BAProblem::write (src/baproblem.rs:768-785); bal9 = to_vec of every camera (c2b_problem_download_bal) */
int c2b_bal_write(const char *path, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                  const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv);
/* the format chosen by the caller instead of by the extension: format 0 = text (from_file_text :580, write_text
 * :709), 1 = binary (from_file_binary :632, write_binary :736), -1 = by extension */
int c2b_bal_read_as(const char *path, int format, c2b_balfile **out);
int c2b_bal_write_as(const char *path, int format, int64_t n_cam, const double *bal9, int64_t n_pts,
                     const double *pts3, const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv);
/* one f64 as write_text prints it -- Rust's `{}` (src/baproblem.rs:713-731): the shortest digits that read back to the
 * same double, no exponent, "-0", "NaN", "inf".  n values -> their texts back to back in buf, each followed by '\n';
 * *len = bytes written.  C2B_ERR_INVALID_ARGUMENT if cap is too small (330 bytes per value always suffice).  Host code
 * (csrc/decimal.hpp); the device writer of c2b_problem_write runs the same functions. */
int c2b_format_f64(int64_t n, const double *values, char *buf, int64_t cap, int64_t *len);
/* the other direction, as from_file_text reads a number (nom's `double`, i.e. str::parse::<f64>: correctly rounded;
 * src/baproblem.rs:580-629): `text` holds n whitespace-separated tokens; values[i] = token i, status[i] = 0 parsed,
 * 1 a spelling this parser leaves to strtod (anything but [+-]digits[.digits][(e|E)[+-]digits], e.g. "NaN"), 2 more
 * than 19 significant digits or a rounding its 128-bit arithmetic cannot decide (the device reader hands such files to
 * the host parser).  Host code (csrc/decimal.hpp); the device reader of c2b_problem_read runs the same functions. */
int c2b_parse_f64(const char *text, int64_t len, int64_t n, double *values, int32_t *status);
/* write_cameras of the `ply` subcommand (src/bin/city2ba.rs:359-439): ASCII PLY with one red vertex per camera
 * centre, one green vertex per point (f32) and one edge per observation (camera, n_cam + point) */
int c2b_ply_write(const char *path, int64_t n_cam, const double *centers3, int64_t n_pts, const double *pts3,
                  const uint64_t *row_ptr, const uint64_t *pt_idx);

/* the bounding-volume hierarchy c2b_occlusion_filter_bvh / c2b_problem_visibility_dense_occlude_bvh traverse (city2ba_hip.h
 * describes its use): built on the HOST over host triangles tri9 [n_tri][9] (f32); _copy fills host buffers of n_nodes *
 * C2B_BVH_NODE_BYTES and n_slots * C2B_BVH_TRI_BYTES bytes (and, optionally, order[n_slots]) for the caller to place in
 * device memory.  Replaces the Embree scene commit of src/bin/city2ba.rs:515-521. */
int c2b_bvh_build(const float *tri9, int64_t n_tri, c2b_bvh **out);
int c2b_bvh_sizes(const c2b_bvh *b, int64_t *n_nodes, int64_t *n_slots, int *depth);
int c2b_bvh_copy(const c2b_bvh *b, void *nodes, void *tris, uint32_t *order);
void c2b_bvh_free(c2b_bvh *b);

#ifdef __cplusplus
}
#endif
#endif /* CITY2BA_HIP_HOST_H */
