// capi_level0.hpp -- Level 0 of include/city2ba_hip.h: stateless asynchronous launchers over device pointers, the placed Jacobian outputs, the RCCL communicator and the sharded statistics / noise entries
// Part of the one translation unit of the C ABI: included by capi.hip (inside its extern "C" block, after its helpers and
// launchers), never compiled or included on its own.

/* ------------------------------- level 0 --------------------------------------------- */

int c2b_cameras_from_bal(const double *bal9, int64_t n, double *cam15, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!bal9 || !cam15))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_from_bal: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_from_bal, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), bal9, n, cam15);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_from_bal")
}

int c2b_cameras_to_bal(const double *cam15, int64_t n, double *bal9, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!bal9 || !cam15))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_to_bal: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_to_bal, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, n, bal9);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_to_bal")
}

int64_t c2b_camblk_doubles(int64_t n_cam) { return n_cam > 0 ? cam_table_doubles(n_cam) : 0; }

int c2b_camblk_from_state(const double *cam15, int64_t n, double *camblk, int64_t camblk_doubles, double *cen4, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !camblk))) return fail(C2B_ERR_INVALID_ARGUMENT, "camblk_from_state: bad arguments");
    if (!n) return C2B_OK;
    if (camblk_doubles < cam_table_doubles(n))
        return fail(C2B_ERR_INVALID_ARGUMENT, "camblk_from_state: the table holds whole groups of 8 cameras -- %lld doubles for %lld cameras, the buffer has %lld",
                    (long long)cam_table_doubles(n), (long long)n, (long long)camblk_doubles);
    if (!aligned16(camblk) || !aligned16(cen4)) return fail(C2B_ERR_INVALID_ARGUMENT, "camblk / cen4 must be 16-byte aligned");
    hipLaunchKernelGGL(k_cameras_prepare<false>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, n, camblk, cen4);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("camblk_from_state")
}

int c2b_camblk_from_bal(const double *bal9, int64_t n, double *camblk, int64_t camblk_doubles, double *cen4, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!bal9 || !camblk))) return fail(C2B_ERR_INVALID_ARGUMENT, "camblk_from_bal: bad arguments");
    if (!n) return C2B_OK;
    if (camblk_doubles < cam_table_doubles(n))
        return fail(C2B_ERR_INVALID_ARGUMENT, "camblk_from_bal: the table holds whole groups of 8 cameras -- %lld doubles for %lld cameras, the buffer has %lld",
                    (long long)cam_table_doubles(n), (long long)n, (long long)camblk_doubles);
    if (!aligned16(camblk) || !aligned16(cen4)) return fail(C2B_ERR_INVALID_ARGUMENT, "camblk / cen4 must be 16-byte aligned");
    hipLaunchKernelGGL(k_cameras_prepare<true>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), bal9, n, camblk, cen4);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("camblk_from_bal")
}

int c2b_cameras_from_position_direction(const double *pos3, const double *dir9, int64_t n, double *cam15,
                                        void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!pos3 || !dir9 || !cam15)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_from_position_direction: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_from_position_direction, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), pos3, dir9,
                       n, cam15);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_from_position_direction")
}

int c2b_project_world(const double *cam15, const uint32_t *cam_idx, const double *p3, int64_t n, double *out3,
                      void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !cam_idx || !p3 || !out3))) return fail(C2B_ERR_INVALID_ARGUMENT, "project_world: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_camera_point_map<false>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, cam_idx, p3, n, out3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("project_world")
}

int c2b_to_world(const double *cam15, const uint32_t *cam_idx, const double *p3, int64_t n, double *out3, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !cam_idx || !p3 || !out3))) return fail(C2B_ERR_INVALID_ARGUMENT, "to_world: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_camera_point_map<true>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, cam_idx, p3, n, out3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("to_world")
}

int c2b_cameras_transform(double *cam15, const double *delta_dir9, const double *delta_loc3, int64_t n, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !delta_dir9 || !delta_loc3))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_transform: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_transform, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, delta_dir9, delta_loc3, n);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_transform")
}

int c2b_points_pad(const double *pts3, int64_t n, double *pts4, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!pts3 || !pts4))) return fail(C2B_ERR_INVALID_ARGUMENT, "points_pad: bad arguments");
    if (!n) return C2B_OK;
    if (!aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "pts4 must be 16-byte aligned");
    hipLaunchKernelGGL(k_points_pad, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), pts3, n,
                       reinterpret_cast<double4 *>(pts4));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("points_pad")
}

int c2b_points_unpad(const double *pts4, int64_t n, double *pts3, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!pts3 || !pts4))) return fail(C2B_ERR_INVALID_ARGUMENT, "points_unpad: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_points_unpad, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream),
                       reinterpret_cast<const double4 *>(pts4), n, pts3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("points_unpad")
}

int c2b_expand_rows(const uint64_t *row_ptr, int64_t n_cam, int64_t obs_base, int64_t n_obs,
                    uint32_t *cam_idx, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_obs < 0 || obs_base < 0 || (n_obs && (!row_ptr || !cam_idx)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "expand_rows: bad arguments");
    if (n_cam >= (int64_t)1 << 32) return fail(C2B_ERR_INVALID_ARGUMENT, "expand_rows: n_cam exceeds u32");
    if (!n_obs) return C2B_OK;
    hipLaunchKernelGGL(k_expand_rows, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream), row_ptr, n_cam,
                       obs_base, n_obs, cam_idx);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("expand_rows")
}

static int check_obs_args(const char *who, const void *camblk, const void *pts4, const void *cam_idx,
                          const void *pt_idx, int64_t n) {
    if (n < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: negative count", who);
    if (n > (int64_t)0x7fffffff - 4096 * 64)       // 32-bit observation indices on the device; 2^31 observations are 34 GB of indices and uv alone
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: more than 2^31 observations in one launch", who);
    if (n && (!camblk || !pts4 || !cam_idx || !pt_idx)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: NULL input", who);
    if (n && (!aligned16(camblk) || !aligned16(pts4)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: camblk/pts4 must be 16-byte aligned", who);
    return C2B_OK;
}

int c2b_project(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                int64_t n_obs, double *uv_out, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("project", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!uv_out || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "project: uv_out NULL or misaligned");
    rc = launch_obs<MODE_PROJECT>(camblk, pts4, cam_idx, pt_idx, nullptr, n_obs, 0.0, 0.0, uv_out, nullptr, nullptr, nullptr, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("project")
}

int c2b_reprojection_error_sum(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                               const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs, double norm,
                               void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("reprojection_error_sum", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum: out_sum is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !aligned16(uv_obs) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum: uv_obs/workspace NULL or misaligned");
    rc = launch_obs<MODE_ERROR>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, norm, 0.0, nullptr, nullptr, workspace, out_sum, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("reprojection_error_sum")
}

// ---- camera-major lists addressed through the row structure (no per-observation camera index) ----
int64_t c2b_rows_tiles_bytes(int64_t n_obs) { return n_obs <= 0 ? 0 : (n_obs + 63) / 64 * 16; }

static int check_rows_args(const char *who, const uint64_t *row_ptr, int64_t n_cam, const void *tiles, int64_t n) {
    if (n_cam < 0 || n_cam >= (int64_t)1 << 31) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: n_cam out of range", who);
    if (n && (!row_ptr || !tiles || n_cam == 0)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: NULL row_ptr / tiles, or no cameras", who);
    if (n && (!aligned16(tiles) || (reinterpret_cast<uintptr_t>(row_ptr) & 7)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: tiles must be 16-byte aligned, row_ptr 8-byte aligned", who);
    return C2B_OK;
}

int c2b_rows_pack(const uint64_t *row_ptr, int64_t n_cam, int64_t n_obs, void *tiles, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || n_obs > (int64_t)0x7fffffff - 4096 * 64) return fail(C2B_ERR_INVALID_ARGUMENT, "rows_pack: observation count out of range");
    int rc = check_rows_args("rows_pack", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    const int64_t n_tiles = (n_obs + 63) / 64;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(k_rows_pack_tiles, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, row_ptr, (int)n_cam, (int)n_obs,
                       reinterpret_cast<uint4 *>(tiles));
    hipLaunchKernelGGL(k_rows_pack_marks, dim3((unsigned)((n_cam + 255) / 256)), dim3(256), 0, st, row_ptr, (int)n_cam, (int)n_obs,
                       reinterpret_cast<uint32_t *>(tiles));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("rows_pack")
}

int c2b_project_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam, const void *tiles,
                     const uint32_t *pt_idx, int64_t n_obs, double *uv_out, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("project_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("project_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!uv_out || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "project_rows: uv_out NULL or misaligned");
    rc = launch_obs<MODE_PROJECT>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_obs, 0.0, 0.0, uv_out,
                                  nullptr, nullptr, nullptr, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("project_rows")
}

int c2b_reprojection_error_sum_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                    const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                    double norm, void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("reprojection_error_sum_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("reprojection_error_sum_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum_rows: out_sum is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !aligned16(uv_obs) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum_rows: uv_obs/workspace NULL or misaligned");
    rc = launch_obs<MODE_ERROR>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, uv_obs, n_obs, norm, 0.0, nullptr,
                                nullptr, workspace, out_sum, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("reprojection_error_sum_rows")
}

// L1 and L2 in one pass: out_sums[0] = sum |du| + |dv|, out_sums[1] = sum du^2 + dv^2 -- each bit-identical to what
// c2b_reprojection_error_sum_rows returns for that norm (same grid, same fold order per sum, one arrival count).
int c2b_reprojection_error_sums2_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                      const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                      void *workspace, double *out_sums, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("reprojection_error_sums2_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("reprojection_error_sums2_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!out_sums) return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sums2_rows: out_sums is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sums, 0, 2 * sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !aligned16(uv_obs) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sums2_rows: uv_obs/workspace NULL or misaligned");
    rc = launch_obs<MODE_ERROR12>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, uv_obs, n_obs, 0.0, 0.0, nullptr,
                                  nullptr, workspace, out_sums, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("reprojection_error_sums2_rows")
}

// add_noise's observation pass (src/noise.rs:152-170) and the two error sums run_noise evaluates right after it
// (src/bin/city2ba.rs:350-354) in ONE pass over the list: uv is perturbed in place exactly as
// c2b_add_noise_observations would (same draws: counter = obs_base + i), and out_sums = the L1 / L2 sums of the
// PERTURBED observations against the cameras and points as they are now (entity noise first, then this).
int c2b_add_noise_observations_error_sums2_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                                const void *tiles, const uint32_t *pt_idx, double *uv, int64_t n_obs,
                                                int64_t obs_base, double observations_std, uint64_t seed, void *workspace,
                                                double *out_sums, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("add_noise_observations_error_sums2_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("add_noise_observations_error_sums2_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!out_sums || obs_base < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations_error_sums2_rows: bad arguments");
    if (!(observations_std >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise: standard deviations must be >= 0");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sums, 0, 2 * sizeof(double), S(stream))); return C2B_OK; }
    if (!uv || !aligned16(uv) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations_error_sums2_rows: uv/workspace NULL or misaligned");
    rc = launch_obs<MODE_NOISE_ERROR12>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_obs, observations_std,
                                        0.0, uv, nullptr, workspace, out_sums, S(stream), row_ptr, n_cam, obs_base, seed);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("add_noise_observations_error_sums2_rows")
}

int c2b_visibility_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam, const void *tiles,
                        const uint32_t *pt_idx, int64_t n_pairs, double max_dist, double *uv_out, uint8_t *keep, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("visibility_rows", camblk, pts4, tiles, pt_idx, n_pairs);
    if (!rc) rc = check_rows_args("visibility_rows", row_ptr, n_cam, tiles, n_pairs);
    if (rc) return rc;
    if (!n_pairs) return C2B_OK;
    if (!uv_out || !keep || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_rows: NULL or misaligned output");
    rc = launch_obs<MODE_VISIBILITY>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_pairs, 0.0, max_dist,
                                     uv_out, keep, nullptr, nullptr, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_rows")
}

int c2b_visibility_rows_bits(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam, const void *tiles,
                             const uint32_t *pt_idx, int64_t n_pairs, double max_dist, double *uv_out, uint64_t *keep_bits, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("visibility_rows_bits", camblk, pts4, tiles, pt_idx, n_pairs);
    if (!rc) rc = check_rows_args("visibility_rows_bits", row_ptr, n_cam, tiles, n_pairs);
    if (rc) return rc;
    if (!n_pairs) return C2B_OK;
    if (!uv_out || !keep_bits || !aligned16(uv_out) || (reinterpret_cast<uintptr_t>(keep_bits) & 7u))
        return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_rows_bits: NULL or misaligned output");
    rc = launch_obs<MODE_VISIBILITY_BITS>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_pairs, 0.0, max_dist,
                                          uv_out, reinterpret_cast<uint8_t *>(keep_bits), nullptr, nullptr, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_rows_bits")
}

int c2b_residual_jacobian(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                          const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs, double *r, double *Jc,
                          double *Jp, double norm, void *workspace, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("residual_jacobian", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!uv_obs || !r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian: NULL buffer");
    if (!aligned16(uv_obs) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian: uv/r/Jc/Jp must be 16-byte aligned");
    // with a workspace the fused error sum lands in the workspace's result slot (c2b_error_sum_finish copies it out)
    double *slot = workspace ? reinterpret_cast<double *>(workspace) + kWsFinal : nullptr;
    if (workspace) rc = launch_jacobian<true>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace, slot, S(stream));
    else rc = launch_jacobian<false>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, nullptr, nullptr, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("residual_jacobian")
}

int c2b_residual_jacobian_sum(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                              const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs, double *r, double *Jc,
                              double *Jp, double norm, void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("residual_jacobian_sum", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_sum: out_sum is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !r || !Jc || !Jp || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_sum: NULL buffer");
    if (!aligned16(uv_obs) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_sum: uv/r/Jc/Jp must be 16-byte aligned");
    rc = launch_jacobian<true>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace, out_sum, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("residual_jacobian_sum")
}

int c2b_jacobian_stream_policy(int64_t n_obs, int64_t n_cam, int64_t n_pts) { return jacobian_stream_policy(n_obs, n_cam, n_pts); }
int c2b_jacobian_tiles_per_wave(int64_t n_obs) { return jacobian_shape(n_obs, 0.0).opl; }
int c2b_jacobian_launch_shape(int64_t n_obs, double store_GBs, int *waves_per_workgroup, int *tiles_per_wave) {
    const JacShape s = jacobian_shape(n_obs, store_GBs);
    if (waves_per_workgroup) *waves_per_workgroup = s.wpb;
    if (tiles_per_wave) *tiles_per_wave = s.opl;
    return C2B_OK;
}

// store_GBs: the measured streaming-store rate of the set r / Jc / Jp live in (0 = unknown) -> the workgroup shape
static int rows_jacobian_impl(const double *camblk, const double *pts4, int64_t n_pts, const uint64_t *row_ptr, int64_t n_cam,
                              const void *tiles, int64_t obs_base, const uint32_t *pt_idx, const double *uv_obs,
                              int64_t n_obs, double *r, double *Jc, double *Jp, double norm, void *workspace,
                              double *out_sum, void *stream, double store_GBs) {
    int rc = check_obs_args("residual_jacobian_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("residual_jacobian_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (obs_base < 0 || (obs_base & 63)) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: obs_base must be a non-negative multiple of 64");
    if (out_sum && !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: out_sum needs a workspace");
    if (!n_obs) {
        if (out_sum) HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream)));
        return C2B_OK;
    }
    if (!uv_obs || !r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: NULL buffer");
    if (!aligned16(uv_obs) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: uv/r/Jc/Jp must be 16-byte aligned");
    const uint32_t *rec = reinterpret_cast<const uint32_t *>(tiles);
    if (workspace) {
        double *dst = out_sum ? out_sum : reinterpret_cast<double *>(workspace) + kWsFinal;
        rc = launch_jacobian<true>(camblk, pts4, rec, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace, dst, S(stream), row_ptr, n_cam, obs_base, n_pts, store_GBs);
    } else {
        rc = launch_jacobian<false>(camblk, pts4, rec, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, nullptr, nullptr, S(stream), row_ptr, n_cam, obs_base, n_pts, store_GBs);
    }
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
}

int c2b_residual_jacobian_rows(const double *camblk, const double *pts4, int64_t n_pts, const uint64_t *row_ptr, int64_t n_cam,
                               const void *tiles, int64_t obs_base, const uint32_t *pt_idx, const double *uv_obs,
                               int64_t n_obs, double *r, double *Jc, double *Jp, double norm, void *workspace,
                               double *out_sum, void *stream) {
    C2B_API_BEGIN
    return rows_jacobian_impl(camblk, pts4, n_pts, row_ptr, n_cam, tiles, obs_base, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace,
                              out_sum, stream, 0.0);
    C2B_API_END("residual_jacobian_rows")
}

int c2b_calib_store_pattern(int64_t n_obs, double *r, double *Jc, double *Jp, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || (n_obs && (!r || !Jc || !Jp)) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "calib_store_pattern: bad arguments");
    if (n_obs < 64) return C2B_OK;
    const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
    hipLaunchKernelGGL((k_store_pattern<true, 8>), dim3((unsigned)bt), dim3(512), 0, S(stream), n_obs, bt,
                       reinterpret_cast<double2 *>(r), Jc, Jp);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("calib_store_pattern")
}

int c2b_calib_copy(const void *src, void *dst, int64_t bytes, void *stream) {
    C2B_API_BEGIN
    if (bytes < 0 || (bytes && (!src || !dst)) || !aligned16(src) || !aligned16(dst) || (bytes & 15))
        return fail(C2B_ERR_INVALID_ARGUMENT, "calib_copy: NULL, misaligned or not a multiple of 16 bytes");
    if (!bytes) return C2B_OK;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, S(stream), reinterpret_cast<const double2 *>(src),
                       reinterpret_cast<double2 *>(dst), bytes / 16);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("calib_copy")
}

// ---- output arrays of the residual + Jacobian launch, placed for streaming stores -------------------------------
// Measured on MI355X ("what the spread really is"): the same kernel writing the same bytes takes
// 690 or 860 us depending only on WHICH device allocation r / Jc / Jp live in -- the store pattern alone streams at
// ~7.0 TB/s into some allocations and ~5.7 TB/s into others of identical size and alignment, in one process on one
// device; a freed and re-made allocation keeps its speed, a different one rolls again.  Nothing visible from user
// space predicts it, so the placement is chosen by measurement: allocate, time the kernel's own store pattern
// (~0.6 ms per repetition), keep the set if it streams at fast_store_GBs or better, otherwise HOLD it (so that the
// allocator cannot hand the same memory back) and try again; the best of max_attempts wins, the rest are freed.
// r05 (tools/probes/vram_store_map.py, profiles/r05ao): the rate belongs to WHERE in the device memory a set lies -- of 56-60
// consecutive 4-GB sets of a fresh process 8-12 stream at 7.0-7.2 TB/s on every one of 17 devices mapped, in windows that
// recur with a period of ~64 GiB, the rest at 5.6-5.9 -- so a search that goes deep enough finds a fast set on devices whose
// first eight sets (32 GB: the search depth of rounds 2-4) are all slow.  Hence up to kMaxPlacementAttempts.
// The windows are several GB wide also for small sets (0.5-GB sets of a rank's eighth: 5.3-5.6 TB/s outside, 6.7-7.0 inside, the
// first window 6-37 GB into the memory by device), so a search over sets smaller than kPlacementStride skips ahead: after a
// rejected set it allocates -- and holds, untouched -- a filler that brings the step to kPlacementStride bytes.
// Held memory is bounded (max_attempts steps of max(208 B per observation, kPlacementStride), and never more than three
// quarters of the memory that was free at the call) and an out-of-memory attempt ends the search with the best set so far
// instead of failing.
constexpr int kMaxPlacementAttempts = 64;
constexpr size_t kPlacementStride = (size_t)2 << 30;
struct c2b_jacobian_outputs {
    int device = 0;
    int64_t n_obs = 0;
    double *r = nullptr, *Jc = nullptr, *Jp = nullptr;
    int attempts = 0, chosen = -1;
    double rate[kMaxPlacementAttempts] = {};
};

namespace {
struct OutSet {
    double *r = nullptr, *Jc = nullptr, *Jp = nullptr;
    void free_all() { if (r) (void)hipFree(r); if (Jc) (void)hipFree(Jc); if (Jp) (void)hipFree(Jp); r = Jc = Jp = nullptr; }
};
hipError_t alloc_set(int64_t n, OutSet *s) {
    const size_t k = (size_t)(n > 0 ? n : 1);
    hipError_t e = hipMalloc((void **)&s->r, k * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&s->Jc, k * 144);
    if (e == hipSuccess) e = hipMalloc((void **)&s->Jp, k * 48);
    if (e != hipSuccess) s->free_all();
    return e;
}
}  // namespace

int c2b_jacobian_outputs_alloc(int64_t n_obs, int max_attempts, double fast_store_GBs, void *stream, c2b_jacobian_outputs **out) {
    C2B_API_BEGIN
    if (!out || n_obs < 0 || n_obs >= ((int64_t)1 << 31)) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_alloc: bad arguments");
    *out = nullptr;
    if (max_attempts < 1) max_attempts = 1;
    if (max_attempts > kMaxPlacementAttempts) max_attempts = kMaxPlacementAttempts;
    if (!(fast_store_GBs > 0.0)) fast_store_GBs = 7000.0;
    hipStream_t st = S(stream);
    std::unique_ptr<c2b_jacobian_outputs> h(new c2b_jacobian_outputs);
    HIP_TRY(hipGetDevice(&h->device));
    h->n_obs = n_obs;
    size_t free_at_call = 0;
    if (max_attempts > 1) {                            // the rejects are HELD during the search: at most 3/4 of what is free now
        size_t total_b = 0;
        if (hipMemGetInfo(&free_at_call, &total_b) == hipSuccess) {
            const size_t set_b = (size_t)(n_obs > 0 ? n_obs : 1) * 208;
            const size_t fit = free_at_call / 4 * 3 / (set_b > kPlacementStride ? set_b : kPlacementStride);
            if ((size_t)max_attempts > fit) max_attempts = fit < 1 ? 1 : (int)fit;
        } else {
            free_at_call = 0;
        }
    }
    std::vector<OutSet> sets((size_t)max_attempts);
    std::vector<void *> fillers;
    const size_t set_bytes = (size_t)(n_obs > 0 ? n_obs : 1) * 208;
    auto free_sets = [&]() {
        for (auto &q : sets) q.free_all();
        for (void *f : fillers) (void)hipFree(f);
        fillers.clear();
    };
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    // below a million observations the store rate means nothing; above, even a caller that takes the first set
    // (max_attempts = 1) gets its rate measured (~4 ms): c2b_residual_jacobian_rows_placed picks its workgroup shape by it
    const bool measure = n_obs >= 1000000;
    if (measure && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess)) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return fail(C2B_ERR_HIP, "jacobian_outputs_alloc: hipEventCreate failed");
    }
    int best = -1;
    hipError_t err = hipSuccess;
    for (int a = 0; a < (measure ? max_attempts : 1); ++a) {
        // the bound above was computed once; another process, rank or allocator sharing the device may have taken
        // memory since.  Asked again before every further attempt: the search stops, with the best set so far, as soon as less
        // than a quarter of what was free at the call (plus one more set) is left.
        if (a > 0 && free_at_call) {
            size_t free_now = 0, total_b = 0;
            if (hipMemGetInfo(&free_now, &total_b) == hipSuccess && free_now < free_at_call / 4 + set_bytes) break;
        }
        err = alloc_set(n_obs, &sets[a]);
        if (err != hipSuccess) {
            if (best >= 0 && err == hipErrorOutOfMemory) { (void)hipGetLastError(); err = hipSuccess; }   // keep the best so far
            break;
        }
        h->attempts = a + 1;
        if (!measure) { best = a; break; }
        const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
        auto pattern = [&]() {
            hipLaunchKernelGGL((k_store_pattern<true, 8>), dim3((unsigned)bt), dim3(512), 0, st, n_obs, bt,
                               reinterpret_cast<double2 *>(sets[a].r), sets[a].Jc, sets[a].Jp);
        };
        pattern(); pattern();
        // two groups of two repetitions, the faster one counts: one hiccup of the device (a 10 x slower reading was seen once in
        // six rounds, r06: a fresh set's first touches or a clock ramp) must not label a set, or steer the launch shape
        err = hipEventRecord(e0, st);
        pattern(); pattern();
        if (err == hipSuccess) err = hipEventRecord(e1, st);
        pattern(); pattern();
        if (err == hipSuccess) err = hipEventRecord(e2, st);
        if (err == hipSuccess) err = hipEventSynchronize(e2);
        float ms_a = 0.f, ms_b = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms_a, e0, e1);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms_b, e1, e2);
        if (err == hipSuccess) err = launch_error();
        if (err != hipSuccess) break;
        const float ms = ms_a < ms_b ? ms_a : ms_b;
        h->rate[a] = (double)n_obs * 208.0 / ((double)ms / 2.0 * 1e-3) / 1e9;
        // a later set replaces the incumbent only if it is clearly faster (2 %): between sets of the same class the
        // measured rate differs by noise, and the kernel's own time does not follow differences that small
        if (best < 0 || h->rate[a] > h->rate[best] * 1.02) best = a;
        if (h->rate[a] >= fast_store_GBs) break;
        if (set_bytes < kPlacementStride && a + 1 < max_attempts) {       // skip ahead: the next set starts a stride further on
            void *f = nullptr;
            if (hipMalloc(&f, kPlacementStride - set_bytes) == hipSuccess) fillers.push_back(f);
            else { (void)hipGetLastError(); break; }                      // memory is full: the best so far is it
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (err != hipSuccess || best < 0) {
        free_sets();
        return fail(err == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "jacobian_outputs_alloc: %s",
                    hipGetErrorString(err == hipSuccess ? hipErrorUnknown : err));
    }
    h->r = sets[best].r; h->Jc = sets[best].Jc; h->Jp = sets[best].Jp;
    sets[best] = OutSet();
    free_sets();
    h->chosen = best;
    *out = h.release();
    return C2B_OK;
    C2B_API_END("jacobian_outputs_alloc")
}

int c2b_jacobian_outputs_pointers(const c2b_jacobian_outputs *h, double **r, double **Jc, double **Jp) {
    if (!h || !r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_pointers: NULL argument");
    *r = h->r; *Jc = h->Jc; *Jp = h->Jp;
    return C2B_OK;
}

int c2b_jacobian_outputs_log(const c2b_jacobian_outputs *h, double *store_GBs_per_attempt, int capacity, int *attempts, int *chosen) {
    if (!h || capacity < 0 || (capacity && !store_GBs_per_attempt)) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_log: bad arguments");
    for (int a = 0; a < h->attempts && a < capacity; ++a) store_GBs_per_attempt[a] = h->rate[a];
    if (attempts) *attempts = h->attempts;
    if (chosen) *chosen = h->chosen;
    return C2B_OK;
}

// The whole-list launch INTO a placed output set: c2b_residual_jacobian_rows with r / Jc / Jp taken from the handle and
// the workgroup shape chosen by the store rate c2b_jacobian_outputs_alloc measured for that set (jacobian_shape).
int c2b_residual_jacobian_rows_placed(const double *camblk, const double *pts4, int64_t n_pts, const uint64_t *row_ptr, int64_t n_cam,
                                      const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                      const c2b_jacobian_outputs *outputs, double norm, void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    if (!outputs) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows_placed: outputs is NULL");
    if (outputs->n_obs != n_obs)
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows_placed: the output set holds %lld observations, the launch %lld",
                    (long long)outputs->n_obs, (long long)n_obs);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != outputs->device)
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows_placed: the current device (%d) is not the output set's (%d)", cur, outputs->device);
    const double rate = outputs->chosen >= 0 && outputs->chosen < kMaxPlacementAttempts ? outputs->rate[outputs->chosen] : 0.0;
    return rows_jacobian_impl(camblk, pts4, n_pts, row_ptr, n_cam, tiles, 0, pt_idx, uv_obs, n_obs, outputs->r, outputs->Jc, outputs->Jp,
                              norm, workspace, out_sum, stream, rate);
    C2B_API_END("residual_jacobian_rows_placed")
}

int c2b_jacobian_outputs_store_rate(const c2b_jacobian_outputs *h, double *store_GBs) {
    if (!h || !store_GBs) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_store_rate: NULL argument");
    *store_GBs = h->chosen >= 0 && h->chosen < kMaxPlacementAttempts ? h->rate[h->chosen] : 0.0;
    return C2B_OK;
}

// a caller that timed the set itself (or wants a particular shape) replaces the recorded rate; <= 0 = "unknown"
int c2b_jacobian_outputs_set_store_rate(c2b_jacobian_outputs *h, double store_GBs) {
    if (!h) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_set_store_rate: NULL handle");
    if (h->chosen < 0 || h->chosen >= kMaxPlacementAttempts) h->chosen = 0;
    h->rate[h->chosen] = store_GBs > 0.0 ? store_GBs : 0.0;
    return C2B_OK;
}

void c2b_jacobian_outputs_free(c2b_jacobian_outputs *h) {
    if (!h) return;
    int prev = 0;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != h->device && hipSetDevice(h->device) == hipSuccess;
    if (h->r) (void)hipFree(h->r);
    if (h->Jc) (void)hipFree(h->Jc);
    if (h->Jp) (void)hipFree(h->Jp);
    if (sw) (void)hipSetDevice(prev);
    delete h;
}

int c2b_error_sum_finish(const void *workspace, int64_t n_obs, double *out_sum, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || !out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "error_sum_finish: bad arguments");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "error_sum_finish: workspace is NULL");
    HIP_TRY(hipMemcpyAsync(out_sum, reinterpret_cast<const double *>(workspace) + kWsFinal, sizeof(double),
                           hipMemcpyDeviceToDevice, S(stream)));
    return C2B_OK;
    C2B_API_END("error_sum_finish")
}

int c2b_visibility_pairs(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                         const uint32_t *pt_idx, int64_t n_pairs, double max_dist, double *uv_out,
                         uint8_t *keep, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("visibility_pairs", camblk, pts4, cam_idx, pt_idx, n_pairs);
    if (rc) return rc;
    if (!n_pairs) return C2B_OK;
    if (!uv_out || !keep || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_pairs: NULL/misaligned output");
    rc = launch_obs<MODE_VISIBILITY>(camblk, pts4, cam_idx, pt_idx, nullptr, n_pairs, 0.0, max_dist, uv_out, keep, nullptr, nullptr, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_pairs")
}

int c2b_occlusion_filter(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                         int64_t n_obs, const float *tri9, int64_t n_tri, uint8_t *keep, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("occlusion_filter", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!keep || n_tri < 0 || (n_tri && !tri9)) return fail(C2B_ERR_INVALID_ARGUMENT, "occlusion_filter: bad arguments");
    hipLaunchKernelGGL(k_occlusion, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream), camblk,
                       reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx, n_obs, tri9, n_tri, keep);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("occlusion_filter")
}

struct c2b_bvh {
    c2b_host::Bvh b;
};

int c2b_bvh_build(const float *tri9, int64_t n_tri, c2b_bvh **out) {
    C2B_API_BEGIN
    if (!out || n_tri < 0 || (n_tri && !tri9)) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: bad arguments");
    *out = nullptr;
    if (n_tri >= ((int64_t)1 << 28)) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: more than 2^28 triangles");
    for (int64_t k = 0; k < 9 * n_tri; ++k)
        if (!std::isfinite(tri9[k])) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: triangle %lld is not finite", (long long)(k / 9));
    c2b_bvh *h = new (std::nothrow) c2b_bvh();
    if (!h) return fail(C2B_ERR_OOM, "bvh_build: host allocation failed");
    try {
        c2b_host::bvh_build(tri9, n_tri, h->b);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(C2B_ERR_OOM, "bvh_build: out of host memory");
    }
    if (h->b.depth >= kBvhStack) {
        const int d = h->b.depth;
        delete h;
        return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: hierarchy depth %d exceeds the traversal stack", d);
    }
    *out = h;
    return C2B_OK;
    C2B_API_END("bvh_build")
}

int c2b_bvh_sizes(const c2b_bvh *b, int64_t *n_nodes, int64_t *n_slots, int *depth) {
    C2B_API_BEGIN
    if (!b) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_sizes: bvh is NULL");
    if (n_nodes) *n_nodes = (int64_t)b->b.nodes.size();
    if (n_slots) *n_slots = (int64_t)b->b.order.size();
    if (depth) *depth = b->b.depth;
    return C2B_OK;
    C2B_API_END("bvh_sizes")
}

int c2b_bvh_copy(const c2b_bvh *b, void *nodes, void *tris, uint32_t *order) {
    C2B_API_BEGIN
    if (!b) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_copy: bvh is NULL");
    if (nodes) std::memcpy(nodes, b->b.nodes.data(), b->b.nodes.size() * sizeof(c2b_host::BvhNode));
    if (tris && !b->b.tris.empty()) std::memcpy(tris, b->b.tris.data(), b->b.tris.size() * sizeof(float));
    if (order && !b->b.order.empty()) std::memcpy(order, b->b.order.data(), b->b.order.size() * sizeof(uint32_t));
    return C2B_OK;
    C2B_API_END("bvh_copy")
}

void c2b_bvh_free(c2b_bvh *b) { delete b; }

int c2b_occlusion_filter_bvh(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                             int64_t n_obs, const void *nodes, int64_t n_nodes, const void *tris, int64_t n_slots,
                             uint8_t *keep, uint32_t *overflow, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("occlusion_filter_bvh", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!keep || !overflow || !nodes || n_nodes < 1 || n_slots < 0 || (n_slots && !tris) || !aligned16(nodes) || (tris && !aligned16(tris)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "occlusion_filter_bvh: NULL/misaligned buffer");
    hipLaunchKernelGGL(k_occlusion_bvh, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream), camblk,
                       reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx, n_obs, reinterpret_cast<const float4 *>(nodes),
                       reinterpret_cast<const float4 *>(tris), keep, overflow);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("occlusion_filter_bvh")
}

int c2b_stats(const double *camblk, const double *cen4, int64_t n_cam, const double *pts4, int64_t n_pts, void *workspace,
              double *stats, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || !stats || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: bad arguments");
    if (n_cam + n_pts == 0) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: empty problem (the reference's fold1().unwrap() panics)");
    if ((n_cam && !camblk && !cen4) || (n_pts && !pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: NULL input");
    if (!aligned16(camblk) || !aligned16(cen4) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: camblk/cen4/pts4 must be 16-byte aligned");
    const SrcBlk src = SrcBlk::make(camblk, cen4, pts4, n_cam);
    return stats_impl(src, n_cam + n_pts, workspace, stats, S(stream));
    C2B_API_END("stats")
}

int c2b_stats_partial_pass1(const double *camblk, const double *cen4, int64_t n_cam, int64_t cam_base, int64_t n_cam_global,
                            const double *pts4, int64_t n_pts, int64_t pt_base, int64_t n_entities_global,
                            void *workspace, double *part, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || cam_base < 0 || pt_base < 0 || n_cam_global < cam_base + n_cam || n_entities_global < 1 ||
        !part || !workspace || (n_cam && !camblk && !cen4) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass1: bad arguments");
    if (!aligned16(camblk) || !aligned16(cen4) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass1: camblk/cen4/pts4 must be 16-byte aligned");
    const SrcBlk src = SrcBlk::make(camblk, cen4, pts4, n_cam);
    const int64_t n = n_cam + n_pts;
    double *rec = reinterpret_cast<double *>(workspace);
    const ShardMap map{n_cam, cam_base, n_cam_global, pt_base};
    hipLaunchKernelGGL((k_stats_pass1<SrcBlk, false>), dim3(stats_grid(n, kStatBlock)), dim3(kStatBlock), 0, S(stream), src, n, (double)n_entities_global, rec,
                       ws_ticket(workspace), map, part);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("stats_partial_pass1")
}

int c2b_stats_partial_pass2(const double *camblk, const double *cen4, int64_t n_cam, const double *pts4, int64_t n_pts, const double *mean3,
                            void *workspace, double *sumsq3, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || !mean3 || !sumsq3 || !workspace || (n_cam && !camblk && !cen4) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass2: bad arguments");
    if (!aligned16(camblk) || !aligned16(cen4) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass2: camblk/cen4/pts4 must be 16-byte aligned");
    const SrcBlk src = SrcBlk::make(camblk, cen4, pts4, n_cam);
    const int64_t n = n_cam + n_pts;
    double *rec = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL((k_stats_pass2<SrcBlk, true>), dim3(stats_grid(n)), dim3(kBlock), 0, S(stream), src, n, mean3, rec,
                       ws_ticket(workspace), sumsq3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("stats_partial_pass2")
}

// ---- collectives of the sharded path (comm_rccl.hpp): RCCL behind the C ABI -----------------------------------
#define RCCL_TRY(who, expr)                                                                                  \
    do {                                                                                                     \
        const ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                               \
            return fail(C2B_ERR_RCCL, who ": %s: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
    } while (0)
#define NEED_RCCL(who)                                                                                       \
    if (!rccl().ok()) return fail(C2B_ERR_RCCL, who ": %s", rccl().error.c_str())

const char *c2b_comm_backend(void) {
    static thread_local char text[256];
    if (!rccl().ok()) { snprintf(text, sizeof text, "unavailable: %s", rccl().error.c_str()); return text; }
    int v = 0;
    (void)rccl().GetVersion(&v);
    snprintf(text, sizeof text, "RCCL %d.%d.%d (%s)", v / 10000, (v / 100) % 100, v % 100, rccl().path.c_str());
    return text;
}

int c2b_comm_unique_id(void *id128) {
    C2B_API_BEGIN
    if (!id128) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_unique_id: NULL argument");
    NEED_RCCL("comm_unique_id");
    static_assert(sizeof(ncclUniqueId) == C2B_COMM_ID_BYTES, "C2B_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    RCCL_TRY("comm_unique_id", rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return C2B_OK;
    C2B_API_END("comm_unique_id")
}

int c2b_comm_init_rank(const void *id128, int rank, int world, int device, c2b_comm **out) {
    C2B_API_BEGIN
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world || device < 0)
        return fail(C2B_ERR_INVALID_ARGUMENT, "comm_init_rank: bad arguments");
    *out = nullptr;
    NEED_RCCL("comm_init_rank");
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    std::unique_ptr<c2b_comm> c(new c2b_comm);
    c->rank = rank; c->world = world; c->device = device;
    RCCL_TRY("comm_init_rank", rccl().CommInitRank(&c->comm, world, id, rank));
    *out = c.release();
    return C2B_OK;
    C2B_API_END("comm_init_rank")
}

int c2b_comm_init_all(int n_dev, const int *dev_ids, c2b_comm **out) {
    C2B_API_BEGIN
    if (n_dev < 1 || n_dev > 64 || !out) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_init_all: bad arguments");
    for (int i = 0; i < n_dev; ++i) out[i] = nullptr;
    NEED_RCCL("comm_init_all");
    int devs[64];
    ncclComm_t comms[64];
    for (int i = 0; i < n_dev; ++i) devs[i] = dev_ids ? dev_ids[i] : i;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    const ncclResult_t r = rccl().CommInitAll(comms, n_dev, devs);      // switches the current device as it goes
    (void)hipSetDevice(prev);
    if (r != ncclSuccess) return fail(C2B_ERR_RCCL, "comm_init_all: ncclCommInitAll: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    int made = 0;
    for (; made < n_dev; ++made) {
        out[made] = new (std::nothrow) c2b_comm;
        if (!out[made]) break;
        out[made]->comm = comms[made]; out[made]->rank = made; out[made]->world = n_dev; out[made]->device = devs[made];
    }
    if (made < n_dev) {                                       // out of host memory half way: give every communicator back
        for (int i = 0; i < made; ++i) { c2b_comm_destroy(out[i]); out[i] = nullptr; }
        for (int i = made; i < n_dev; ++i) {
            const bool sw = hipSetDevice(devs[i]) == hipSuccess;
            (void)rccl().CommDestroy(comms[i]);
            if (sw) (void)hipSetDevice(prev);
        }
        (void)hipSetDevice(prev);
        return fail(C2B_ERR_OOM, "comm_init_all: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("comm_init_all")
}

int c2b_comm_info(const c2b_comm *c, int *rank, int *world, int *device) {
    if (!c) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_info: NULL communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return C2B_OK;
}

int c2b_comm_group_start(void) {
    C2B_API_BEGIN
    NEED_RCCL("comm_group_start");
    RCCL_TRY("comm_group_start", rccl().GroupStart());
    return C2B_OK;
    C2B_API_END("comm_group_start")
}

int c2b_comm_group_end(void) {
    C2B_API_BEGIN
    NEED_RCCL("comm_group_end");
    RCCL_TRY("comm_group_end", rccl().GroupEnd());
    return C2B_OK;
    C2B_API_END("comm_group_end")
}

int c2b_comm_all_reduce_sum_f64(c2b_comm *c, double *buf, int64_t n, void *stream) {
    C2B_API_BEGIN
    if (!c || !c->comm || n < 0 || (n && !buf)) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_all_reduce_sum_f64: bad arguments");
    if (!n) return C2B_OK;
    RCCL_TRY("comm_all_reduce_sum_f64", rccl().AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->comm, S(stream)));
    return C2B_OK;
    C2B_API_END("comm_all_reduce_sum_f64")
}

int c2b_comm_all_gather_f64(c2b_comm *c, const double *send, int64_t n_per_rank, double *recv, void *stream) {
    C2B_API_BEGIN
    if (!c || !c->comm || n_per_rank < 0 || (n_per_rank && (!send || !recv)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "comm_all_gather_f64: bad arguments");
    if (!n_per_rank) return C2B_OK;
    RCCL_TRY("comm_all_gather_f64", rccl().AllGather(send, recv, (size_t)n_per_rank, ncclDouble, c->comm, S(stream)));
    return C2B_OK;
    C2B_API_END("comm_all_gather_f64")
}

void c2b_comm_destroy(c2b_comm *c) {
    if (!c) return;
    if (c->comm && rccl().ok()) {
        int prev = 0;
        const bool sw = hipGetDevice(&prev) == hipSuccess && prev != c->device && hipSetDevice(c->device) == hipSuccess;
        (void)rccl().CommDestroy(c->comm);
        if (sw) (void)hipSetDevice(prev);
    }
    delete c;
}

// ---- host halves of the statistics over sharded cameras ------------------------------------
// shares [world][20] = every rank's c2b_stats_partial_pass1 record in rank order.  mean: the shares summed in rank
// order; origin: smallest distance, ties to the LARGER global index (fold1 with strict <, src/noise.rs:80-86).
int c2b_stats_combine_shares(const double *shares, int world, double *stats) {
    C2B_API_BEGIN
    if (!shares || !stats || world < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_combine_shares: bad arguments");
    const double inf = std::numeric_limits<double>::infinity();
    double mean[3] = {0, 0, 0}, mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    const double *best = nullptr;
    for (int r = 0; r < world; ++r) {
        const double *p = shares + 20 * (size_t)r;
        for (int k = 0; k < 3; ++k) {
            mean[k] = mean[k] + p[k];
            mn[k] = std::fmin(mn[k], p[6 + k]);
            mx[k] = std::fmax(mx[k], p[9 + k]);
        }
        if (p[18] < 0) continue;
        if (!best || p[19] < best[19] || (p[19] == best[19] && p[18] > best[18])) best = p;
    }
    if (!best) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: empty problem");
    for (int k = 0; k < 20; ++k) stats[k] = 0.0;
    for (int k = 0; k < 3; ++k) {
        stats[k] = mean[k]; stats[6 + k] = mn[k]; stats[9 + k] = mx[k]; stats[12 + k] = mx[k] - mn[k];
        stats[15 + k] = best[15 + k];
    }
    stats[18] = best[18];
    stats[19] = best[19];
    return C2B_OK;
    C2B_API_END("stats_combine_shares")
}

// sumsq [world][3] = every rank's c2b_stats_partial_pass2 sums, rank order -> stats[3..5] = std, stats[19] = |std|
int c2b_stats_finish_shares(const double *sumsq, int world, int64_t n_entities, double *stats) {
    C2B_API_BEGIN
    if (!sumsq || !stats || world < 1 || n_entities < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_finish_shares: bad arguments");
    double t[3] = {0, 0, 0};
    for (int r = 0; r < world; ++r)
        for (int k = 0; k < 3; ++k) t[k] = t[k] + sumsq[3 * (size_t)r + k];
    const double num = (double)n_entities;
    for (int k = 0; k < 3; ++k) stats[3 + k] = std::sqrt(t[k] / num);
    stats[19] = std::sqrt((stats[3] * stats[3] + stats[4] * stats[4]) + stats[5] * stats[5]);
    return C2B_OK;
    C2B_API_END("stats_finish_shares")
}

// BAProblem::mean/std/extent/dimensions + add_drift's origin when cameras are sharded: this rank's camblk holds cameras
// [cam_base, cam_base + n_cam) of n_cam_global, pts4 is the whole replicated table and rank r of W reduces its r-th
// slice.  Two all-gathers (20 and 3 doubles per rank) through the communicator; sums in rank order on the host, so
// every rank ends with the same bits.  Synchronous; `stats` (device, 20 doubles) is complete on return.
int c2b_stats_sharded(c2b_comm *c, const double *camblk, const double *cen4, int64_t n_cam, int64_t cam_base, int64_t n_cam_global,
                      const double *pts4, int64_t n_pts, void *workspace, double *stats, void *stream) {
    C2B_API_BEGIN
    if (!c || !c->comm || !workspace || !stats || n_cam < 0 || n_pts < 0 || cam_base < 0 || n_cam_global < cam_base + n_cam)
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: bad arguments");
    if (!aligned16(camblk) || !aligned16(cen4) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: camblk/cen4/pts4 must be 16-byte aligned");
    const int W = c->world, R = c->rank;
    const int64_t lo = n_pts * R / W, hi = n_pts * (R + 1) / W, n_ent = n_cam_global + n_pts;
    if (n_ent < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: empty problem");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device)
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: the current device (%d) is not the communicator's (%d)", cur, c->device);
    hipStream_t st = S(stream);
    // scratch [20 mine | W x 20 | 3 mean | 3 mine | W x 3] carved out of the workspace's partial slots, which the
    // statistics kernels do not use (they keep their records in front of them): no device allocation per call -- a
    // hipMalloc / hipFree pair synchronises the device, ~1 ms each, while the peers' collectives are in flight
    const size_t n_dev = 20 + 20 * (size_t)W + 3 + 3 + 3 * (size_t)W;
    if ((int64_t)n_dev > block_part_slots(0)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: more than %d ranks", 150);
    double *dev = reinterpret_cast<double *>(workspace) + kWsBlockPart;
    double *d_mine = dev, *d_all = dev + 20, *d_mean = d_all + 20 * (size_t)W, *d_sq = d_mean + 3, *d_sqall = d_sq + 3;
    std::vector<double> shares(20 * (size_t)W), sq(3 * (size_t)W);
    double host_stats[20];
    for (double &v : host_stats) v = std::numeric_limits<double>::quiet_NaN();
    // A rank that fails locally keeps taking part in BOTH all-gathers (its peers are already waiting in them) and
    // reports its first error afterwards: `first` carries it.
    int first = C2B_OK;
    char first_msg[sizeof g_err] = "";
    auto note = [&](int rc) { if (rc && !first) { first = rc; std::snprintf(first_msg, sizeof first_msg, "%s", g_err); } return rc; };
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) note(fail(C2B_ERR_HIP, "stats_sharded: %s: %s", what, hipGetErrorString(e)));
        return e == hipSuccess;
    };
    note(c2b_stats_partial_pass1(camblk, cen4, n_cam, cam_base, n_cam_global, pts4 + 4 * lo, hi - lo, lo, n_ent, workspace, d_mine, stream));
    note(c2b_comm_all_gather_f64(c, d_mine, 20, d_all, stream));
    if (hip(hipMemcpyAsync(shares.data(), d_all, shares.size() * sizeof(double), hipMemcpyDeviceToHost, st), "copy of the shares") &&
        hip(hipStreamSynchronize(st), "synchronize") && !first)
        note(c2b_stats_combine_shares(shares.data(), W, host_stats));
    hip(hipMemcpyAsync(d_mean, host_stats, 3 * sizeof(double), hipMemcpyHostToDevice, st), "upload of the mean");
    if (!first) note(c2b_stats_partial_pass2(camblk, cen4, n_cam, pts4 + 4 * lo, hi - lo, d_mean, workspace, d_sq, stream));
    note(c2b_comm_all_gather_f64(c, d_sq, 3, d_sqall, stream));
    if (hip(hipMemcpyAsync(sq.data(), d_sqall, sq.size() * sizeof(double), hipMemcpyDeviceToHost, st), "copy of the squared sums") &&
        hip(hipStreamSynchronize(st), "synchronize") && !first)
        note(c2b_stats_finish_shares(sq.data(), W, n_ent, host_stats));
    if (first) { std::snprintf(g_err, sizeof g_err, "%s", first_msg); return first; }
    HIP_TRY(hipMemcpyAsync(stats, host_stats, sizeof host_stats, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return C2B_OK;
    C2B_API_END("stats_sharded")
}

int c2b_add_drift_sharded(double *cam15, int64_t n_cam, int64_t cam_base, double *pts4, int64_t n_pts, const double *stats,
                          int normalized, double strength, double angle_strength, double std, double dir_x, double dir_y,
                          double dir_z, uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (!stats || cam_base < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "add_drift_sharded: bad arguments");
    return drift_impl<double>("add_drift_sharded", cam15, n_cam, pts4, n_pts, stats + 15, normalized ? stats : nullptr,
                              strength, angle_strength, std, dir_x, dir_y, dir_z, seed, S(stream), cam_base);
    C2B_API_END("add_drift_sharded")
}

int c2b_add_noise_entities_sharded(double *cam15, int64_t n_cam, int64_t cam_base, double *pts4, int64_t n_pts,
                                   const double *stats, double translation_std, double rotation_std, double point_std,
                                   uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (cam_base < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_entities_sharded: bad arguments");
    return noise_entities_impl<double>("add_noise_entities_sharded", cam15, n_cam, pts4, n_pts, stats, translation_std,
                                       rotation_std, point_std, seed, S(stream), cam_base);
    C2B_API_END("add_noise_entities_sharded")
}

int c2b_stats_f32(const float *cam15, int64_t n_cam, const float *pts4, int64_t n_pts, void *workspace,
                  double *stats, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || !stats || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_f32: bad arguments");
    if (n_cam + n_pts == 0) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_f32: empty problem");
    if ((n_cam && !cam15) || (n_pts && !pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_f32: NULL input");
    const SrcState32 src{cam15, reinterpret_cast<const float4 *>(pts4), n_cam};
    return stats_impl(src, n_cam + n_pts, workspace, stats, S(stream));
    C2B_API_END("stats_f32")
}

int c2b_convert_f64_to_f32(const double *src, int64_t n, float *dst, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!src || !dst))) return fail(C2B_ERR_INVALID_ARGUMENT, "convert_f64_to_f32: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), src, n, dst);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("convert_f64_to_f32")
}

int c2b_convert_f32_to_f64(const float *src, int64_t n, double *dst, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!src || !dst))) return fail(C2B_ERR_INVALID_ARGUMENT, "convert_f32_to_f64: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), src, n, dst);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("convert_f32_to_f64")
}

int c2b_add_drift(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *origin,
                  double strength, double angle_strength, double std, double dir_x, double dir_y, double dir_z,
                  uint64_t seed, void *stream) {
    C2B_API_BEGIN
    return drift_impl<double>("add_drift", cam15, n_cam, pts4, n_pts, origin, nullptr, strength, angle_strength, std,
                              dir_x, dir_y, dir_z, seed, S(stream));
    C2B_API_END("add_drift")
}
int c2b_add_drift_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *origin,
                      double strength, double angle_strength, double std, double dir_x, double dir_y, double dir_z,
                      uint64_t seed, void *stream) {
    C2B_API_BEGIN
    return drift_impl<float>("add_drift_f32", cam15, n_cam, pts4, n_pts, origin, nullptr, strength, angle_strength, std,
                             dir_x, dir_y, dir_z, seed, S(stream));
    C2B_API_END("add_drift_f32")
}
int c2b_add_drift_normalized(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *stats,
                             double strength, double angle_strength, double std, uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "add_drift_normalized: stats is NULL");
    return drift_impl<double>("add_drift_normalized", cam15, n_cam, pts4, n_pts, stats + 15, stats, strength,
                              angle_strength, std, 0.0, 0.0, 0.0, seed, S(stream));
    C2B_API_END("add_drift_normalized")
}
int c2b_add_drift_normalized_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats,
                                 double strength, double angle_strength, double std, uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "add_drift_normalized_f32: stats is NULL");
    return drift_impl<float>("add_drift_normalized_f32", cam15, n_cam, pts4, n_pts, stats + 15, stats, strength,
                             angle_strength, std, 0.0, 0.0, 0.0, seed, S(stream));
    C2B_API_END("add_drift_normalized_f32")
}

int c2b_add_noise_entities(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *stats,
                           double translation_std, double rotation_std, double point_std, uint64_t seed,
                           void *stream) {
    C2B_API_BEGIN
    return noise_entities_impl<double>("add_noise_entities", cam15, n_cam, pts4, n_pts, stats, translation_std,
                                       rotation_std, point_std, seed, S(stream));
    C2B_API_END("add_noise_entities")
}
int c2b_add_noise_entities_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats,
                               double translation_std, double rotation_std, double point_std, uint64_t seed,
                               void *stream) {
    C2B_API_BEGIN
    return noise_entities_impl<float>("add_noise_entities_f32", cam15, n_cam, pts4, n_pts, stats, translation_std,
                                      rotation_std, point_std, seed, S(stream));
    C2B_API_END("add_noise_entities_f32")
}

int c2b_add_noise_observations(double *uv, int64_t n_obs, int64_t obs_base, double observations_std, uint64_t seed,
                               void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || obs_base < 0 || (n_obs && !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations: bad arguments");
    if (!(observations_std >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise: standard deviations must be >= 0");
    if (!n_obs) return C2B_OK;
    if (!aligned16(uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations: uv must be 16-byte aligned");
    const int64_t n_tiles = (n_obs + 63) / 64, wgs = (n_tiles + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(k_add_noise_observations, dim3((unsigned)std::min<int64_t>(wgs, kNoiseGrid)), dim3(kBlock), 0, S(stream),
                       reinterpret_cast<double2 *>(uv), n_obs, n_tiles, obs_base, observations_std, seed);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("add_noise_observations")
}

int c2b_add_sin_noise(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *stats, double dir_x,
                      double dir_y, double dir_z, double ndir_x, double ndir_y, double ndir_z, double strength,
                      double frequency, void *stream) {
    C2B_API_BEGIN
    return sin_impl<double>("add_sin_noise", cam15, n_cam, pts4, n_pts, stats, dir_x, dir_y, dir_z, ndir_x, ndir_y, ndir_z,
                            strength, frequency, S(stream));
    C2B_API_END("add_sin_noise")
}
int c2b_add_sin_noise_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats, double dir_x,
                          double dir_y, double dir_z, double ndir_x, double ndir_y, double ndir_z, double strength,
                          double frequency, void *stream) {
    C2B_API_BEGIN
    return sin_impl<float>("add_sin_noise_f32", cam15, n_cam, pts4, n_pts, stats, dir_x, dir_y, dir_z, ndir_x, ndir_y,
                           ndir_z, strength, frequency, S(stream));
    C2B_API_END("add_sin_noise_f32")
}

int c2b_partition_cameras(const uint64_t *row_ptr, int64_t n_cam, int n_parts, int64_t *bounds) {
    C2B_API_BEGIN
    if (!row_ptr || !bounds || n_cam < 0 || n_parts < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "partition_cameras: bad arguments");
    const uint64_t total = row_ptr[n_cam];
    bounds[0] = 0;
    int64_t c = 0;
    for (int k = 1; k < n_parts; ++k) {
        // first camera whose prefix reaches k/n_parts of the observations
        const uint64_t target = (uint64_t)(((__uint128_t)total * (unsigned)k) / (unsigned)n_parts);
        int64_t lo = c, hi = n_cam;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (row_ptr[mid] < target) lo = mid + 1; else hi = mid;
        }
        c = lo;
        bounds[k] = c;
    }
    bounds[n_parts] = n_cam;
    return C2B_OK;
    C2B_API_END("partition_cameras")
}

int64_t c2b_visibility_dense_tiles(int64_t n_pts) { return n_pts <= 0 ? 0 : (n_pts + kDenseTile - 1) / kDenseTile; }

static void dense_grid(int64_t n_cam, int64_t n_tiles, dim3 *grid, int64_t *cams_per_chunk) {
    const int64_t bx = (n_tiles + kDenseWPB - 1) / kDenseWPB;
    // enough waves to fill 256 CUs a few times over, camera chunks in multiples of the LDS tile
    int64_t chunks = (16384 + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
    const int64_t max_chunks = (n_cam + kDenseCamTile - 1) / kDenseCamTile;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 65535) chunks = 65535;
    if (chunks < 1) chunks = 1;
    int64_t per = (n_cam + chunks - 1) / chunks;
    per = (per + kDenseCamTile - 1) / kDenseCamTile * kDenseCamTile;
    chunks = (n_cam + per - 1) / per;
    *grid = dim3((unsigned)bx, (unsigned)chunks);
    *cams_per_chunk = per;
}

static int dense_check(const char *who, const void *camblk, int64_t n_cam, const void *pts4, int64_t n_pts) {
    if (n_cam < 0 || n_pts < 0 || (n_cam && !camblk) || (n_pts && !pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    if (n_pts >= ((int64_t)1 << 32)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: point indices are 32-bit", who);
    if ((n_cam && !aligned16(camblk)) || (n_pts && !aligned16(pts4))) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: camblk/pts4 must be 16-byte aligned", who);
    return C2B_OK;
}

int c2b_visibility_dense_count(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts, double max_dist,
                               uint32_t *tile_counts, uint64_t *cam_total, uint64_t *row_ptr, void *stream) {
    C2B_API_BEGIN
    int rc = dense_check("visibility_dense_count", camblk, n_cam, pts4, n_pts);
    if (rc) return rc;
    if (!row_ptr) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_dense_count: row_ptr is NULL");
    const int64_t n_tiles = c2b_visibility_dense_tiles(n_pts);
    if (!n_cam || !n_tiles) {
        HIP_TRY(hipMemsetAsync(row_ptr, 0, sizeof(uint64_t) * (size_t)(n_cam + 1), S(stream)));
        return C2B_OK;
    }
    if (!tile_counts || !cam_total) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_dense_count: NULL scratch");
    dim3 grid;
    int64_t per;
    dense_grid(n_cam, n_tiles, &grid, &per);
    // the count pass writes non-empty (camera, tile) cells only
    HIP_TRY(hipMemsetAsync(tile_counts, 0, sizeof(uint32_t) * (size_t)n_cam * (size_t)n_tiles, S(stream)));
    hipLaunchKernelGGL(k_visibility_dense<false>, grid, dim3(kDenseWPB * 64), 0, S(stream), camblk, n_cam, per,
                       reinterpret_cast<const double4 *>(pts4), n_pts, n_tiles, max_dist, tile_counts,
                       (const uint64_t *)nullptr, (uint32_t *)nullptr, (double2 *)nullptr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dense_row_scan, dim3((unsigned)n_cam), dim3(256), 0, S(stream), tile_counts, n_tiles, cam_total);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dense_cam_scan, dim3(1), dim3(256), 0, S(stream), (const uint64_t *)cam_total, n_cam, row_ptr);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_dense_count")
}

int c2b_visibility_dense_fill(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts, double max_dist,
                              const uint32_t *tile_offsets, const uint64_t *row_ptr, uint32_t *pt_idx, double *uv,
                              void *stream) {
    C2B_API_BEGIN
    int rc = dense_check("visibility_dense_fill", camblk, n_cam, pts4, n_pts);
    if (rc) return rc;
    const int64_t n_tiles = c2b_visibility_dense_tiles(n_pts);
    if (!n_cam || !n_tiles) return C2B_OK;
    if (!tile_offsets || !row_ptr || !pt_idx || !uv || !aligned16(uv))
        return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_dense_fill: NULL/misaligned buffer");
    dim3 grid;
    int64_t per;
    dense_grid(n_cam, n_tiles, &grid, &per);
    hipLaunchKernelGGL(k_visibility_dense<true>, grid, dim3(kDenseWPB * 64), 0, S(stream), camblk, n_cam, per,
                       reinterpret_cast<const double4 *>(pts4), n_pts, n_tiles, max_dist,
                       const_cast<uint32_t *>(tile_offsets), row_ptr, pt_idx, reinterpret_cast<double2 *>(uv));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_dense_fill")
}

