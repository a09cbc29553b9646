// capi_problem.hpp -- Level 1 of include/city2ba_hip.h: a BAProblem resident on one device (c2b_problem_*), its device-side generators, cull, file images, noise functions, and the *_sharded forms
// Part of the one translation unit of the C ABI: included by capi.hip (inside its extern "C" block, after its helpers and
// launchers), never compiled or included on its own.

/* ------------------------------- level 1 --------------------------------------------- */

struct c2b_problem {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n_cam = 0, n_pts = 0, n_obs = 0;
    double *cam15 = nullptr, *bal9 = nullptr, *camblk = nullptr, *cen4 = nullptr, *pts4 = nullptr, *uv = nullptr;
    // cen4[n_cam][4]: the cameras' centres as 32-byte rows (derived with camblk, valid when blk_valid): what the statistics
    // and the centre-keyed cell lists read -- a 128-byte line of camblk per camera otherwise
    uint32_t *cam_idx = nullptr, *pt_idx = nullptr;
    void *ws = nullptr;
    double *stats = nullptr, *scalar = nullptr;
    // this problem as ONE SHARD of a larger one (c2b_problem_set_shard): its cameras are [shard_cam_base, + n_cam) of
    // shard_n_cam_global (< 0: not a shard), its first observation is observation shard_obs_base of the whole list
    int64_t shard_cam_base = 0, shard_n_cam_global = -1, shard_obs_base = 0;
    c2b_problem_options opt{0, 0, 0, 0, 0, 0, -1};   // c2b_problem_set_options; survives uploads / reads (it is the handle's, not the data's)
    bool bal_valid = false;     // bal9 still describes the cameras (no mutation since upload_bal)
    bool blk_valid = false;     // camblk matches cam15 (and bal_valid mode)
    bool bal9_fresh = false;    // !bal_valid, but bal9 holds to_vec of the current cameras (the last write / download_bal computed it)
    // the row structure of the observation list for the *_rows launchers, rebuilt on demand after the list changed
    uint64_t *rows_ptr = nullptr;
    void *rows_tiles = nullptr;
    bool rows_valid = false;
    uint32_t *dense_pt = nullptr;   // survivors of the last dense visibility sweep
    double *dense_uv = nullptr;
    uint64_t *dense_row = nullptr;  // its CSR row pointer [n_cam + 1], kept for the occlusion filter
    int64_t dense_n = 0;
    // residual + Jacobian to host buffers: a ring of chunk-sized device buffers, a copy stream, per-slot events
    static constexpr int kJacSlots = 3;
    static constexpr int64_t kJacChunk = 256 * 1024;       // observations per chunk (53 MB of results)
    double *jac_ring = nullptr;                            // kJacSlots x kJacChunk x 26 doubles
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_done[kJacSlots] = {nullptr, nullptr, nullptr}, ev_free[kJacSlots] = {nullptr, nullptr, nullptr};
};

static void free_dense(c2b_problem *p) {
    if (p->dense_pt) (void)hipFree(p->dense_pt);
    if (p->dense_uv) (void)hipFree(p->dense_uv);
    if (p->dense_row) (void)hipFree(p->dense_row);
    p->dense_pt = nullptr; p->dense_uv = nullptr; p->dense_row = nullptr; p->dense_n = 0;
}

// the observation list changed (upload, cull, adopted visibility): its row structure is rebuilt by the next user
static void drop_rows(c2b_problem *p) {
    if (p->rows_ptr) (void)hipFree(p->rows_ptr);
    if (p->rows_tiles) (void)hipFree(p->rows_tiles);
    p->rows_ptr = nullptr; p->rows_tiles = nullptr; p->rows_valid = false;
}

static void free_buffers(c2b_problem *p) {
    void *ptrs[] = {p->cam15, p->bal9, p->camblk, p->cen4, p->pts4, p->uv, p->cam_idx, p->pt_idx, p->ws, p->stats, p->scalar, p->jac_ring};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    p->jac_ring = nullptr;
    free_dense(p);
    drop_rows(p);
    p->cam15 = p->bal9 = p->camblk = p->cen4 = p->pts4 = p->uv = nullptr;
    p->cam_idx = p->pt_idx = nullptr;
    p->ws = nullptr; p->stats = p->scalar = nullptr;
    p->n_cam = p->n_pts = p->n_obs = 0;
    p->bal_valid = p->blk_valid = p->bal9_fresh = false;
}

void c2b_problem_options_init(c2b_problem_options *o) {
    if (o) *o = c2b_problem_options{0, 0, 0, 0, 0, 0, -1};
}
int c2b_problem_set_options(c2b_problem *p, const c2b_problem_options *o) {
    if (!p || !o) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_set_options: NULL argument");
    if (o->read_threads < 0 || o->read_threads > 64 || o->io_threads < 0 || o->io_threads > 64 || o->rank_sort_max_row < 0 || o->reserved != 0)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_set_options: thread counts must be in [0, 64], rank_sort_max_row >= 0, reserved 0");
    p->opt = *o;
    return C2B_OK;
}
int c2b_problem_get_options(const c2b_problem *p, c2b_problem_options *o) {
    if (!p || !o) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_get_options: NULL argument");
    *o = p->opt;
    return C2B_OK;
}
void c2b_host_set_io_threads(int n) { c2b_host::io_threads_setting().store(n < 1 ? 0 : (n > 64 ? 64 : n), std::memory_order_relaxed); }

int c2b_problem_create(int device, c2b_problem **out) {
    C2B_API_BEGIN
    if (!out) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(C2B_ERR_NO_DEVICE, "problem_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_create: device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    c2b_problem *p = new (std::nothrow) c2b_problem();
    if (!p) return fail(C2B_ERR_OOM, "problem_create: host allocation failed");
    p->device = device;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete p; return fail(C2B_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = p;
    return C2B_OK;
    C2B_API_END("problem_create")
}

void c2b_problem_destroy(c2b_problem *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    free_buffers(p);
    for (int k = 0; k < c2b_problem::kJacSlots; ++k) {
        if (p->ev_done[k]) (void)hipEventDestroy(p->ev_done[k]);
        if (p->ev_free[k]) (void)hipEventDestroy(p->ev_free[k]);
    }
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static int ensure_camblk(c2b_problem *p) {
    if (p->blk_valid) return C2B_OK;
    int rc = p->bal_valid ? c2b_camblk_from_bal(p->bal9, p->n_cam, p->camblk, cam_table_doubles(p->n_cam), p->cen4, p->stream)
                          : c2b_camblk_from_state(p->cam15, p->n_cam, p->camblk, cam_table_doubles(p->n_cam), p->cen4, p->stream);
    if (rc) return rc;
    p->blk_valid = true;
    return C2B_OK;
}

// the resident arrays of a problem with these sizes (whatever it held before is freed); contents undefined
static int alloc_problem(c2b_problem *p, int64_t n_cam, int64_t n_pts, int64_t n_obs) {
    HIP_TRY(hipSetDevice(p->device));
    free_buffers(p);
    auto dalloc = [&](void **q, size_t bytes) -> hipError_t { return hipMalloc(q, bytes ? bytes : 16); };
    HIP_TRY(dalloc((void **)&p->cam15, sizeof(double) * 15 * n_cam));
    HIP_TRY(dalloc((void **)&p->bal9, sizeof(double) * 9 * n_cam));
    HIP_TRY(dalloc((void **)&p->camblk, sizeof(double) * (size_t)cam_table_doubles(n_cam)));      // whole groups of 8 cameras
    HIP_TRY(dalloc((void **)&p->cen4, sizeof(double) * 4 * n_cam));
    HIP_TRY(dalloc((void **)&p->pts4, sizeof(double) * 4 * n_pts));
    HIP_TRY(dalloc((void **)&p->uv, sizeof(double) * 2 * n_obs));
    HIP_TRY(dalloc((void **)&p->cam_idx, sizeof(uint32_t) * n_obs));
    HIP_TRY(dalloc((void **)&p->pt_idx, sizeof(uint32_t) * n_obs));
    HIP_TRY(dalloc(&p->ws, (size_t)c2b_workspace_bytes(n_obs)));
    if (int rc = c2b_workspace_init(p->ws, p->stream)) return rc;
    HIP_TRY(dalloc((void **)&p->stats, sizeof(double) * C2B_STATS_DOUBLES));
    HIP_TRY(dalloc((void **)&p->scalar, sizeof(double) * 2));
    p->n_cam = n_cam; p->n_pts = n_pts; p->n_obs = n_obs;
    return C2B_OK;
}

static int upload_common(c2b_problem *p, int64_t n_cam, const double *cams, bool is_bal, int64_t n_pts,
                         const double *pts3, const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: problem is NULL");
    if (n_cam < 0 || n_pts < 0 || (n_cam && !cams) || (n_pts && !pts3) || !row_ptr)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: bad arguments");
    if (n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: device indices are 32-bit");
    // assert!(cams.len() == obs.len()) is structural here; row_ptr must be a monotone prefix
    if (row_ptr[0] != 0) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: row_ptr[0] != 0");
    for (int64_t c = 0; c < n_cam; ++c)
        if (row_ptr[c + 1] < row_ptr[c]) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: row_ptr not monotone at camera %lld", (long long)c);
    const int64_t n_obs = (int64_t)row_ptr[n_cam];
    if (n_obs && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: NULL observations");
    std::vector<uint32_t> pi32((size_t)n_obs);
    for (int64_t o = 0; o < n_obs; ++o) {
        // assert!(ci < &points.len()), src/baproblem.rs:368
        if (pt_idx[o] >= (uint64_t)n_pts)
            return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "problem_upload: observation %lld refers to point %llu >= %lld",
                        (long long)o, (unsigned long long)pt_idx[o], (long long)n_pts);
        pi32[(size_t)o] = (uint32_t)pt_idx[o];
    }
    if (int rc = alloc_problem(p, n_cam, n_pts, n_obs)) return rc;
    auto dalloc = [&](void **q, size_t bytes) -> hipError_t { return hipMalloc(q, bytes ? bytes : 16); };

    // staging through temporary device buffers (row_ptr, packed points)
    uint64_t *d_row = nullptr;
    double *d_p3 = nullptr;
    HIP_TRY(dalloc((void **)&d_row, sizeof(uint64_t) * (n_cam + 1)));
    hipError_t e = dalloc((void **)&d_p3, sizeof(double) * 3 * n_pts);
    if (e != hipSuccess) { (void)hipFree(d_row); return fail(C2B_ERR_OOM, "problem_upload: %s", hipGetErrorString(e)); }
    int rc = C2B_OK;
    do {
#define UP_TRY(expr) { hipError_t e2 = (expr); if (e2 != hipSuccess) { rc = fail(C2B_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e2)); break; } }
        UP_TRY(hipMemcpyAsync(d_row, row_ptr, sizeof(uint64_t) * (n_cam + 1), hipMemcpyHostToDevice, p->stream));
        if (n_pts) UP_TRY(hipMemcpyAsync(d_p3, pts3, sizeof(double) * 3 * n_pts, hipMemcpyHostToDevice, p->stream));
        if (n_obs) {
            UP_TRY(hipMemcpyAsync(p->pt_idx, pi32.data(), sizeof(uint32_t) * n_obs, hipMemcpyHostToDevice, p->stream));
            UP_TRY(hipMemcpyAsync(p->uv, uv, sizeof(double) * 2 * n_obs, hipMemcpyHostToDevice, p->stream));
        }
        if (is_bal) {
            if (n_cam) UP_TRY(hipMemcpyAsync(p->bal9, cams, sizeof(double) * 9 * n_cam, hipMemcpyHostToDevice, p->stream));
            if ((rc = c2b_cameras_from_bal(p->bal9, n_cam, p->cam15, p->stream))) break;
        } else {
            if (n_cam) UP_TRY(hipMemcpyAsync(p->cam15, cams, sizeof(double) * 15 * n_cam, hipMemcpyHostToDevice, p->stream));
        }
        if ((rc = c2b_points_pad(d_p3, n_pts, p->pts4, p->stream))) break;
        if ((rc = c2b_expand_rows(d_row, n_cam, 0, n_obs, p->cam_idx, p->stream))) break;
        UP_TRY(hipStreamSynchronize(p->stream));
#undef UP_TRY
    } while (0);
    (void)hipFree(d_row);
    (void)hipFree(d_p3);
    if (rc) return rc;
    p->bal_valid = is_bal;
    p->blk_valid = false;
    return C2B_OK;
}

int c2b_problem_upload(c2b_problem *p, int64_t n_cam, const double *cams15, int64_t n_pts, const double *pts3,
                       const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    return upload_common(p, n_cam, cams15, false, n_pts, pts3, row_ptr, pt_idx, uv);
    C2B_API_END("problem_upload")
}

int c2b_problem_upload_bal(c2b_problem *p, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                           const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    return upload_common(p, n_cam, bal9, true, n_pts, pts3, row_ptr, pt_idx, uv);
    C2B_API_END("problem_upload_bal")
}

// synthetic_grid's / synthetic_line's layout loops (src/synthetic.rs:178-258, :323-344) straight into the resident problem:
// cameras by Camera::from_position_direction, points, no observations yet (the visibility loop adds them).  Entity for
// entity and bit for bit what c2b_synthetic_grid_layout + c2b_problem_from_position_direction + c2b_problem_upload give,
// without 2 x 112 MB crossing PCIe.  The orientations' sines and cosines come from the host's libm like the host
// layout's (Basis3::from_angle_y(Deg(..)), :191-205).
static GridDirs layout_dirs() {
    GridDirs d;
    c2b_host::basis_from_angle_y_deg(-90.0, d.m[0]);
    c2b_host::basis_from_angle_y_deg(90.0, d.m[1]);
    c2b_host::basis_from_angle_y_deg(180.0, d.m[2]);
    const double one[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::copy(one, one + 9, d.m[3]);
    return d;
}

int c2b_problem_synthetic_grid_layout(c2b_problem *p, int64_t cpb, int64_t ppb, int64_t blocks, double block_length,
                                      double block_inset, double camera_height, double point_height) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_grid_layout: problem is NULL");
    if (cpb < 0 || ppb < 0 || blocks < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_grid_layout: bad arguments");
    // assert!(block_inset * 2. < block_length, ...), src/synthetic.rs:177
    if (!(block_inset * 2.0 < block_length))
        return fail(C2B_ERR_INVALID_ARGUMENT,
                    "Block inset (%g) must be less than half the block length (%g), to not violate physical constraints.",
                    block_inset, block_length);
    int64_t n_cam = 0, n_pts = 0;
    c2b_host::grid_sizes(cpb, ppb, blocks, &n_cam, &n_pts);
    if (n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_grid_layout: device indices are 32-bit");
    int rc = alloc_problem(p, n_cam, n_pts, 0);
    if (rc) return rc;
    if (n_cam) hipLaunchKernelGGL(k_grid_cameras, dim3(blocks_for(n_cam, 256)), dim3(256), 0, p->stream, n_cam, cpb, blocks, block_length,
                                  camera_height, layout_dirs(), p->cam15);
    if (n_pts) hipLaunchKernelGGL(k_grid_points, dim3(blocks_for(n_pts, 256)), dim3(256), 0, p->stream, n_pts, ppb, blocks, block_length,
                                  block_inset, point_height, reinterpret_cast<double4 *>(p->pts4));
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->bal_valid = false; p->blk_valid = false;
    return C2B_OK;
    C2B_API_END("problem_synthetic_grid_layout")
}

int c2b_problem_synthetic_line_layout(c2b_problem *p, int64_t n_cam, int64_t n_pts, double length, double point_offset,
                                      double camera_height, double point_height) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_line_layout: problem is NULL");
    if (n_cam < 0 || n_pts < 0 || n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_line_layout: bad arguments");
    int rc = alloc_problem(p, n_cam, n_pts, 0);
    if (rc) return rc;
    const int64_t n = std::max(n_cam, n_pts);
    if (n) hipLaunchKernelGGL(k_line_layout, dim3(blocks_for(n, 256)), dim3(256), 0, p->stream, n_cam, n_pts, length, point_offset,
                              camera_height, point_height, layout_dirs(), p->cam15, reinterpret_cast<double4 *>(p->pts4));
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->bal_valid = false; p->blk_valid = false;
    return C2B_OK;
    C2B_API_END("problem_synthetic_line_layout")
}

int c2b_problem_sizes(const c2b_problem *p, int64_t *n_cam, int64_t *n_pts, int64_t *n_obs) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_sizes: problem is NULL");
    if (n_cam) *n_cam = p->n_cam;
    if (n_pts) *n_pts = p->n_pts;
    if (n_obs) *n_obs = p->n_obs;
    return C2B_OK;
    C2B_API_END("problem_sizes")
}

#define NEED_UPLOADED(p, who)                                                              \
    if (!(p)) return fail(C2B_ERR_INVALID_ARGUMENT, who ": problem is NULL");              \
    if (!(p)->ws) return fail(C2B_ERR_INVALID_ARGUMENT, who ": nothing uploaded");         \
    HIP_TRY(hipSetDevice((p)->device));

int c2b_problem_download(c2b_problem *p, double *cams15, double *pts3, double *uv) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_download");
    if (cams15 && p->n_cam)
        HIP_TRY(hipMemcpyAsync(cams15, p->cam15, sizeof(double) * 15 * p->n_cam, hipMemcpyDeviceToHost, p->stream));
    if (uv && p->n_obs)
        HIP_TRY(hipMemcpyAsync(uv, p->uv, sizeof(double) * 2 * p->n_obs, hipMemcpyDeviceToHost, p->stream));
    double *d_p3 = nullptr;
    if (pts3 && p->n_pts) {
        HIP_TRY(hipMalloc((void **)&d_p3, sizeof(double) * 3 * p->n_pts));
        int rc = c2b_points_unpad(p->pts4, p->n_pts, d_p3, p->stream);
        if (rc) { (void)hipFree(d_p3); return rc; }
        hipError_t e = hipMemcpyAsync(pts3, d_p3, sizeof(double) * 3 * p->n_pts, hipMemcpyDeviceToHost, p->stream);
        if (e != hipSuccess) { (void)hipFree(d_p3); return fail(C2B_ERR_HIP, "download points: %s", hipGetErrorString(e)); }
    }
    hipError_t e = hipStreamSynchronize(p->stream);
    if (d_p3) (void)hipFree(d_p3);
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_download: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_download")
}

int c2b_problem_download_bal(c2b_problem *p, double *bal9) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_download_bal");
    if (!bal9) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_download_bal: bal9 is NULL");
    if (!p->n_cam) return C2B_OK;
    if (!p->bal_valid && !p->bal9_fresh) {
        // to_vec (src/baproblem.rs:189-202) of the current state
        int rc = c2b_cameras_to_bal(p->cam15, p->n_cam, p->bal9, p->stream);
        if (rc) return rc;
        p->bal9_fresh = true;
    }
    HIP_TRY(hipMemcpyAsync(bal9, p->bal9, sizeof(double) * 9 * p->n_cam, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_download_bal")
}

int c2b_problem_from_position_direction(c2b_problem *p, int64_t n_cam, const double *pos3, const double *dir9,
                                        double *cams15) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_from_position_direction: problem is NULL");
    if (n_cam < 0 || (n_cam && (!pos3 || !dir9 || !cams15)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_from_position_direction: bad arguments");
    if (!n_cam) return C2B_OK;
    HIP_TRY(hipSetDevice(p->device));
    double *d_pos = nullptr, *d_dir = nullptr, *d_cam = nullptr;
    int rc = C2B_OK;
    hipError_t e = hipMalloc((void **)&d_pos, sizeof(double) * 3 * n_cam);
    if (e == hipSuccess) e = hipMalloc((void **)&d_dir, sizeof(double) * 9 * n_cam);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cam, sizeof(double) * 15 * n_cam);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pos, pos3, sizeof(double) * 3 * n_cam, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dir, dir9, sizeof(double) * 9 * n_cam, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        rc = c2b_cameras_from_position_direction(d_pos, d_dir, n_cam, d_cam, p->stream);
        if (!rc) e = hipMemcpyAsync(cams15, d_cam, sizeof(double) * 15 * n_cam, hipMemcpyDeviceToHost, p->stream);
        hipError_t e2 = hipStreamSynchronize(p->stream);
        if (e == hipSuccess) e = e2;
    }
    if (d_pos) (void)hipFree(d_pos);
    if (d_dir) (void)hipFree(d_dir);
    if (d_cam) (void)hipFree(d_cam);
    if (rc) return rc;
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_from_position_direction: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_from_position_direction")
}

int c2b_problem_centers(c2b_problem *p, double *centers3) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_centers");
    if (!p->n_cam) return C2B_OK;
    if (!centers3) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_centers: centers3 is NULL");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    // the compact centre table: 32-byte rows, the centre in the first three doubles
    HIP_TRY(hipMemcpy2DAsync(centers3, 3 * sizeof(double), p->cen4, 4 * sizeof(double),
                             3 * sizeof(double), (size_t)p->n_cam, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_centers")
}

// row_ptr (from the camera-major cam_idx) and the tile records of the current observation list
static int ensure_rows(c2b_problem *p) {
    if (p->rows_valid || !p->n_obs) return C2B_OK;
    drop_rows(p);
    HIP_TRY(hipMalloc((void **)&p->rows_ptr, sizeof(uint64_t) * (size_t)(p->n_cam + 1)));
    HIP_TRY(hipMalloc(&p->rows_tiles, (size_t)c2b_rows_tiles_bytes(p->n_obs)));
    hipLaunchKernelGGL(k_rows_from_sorted, dim3(blocks_for(p->n_obs + 1)), dim3(kBlock), 0, p->stream, (const uint32_t *)p->cam_idx,
                       p->n_obs, p->n_cam, p->rows_ptr);
    LAUNCH_CHECK();
    const int rc = c2b_rows_pack(p->rows_ptr, p->n_cam, p->n_obs, p->rows_tiles, p->stream);
    if (rc) return rc;
    p->rows_valid = true;
    return C2B_OK;
}

int c2b_problem_project(c2b_problem *p, double *uv_out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_project");
    if (!p->n_obs) return C2B_OK;
    if (!uv_out) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_project: uv_out is NULL");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    double *d_uv = nullptr;
    HIP_TRY(hipMalloc((void **)&d_uv, sizeof(double) * 2 * p->n_obs));
    rc = c2b_project_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->n_obs, d_uv, p->stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(uv_out, d_uv, sizeof(double) * 2 * p->n_obs, hipMemcpyDeviceToHost, p->stream);
    hipError_t e2 = hipStreamSynchronize(p->stream);
    (void)hipFree(d_uv);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(C2B_ERR_HIP, "problem_project: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return C2B_OK;
    C2B_API_END("problem_project")
}

int c2b_problem_total_reprojection_error(c2b_problem *p, double norm, double *out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_error");
    if (!out) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_error: out is NULL");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    rc = c2b_reprojection_error_sum_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                         norm, p->ws, p->scalar, p->stream);
    if (rc) return rc;
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, p->scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *out = std::pow(sum, 1.0 / norm);          // .powf(1. / norm), src/baproblem.rs:278
    return C2B_OK;
    C2B_API_END("problem_total_reprojection_error")
}

// The same for a problem that is one SHARD (a contiguous camera range) of a larger one: the local sum, one 8-byte
// all-reduce through the communicator on the problem's stream, then .powf(1/norm) -- every rank returns the global
// error (src/baproblem.rs:265-279 over all shards).  Collective: every rank of the communicator must call it.
int c2b_problem_total_reprojection_error_sharded(c2b_problem *p, c2b_comm *comm, double norm, double *out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_error_sharded");
    if (!out || !comm) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_error_sharded: NULL argument");
    if (comm->device != p->device) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_error_sharded: communicator and problem live on different devices");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    if (p->n_obs > 0)
        rc = c2b_reprojection_error_sum_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                             norm, p->ws, p->scalar, p->stream);
    else
        HIP_TRY(hipMemsetAsync(p->scalar, 0, sizeof(double), p->stream));          // an empty shard still takes part
    if (!rc) rc = c2b_comm_all_reduce_sum_f64(comm, p->scalar, 1, p->stream);
    if (rc) return rc;
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, p->scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *out = std::pow(sum, 1.0 / norm);
    return C2B_OK;
    C2B_API_END("problem_total_reprojection_error_sharded")
}

// Both norms run_noise prints (src/bin/city2ba.rs:283-287, 350-354) from ONE pass over the observations.
static int errors_l1_l2_impl(c2b_problem *p, c2b_comm *comm, double *l1, double *l2) {
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    if (p->n_obs > 0)
        rc = c2b_reprojection_error_sums2_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                               p->ws, p->scalar, p->stream);
    else
        HIP_TRY(hipMemsetAsync(p->scalar, 0, 2 * sizeof(double), p->stream));      // an empty shard still takes part
    if (!rc && comm) rc = c2b_comm_all_reduce_sum_f64(comm, p->scalar, 2, p->stream);   // ONE 2-element all-reduce
    if (rc) return rc;
    double sums[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(sums, p->scalar, 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *l1 = std::pow(sums[0], 1.0 / 1.0);        // .powf(1. / norm), src/baproblem.rs:278
    *l2 = std::pow(sums[1], 1.0 / 2.0);
    return C2B_OK;
}

int c2b_problem_total_reprojection_errors_l1_l2(c2b_problem *p, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_errors_l1_l2");
    if (!l1 || !l2) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_errors_l1_l2: NULL output");
    return errors_l1_l2_impl(p, nullptr, l1, l2);
    C2B_API_END("problem_total_reprojection_errors_l1_l2")
}

int c2b_problem_total_reprojection_errors_l1_l2_sharded(c2b_problem *p, c2b_comm *comm, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_errors_l1_l2_sharded");
    if (!l1 || !l2 || !comm) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_errors_l1_l2_sharded: NULL argument");
    if (comm->device != p->device) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_errors_l1_l2_sharded: communicator and problem live on different devices");
    return errors_l1_l2_impl(p, comm, l1, l2);
    C2B_API_END("problem_total_reprojection_errors_l1_l2_sharded")
}

// Results leave in chunks of kJacChunk observations through a ring of kJacSlots device buffers: the kernel of chunk
// k + 1 is queued before the copies of chunk k start, copies run on their own stream, so PCIe and the kernel overlap
// and the device never holds more than the ring (159 MB) whatever the problem size.  Host buffers from
// c2b_host_alloc (pinned) take the copies at link speed; ordinary pageable memory works too, at the runtime's staged
// rate.  The ring, the copy stream and the events are created on first use and live as long as the problem.
int c2b_problem_residual_jacobian(c2b_problem *p, double *r, double *Jc, double *Jp) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_residual_jacobian");
    if (!p->n_obs) return C2B_OK;
    if (!r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_residual_jacobian: NULL output");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    constexpr int kSlots = c2b_problem::kJacSlots;
    constexpr int64_t kChunk = c2b_problem::kJacChunk;
    if (!p->jac_ring) HIP_TRY(hipMalloc((void **)&p->jac_ring, sizeof(double) * 26 * (size_t)kChunk * kSlots));
    if (!p->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < kSlots; ++k) {
        if (!p->ev_done[k]) HIP_TRY(hipEventCreateWithFlags(&p->ev_done[k], hipEventDisableTiming));
        if (!p->ev_free[k]) HIP_TRY(hipEventCreateWithFlags(&p->ev_free[k], hipEventDisableTiming));
    }
    const int64_t n = p->n_obs, n_chunks = (n + kChunk - 1) / kChunk;
    auto slot_r = [&](int s) { return p->jac_ring + (size_t)s * 26 * kChunk; };
    auto slot_Jc = [&](int s) { return slot_r(s) + 2 * kChunk; };
    auto slot_Jp = [&](int s) { return slot_r(s) + 20 * kChunk; };
    auto launch = [&](int64_t k) -> int {
        const int s = (int)(k % kSlots);
        const int64_t o0 = k * kChunk, m = (n - o0 < kChunk) ? n - o0 : kChunk;
        if (k >= kSlots) { HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_free[s], 0)); }      // its previous copies are out
        // kJacChunk is a multiple of 64: every chunk starts on a tile record
        // (n_pts = 0: a chunk's working set is small and this path is bound by the PCIe copies; loads stay cached)
        int rc2 = c2b_residual_jacobian_rows(p->camblk, p->pts4, 0, p->rows_ptr, p->n_cam, (const char *)p->rows_tiles + (o0 >> 6) * 16, o0,
                                             p->pt_idx + o0, p->uv + 2 * o0, m, slot_r(s), slot_Jc(s), slot_Jp(s), 2.0, nullptr,
                                             nullptr, p->stream);
        if (rc2) return rc2;
        HIP_TRY(hipEventRecord(p->ev_done[s], p->stream));
        return C2B_OK;
    };
    hipError_t e = hipSuccess;
    rc = launch(0);
    for (int64_t k = 0; k < n_chunks && !rc && e == hipSuccess; ++k) {
        if (k + 1 < n_chunks) rc = launch(k + 1);            // queued BEFORE chunk k's copies: they overlap
        if (rc) break;
        const int s = (int)(k % kSlots);
        const int64_t o0 = k * kChunk, m = (n - o0 < kChunk) ? n - o0 : kChunk;
        e = hipStreamWaitEvent(p->copy_stream, p->ev_done[s], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(r + 2 * o0, slot_r(s), sizeof(double) * 2 * m, hipMemcpyDeviceToHost, p->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(Jc + 18 * o0, slot_Jc(s), sizeof(double) * 18 * m, hipMemcpyDeviceToHost, p->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(Jp + 6 * o0, slot_Jp(s), sizeof(double) * 6 * m, hipMemcpyDeviceToHost, p->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(p->ev_free[s], p->copy_stream);
    }
    const hipError_t e1 = hipStreamSynchronize(p->copy_stream), e2 = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_residual_jacobian: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_residual_jacobian")
}

// The same launch with the results left ON THE DEVICE, in output arrays placed for streaming stores: what a
// BAProblem-level caller that consumes the Jacobian on the GPU (a solver's normal equations) calls in its loop.  The
// whole list in ONE launch -- residual, both blocks and the folded sum of squared residuals -- at the Level-0 headline
// rate; nothing crosses PCIe but the 8-byte sum.  *outputs == NULL: a set is allocated by c2b_jacobian_outputs_alloc
// (max_attempts placements tried, as there) and handed to the caller, who passes it back on later calls (it is reused as
// long as the observation count matches) and frees it with c2b_jacobian_outputs_free.
int c2b_problem_residual_jacobian_device(c2b_problem *p, int max_attempts, c2b_jacobian_outputs **outputs, double *sum_sq) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_residual_jacobian_device");
    if (!outputs) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_residual_jacobian_device: outputs is NULL");
    if (*outputs && ((*outputs)->n_obs != p->n_obs || (*outputs)->device != p->device))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_residual_jacobian_device: the output set holds %lld observations on device %d, the problem %lld on device %d",
                    (long long)(*outputs)->n_obs, (*outputs)->device, (long long)p->n_obs, p->device);
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    const bool mine = *outputs == nullptr;
    if (mine) {
        rc = c2b_jacobian_outputs_alloc(p->n_obs, max_attempts, 0.0, p->stream, outputs);
        if (rc) return rc;
    }
    c2b_jacobian_outputs *h = *outputs;
    // into the placed set: the workgroup shape follows the store rate measured for it
    rc = c2b_residual_jacobian_rows_placed(p->camblk, p->pts4, p->n_pts, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                           h, 2.0, p->ws, p->scalar, p->stream);
    double sum = 0.0;
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(&sum, p->scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream);
    const hipError_t e2 = hipStreamSynchronize(p->stream);
    if (rc || e != hipSuccess || e2 != hipSuccess) {
        if (mine) { c2b_jacobian_outputs_free(h); *outputs = nullptr; }
        if (rc) return rc;
        return fail(C2B_ERR_HIP, "problem_residual_jacobian_device: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    if (sum_sq) *sum_sq = sum;
    return C2B_OK;
    C2B_API_END("problem_residual_jacobian_device")
}

int c2b_host_alloc(void **ptr, int64_t bytes) {
    C2B_API_BEGIN
    if (!ptr || bytes < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "host_alloc: bad arguments");
    *ptr = nullptr;
    if (!bytes) return C2B_OK;
    HIP_TRY(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return C2B_OK;
    C2B_API_END("host_alloc")
}

void c2b_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

static int compute_stats(c2b_problem *p) {
    // The statistics read the centre table only.  When the camera table is stale because the in-memory cameras MOVED (drift, noise:
    // bal_valid is off then), the centres alone are derived from the state -- k_cameras_centers, the bits k_cameras_prepare would write
    // -- instead of the whole table: run_noise's add_noise asks for std() right after add_drift (src/noise.rs:133) and its entity
    // pass invalidates a camera table again before any pass reads it (38.6 -> ~20 us of that flow, r06).  camblk stays stale.
    if (!p->blk_valid && !p->bal_valid && p->n_cam > 0) {
        hipLaunchKernelGGL(k_cameras_centers, dim3(blocks_for(p->n_cam)), dim3(kBlock), 0, p->stream, p->cam15, p->n_cam, p->cen4);
        LAUNCH_CHECK();
    } else {
        int rc = ensure_camblk(p);
        if (rc) return rc;
    }
    return c2b_stats(p->camblk, p->cen4, p->n_cam, p->pts4, p->n_pts, p->ws, p->stats, p->stream);
}

int c2b_problem_stats(c2b_problem *p, double *stats) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_stats");
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_stats: stats is NULL");
    int rc = compute_stats(p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof(double) * C2B_STATS_DOUBLES, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_stats")
}

int c2b_problem_visibility_pairs(c2b_problem *p, int64_t n_pairs, const uint32_t *cam_idx, const uint32_t *pt_idx,
                                 double max_dist, double *uv_out, uint8_t *keep) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_pairs");
    if (n_pairs < 0 || (n_pairs && (!cam_idx || !pt_idx || !uv_out || !keep)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_pairs: bad arguments");
    if (!n_pairs) return C2B_OK;
    for (int64_t i = 0; i < n_pairs; ++i)
        if (cam_idx[i] >= (uint64_t)p->n_cam || pt_idx[i] >= (uint64_t)p->n_pts)
            return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "problem_visibility_pairs: pair %lld out of range", (long long)i);
    int rc = ensure_camblk(p);
    if (rc) return rc;
    uint32_t *d_c = nullptr, *d_p = nullptr;
    double *d_uv = nullptr;
    uint8_t *d_k = nullptr;
    hipError_t e = hipMalloc((void **)&d_c, sizeof(uint32_t) * n_pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_p, sizeof(uint32_t) * n_pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_uv, sizeof(double) * 2 * n_pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_k, n_pairs);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, cam_idx, sizeof(uint32_t) * n_pairs, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_p, pt_idx, sizeof(uint32_t) * n_pairs, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        rc = c2b_visibility_pairs(p->camblk, p->pts4, d_c, d_p, n_pairs, max_dist, d_uv, d_k, p->stream);
        if (!rc) {
            e = hipMemcpyAsync(uv_out, d_uv, sizeof(double) * 2 * n_pairs, hipMemcpyDeviceToHost, p->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(keep, d_k, n_pairs, hipMemcpyDeviceToHost, p->stream);
        }
        hipError_t e2 = hipStreamSynchronize(p->stream);
        if (e == hipSuccess) e = e2;
    }
    if (d_c) (void)hipFree(d_c);
    if (d_p) (void)hipFree(d_p);
    if (d_uv) (void)hipFree(d_uv);
    if (d_k) (void)hipFree(d_k);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_pairs: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_pairs")
}

/* ---- BAProblem::cull on the device (src/baproblem.rs:538-549) ---- */
extern "C++" {
namespace {

// device allocation that frees itself (the cull pipeline holds ~20 scratch arrays)
struct DevBuf {
    void *ptr = nullptr;
    bool owned = true;                         // false: a view into an arena (below), never freed or released by itself
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (ptr && owned) (void)hipFree(ptr); }
    hipError_t alloc(size_t bytes) { owned = true; return hipMalloc(&ptr, bytes ? bytes : 16); }
    void view(void *q) { ptr = q; owned = false; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(ptr); }
    void *release() { void *q = ptr; ptr = nullptr; return q; }
};

// Temporaries of one call carved out of ONE allocation: a device malloc / free pair costs ~1 ms at these sizes (the free
// synchronises), and cull used to make ~25 of each -- most of its 45 ms at --blocks 128 once its kernels took 10.
struct DevArena {
    DevBuf block;
    size_t used = 0, cap = 0;
    static size_t rounded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    hipError_t reserve(size_t bytes) { cap = bytes; return block.alloc(bytes); }
    void *take(size_t bytes) {
        void *q = static_cast<char *>(block.ptr) + used;
        used += rounded(bytes ? bytes : 16);
        return used <= cap ? q : nullptr;
    }
};

unsigned blocks_of(int64_t n, int per) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }

// exclusive scan of n 0/1 flags into pos; *total_host = number of set flags.  Synchronises.
hipError_t scan_flags(hipStream_t st, const uint32_t *flags, int64_t n, uint32_t *pos, uint32_t *tile_scratch, uint32_t *d_total,
                      uint32_t *total_host) {
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (n > 0) hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, flags, n, pos, tile_scratch);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kScanBlock), 0, st, tile_scratch, tiles, d_total);
    if (n > 0) hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, pos, n, (const uint32_t *)tile_scratch);
    hipError_t e = launch_error();
    if (e == hipSuccess) e = hipMemcpyAsync(total_host, d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

}  // namespace
}  // extern "C++"

// mode 0: cull() = both passes to a fixed point; 1: largest_connected_component() once; 2: remove_singletons() once
static int cull_impl(c2b_problem *p, int faithful, int mode) {
    NEED_UPLOADED(p, "problem_cull");
    if (p->n_obs >= ((int64_t)1 << 32) || p->n_cam + p->n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_cull: more than 2^32 observations or entities");
    free_dense(p);
    hipStream_t st = p->stream;
    const int64_t nc0 = p->n_cam, np0 = p->n_pts, no0 = p->n_obs;
    const int64_t nodes0 = nc0 + np0, big0 = std::max(std::max(nc0, np0), no0);
    // current graph (ping-pong pairs) + where everything came from
    DevBuf cam[2], pt[2], eorig[2], corig[2], porig[2];
    DevBuf parent, sets, size, keep_c, keep_p, keep_o, pos_c, pos_p, pos_o, tiles, best, total, deg, cnt;
    hipError_t e = hipSuccess;
    auto A = [&](DevBuf &b, size_t bytes) { if (e == hipSuccess) e = b.alloc(bytes); };
    // cam / pt are allocations of their own (one of each pair becomes the problem's index array); every other temporary
    // is a view into one arena
    for (int k = 0; k < 2; ++k) { A(cam[k], 4 * (size_t)no0); A(pt[k], 4 * (size_t)no0); }
    DevArena arena;
    struct Want { DevBuf *b; size_t bytes; };
    const Want wants[] = {
        {&eorig[0], 4 * (size_t)no0}, {&eorig[1], 4 * (size_t)no0}, {&corig[0], 4 * (size_t)nc0}, {&corig[1], 4 * (size_t)nc0},
        {&porig[0], 4 * (size_t)np0}, {&porig[1], 4 * (size_t)np0},
        {&parent, 4 * (size_t)nodes0}, {&sets, 4 * (size_t)nodes0}, {&size, 4 * (size_t)nodes0},
        {&keep_c, 4 * (size_t)nc0}, {&keep_p, 4 * (size_t)np0}, {&keep_o, 4 * (size_t)no0},
        {&pos_c, 4 * (size_t)nc0}, {&pos_p, 4 * (size_t)np0}, {&pos_o, 4 * (size_t)no0},
        {&tiles, 4 * (size_t)(big0 / kScanTile + 2)}, {&best, 8}, {&total, 4}, {&deg, 4 * (size_t)nc0}, {&cnt, 4 * (size_t)np0}};
    size_t arena_bytes = 0;
    for (const Want &w : wants) arena_bytes += DevArena::rounded(w.bytes ? w.bytes : 16);
    if (e == hipSuccess) e = arena.reserve(arena_bytes);
    if (e == hipSuccess)
        for (const Want &w : wants) w.b->view(arena.take(w.bytes));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_cull: %s", hipGetErrorString(e));

    int cur = 0;
    int64_t nc = nc0, np = np0, no = no0;
    if (no) {
        e = hipMemcpyAsync(cam[0].ptr, p->cam_idx, 4 * (size_t)no, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(pt[0].ptr, p->pt_idx, 4 * (size_t)no, hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, eorig[0].as<uint32_t>(), no);   // iota
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(nc, kBlock)), dim3(kBlock), 0, st, corig[0].as<uint32_t>(), nc);
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(np, kBlock)), dim3(kBlock), 0, st, porig[0].as<uint32_t>(), np);
        e = launch_error();
    }

    // renumber by the keep flags currently in keep_c / keep_p / keep_o
    auto compact = [&]() -> hipError_t {
        uint32_t nc_new = 0, np_new = 0, no_new = 0;
        hipError_t s = scan_flags(st, keep_c.as<uint32_t>(), nc, pos_c.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &nc_new);
        if (s == hipSuccess) s = scan_flags(st, keep_p.as<uint32_t>(), np, pos_p.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &np_new);
        if (s == hipSuccess) s = scan_flags(st, keep_o.as<uint32_t>(), no, pos_o.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &no_new);
        if (s != hipSuccess) return s;
        const int nxt = cur ^ 1;
        hipLaunchKernelGGL(k_cull_move_nodes, dim3(blocks_of(nc, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)keep_c.as<uint32_t>(),
                           (const uint32_t *)pos_c.as<uint32_t>(), nc, (const uint32_t *)corig[cur].as<uint32_t>(), corig[nxt].as<uint32_t>());
        hipLaunchKernelGGL(k_cull_move_nodes, dim3(blocks_of(np, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)keep_p.as<uint32_t>(),
                           (const uint32_t *)pos_p.as<uint32_t>(), np, (const uint32_t *)porig[cur].as<uint32_t>(), porig[nxt].as<uint32_t>());
        hipLaunchKernelGGL(k_cull_move_edges, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)keep_o.as<uint32_t>(),
                           (const uint32_t *)pos_o.as<uint32_t>(), no, (const uint32_t *)cam[cur].as<uint32_t>(),
                           (const uint32_t *)pt[cur].as<uint32_t>(), (const uint32_t *)eorig[cur].as<uint32_t>(),
                           (const uint32_t *)pos_c.as<uint32_t>(), (const uint32_t *)pos_p.as<uint32_t>(), cam[nxt].as<uint32_t>(),
                           pt[nxt].as<uint32_t>(), eorig[nxt].as<uint32_t>());
        cur = nxt;
        nc = nc_new; np = np_new; no = no_new;
        return launch_error();
    };
    auto lcc_pass = [&]() -> hipError_t {
        if (nc == 0) return hipSuccess;                      // largest_connected_component returns self (:457-459)
        const int64_t nodes = nc + np, big = std::max(std::max(nc, np), no);
        hipError_t s = hipMemsetAsync(size.ptr, 0, 4 * (size_t)nodes, st);
        if (s == hipSuccess) s = hipMemsetAsync(best.ptr, 0, 8, st);
        if (s != hipSuccess) return s;
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(nodes, kBlock)), dim3(kBlock), 0, st, parent.as<uint32_t>(), nodes);
        if (no) hipLaunchKernelGGL(k_uf_union, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, parent.as<uint32_t>(),
                                   (const uint32_t *)cam[cur].as<uint32_t>(), (const uint32_t *)pt[cur].as<uint32_t>(), no, (uint32_t)nc);
        hipLaunchKernelGGL(k_uf_flatten, dim3(blocks_of(nodes, kBlock)), dim3(kBlock), 0, st, parent.as<uint32_t>(), nodes,
                           sets.as<uint32_t>(), size.as<uint32_t>());
        hipLaunchKernelGGL(k_uf_largest, dim3(blocks_of(nodes, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)sets.as<uint32_t>(),
                           (const uint32_t *)size.as<uint32_t>(), nodes, best.as<unsigned long long>());
        hipLaunchKernelGGL(k_lcc_flags, dim3(blocks_of(big, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)sets.as<uint32_t>(),
                           (const unsigned long long *)best.as<unsigned long long>(), (uint32_t)nc, (uint32_t)np,
                           (const uint32_t *)cam[cur].as<uint32_t>(), (const uint32_t *)pt[cur].as<uint32_t>(), no, faithful ? 1 : 0,
                           keep_c.as<uint32_t>(), keep_p.as<uint32_t>(), keep_o.as<uint32_t>());
        s = launch_error();
        return s == hipSuccess ? compact() : s;
    };
    auto singleton_pass = [&]() -> hipError_t {
        const int64_t big = std::max(std::max(nc, np), no);
        hipError_t s = hipMemsetAsync(deg.ptr, 0, 4 * (size_t)(nc ? nc : 1), st);
        if (s == hipSuccess) s = hipMemsetAsync(cnt.ptr, 0, 4 * (size_t)(np ? np : 1), st);
        if (s != hipSuccess) return s;
        if (no) hipLaunchKernelGGL(k_degree, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)cam[cur].as<uint32_t>(),
                                   (const uint32_t *)pt[cur].as<uint32_t>(), no, deg.as<uint32_t>(), cnt.as<uint32_t>());
        hipLaunchKernelGGL(k_singleton_flags, dim3(blocks_of(big, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)deg.as<uint32_t>(),
                           (const uint32_t *)cnt.as<uint32_t>(), (uint32_t)nc, (uint32_t)np, (const uint32_t *)cam[cur].as<uint32_t>(),
                           (const uint32_t *)pt[cur].as<uint32_t>(), no, keep_c.as<uint32_t>(), keep_p.as<uint32_t>(), keep_o.as<uint32_t>());
        s = launch_error();
        return s == hipSuccess ? compact() : s;
    };
    // culled = lcc().remove_singletons(); while the counts change: again (src/baproblem.rs:541-547)
    int64_t pnc = nc, pnp = np;
    if (e == hipSuccess && mode != 2) e = lcc_pass();
    if (e == hipSuccess && mode != 1) e = singleton_pass();
    while (mode == 0 && e == hipSuccess && (nc != pnc || np != pnp)) {
        pnc = nc; pnp = np;
        e = lcc_pass();
        if (e == hipSuccess) e = singleton_pass();
    }

    // gather the payloads once and swap them in
    DevBuf n_cam15, n_bal9, n_camblk, n_cen4, n_pts4, n_uv, n_ws;
    if (e == hipSuccess) {
        A(n_cam15, sizeof(double) * 15 * (size_t)nc); A(n_bal9, sizeof(double) * 9 * (size_t)nc);
        A(n_camblk, sizeof(double) * (size_t)cam_table_doubles(nc)); A(n_cen4, sizeof(double) * 4 * (size_t)nc); A(n_pts4, sizeof(double) * 4 * (size_t)np);
        A(n_uv, sizeof(double) * 2 * (size_t)no); A(n_ws, (size_t)c2b_workspace_bytes(no));
    }
    if (e == hipSuccess && c2b_workspace_init(n_ws.ptr, st) != C2B_OK) e = hipErrorUnknown;
    if (e == hipSuccess) {
        auto gather = [&](const double *in, const DevBuf &orig, int64_t n, int width, DevBuf &out) {
            if (n) hipLaunchKernelGGL(k_gather_rows, dim3(blocks_of(n * width, kBlock)), dim3(kBlock), 0, st, in,
                                      (const uint32_t *)orig.as<uint32_t>(), n, width, out.as<double>());
        };
        gather(p->cam15, corig[cur], nc, 15, n_cam15);
        if (p->bal_valid) gather(p->bal9, corig[cur], nc, 9, n_bal9);
        gather(p->pts4, porig[cur], np, 4, n_pts4);
        gather(p->uv, eorig[cur], no, 2, n_uv);
        e = launch_error();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_cull: %s", hipGetErrorString(e));
    }
    void *old[] = {p->cam15, p->bal9, p->camblk, p->cen4, p->pts4, p->uv, p->cam_idx, p->pt_idx, p->ws};
    for (void *q : old) if (q) (void)hipFree(q);
    p->cam15 = (double *)n_cam15.release(); p->bal9 = (double *)n_bal9.release(); p->camblk = (double *)n_camblk.release();
    p->cen4 = (double *)n_cen4.release();
    p->pts4 = (double *)n_pts4.release(); p->uv = (double *)n_uv.release(); p->ws = n_ws.release();
    p->cam_idx = (uint32_t *)cam[cur].release(); p->pt_idx = (uint32_t *)pt[cur].release();
    drop_rows(p);
    p->n_cam = nc; p->n_pts = np; p->n_obs = no;
    p->blk_valid = false;                                     // camblk is rebuilt on demand from the gathered cameras
    p->bal9_fresh = false;                                    // (bal9 was gathered only when it was the truth)
    return C2B_OK;
}

int c2b_problem_cull(c2b_problem *p, int faithful) { return cull_impl(p, faithful, 0); }
int c2b_problem_largest_connected_component(c2b_problem *p, int faithful) { return cull_impl(p, faithful, 1); }
int c2b_problem_remove_singletons(c2b_problem *p) { return cull_impl(p, 1, 2); }

int c2b_problem_adopt_visibility(c2b_problem *p) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_adopt_visibility");
    if (!p->dense_pt || !p->dense_uv || !p->dense_row)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_adopt_visibility: no pending visibility result");
    const int64_t n = p->dense_n;
    if (n >= ((int64_t)1 << 32)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_adopt_visibility: more than 2^32 observations");
    DevBuf cam_idx, ws;
    hipError_t e = cam_idx.alloc(sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = ws.alloc((size_t)c2b_workspace_bytes(n));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_adopt_visibility: %s", hipGetErrorString(e));
    int rc = c2b_expand_rows(p->dense_row, p->n_cam, 0, n, cam_idx.as<uint32_t>(), p->stream);
    if (!rc) rc = c2b_workspace_init(ws.ptr, p->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    void *old[] = {p->uv, p->cam_idx, p->pt_idx, p->ws, p->dense_row};
    for (void *q : old) if (q) (void)hipFree(q);
    p->cam_idx = (uint32_t *)cam_idx.release();
    drop_rows(p);
    p->ws = ws.release();
    p->pt_idx = p->dense_pt; p->uv = p->dense_uv; p->n_obs = n;
    p->dense_pt = nullptr; p->dense_uv = nullptr; p->dense_row = nullptr; p->dense_n = 0;
    return C2B_OK;
    C2B_API_END("problem_adopt_visibility")
}

int c2b_problem_download_graph(c2b_problem *p, uint64_t *row_ptr, uint64_t *pt_idx) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_download_graph");
    if (!row_ptr || (p->n_obs && !pt_idx)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_download_graph: bad arguments");
    const int64_t n_cam = p->n_cam, n_obs = p->n_obs;
    // the point indices are widened to the host's u64 on the device and leave in ONE copy (r01-r02: a u32 copy into a
    // fresh host vector, then a serial widening loop over 19 M entries -- a third of the 120-ms download at --blocks 128)
    DevBuf d_row, d_pt64;
    hipError_t e = d_row.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess && n_obs) e = d_pt64.alloc(sizeof(uint64_t) * (size_t)n_obs);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rows_from_sorted, dim3(blocks_for(n_obs + 1)), dim3(kBlock), 0, p->stream, (const uint32_t *)p->cam_idx, n_obs,
                           n_cam, d_row.as<uint64_t>());
        if (n_obs) hipLaunchKernelGGL(k_widen_u32, dim3(blocks_for(n_obs)), dim3(kBlock), 0, p->stream, (const uint32_t *)p->pt_idx, n_obs,
                                      d_pt64.as<uint64_t>());
        e = launch_error();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(row_ptr, d_row.ptr, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && n_obs) e = hipMemcpyAsync(pt_idx, d_pt64.ptr, sizeof(uint64_t) * (size_t)n_obs, hipMemcpyDeviceToHost, p->stream);
    hipError_t e2 = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_download_graph: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_download_graph")
}

// The bridge from a problem that was BORN on the device (layout, visibility loop, cull, file read) to the Level-0
// launchers: the resident arrays -- or the slice that belongs to the camera range [cam_lo, cam_hi) -- copied device to
// device into buffers the caller owns.  No byte crosses PCIe but the two row-pointer words that size the slice.
__global__ __launch_bounds__(256) void k_rows_rebase(const uint64_t *__restrict__ in, int64_t n, uint64_t base, uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] - base;
}

int c2b_problem_export_device(c2b_problem *p, int64_t cam_lo, int64_t cam_hi, double *cam15, double *pts4, uint64_t *row_ptr,
                              uint32_t *pt_idx, double *uv, int64_t *obs_lo, int64_t *n_obs_slice) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_export_device");
    if (cam_lo < 0 || cam_hi < cam_lo || cam_hi > p->n_cam)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_export_device: camera range [%lld, %lld) outside [0, %lld]", (long long)cam_lo,
                    (long long)cam_hi, (long long)p->n_cam);
    int rc = ensure_rows(p);
    if (rc) return rc;
    hipStream_t st = p->stream;
    uint64_t ends[2] = {0, 0};
    if (p->n_obs) {
        HIP_TRY(hipMemcpyAsync(&ends[0], p->rows_ptr + cam_lo, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&ends[1], p->rows_ptr + cam_hi, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const int64_t o0 = (int64_t)ends[0], n = (int64_t)(ends[1] - ends[0]), nc = cam_hi - cam_lo;
    if (obs_lo) *obs_lo = o0;
    if (n_obs_slice) *n_obs_slice = n;
    if (cam15 && nc) HIP_TRY(hipMemcpyAsync(cam15, p->cam15 + 15 * cam_lo, sizeof(double) * 15 * (size_t)nc, hipMemcpyDeviceToDevice, st));
    if (pts4 && p->n_pts) HIP_TRY(hipMemcpyAsync(pts4, p->pts4, sizeof(double) * 4 * (size_t)p->n_pts, hipMemcpyDeviceToDevice, st));
    if (row_ptr) {
        if (p->n_obs) {
            hipLaunchKernelGGL(k_rows_rebase, dim3(blocks_of(nc + 1, 256)), dim3(256), 0, st, (const uint64_t *)(p->rows_ptr + cam_lo), nc + 1,
                               ends[0], row_ptr);
            LAUNCH_CHECK();
        } else {
            HIP_TRY(hipMemsetAsync(row_ptr, 0, sizeof(uint64_t) * (size_t)(nc + 1), st));
        }
    }
    if (pt_idx && n) HIP_TRY(hipMemcpyAsync(pt_idx, p->pt_idx + o0, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    if (uv && n) HIP_TRY(hipMemcpyAsync(uv, p->uv + 2 * o0, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return C2B_OK;
    C2B_API_END("problem_export_device")
}

// Stable compaction of CSR lists (point index, uv) by a keep mask, on the device: kept count per row -> row scan ->
// one wave per camera scatters in order.  The new row pointer goes to row_ptr_host; on success the three new device
// buffers are handed to the caller (who owns them), *w = kept count.  Synchronises the problem's stream.
static hipError_t compact_rows_on_device(c2b_problem *p, const uint64_t *d_row_old, const uint8_t *d_keep, const uint32_t *d_pt,
                                         const double *d_uv, int64_t n_cam, uint64_t *row_ptr_host, uint64_t **d_row_new,
                                         uint32_t **d_pt_new, double **d_uv_new, int64_t *w) {
    uint64_t *d_tot = nullptr;
    *d_row_new = nullptr; *d_pt_new = nullptr; *d_uv_new = nullptr; *w = 0;
    hipError_t e = hipMalloc((void **)&d_tot, sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess) e = hipMalloc((void **)d_row_new, sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess) {
        const unsigned row_blocks = (unsigned)((n_cam + 3) / 4);
        if (n_cam) hipLaunchKernelGGL(k_keep_row_counts, dim3(row_blocks), dim3(256), 0, p->stream, d_row_old, d_keep, n_cam, d_tot);
        hipLaunchKernelGGL(k_dense_cam_scan, dim3(1), dim3(256), 0, p->stream, (const uint64_t *)d_tot, n_cam, *d_row_new);
        e = hipMemcpyAsync(row_ptr_host, *d_row_new, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (e == hipSuccess) {
            *w = (int64_t)row_ptr_host[n_cam];
            e = hipMalloc((void **)d_pt_new, sizeof(uint32_t) * (size_t)(*w ? *w : 4));
            if (e == hipSuccess) e = hipMalloc((void **)d_uv_new, sizeof(double) * 2 * (size_t)(*w ? *w : 1));
            if (e == hipSuccess && n_cam) {
                hipLaunchKernelGGL(k_keep_row_scatter, dim3(row_blocks), dim3(256), 0, p->stream, d_row_old,
                                   (const uint64_t *)*d_row_new, d_keep, d_pt, reinterpret_cast<const double2 *>(d_uv), n_cam,
                                   *d_pt_new, reinterpret_cast<double2 *>(*d_uv_new));
                e = launch_error();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        }
    }
    if (d_tot) (void)hipFree(d_tot);
    if (e != hipSuccess) {
        if (*d_row_new) (void)hipFree(*d_row_new);
        if (*d_pt_new) (void)hipFree(*d_pt_new);
        if (*d_uv_new) (void)hipFree(*d_uv_new);
        *d_row_new = nullptr; *d_pt_new = nullptr; *d_uv_new = nullptr;
    }
    return e;
}

int c2b_problem_visibility_pairs_compact(c2b_problem *p, int64_t n_pairs, const uint32_t *cam_idx, const uint32_t *pt_idx,
                                         double max_dist, uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_pairs_compact");
    if (n_pairs < 0 || !row_ptr || (n_pairs && (!cam_idx || !pt_idx)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_pairs_compact: bad arguments");
    for (int64_t i = 0; i < n_pairs; ++i) {
        if (cam_idx[i] >= (uint64_t)p->n_cam || pt_idx[i] >= (uint64_t)p->n_pts)
            return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "problem_visibility_pairs_compact: pair %lld out of range", (long long)i);
        if (i && cam_idx[i] < cam_idx[i - 1])
            return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_pairs_compact: cam_idx must be non-decreasing (pair %lld)", (long long)i);
    }
    int rc = ensure_camblk(p);
    if (rc) return rc;
    free_dense(p);
    const int64_t n_cam = p->n_cam;
    const size_t n = (size_t)(n_pairs ? n_pairs : 1);
    uint32_t *d_c = nullptr, *d_p = nullptr, *d_pt_new = nullptr;
    double *d_uv = nullptr, *d_uv_new = nullptr;
    uint8_t *d_k = nullptr;
    uint64_t *d_row = nullptr, *d_row_new = nullptr;
    int64_t w = 0;
    hipError_t e = hipMalloc((void **)&d_c, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_p, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_uv, sizeof(double) * 2 * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_k, n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_row, sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess && n_pairs) e = hipMemcpyAsync(d_c, cam_idx, sizeof(uint32_t) * n, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess && n_pairs) e = hipMemcpyAsync(d_p, pt_idx, sizeof(uint32_t) * n, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        if (n_pairs) rc = c2b_visibility_pairs(p->camblk, p->pts4, d_c, d_p, n_pairs, max_dist, d_uv, d_k, p->stream);
        if (!rc) {
            hipLaunchKernelGGL(k_rows_from_sorted, dim3(blocks_for(n_pairs + 1)), dim3(kBlock), 0, p->stream, (const uint32_t *)d_c,
                               n_pairs, n_cam, d_row);
            e = launch_error();
            if (e == hipSuccess)
                e = compact_rows_on_device(p, d_row, d_k, d_p, d_uv, n_cam, row_ptr, &d_row_new, &d_pt_new, &d_uv_new, &w);
        }
    }
    if (rc || e != hipSuccess) (void)hipStreamSynchronize(p->stream);
    if (!rc && e == hipSuccess) {                 // becomes the pending visibility result (fetch with _dense_fetch)
        p->dense_row = d_row_new; p->dense_pt = d_pt_new; p->dense_uv = d_uv_new; p->dense_n = w;
    }
    if (d_c) (void)hipFree(d_c);
    if (d_p) (void)hipFree(d_p);
    if (d_uv) (void)hipFree(d_uv);
    if (d_k) (void)hipFree(d_k);
    if (d_row) (void)hipFree(d_row);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_pairs_compact: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_pairs_compact")
}

// The generators' whole visibility loop (src/synthetic.rs:268-297, :353-378) on the device: candidates by a cell list
// (rstar's locate_within_distance), the sight line against the buildings (hits_building), the predicate, and the kept
// (point, uv) lists compacted per camera in ascending point index -- csrc/cell_kernels.hpp.  The result becomes the
// pending visibility result like c2b_problem_visibility_pairs_compact's (adopt / fetch it the same way); row_ptr (host,
// n_cam + 1) may be NULL.
int c2b_problem_visibility_within_distance(c2b_problem *p, double max_dist, int occlusion, double block_length, double block_inset,
                                           uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_within_distance");
    if (!(max_dist >= 0.0) || (occlusion && !(block_length > 0.0)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: max_dist must be >= 0 (and block_length > 0 with occlusion)");
    const int64_t n_cam = p->n_cam, n_pts = p->n_pts;
    if (n_pts >= ((int64_t)1 << 32) || n_cam >= ((int64_t)1 << 31))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: too many cameras or points");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    free_dense(p);
    hipStream_t st = p->stream;
    if (n_cam == 0 || n_pts == 0) {                                      // nothing can be seen: an empty graph
        DevBuf row0, pt0, uv0;
        hipError_t e0 = row0.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
        if (e0 == hipSuccess) e0 = pt0.alloc(4);
        if (e0 == hipSuccess) e0 = uv0.alloc(16);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(row0.ptr, 0, sizeof(uint64_t) * (size_t)(n_cam + 1), st);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) return fail(C2B_ERR_HIP, "problem_visibility_within_distance: %s", hipGetErrorString(e0));
        if (row_ptr) std::fill(row_ptr, row_ptr + n_cam + 1, (uint64_t)0);
        p->dense_row = (uint64_t *)row0.release(); p->dense_pt = (uint32_t *)pt0.release(); p->dense_uv = (double *)uv0.release();
        p->dense_n = 0;
        return C2B_OK;
    }
    // extent of cameras and points -> the cell grid.  Cells are a hair wider than max_dist so that rounding in the cell
    // arithmetic can never separate a camera from a point within max_dist by more than one cell.
    double stats[C2B_STATS_DOUBLES];
    rc = compute_stats(p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof stats, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    CellGrid g;
    g.x0 = stats[6]; g.z0 = stats[8];
    const double ex = stats[9] - stats[6], ez = stats[11] - stats[8];
    double cs = (max_dist > 0.0 ? max_dist : 1.0) * (1.0 + 0x1.0p-20);
    if (!(ex >= 0.0) || !(ez >= 0.0) || !std::isfinite(ex) || !std::isfinite(ez) || !std::isfinite(cs))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: non-finite coordinates");
    auto cells = [&](double c) { return (std::floor(ex / c) + 1.0) * (std::floor(ez / c) + 1.0); };
    while (cells(cs) > (double)(1 << 24)) cs *= 2.0;                     // wider cells stay correct, only slower
    g.inv_cs = 1.0 / cs;
    g.ncx = (int)std::floor(ex / cs) + 1; g.ncz = (int)std::floor(ez / cs) + 1;
    const int64_t n_cells = (int64_t)g.ncx * g.ncz;

    DevArena arena;
    DevBuf cell_of, counts, cursor, sorted, tiles, total, cam_count, pos, sum64, max32, row64;
    struct Want { DevBuf *b; size_t bytes; };
    const int64_t big = std::max(n_cells + 1, n_cam + 1);
    const Want wants[] = {{&cell_of, 4 * (size_t)n_pts}, {&counts, 4 * (size_t)(n_cells + 1)}, {&cursor, 4 * (size_t)(n_cells + 1)},
                          {&sorted, 4 * (size_t)n_pts}, {&tiles, 4 * (size_t)(big / kScanTile + 2)}, {&total, 4},
                          {&cam_count, 4 * (size_t)(n_cam + 1)}, {&pos, 4 * (size_t)(n_cam + 1)}, {&sum64, 8}, {&max32, 4}};
    size_t arena_bytes = 0;
    for (const Want &w : wants) arena_bytes += DevArena::rounded(w.bytes ? w.bytes : 16);
    hipError_t e = arena.reserve(arena_bytes);
    if (e == hipSuccess) e = row64.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_within_distance: %s", hipGetErrorString(e));
    for (const Want &w : wants) w.b->view(arena.take(w.bytes));

    // cell list: count, exclusive scan (start[n_cells] = n_pts), fill
    HIP_TRY(hipMemsetAsync(counts.ptr, 0, 4 * (size_t)(n_cells + 1), st));
    HIP_TRY(hipMemsetAsync(cursor.ptr, 0, 4 * (size_t)(n_cells + 1), st));
    HIP_TRY(hipMemsetAsync(cam_count.ptr, 0, 4 * (size_t)(n_cam + 1), st));
    HIP_TRY(hipMemsetAsync(sum64.ptr, 0, 8, st));
    HIP_TRY(hipMemsetAsync(max32.ptr, 0, 4, st));
    if (n_pts) hipLaunchKernelGGL(k_cells_assign, dim3(blocks_of(n_pts, 256)), dim3(256), 0, st, reinterpret_cast<const double4 *>(p->pts4), n_pts,
                                  g, cell_of.as<uint32_t>(), counts.as<uint32_t>());
    uint32_t n_sorted = 0, n_kept32 = 0;
    uint32_t *start = cursor.as<uint32_t>();                             // scanned counts; the fill's cursors live in `counts` afterwards
    e = scan_flags(st, counts.as<uint32_t>(), n_cells + 1, start, tiles.as<uint32_t>(), total.as<uint32_t>(), &n_sorted);
    if (e == hipSuccess && (int64_t)n_sorted != n_pts) return fail(C2B_ERR_HIP, "problem_visibility_within_distance: cell counts do not add up");
    if (e == hipSuccess) e = hipMemsetAsync(counts.ptr, 0, 4 * (size_t)(n_cells + 1), st);
    if (e == hipSuccess && n_pts)
        hipLaunchKernelGGL(k_cells_fill, dim3(blocks_of(n_pts, 256)), dim3(256), 0, st, (const uint32_t *)cell_of.as<uint32_t>(), n_pts,
                           (const uint32_t *)start, counts.as<uint32_t>(), sorted.as<uint32_t>());
    // pass 1: survivors per camera; scan; total
    const unsigned cam_blocks = blocks_of(n_cam, kCellWPB);
    if (e == hipSuccess && n_cam && n_pts) {
        hipLaunchKernelGGL((k_cells_visibility<false>), dim3(cam_blocks), dim3(kCellWPB * 64), 0, st, (const double *)p->camblk, n_cam,
                           reinterpret_cast<const double4 *>(p->pts4), g, (const uint32_t *)start, (const uint32_t *)sorted.as<uint32_t>(),
                           max_dist, occlusion ? 1 : 0, block_length, block_inset, cam_count.as<uint32_t>(), (const uint64_t *)nullptr,
                           (uint32_t *)nullptr, (double2 *)nullptr);
        hipLaunchKernelGGL(k_sum_u32_u64, dim3(256), dim3(256), 0, st, (const uint32_t *)cam_count.as<uint32_t>(), n_cam,
                           sum64.as<unsigned long long>());
        hipLaunchKernelGGL(k_max_u32, dim3(256), dim3(256), 0, st, (const uint32_t *)cam_count.as<uint32_t>(), n_cam, max32.as<uint32_t>());
    }
    if (e == hipSuccess) e = scan_flags(st, cam_count.as<uint32_t>(), n_cam + 1, pos.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &n_kept32);
    unsigned long long n_kept = 0;
    uint32_t longest = 0;
    if (e == hipSuccess) e = hipMemcpy(&n_kept, sum64.ptr, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&longest, max32.ptr, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_kept != (unsigned long long)n_kept32)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: more than 2^32 observations");
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_widen_u32, dim3(blocks_for(n_cam + 1)), dim3(kBlock), 0, st, (const uint32_t *)pos.as<uint32_t>(), n_cam + 1,
                           row64.as<uint64_t>());
        e = launch_error();
    }
    // pass 2: fill in meeting order, then every row into ascending point index
    DevBuf tmp_pt, tmp_uv, out_pt, out_uv;
    const size_t w = (size_t)n_kept;
    if (e == hipSuccess) e = tmp_pt.alloc(4 * w);
    if (e == hipSuccess) e = tmp_uv.alloc(16 * w);
    if (e == hipSuccess) e = out_pt.alloc(4 * w);
    if (e == hipSuccess) e = out_uv.alloc(16 * w);
    if (e == hipSuccess && w) {
        hipLaunchKernelGGL((k_cells_visibility<true>), dim3(cam_blocks), dim3(kCellWPB * 64), 0, st, (const double *)p->camblk, n_cam,
                           reinterpret_cast<const double4 *>(p->pts4), g, (const uint32_t *)start, (const uint32_t *)sorted.as<uint32_t>(),
                           max_dist, occlusion ? 1 : 0, block_length, block_inset, (uint32_t *)nullptr, (const uint64_t *)row64.as<uint64_t>(),
                           tmp_pt.as<uint32_t>(), tmp_uv.as<double2>());
        // The device row sort is quadratic in the row length: right for the generators' rows (a few dozen entries), wrong
        // for a radius that makes one camera see tens of thousands of points.  Beyond `long_row` entries the rows are sorted on the
        // host instead (threads over cameras) -- a path for odd inputs, not a fast one; such problems belong to the dense
        // sweep (c2b_problem_visibility_dense).
        // (2 048: a lane of the rank sort then makes at most 64 k dependent compares -- ~0.1 ms per row-wave; at the 8 192 of
        // round 4 a scene with thousands of cameras seeing several thousand points each was a multi-second cliff)
        const uint32_t long_row = p->opt.rank_sort_max_row > 0 ? (uint32_t)p->opt.rank_sort_max_row : 2048u;
        if (longest <= long_row) {
            hipLaunchKernelGGL(k_rows_rank_sort, dim3(cam_blocks), dim3(kCellWPB * 64), 0, st, (const uint64_t *)row64.as<uint64_t>(), n_cam,
                               (const uint32_t *)tmp_pt.as<uint32_t>(), (const double2 *)tmp_uv.as<double2>(), out_pt.as<uint32_t>(),
                               out_uv.as<double2>());
            e = launch_error();
        } else {
            e = launch_error();
            std::vector<uint64_t> rp((size_t)n_cam + 1);
            std::vector<uint32_t> hp(w), hq(w);
            std::vector<double> hu(2 * w), hv(2 * w);
            if (e == hipSuccess) e = hipMemcpyAsync(rp.data(), row64.ptr, 8 * ((size_t)n_cam + 1), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(hp.data(), tmp_pt.ptr, 4 * w, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(hu.data(), tmp_uv.ptr, 16 * w, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) {
                const int T = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), std::max<int64_t>(1, n_cam));
                c2b_host::run_threads(T, [&](int t) {
                        std::vector<uint32_t> order;
                        for (int64_t c = n_cam * t / T; c < n_cam * (t + 1) / T; ++c) {
                            const size_t b = (size_t)rp[(size_t)c], k = (size_t)(rp[(size_t)c + 1] - rp[(size_t)c]);
                            order.resize(k);
                            for (size_t i = 0; i < k; ++i) order[i] = (uint32_t)i;
                            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return hp[b + x] < hp[b + y]; });
                            for (size_t i = 0; i < k; ++i) {
                                hq[b + i] = hp[b + order[i]];
                                hv[2 * (b + i)] = hu[2 * (b + order[i])];
                                hv[2 * (b + i) + 1] = hu[2 * (b + order[i]) + 1];
                            }
                        }
                    });
                e = hipMemcpyAsync(out_pt.ptr, hq.data(), 4 * w, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(out_uv.ptr, hv.data(), 16 * w, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
            }
        }
    }
    if (e == hipSuccess && row_ptr)
        e = hipMemcpyAsync(row_ptr, row64.ptr, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_within_distance: %s", hipGetErrorString(e));
    }
    p->dense_row = (uint64_t *)row64.release();
    p->dense_pt = (uint32_t *)out_pt.release();
    p->dense_uv = (double *)out_uv.release();
    p->dense_n = (int64_t)n_kept;
    return C2B_OK;
    C2B_API_END("problem_visibility_within_distance")
}

// generate_world_points_uniform (src/generate.rs:356-420) for the cameras of the resident problem, on the device
// (cell_kernels.hpp: k_world_*): the problem's points are REPLACED by the sampled ones (its observations must be empty).
// Candidate k draws from the k-th splitmix64 stream of `seed` exactly like c2b_generate_world_points, and candidates are
// accepted in order, so the points are that function's, bit for bit.  tri9: HOST triangles [n_tri][9] f32.
int c2b_problem_generate_world_points(c2b_problem *p, const float *tri9, int64_t n_tri, int64_t num_points, double max_dist,
                                      uint64_t seed, int64_t *n_out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_generate_world_points");
    if (!tri9 || n_tri < 0 || num_points < 0 || !n_out || num_points >= ((int64_t)1 << 31))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_generate_world_points: bad arguments");
    if (p->n_obs != 0) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_generate_world_points: the problem already has observations");
    if (p->n_cam == 0)
        return fail(C2B_ERR_INVALID_ARGUMENT, "Cannot generate world points with 0 cameras. Try increasing the number of cameras generated (via --cameras).");
    if (n_tri == 0) return fail(C2B_ERR_INVALID_ARGUMENT, "the model has no triangles");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    hipStream_t st = p->stream;
    const int64_t n_cam = p->n_cam;
    // triangle areas and their running sum, on the host in the host sampler's own arithmetic (a sequential sum: 1 ms)
    std::vector<double> cum((size_t)n_tri);
    {
        double acc = 0.0;
        for (int64_t t = 0; t < n_tri; ++t) {
            const float *q = tri9 + 9 * t;
            const double a[3] = {(double)q[3] - (double)q[0], (double)q[4] - (double)q[1], (double)q[5] - (double)q[2]};
            const double b[3] = {(double)q[6] - (double)q[0], (double)q[7] - (double)q[1], (double)q[8] - (double)q[2]};
            const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            acc += std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) / 2.0;
            cum[(size_t)t] = acc;
        }
    }
    // the cell list over the camera centres
    DevBuf d_tri, d_cum;
    const double4 *centres = reinterpret_cast<const double4 *>(p->cen4);     // written with camblk (ensure_camblk above)
    hipError_t e = d_tri.alloc(36 * (size_t)n_tri);
    if (e == hipSuccess) e = d_cum.alloc(8 * (size_t)n_tri);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tri.ptr, tri9, 36 * (size_t)n_tri, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cum.ptr, cum.data(), 8 * (size_t)n_tri, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_generate_world_points: %s", hipGetErrorString(e));
    // the centres' extent (a reduction the statistics kernel already is: cameras only)
    double stats[C2B_STATS_DOUBLES];
    rc = c2b_stats(p->camblk, p->cen4, n_cam, p->pts4, 0, p->ws, p->stats, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof stats, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    CellGrid g;
    g.x0 = stats[6]; g.z0 = stats[8];
    const double ex = stats[9] - stats[6], ez = stats[11] - stats[8];
    double cs = (max_dist > 0.0 ? max_dist : 1.0) * (1.0 + 0x1.0p-20);
    if (!(ex >= 0.0) || !(ez >= 0.0) || !std::isfinite(ex) || !std::isfinite(ez) || !std::isfinite(cs))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_generate_world_points: non-finite coordinates");
    auto cells = [&](double c) { return (std::floor(ex / c) + 1.0) * (std::floor(ez / c) + 1.0); };
    while (cells(cs) > (double)(1 << 24)) cs *= 2.0;
    g.inv_cs = 1.0 / cs;
    g.ncx = (int)std::floor(ex / cs) + 1; g.ncz = (int)std::floor(ez / cs) + 1;
    const int64_t n_cells = (int64_t)g.ncx * g.ncz;
    const int64_t chunk = std::max<int64_t>(4096, std::min<int64_t>(num_points, (int64_t)1 << 20));      // the host sampler's chunks
    DevBuf cell_of, counts, startb, sorted, tiles, total, cand, ok, pos, cutoff, out;
    const int64_t big = std::max(n_cells + 1, chunk);
    e = cell_of.alloc(4 * (size_t)n_cam);
    if (e == hipSuccess) e = counts.alloc(4 * (size_t)(n_cells + 1));
    if (e == hipSuccess) e = startb.alloc(4 * (size_t)(n_cells + 1));
    if (e == hipSuccess) e = sorted.alloc(4 * (size_t)n_cam);
    if (e == hipSuccess) e = tiles.alloc(4 * (size_t)(big / kScanTile + 2));
    if (e == hipSuccess) e = total.alloc(4);
    if (e == hipSuccess) e = cand.alloc(32 * (size_t)chunk);
    if (e == hipSuccess) e = ok.alloc(4 * (size_t)chunk);
    if (e == hipSuccess) e = pos.alloc(4 * (size_t)chunk);
    if (e == hipSuccess) e = cutoff.alloc(8);
    if (e == hipSuccess) e = out.alloc(32 * (size_t)std::max<int64_t>(num_points, 1));
    if (e == hipSuccess) e = hipMemsetAsync(counts.ptr, 0, 4 * (size_t)(n_cells + 1), st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_generate_world_points: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_cells_assign, dim3(blocks_of(n_cam, 256)), dim3(256), 0, st, centres, n_cam, g,
                       cell_of.as<uint32_t>(), counts.as<uint32_t>());
    uint32_t n_sorted = 0;
    e = scan_flags(st, counts.as<uint32_t>(), n_cells + 1, startb.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &n_sorted);
    if (e == hipSuccess && (int64_t)n_sorted != n_cam) return fail(C2B_ERR_HIP, "problem_generate_world_points: cell counts do not add up");
    if (e == hipSuccess) e = hipMemsetAsync(counts.ptr, 0, 4 * (size_t)(n_cells + 1), st);
    if (e == hipSuccess)
        hipLaunchKernelGGL(k_cells_fill, dim3(blocks_of(n_cam, 256)), dim3(256), 0, st, (const uint32_t *)cell_of.as<uint32_t>(), n_cam,
                           (const uint32_t *)startb.as<uint32_t>(), counts.as<uint32_t>(), sorted.as<uint32_t>());
    if (e == hipSuccess) e = launch_error();
    // the reference's loop, a chunk of candidates at a time
    int64_t accepted = 0, failed = 0;
    const int64_t fail_threshold = 10 * num_points;
    for (int64_t k0 = 0; e == hipSuccess && accepted < num_points && failed < fail_threshold; k0 += chunk) {
        const unsigned long long all = (unsigned long long)chunk;
        e = hipMemcpyAsync(cutoff.ptr, &all, 8, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_world_candidates, dim3(blocks_of(chunk, 256)), dim3(256), 0, st, (const float *)d_tri.as<float>(), n_tri,
                           (const double *)d_cum.as<double>(), seed, k0, chunk, centres, g,
                           (const uint32_t *)startb.as<uint32_t>(), (const uint32_t *)sorted.as<uint32_t>(), max_dist, cand.as<double4>(),
                           ok.as<uint32_t>());
        uint32_t n_ok = 0;
        e = scan_flags(st, ok.as<uint32_t>(), chunk, pos.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &n_ok);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_world_cutoff, dim3(blocks_of(chunk, 256)), dim3(256), 0, st, (const uint32_t *)pos.as<uint32_t>(), chunk,
                           (uint64_t)(num_points - accepted), (uint64_t)(fail_threshold - failed), cutoff.as<unsigned long long>());
        hipLaunchKernelGGL(k_world_accept, dim3(blocks_of(chunk, 256)), dim3(256), 0, st, (const double4 *)cand.as<double4>(),
                           (const uint32_t *)ok.as<uint32_t>(), (const uint32_t *)pos.as<uint32_t>(), chunk,
                           (const unsigned long long *)cutoff.as<unsigned long long>(), out.as<double4>() + accepted);
        e = launch_error();
        unsigned long long cut = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&cut, cutoff.ptr, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) break;
        uint32_t took = n_ok;                                 // accepted among the candidates before the cutoff
        if ((int64_t)cut < chunk) {
            e = hipMemcpy(&took, pos.as<uint32_t>() + cut, 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) break;
        }
        accepted += (int64_t)took;
        failed += (int64_t)cut - (int64_t)took;
    }
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_generate_world_points: %s", hipGetErrorString(e));
    if (failed >= fail_threshold && num_points > 0)
        return fail(C2B_ERR_INVALID_ARGUMENT, "Failed to generate enough points. %lld successes, %lld failures, %lld requested points.",
                    (long long)accepted, (long long)failed, (long long)num_points);
    // the sampled points become the problem's
    free_dense(p);
    drop_rows(p);
    if (p->pts4) (void)hipFree(p->pts4);
    p->pts4 = (double *)out.release();
    p->n_pts = accepted;
    *n_out = accepted;
    return C2B_OK;
    C2B_API_END("problem_generate_world_points")
}

#include "capi_files.hpp"        // c2b_problem_write / c2b_problem_read: both file forms assembled / taken apart on the device

int c2b_problem_visibility_dense(c2b_problem *p, double max_dist, uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_dense");
    if (!row_ptr) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense: row_ptr is NULL");
    if (!(max_dist >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense: max_dist must be >= 0");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    free_dense(p);
    const int64_t n_tiles = c2b_visibility_dense_tiles(p->n_pts);
    const int64_t cells = p->n_cam * n_tiles;
    if (cells > ((int64_t)1 << 33))
        return fail(C2B_ERR_INVALID_ARGUMENT,
                    "problem_visibility_dense: %lld cameras x %lld point tiles is too large for the dense sweep; "
                    "use candidate pairs + c2b_problem_visibility_pairs", (long long)p->n_cam, (long long)n_tiles);
    uint32_t *d_counts = nullptr;
    uint64_t *d_tot = nullptr, *d_row = nullptr;
    hipError_t e = hipMalloc((void **)&d_counts, sizeof(uint32_t) * (size_t)(cells ? cells : 4));
    if (e == hipSuccess) e = hipMalloc((void **)&d_tot, sizeof(uint64_t) * (size_t)(p->n_cam + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_row, sizeof(uint64_t) * (size_t)(p->n_cam + 1));
    if (e == hipSuccess) {
        rc = c2b_visibility_dense_count(p->camblk, p->n_cam, p->pts4, p->n_pts, max_dist, d_counts, d_tot, d_row, p->stream);
        if (!rc) e = hipMemcpyAsync(row_ptr, d_row, sizeof(uint64_t) * (size_t)(p->n_cam + 1), hipMemcpyDeviceToHost, p->stream);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (!rc && e == hipSuccess) {
            const int64_t total = (int64_t)row_ptr[p->n_cam];
            e = hipMalloc((void **)&p->dense_pt, sizeof(uint32_t) * (size_t)(total ? total : 4));
            if (e == hipSuccess) e = hipMalloc((void **)&p->dense_uv, sizeof(double) * 2 * (size_t)(total ? total : 1));
            if (e == hipSuccess && total) {
                rc = c2b_visibility_dense_fill(p->camblk, p->n_cam, p->pts4, p->n_pts, max_dist, d_counts, d_row, p->dense_pt,
                                               p->dense_uv, p->stream);
                if (!rc) e = hipStreamSynchronize(p->stream);
            }
            if (!rc && e == hipSuccess) { p->dense_n = total; p->dense_row = d_row; d_row = nullptr; }
        }
    }
    if (d_counts) (void)hipFree(d_counts);
    if (d_tot) (void)hipFree(d_tot);
    if (d_row) (void)hipFree(d_row);
    if (rc || e != hipSuccess) free_dense(p);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_dense: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_dense")
}

// `ready`: a hierarchy the caller built over the same triangles (c2b_bvh_build), or NULL: built here when the mesh is
// large enough for one
static int occlude_impl(c2b_problem *p, const float *tri9, int64_t n_tri, const c2b_bvh *ready, uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_dense_occlude");
    if (!p->dense_pt || !p->dense_uv || !p->dense_row)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude: no sweep result");
    if (!row_ptr || n_tri < 0 || (n_tri && !tri9 && !ready)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude: bad arguments");
    const int64_t n = p->dense_n, n_cam = p->n_cam;
    if (!n || !n_tri) {
        HIP_TRY(hipMemcpyAsync(row_ptr, p->dense_row, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return C2B_OK;
    }
    float *d_tri = nullptr;
    uint32_t *d_cam = nullptr, *d_pt_new = nullptr, *d_flag = nullptr;
    uint32_t stack_overflow = 0;
    uint8_t *d_keep = nullptr;
    uint64_t *d_tot = nullptr, *d_row_new = nullptr;
    double *d_uv_new = nullptr;
    int rc = C2B_OK;
    // small meshes: every ray against every triangle; larger ones through a hierarchy built here on the host
    const bool use_bvh = ready || n_tri >= kBvhMinTriangles;
    c2b_bvh *built = nullptr;
    void *d_nodes = nullptr;
    int64_t n_nodes = 0;
    if (use_bvh && !ready) {
        rc = c2b_bvh_build(tri9, n_tri, &built);
        if (rc) return rc;
    }
    const c2b_bvh *bvh = ready ? ready : built;
    if (use_bvh) n_nodes = (int64_t)bvh->b.nodes.size();
    const size_t tri_bytes = use_bvh ? (size_t)C2B_BVH_TRI_BYTES * (size_t)n_tri : sizeof(float) * 9 * (size_t)n_tri;
    const void *tri_src = use_bvh ? (const void *)bvh->b.tris.data() : (const void *)tri9;
    hipError_t e = hipMalloc((void **)&d_tri, tri_bytes);
    if (e == hipSuccess && use_bvh) e = hipMalloc(&d_nodes, (size_t)C2B_BVH_NODE_BYTES * (size_t)n_nodes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cam, sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_keep, (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_flag, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(uint32_t), p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tri, tri_src, tri_bytes, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess && use_bvh)
        e = hipMemcpyAsync(d_nodes, bvh->b.nodes.data(), (size_t)C2B_BVH_NODE_BYTES * (size_t)n_nodes, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        rc = c2b_expand_rows(p->dense_row, n_cam, 0, n, d_cam, p->stream);
        if (!rc)
            rc = use_bvh ? c2b_occlusion_filter_bvh(p->camblk, p->pts4, d_cam, p->dense_pt, n, d_nodes, n_nodes, d_tri, n_tri, d_keep, d_flag, p->stream)
                         : c2b_occlusion_filter(p->camblk, p->pts4, d_cam, p->dense_pt, n, d_tri, n_tri, d_keep, p->stream);
        // Stable compaction of the survivor lists on the device (per-camera order of the sweep is kept): kept count per
        // row, row scan, scatter.  Only the new row pointer travels to the host.
        if (!rc) {                                         // a traversal-stack overflow invalidates the whole mask
            e = hipMemcpyAsync(&stack_overflow, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
            if (e == hipSuccess && stack_overflow)
                rc = fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude: hierarchy deeper than the traversal stack");
        }
        if (!rc && e == hipSuccess) {
            int64_t w = 0;
            e = compact_rows_on_device(p, p->dense_row, d_keep, p->dense_pt, p->dense_uv, n_cam, row_ptr, &d_row_new, &d_pt_new,
                                       &d_uv_new, &w);
            if (e == hipSuccess) {                           // the filtered lists replace the sweep's
                std::swap(p->dense_pt, d_pt_new);
                std::swap(p->dense_uv, d_uv_new);
                std::swap(p->dense_row, d_row_new);
                p->dense_n = w;
            }
        }
    }
    if (rc || e != hipSuccess) (void)hipStreamSynchronize(p->stream);   // nothing below may free what a copy still reads
    if (d_tri) (void)hipFree(d_tri);
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_cam) (void)hipFree(d_cam);
    if (d_keep) (void)hipFree(d_keep);
    if (d_flag) (void)hipFree(d_flag);
    if (d_tot) (void)hipFree(d_tot);
    if (d_row_new) (void)hipFree(d_row_new);
    if (d_pt_new) (void)hipFree(d_pt_new);
    if (d_uv_new) (void)hipFree(d_uv_new);
    c2b_bvh_free(built);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_dense_occlude: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_dense_occlude")
}

int c2b_problem_visibility_dense_occlude(c2b_problem *p, const float *tri9, int64_t n_tri, uint64_t *row_ptr) {
    return occlude_impl(p, tri9, n_tri, nullptr, row_ptr);
}
int c2b_problem_visibility_dense_occlude_bvh(c2b_problem *p, const c2b_bvh *bvh, uint64_t *row_ptr) {
    if (!bvh) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude_bvh: bvh is NULL");
    return occlude_impl(p, nullptr, (int64_t)(bvh->b.tris.size() / (C2B_BVH_TRI_BYTES / sizeof(bvh->b.tris[0]))), bvh, row_ptr);
}

int c2b_problem_visibility_dense_fetch(c2b_problem *p, uint64_t *pt_idx, double *uv) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_dense_fetch");
    if (!p->dense_pt || !p->dense_uv) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_fetch: no sweep result");
    const int64_t n = p->dense_n;
    if (!n) return C2B_OK;
    if (uv) HIP_TRY(hipMemcpyAsync(uv, p->dense_uv, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, p->stream));
    if (pt_idx) {
        std::vector<uint32_t> tmp((size_t)n);
        HIP_TRY(hipMemcpyAsync(tmp.data(), p->dense_pt, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (int64_t i = 0; i < n; ++i) pt_idx[i] = tmp[(size_t)i];
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_visibility_dense_fetch")
}

static void cameras_mutated(c2b_problem *p) { p->bal_valid = false; p->blk_valid = false; p->bal9_fresh = false; }

int c2b_problem_add_drift(c2b_problem *p, double strength, double angle_strength, double std, const double dir[3],
                          uint64_t seed) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_drift");
    if (!dir) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_drift: dir is NULL");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_drift(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats + 15, strength, angle_strength, std, dir[0],
                       dir[1], dir[2], seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_drift")
}

int c2b_problem_add_drift_normalized(c2b_problem *p, double strength, double angle_strength, double std,
                                     uint64_t seed) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_drift_normalized");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_drift_normalized(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, strength, angle_strength, std, seed,
                                  p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_drift_normalized")
}

int c2b_problem_add_noise(c2b_problem *p, double translation_std, double rotation_std, double point_std,
                          double observations_std, uint64_t seed) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_noise");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_noise_entities(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, translation_std, rotation_std,
                                point_std, seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    rc = c2b_add_noise_observations(p->uv, p->n_obs, 0, observations_std, seed, p->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_noise")
}

// add_noise followed by the L1 / L2 errors of the result -- run_noise's tail (src/bin/city2ba.rs:334-354) -- with the
// observation pass and both error sums in one launch.  comm != NULL: the problem is a shard (statistics and the
// 2-element sum go through the communicator).
static int sharded_stats(c2b_problem *p, c2b_comm *comm);
static int add_noise_errors_impl(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std, double point_std,
                                 double observations_std, uint64_t seed, double *l1, double *l2) {
    int rc = comm ? sharded_stats(p, comm) : compute_stats(p);
    if (rc) return rc;
    rc = comm ? c2b_add_noise_entities_sharded(p->cam15, p->n_cam, p->shard_cam_base, p->pts4, p->n_pts, p->stats, translation_std,
                                               rotation_std, point_std, seed, p->stream)
              : c2b_add_noise_entities(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, translation_std, rotation_std, point_std,
                                       seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    rc = ensure_camblk(p);                                 // the perturbed cameras' records
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    if (p->n_obs > 0)
        rc = c2b_add_noise_observations_error_sums2_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv,
                                                         p->n_obs, comm ? p->shard_obs_base : 0, observations_std, seed, p->ws,
                                                         p->scalar, p->stream);
    else
        HIP_TRY(hipMemsetAsync(p->scalar, 0, 2 * sizeof(double), p->stream));
    if (!rc && comm) rc = c2b_comm_all_reduce_sum_f64(comm, p->scalar, 2, p->stream);
    if (rc) return rc;
    double sums[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(sums, p->scalar, 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *l1 = std::pow(sums[0], 1.0 / 1.0);
    *l2 = std::pow(sums[1], 1.0 / 2.0);
    return C2B_OK;
}

int c2b_problem_add_noise_errors_l1_l2(c2b_problem *p, double translation_std, double rotation_std, double point_std,
                                       double observations_std, uint64_t seed, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_noise_errors_l1_l2");
    if (!l1 || !l2) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_noise_errors_l1_l2: NULL output");
    return add_noise_errors_impl(p, nullptr, translation_std, rotation_std, point_std, observations_std, seed, l1, l2);
    C2B_API_END("problem_add_noise_errors_l1_l2")
}

int c2b_problem_add_sin_noise(c2b_problem *p, const double dir[3], const double noise_dir[3], double strength,
                              double frequency) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_sin_noise");
    if (!dir || !noise_dir) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_sin_noise: NULL direction");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_sin_noise(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, dir[0], dir[1], dir[2], noise_dir[0],
                           noise_dir[1], noise_dir[2], strength, frequency, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_sin_noise")
}

// ---- Level 1 for a problem that is ONE SHARD of a larger one -------------------------------
// One c2b_problem per GPU holds a contiguous camera range (c2b_partition_cameras), its slice of the observation list
// and the WHOLE point table.  After c2b_problem_set_shard the *_sharded entries below give, shard by shard, exactly what
// the unsharded calls give on the whole problem: draws are keyed by global indices, the statistics go through the
// communicator (c2b_stats_sharded), every rank perturbs the replicated points identically.  All are collective (every
// rank of the communicator calls them in the same order) and synchronous.
int c2b_problem_set_shard(c2b_problem *p, int64_t cam_base, int64_t n_cam_global, int64_t obs_base) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_set_shard");
    if (cam_base < 0 || obs_base < 0 || n_cam_global < cam_base + p->n_cam)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_set_shard: the shard [%lld, %lld) does not fit %lld cameras",
                    (long long)cam_base, (long long)(cam_base + p->n_cam), (long long)n_cam_global);
    p->shard_cam_base = cam_base; p->shard_n_cam_global = n_cam_global; p->shard_obs_base = obs_base;
    return C2B_OK;
    C2B_API_END("problem_set_shard")
}

#define NEED_SHARD(p, comm, who)                                                                          \
    NEED_UPLOADED(p, who);                                                                                \
    if (!(comm)) return fail(C2B_ERR_INVALID_ARGUMENT, who ": communicator is NULL");                     \
    if ((p)->shard_n_cam_global < 0) return fail(C2B_ERR_INVALID_ARGUMENT, who ": c2b_problem_set_shard first"); \
    if ((comm)->device != (p)->device) return fail(C2B_ERR_INVALID_ARGUMENT, who ": communicator and problem live on different devices")

static int sharded_stats(c2b_problem *p, c2b_comm *comm) {
    int rc = ensure_camblk(p);
    if (rc) return rc;
    return c2b_stats_sharded(comm, p->camblk, p->cen4, p->n_cam, p->shard_cam_base, p->shard_n_cam_global, p->pts4, p->n_pts, p->ws,
                             p->stats, p->stream);
}

int c2b_problem_stats_sharded(c2b_problem *p, c2b_comm *comm, double *stats) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_stats_sharded");
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_stats_sharded: stats is NULL");
    const int rc = sharded_stats(p, comm);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof(double) * C2B_STATS_DOUBLES, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_stats_sharded")
}

// dir == NULL: add_drift_normalized (direction and scale from the global std, src/noise.rs:47-56)
int c2b_problem_add_drift_sharded(c2b_problem *p, c2b_comm *comm, double strength, double angle_strength, double std,
                                  const double *dir, uint64_t seed) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_drift_sharded");
    int rc = sharded_stats(p, comm);
    if (rc) return rc;
    rc = c2b_add_drift_sharded(p->cam15, p->n_cam, p->shard_cam_base, p->pts4, p->n_pts, p->stats, dir ? 0 : 1, strength,
                               angle_strength, std, dir ? dir[0] : 0.0, dir ? dir[1] : 0.0, dir ? dir[2] : 0.0, seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_drift_sharded")
}

int c2b_problem_add_noise_sharded(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std,
                                  double point_std, double observations_std, uint64_t seed) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_noise_sharded");
    int rc = sharded_stats(p, comm);
    if (rc) return rc;
    rc = c2b_add_noise_entities_sharded(p->cam15, p->n_cam, p->shard_cam_base, p->pts4, p->n_pts, p->stats, translation_std,
                                        rotation_std, point_std, seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    rc = c2b_add_noise_observations(p->uv, p->n_obs, p->shard_obs_base, observations_std, seed, p->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_noise_sharded")
}

int c2b_problem_add_noise_errors_l1_l2_sharded(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std,
                                               double point_std, double observations_std, uint64_t seed, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_noise_errors_l1_l2_sharded");
    if (!l1 || !l2) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_noise_errors_l1_l2_sharded: NULL output");
    return add_noise_errors_impl(p, comm, translation_std, rotation_std, point_std, observations_std, seed, l1, l2);
    C2B_API_END("problem_add_noise_errors_l1_l2_sharded")
}

int c2b_problem_add_sin_noise_sharded(c2b_problem *p, c2b_comm *comm, const double dir[3], const double noise_dir[3],
                                      double strength, double frequency) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_sin_noise_sharded");
    if (!dir || !noise_dir) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_sin_noise_sharded: NULL direction");
    int rc = sharded_stats(p, comm);                        // the extent of the WHOLE problem scales the phase
    if (rc) return rc;
    rc = c2b_add_sin_noise(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, dir[0], dir[1], dir[2], noise_dir[0],
                           noise_dir[1], noise_dir[2], strength, frequency, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_sin_noise_sharded")
}

