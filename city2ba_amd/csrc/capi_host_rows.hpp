// capi_host_rows.hpp -- the host-side rows of include/city2ba_hip.h (CPU C++ behind the same ABI): synthetic layout and candidate pairs, .obj / samplers of the mesh generator, host cull, index-corruption noise, .bal / .bbal / .ply files
// Part of the one translation unit of the C ABI: included by capi.hip (inside its extern "C" block, after its helpers and
// launchers), never compiled or included on its own.

/* ------------------------------- host-side generator pieces -------------------------- */

int c2b_synthetic_grid_sizes(int64_t cpb, int64_t ppb, int64_t blocks, int64_t *n_cam, int64_t *n_pts) {
    C2B_API_BEGIN
    if (cpb < 0 || ppb < 0 || blocks < 0 || !n_cam || !n_pts)
        return fail(C2B_ERR_INVALID_ARGUMENT, "synthetic_grid_sizes: bad arguments");
    c2b_host::grid_sizes(cpb, ppb, blocks, n_cam, n_pts);
    return C2B_OK;
    C2B_API_END("synthetic_grid_sizes")
}

int c2b_synthetic_grid_layout(int64_t cpb, int64_t ppb, int64_t blocks, double block_length, double block_inset,
                              double camera_height, double point_height, double *cam_pos3, double *cam_dir9,
                              double *pts3) {
    C2B_API_BEGIN
    if (cpb < 0 || ppb < 0 || blocks < 0 || !cam_pos3 || !cam_dir9 || !pts3)
        return fail(C2B_ERR_INVALID_ARGUMENT, "synthetic_grid_layout: bad arguments");
    // assert!(block_inset * 2. < block_length, ...), src/synthetic.rs:177
    if (!(block_inset * 2.0 < block_length))
        return fail(C2B_ERR_INVALID_ARGUMENT,
                    "Block inset (%g) must be less than half the block length (%g), to not violate physical constraints.",
                    block_inset, block_length);
    c2b_host::grid_layout(cpb, ppb, blocks, block_length, block_inset, camera_height, point_height, cam_pos3,
                          cam_dir9, pts3);
    return C2B_OK;
    C2B_API_END("synthetic_grid_layout")
}

int c2b_synthetic_line_layout(int64_t n_cam, int64_t n_pts, double length, double point_offset, double camera_height,
                              double point_height, double *cam_pos3, double *cam_dir9, double *pts3) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || (n_cam && (!cam_pos3 || !cam_dir9)) || (n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "synthetic_line_layout: bad arguments");
    c2b_host::line_layout(n_cam, n_pts, length, point_offset, camera_height, point_height, cam_pos3, cam_dir9, pts3);
    return C2B_OK;
    C2B_API_END("synthetic_line_layout")
}

struct c2b_pairs {
    c2b_host::Pairs v;
};

int c2b_candidate_pairs(const double *centers3, int64_t n_cam, const double *pts3, int64_t n_pts, double max_dist,
                        int64_t cam_lo, int64_t cam_hi, int occlusion, double block_length, double block_inset,
                        int n_threads, c2b_pairs **out) {
    C2B_API_BEGIN
    if (!out) return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: out is NULL");
    *out = nullptr;
    if (n_cam < 0 || n_pts < 0 || cam_lo < 0 || cam_hi > n_cam || cam_lo > cam_hi || (n_cam && !centers3) ||
        (n_pts && !pts3) || !(max_dist >= 0.0))
        return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: bad arguments");
    if (n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: indices are 32-bit");
    if (occlusion && !(block_length > 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: block_length must be > 0");
    c2b_pairs *p = new (std::nothrow) c2b_pairs();
    if (!p) return fail(C2B_ERR_OOM, "candidate_pairs: host allocation failed");
    try {
        c2b_host::candidate_pairs(centers3, pts3, n_pts, max_dist, cam_lo, cam_hi, occlusion != 0, block_length,
                                  block_inset, n_threads, &p->v);
    } catch (const std::bad_alloc &) {
        delete p;
        return fail(C2B_ERR_OOM, "candidate_pairs: out of host memory");
    } catch (const std::exception &e) {
        delete p;
        return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: %s", e.what());
    }
    *out = p;
    return C2B_OK;
    C2B_API_END("candidate_pairs")
}

int64_t c2b_pairs_count(const c2b_pairs *p) { return p ? (int64_t)p->v.cam.size() : 0; }
const uint32_t *c2b_pairs_cam_idx(const c2b_pairs *p) { return p ? p->v.cam.data() : nullptr; }
const uint32_t *c2b_pairs_pt_idx(const c2b_pairs *p) { return p ? p->v.pt.data() : nullptr; }
void c2b_pairs_free(c2b_pairs *p) { delete p; }

/* ---- mesh generator, host side ---- */
struct c2b_obj {
    std::vector<c2b_host::ObjModel> models;
};

int c2b_obj_load(const char *path, c2b_obj **out) {
    C2B_API_BEGIN
    if (!path || !out) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_load: bad arguments");
    *out = nullptr;
    c2b_obj *o = new (std::nothrow) c2b_obj();
    if (!o) return fail(C2B_ERR_OOM, "obj_load: host allocation failed");
    std::string err;
    bool ok = false;
    try {
        ok = c2b_host::load_obj(path, o->models, &err);
    } catch (const std::bad_alloc &) {
        err = "out of host memory";
    }
    if (!ok) { delete o; return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str()); }
    *out = o;
    return C2B_OK;
    C2B_API_END("obj_load")
}

int64_t c2b_obj_model_count(const c2b_obj *o) { return o ? (int64_t)o->models.size() : 0; }

const char *c2b_obj_model_name(const c2b_obj *o, int64_t m) {
    return (o && m >= 0 && m < (int64_t)o->models.size()) ? o->models[(size_t)m].name.c_str() : nullptr;
}

int c2b_obj_model_sizes(const c2b_obj *o, int64_t m, int64_t *n_positions, int64_t *n_indices, int *is_lines) {
    C2B_API_BEGIN
    if (!o || m < 0 || m >= (int64_t)o->models.size()) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_model_sizes: bad model index");
    const c2b_host::ObjModel &mod = o->models[(size_t)m];
    if (n_positions) *n_positions = (int64_t)(mod.positions.size() / 3);
    if (n_indices) *n_indices = (int64_t)mod.indices.size();
    if (is_lines) *is_lines = mod.lines ? 1 : 0;
    return C2B_OK;
    C2B_API_END("obj_model_sizes")
}

int c2b_obj_model_copy(const c2b_obj *o, int64_t m, float *positions3, uint32_t *indices) {
    C2B_API_BEGIN
    if (!o || m < 0 || m >= (int64_t)o->models.size()) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_model_copy: bad model index");
    const c2b_host::ObjModel &mod = o->models[(size_t)m];
    if (positions3) std::copy(mod.positions.begin(), mod.positions.end(), positions3);
    if (indices) std::copy(mod.indices.begin(), mod.indices.end(), indices);
    return C2B_OK;
    C2B_API_END("obj_model_copy")
}

int c2b_obj_move_to_origin(c2b_obj *o, int64_t skip_model) {
    C2B_API_BEGIN
    if (!o) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_move_to_origin: obj is NULL");
    c2b_host::move_to_origin(o->models, skip_model);
    return C2B_OK;
    C2B_API_END("obj_move_to_origin")
}

int c2b_obj_triangles(const c2b_obj *o, int64_t skip_model, float *tri9, int64_t *n_tri) {
    C2B_API_BEGIN
    if (!o || !n_tri) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_triangles: bad arguments");
    std::vector<c2b_host::ObjModel> use;
    for (int64_t m = 0; m < (int64_t)o->models.size(); ++m)
        if (m != skip_model) use.push_back(o->models[(size_t)m]);
    std::vector<float> t;
    c2b_host::triangles_of(use, t);
    *n_tri = (int64_t)(t.size() / 9);
    if (tri9) std::copy(t.begin(), t.end(), tri9);
    return C2B_OK;
    C2B_API_END("obj_triangles")
}

void c2b_obj_free(c2b_obj *o) { delete o; }

int c2b_generate_cameras_path(const c2b_obj *o, int64_t path_model, int64_t num_cameras, double step_size, uint64_t seed,
                              double *cam_pos3, double *cam_dir9) {
    C2B_API_BEGIN
    if (!o || path_model < 0 || path_model >= (int64_t)o->models.size() || num_cameras < 0 || (num_cameras && (!cam_pos3 || !cam_dir9)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_path: bad arguments");
    const c2b_host::ObjModel &path = o->models[(size_t)path_model];
    if (!path.lines) return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_path: model '%s' is not a polyline", path.name.c_str());
    c2b_host::CameraSamples cs;
    std::string err;
    double total = 0;
    const bool ok = step_size <= 0.0 ? c2b_host::cameras_path(path, num_cameras, seed, cs, &err)
                                     : c2b_host::cameras_path_step(path, num_cameras, step_size, cs, &err, &total);
    if (!ok) return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    std::copy(cs.pos.begin(), cs.pos.end(), cam_pos3);
    std::copy(cs.dir.begin(), cs.dir.end(), cam_dir9);
    return C2B_OK;
    C2B_API_END("generate_cameras_path")
}

static int cameras_poisson_impl(const float *tri9, int64_t n_tri, const c2b_host::Bvh *ready, int64_t num_points, double height, double ground,
                                uint64_t seed, int64_t capacity, double *cam_pos3, double *cam_dir9, int64_t *n_out) {
    C2B_API_BEGIN
    if (!tri9 || n_tri <= 0 || num_points < 0 || capacity < 0 || !n_out || (capacity && (!cam_pos3 || !cam_dir9)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_poisson: bad arguments");
    const std::vector<float> tri(tri9, tri9 + 9 * n_tri);
    c2b_host::CameraSamples cs;
    c2b_host::cameras_poisson(tri, num_points, height, ground, seed, cs, ready);
    const int64_t n = std::min<int64_t>((int64_t)cs.size(), capacity);
    if (n) {
        std::copy(cs.pos.begin(), cs.pos.begin() + 3 * n, cam_pos3);
        std::copy(cs.dir.begin(), cs.dir.begin() + 9 * n, cam_dir9);
    }
    *n_out = (int64_t)cs.size();
    return C2B_OK;
    C2B_API_END("generate_cameras_poisson")
}
int c2b_generate_cameras_poisson(const float *tri9, int64_t n_tri, int64_t num_points, double height, double ground,
                                 uint64_t seed, int64_t capacity, double *cam_pos3, double *cam_dir9, int64_t *n_out) {
    return cameras_poisson_impl(tri9, n_tri, nullptr, num_points, height, ground, seed, capacity, cam_pos3, cam_dir9, n_out);
}
int c2b_generate_cameras_poisson_bvh(const float *tri9, int64_t n_tri, const c2b_bvh *bvh, int64_t num_points, double height,
                                     double ground, uint64_t seed, int64_t capacity, double *cam_pos3, double *cam_dir9,
                                     int64_t *n_out) {
    if (!bvh) return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_poisson_bvh: bvh is NULL");
    return cameras_poisson_impl(tri9, n_tri, &bvh->b, num_points, height, ground, seed, capacity, cam_pos3, cam_dir9, n_out);
}

int c2b_modify_intrinsics(double *cams15, int64_t n_cam, const double start[3], const double end[3], uint64_t seed) {
    C2B_API_BEGIN
    if (n_cam < 0 || (n_cam && !cams15) || !start || !end) return fail(C2B_ERR_INVALID_ARGUMENT, "modify_intrinsics: bad arguments");
    c2b_host::modify_intrinsics(cams15, n_cam, start, end, seed);
    return C2B_OK;
    C2B_API_END("modify_intrinsics")
}

int c2b_generate_world_points(const float *tri9, int64_t n_tri, const double *centers3, int64_t n_cam, int64_t num_points,
                              double max_dist, uint64_t seed, double *pts3, int64_t *n_out) {
    C2B_API_BEGIN
    if (!tri9 || n_tri < 0 || n_cam < 0 || (n_cam && !centers3) || num_points < 0 || !n_out || (num_points && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "generate_world_points: bad arguments");
    const std::vector<float> tri(tri9, tri9 + 9 * n_tri);
    std::vector<double> pts;
    std::string err;
    if (!c2b_host::world_points_uniform(tri, centers3, n_cam, num_points, max_dist, seed, pts, &err))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    std::copy(pts.begin(), pts.end(), pts3);
    *n_out = (int64_t)(pts.size() / 3);
    return C2B_OK;
    C2B_API_END("generate_world_points")
}

static int cull_host(int mode, int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3, uint64_t *row_ptr,
                     uint64_t *pt_idx, double *uv, int faithful) {
    C2B_API_BEGIN
    if (!n_cam || !n_pts || !row_ptr || *n_cam < 0 || *n_pts < 0 || cam_stride < 0 || (*n_cam && cam_stride && !cams) ||
        (*n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "cull: bad arguments");
    const int64_t n_obs = (int64_t)row_ptr[*n_cam];
    if (n_obs && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "cull: NULL observations");
    for (int64_t o = 0; o < n_obs; ++o)
        if (pt_idx[o] >= (uint64_t)*n_pts) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "cull: point index out of range");
    try {
        c2b_host::Graph g;
        g.n_cam = *n_cam; g.n_pts = *n_pts; g.stride = cam_stride;
        g.cams.assign(cams, cams + (size_t)*n_cam * cam_stride);
        g.pts.assign(pts3, pts3 + (size_t)*n_pts * 3);
        g.row_ptr.assign(row_ptr, row_ptr + *n_cam + 1);
        g.pt_idx.assign(pt_idx, pt_idx + n_obs);
        g.uv.assign(uv, uv + 2 * n_obs);
        const c2b_host::Graph c = c2b_host::cull(g, faithful != 0, mode);
        std::copy(c.cams.begin(), c.cams.end(), cams);
        std::copy(c.pts.begin(), c.pts.end(), pts3);
        std::copy(c.row_ptr.begin(), c.row_ptr.end(), row_ptr);
        std::copy(c.pt_idx.begin(), c.pt_idx.end(), pt_idx);
        std::copy(c.uv.begin(), c.uv.end(), uv);
        *n_cam = c.n_cam;
        *n_pts = c.n_pts;
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "cull: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("cull_host")
}

int c2b_cull(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3, uint64_t *row_ptr,
             uint64_t *pt_idx, double *uv, int faithful) {
    C2B_API_BEGIN
    return cull_host(0, n_cam, cams, cam_stride, n_pts, pts3, row_ptr, pt_idx, uv, faithful);
    C2B_API_END("cull")
}
int c2b_largest_connected_component(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3,
                                    uint64_t *row_ptr, uint64_t *pt_idx, double *uv, int faithful) {
    C2B_API_BEGIN
    return cull_host(1, n_cam, cams, cam_stride, n_pts, pts3, row_ptr, pt_idx, uv, faithful);
    C2B_API_END("largest_connected_component")
}
int c2b_remove_singletons(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3, uint64_t *row_ptr,
                          uint64_t *pt_idx, double *uv) {
    C2B_API_BEGIN
    return cull_host(2, n_cam, cams, cam_stride, n_pts, pts3, row_ptr, pt_idx, uv, 1);
    C2B_API_END("remove_singletons")
}

/* ---- index-corruption noise, host side ---- */
static int check_csr(const char *who, int64_t n_cam, const uint64_t *row_ptr) {
    if (n_cam < 0 || !row_ptr) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    if (row_ptr[0] != 0) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: row_ptr[0] != 0", who);
    for (int64_t c = 0; c < n_cam; ++c)
        if (row_ptr[c + 1] < row_ptr[c]) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: row_ptr not monotone at camera %lld", who, (long long)c);
    return C2B_OK;
}

int c2b_add_incorrect_correspondences(int64_t n_cam, const uint64_t *row_ptr, uint64_t *pt_idx, const double *uv,
                                      double mismatch_chance, uint64_t seed) {
    C2B_API_BEGIN
    int rc = check_csr("add_incorrect_correspondences", n_cam, row_ptr);
    if (rc) return rc;
    if (row_ptr[n_cam] && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_incorrect_correspondences: NULL observations");
    std::string err;
    try {
        if (!c2b_host::add_incorrect_correspondences(n_cam, row_ptr, pt_idx, uv, mismatch_chance, seed, &err))
            return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "add_incorrect_correspondences: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("add_incorrect_correspondences")
}

int c2b_drop_features(int64_t n_cam, uint64_t *row_ptr, uint64_t *pt_idx, double *uv, double keep_fraction, uint64_t seed) {
    C2B_API_BEGIN
    int rc = check_csr("drop_features", n_cam, row_ptr);
    if (rc) return rc;
    if (row_ptr[n_cam] && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "drop_features: NULL observations");
    if (keep_fraction != keep_fraction) return fail(C2B_ERR_INVALID_ARGUMENT, "drop_features: keep_fraction is NaN");
    try {
        c2b_host::drop_features(n_cam, row_ptr, pt_idx, uv, keep_fraction, seed);
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "drop_features: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("drop_features")
}

int c2b_split_landmarks(int64_t *n_pts, double *pts3, int64_t pts_capacity, int64_t n_obs, uint64_t *pt_idx,
                        double split_fraction, uint64_t seed) {
    C2B_API_BEGIN
    if (!n_pts || *n_pts < 0 || n_obs < 0 || (*n_pts && !pts3) || (n_obs && !pt_idx) || split_fraction != split_fraction)
        return fail(C2B_ERR_INVALID_ARGUMENT, "split_landmarks: bad arguments");
    const uint64_t n = std::min<uint64_t>(c2b_host::fraction_of(split_fraction, (uint64_t)*n_pts), (uint64_t)*n_pts);
    if (pts_capacity < *n_pts + (int64_t)n)
        return fail(C2B_ERR_INVALID_ARGUMENT, "split_landmarks: pts3 holds %lld rows, %lld needed", (long long)pts_capacity,
                    (long long)(*n_pts + (int64_t)n));
    for (int64_t o = 0; o < n_obs; ++o)
        if (pt_idx[o] >= (uint64_t)*n_pts) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "split_landmarks: point index out of range");
    try {
        *n_pts = c2b_host::split_landmarks(*n_pts, pts3, n_obs, pt_idx, split_fraction, seed);
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "split_landmarks: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("split_landmarks")
}

int c2b_join_landmarks(int64_t n_pts, const double *pts3, int64_t n_obs, uint64_t *pt_idx, double join_fraction, uint64_t seed) {
    C2B_API_BEGIN
    if (n_pts < 0 || n_obs < 0 || (n_pts && !pts3) || (n_obs && !pt_idx) || join_fraction != join_fraction)
        return fail(C2B_ERR_INVALID_ARGUMENT, "join_landmarks: bad arguments");
    for (int64_t o = 0; o < n_obs; ++o)
        if (pt_idx[o] >= (uint64_t)n_pts) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "join_landmarks: point index out of range");
    std::string err;
    try {
        if (!c2b_host::join_landmarks(n_pts, pts3, n_obs, pt_idx, join_fraction, seed, &err))
            return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "join_landmarks: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("join_landmarks")
}

struct c2b_balfile {
    c2b_host::Graph g;
};

// format: 0 text (from_file_text), 1 binary (from_file_binary), -1 by extension (from_file)
static int bal_format(const char *path, int format, bool *binary) {
    if (format == 0 || format == 1) { *binary = format == 1; return C2B_OK; }
    const std::string ext = c2b_host::extension(path);
    if (ext.empty()) return fail(C2B_ERR_INVALID_ARGUMENT, "file does not have an extension");
    if (ext != "bal" && ext != "bbal") return fail(C2B_ERR_INVALID_ARGUMENT, "unknown file extension %s", ext.c_str());
    *binary = ext == "bbal";
    return C2B_OK;
}

int c2b_bal_read_as(const char *path, int format, c2b_balfile **out) {
    C2B_API_BEGIN
    if (!path || !out) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_read: bad arguments");
    *out = nullptr;
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const std::string ext = binary ? "bbal" : "bal";
    c2b_balfile *f = new (std::nothrow) c2b_balfile();
    if (!f) return fail(C2B_ERR_OOM, "bal_read: host allocation failed");
    std::string err;
    bool ok = false;
    try {
        ok = ext == "bal" ? c2b_host::read_text(path, f->g, &err) : c2b_host::read_binary(path, f->g, &err);
    } catch (const std::bad_alloc &) {
        err = "out of host memory";
    }
    if (!ok) {
        delete f;
        const bool range = err.find("assertion failed") != std::string::npos || err.find("out of range") != std::string::npos;
        return fail(range ? C2B_ERR_INDEX_OUT_OF_RANGE : C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    }
    *out = f;
    return C2B_OK;
    C2B_API_END("bal_read_as")
}

int c2b_bal_read(const char *path, c2b_balfile **out) { return c2b_bal_read_as(path, -1, out); }

int c2b_bal_sizes(const c2b_balfile *f, int64_t *n_cam, int64_t *n_pts, int64_t *n_obs) {
    C2B_API_BEGIN
    if (!f) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_sizes: file is NULL");
    if (n_cam) *n_cam = f->g.n_cam;
    if (n_pts) *n_pts = f->g.n_pts;
    if (n_obs) *n_obs = f->g.n_obs();
    return C2B_OK;
    C2B_API_END("bal_sizes")
}

int c2b_bal_copy(const c2b_balfile *f, double *bal9, double *pts3, uint64_t *row_ptr, uint64_t *pt_idx, double *uv) {
    C2B_API_BEGIN
    if (!f) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_copy: file is NULL");
    if (bal9) std::copy(f->g.cams.begin(), f->g.cams.end(), bal9);
    if (pts3) std::copy(f->g.pts.begin(), f->g.pts.end(), pts3);
    if (row_ptr) std::copy(f->g.row_ptr.begin(), f->g.row_ptr.end(), row_ptr);
    if (pt_idx) std::copy(f->g.pt_idx.begin(), f->g.pt_idx.end(), pt_idx);
    if (uv) std::copy(f->g.uv.begin(), f->g.uv.end(), uv);
    return C2B_OK;
    C2B_API_END("bal_copy")
}

void c2b_bal_close(c2b_balfile *f) { delete f; }

int c2b_bal_write_as(const char *path, int format, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                     const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    if (!path || n_cam < 0 || n_pts < 0 || !row_ptr || (n_cam && !bal9) || (n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "bal_write: bad arguments");
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const std::string ext = binary ? "bbal" : "bal";
    const int64_t n_obs = (int64_t)row_ptr[n_cam];
    if (n_obs && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_write: NULL observations");
    std::string err;
    bool ok = false;
    try {
        c2b_host::Graph g;
        g.n_cam = n_cam; g.n_pts = n_pts; g.stride = 9;
        g.cams.assign(bal9, bal9 + (size_t)n_cam * 9);
        g.pts.assign(pts3, pts3 + (size_t)n_pts * 3);
        g.row_ptr.assign(row_ptr, row_ptr + n_cam + 1);
        g.pt_idx.assign(pt_idx, pt_idx + n_obs);
        g.uv.assign(uv, uv + 2 * n_obs);
        ok = ext == "bal" ? c2b_host::write_text(path, g, &err) : c2b_host::write_binary(path, g, &err);
    } catch (const std::bad_alloc &) {
        err = "out of host memory";
    }
    if (!ok) return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    return C2B_OK;
    C2B_API_END("bal_write_as")
}

int c2b_bal_write(const char *path, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                  const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    return c2b_bal_write_as(path, -1, n_cam, bal9, n_pts, pts3, row_ptr, pt_idx, uv);
    C2B_API_END("bal_write")
}

int c2b_format_f64(int64_t n, const double *values, char *buf, int64_t cap, int64_t *len) {
    C2B_API_BEGIN
    if (n < 0 || (n && !values) || !buf || !len) return fail(C2B_ERR_INVALID_ARGUMENT, "format_f64: NULL argument");
    const c2b_dec::Tables &T = c2b_dec::host_tables();
    int64_t at = 0;
    for (int64_t i = 0; i < n; ++i) {
        const c2b_dec::Text t = c2b_dec::describe(values[i], &T);
        if (at + (int64_t)t.len + 1 > cap) return fail(C2B_ERR_INVALID_ARGUMENT, "format_f64: buffer too small");
        c2b_dec::emit(t, buf + at);
        at += t.len;
        buf[at++] = '\n';
    }
    *len = at;
    return C2B_OK;
    C2B_API_END("format_f64")
}

int c2b_parse_f64(const char *text, int64_t len, int64_t n, double *values, int32_t *status) {
    C2B_API_BEGIN
    if (len < 0 || n < 0 || (len && !text) || (n && (!values || !status))) return fail(C2B_ERR_INVALID_ARGUMENT, "parse_f64: bad argument");
    const c2b_dec::ParseTables &T = c2b_dec::host_parse_tables();
    auto ws = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; };
    int64_t i = 0, k = 0;
    for (; k < n; ++k) {
        while (i < len && ws(text[i])) ++i;
        if (i >= len) return fail(C2B_ERR_INVALID_ARGUMENT, "parse_f64: %lld tokens expected, %lld found", (long long)n, (long long)k);
        const int64_t b = i;
        while (i < len && !ws(text[i])) ++i;
        int st = c2b_dec::PARSE_OK;
        values[k] = i - b > 400 ? 0.0 : c2b_dec::parse_f64(text + b, (int32_t)(i - b), &T, &st);
        status[k] = i - b > 400 ? (int32_t)c2b_dec::PARSE_IRREGULAR : (int32_t)st;
    }
    return C2B_OK;
    C2B_API_END("parse_f64")
}

int c2b_ply_write(const char *path, int64_t n_cam, const double *centers3, int64_t n_pts, const double *pts3,
                  const uint64_t *row_ptr, const uint64_t *pt_idx) {
    C2B_API_BEGIN
    if (!path || n_cam < 0 || n_pts < 0 || (n_cam && (!centers3 || !row_ptr)) || (n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "ply_write: bad arguments");
    if (n_cam && row_ptr[n_cam] && !pt_idx) return fail(C2B_ERR_INVALID_ARGUMENT, "ply_write: pt_idx is NULL");
    std::string err;
    if (!c2b_host::write_ply(path, n_cam, centers3, n_pts, pts3, row_ptr, pt_idx, &err))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    return C2B_OK;
    C2B_API_END("ply_write")
}

