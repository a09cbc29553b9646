"""Host-side mirror of the reference's BAProblem<SnavelyCamera> for the hot path, over the C ABI.

Names, argument meaning and failure behaviour follow src/baproblem.rs (methods) and src/noise.rs
(free functions taking and returning a problem).  All arithmetic runs in the HIP library; this
file only moves numpy buffers across the boundary.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class ProblemOptions(C.Structure):
    """c2b_problem_options (include/city2ba_hip.h): which of two equivalent routes a problem's calls take and how many
    host threads they may use -- arguments of the C ABI, not environment variables"""
    _fields_ = [("host_text", C.c_int32), ("text_device_strict", C.c_int32), ("read_threads", C.c_int32), ("io_threads", C.c_int32),
                ("rank_sort_max_row", C.c_int32), ("reserved", C.c_int32), ("text_device_min_bytes", C.c_int64)]


_OPTION_NAMES = ("host_text", "text_device_strict", "read_threads", "io_threads", "rank_sort_max_row", "text_device_min_bytes")
_new_problem_options = {}        # what set_default_options changed: applied to every BAProblem made afterwards


def set_default_options(**kw):
    """options every BAProblem made from now on starts with (BAProblem.set_options changes one problem).  The classmethod
    constructors read files before a caller can reach the new object, so tests choose the parser this way.
    host_text, text_device_strict: bool; read_threads, io_threads, rank_sort_max_row: 0 = the library's default;
    text_device_min_bytes: -1 = the library's default (65 536)."""
    for k in kw:
        if k not in _OPTION_NAMES:
            raise TypeError("unknown option %r (one of %s)" % (k, ", ".join(_OPTION_NAMES)))
    _new_problem_options.update(kw)


def reset_default_options():
    _new_problem_options.clear()


def set_host_io_threads(n):
    """c2b_host_set_io_threads: threads of the host text formatter / parser for read_bal / write_bal and for problems
    whose options leave io_threads at 0 (n < 1: the library's default -- the usable cores, at most 16)"""
    L.lib().c2b_host_set_io_threads(int(n))


class BAProblem:
    """cameras + points + vis_graph (CSR) resident on one MI355X.

    `vis_graph` is the flattening of Vec<Vec<(usize,(f64,f64))>> (src/baproblem.rs:256-260):
    row_ptr[n_cam+1], pt_idx[n_obs], uv[n_obs,2], camera-major, in-camera order preserved."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._device = int(device)
        L.check(L.lib().c2b_problem_create(int(device), C.byref(self._h)))
        self._row_ptr = np.zeros(1, dtype=np.uint64)
        self._pt_idx = np.zeros(0, dtype=np.uint64)
        if _new_problem_options:
            self.set_options(**_new_problem_options)

    def options(self):
        """this problem's c2b_problem_options as a dict"""
        o = ProblemOptions()
        L.check(L.lib().c2b_problem_get_options(self._h, C.byref(o)))
        return {k: getattr(o, k) for k in _OPTION_NAMES}

    def set_options(self, **kw):
        """c2b_problem_set_options: change the named options of this problem (see set_default_options for the names)"""
        o = ProblemOptions()
        L.check(L.lib().c2b_problem_get_options(self._h, C.byref(o)))
        for k, v in kw.items():
            if k not in _OPTION_NAMES:
                raise TypeError("unknown option %r (one of %s)" % (k, ", ".join(_OPTION_NAMES)))
            setattr(o, k, int(v))
        L.check(L.lib().c2b_problem_set_options(self._h, C.byref(o)))
        return self

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib().c2b_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- constructors -------------------------------------------------------------------
    @classmethod
    def from_visibility(cls, cams15, points, row_ptr, pt_idx, uv, device=0):
        """BAProblem::from_visibility (src/baproblem.rs:360-376); cameras as in-memory
        SnavelyCamera records [n,15] (R col-major, t, intrin)."""
        self = cls(device)
        self._upload(_f64(cams15, (-1, 15)), False, points, row_ptr, pt_idx, uv)
        return self

    @classmethod
    def from_bal(cls, bal9, points, row_ptr, pt_idx, uv, device=0):
        """Same with cameras as the 9-vectors a .bal/.bbal file holds (src/baproblem.rs:605-608)."""
        self = cls(device)
        self._upload(_f64(bal9, (-1, 9)), True, points, row_ptr, pt_idx, uv)
        return self

    @classmethod
    def new(cls, bal9, points, obs, device=0):
        """BAProblem::new (src/baproblem.rs:342-355): obs = iterable of (cam, point, u, v)."""
        bal9 = _f64(bal9, (-1, 9))
        points = _f64(points, (-1, 3))
        obs = list(obs)
        n_cam = len(bal9)
        for (ci, pi, _, _) in obs:
            if not (0 <= ci < n_cam):       # assert!(cam_i < cams.len())
                raise L.City2baError(L.ERR_INDEX_OUT_OF_RANGE, "camera index %d out of range" % ci)
            if not (0 <= pi < len(points)):  # assert!(p_i < points.len())
                raise L.City2baError(L.ERR_INDEX_OUT_OF_RANGE, "point index %d out of range" % pi)
        order = sorted(range(len(obs)), key=lambda k: obs[k][0])     # stable: keeps push order
        counts = np.bincount([obs[k][0] for k in order], minlength=n_cam) if obs else np.zeros(n_cam, int)
        row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        pt_idx = np.array([obs[k][1] for k in order], dtype=np.uint64)
        uv = np.array([[obs[k][2], obs[k][3]] for k in order], dtype=np.float64).reshape(-1, 2)
        return cls.from_bal(bal9, points, row_ptr, pt_idx, uv, device)

    def _upload(self, cams, is_bal, points, row_ptr, pt_idx, uv):
        points = _f64(points, (-1, 3))
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint64)
        uv = _f64(uv, (-1, 2))
        if len(row_ptr) != len(cams) + 1:          # assert!(cams.len() == obs.len())
            raise L.City2baError(L.ERR_INVALID_ARGUMENT, "row_ptr must have n_cameras + 1 entries")
        if len(pt_idx) != len(uv) or (len(row_ptr) and int(row_ptr[-1]) != len(pt_idx)):
            raise L.City2baError(L.ERR_INVALID_ARGUMENT, "row_ptr[-1], pt_idx and uv disagree on n_obs")
        fn = L.lib().c2b_problem_upload_bal if is_bal else L.lib().c2b_problem_upload
        L.check(fn(self._h, len(cams), _ptr(cams), len(points), _ptr(points), _ptr(row_ptr), _ptr(pt_idx), _ptr(uv)))
        self._row_ptr, self._pt_idx = row_ptr, pt_idx

    # ---- counters (src/baproblem.rs:378-390) --------------------------------------------------
    def _sizes(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().c2b_problem_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def num_cameras(self):
        return self._sizes()[0]

    def num_points(self):
        return self._sizes()[1]

    def num_observations(self):
        return self._sizes()[2]

    def __str__(self):   # Display, src/baproblem.rs:788-801
        return "Bundle Adjustment Problem with %d cameras, %d points, and %d observations" % self._sizes()

    # ---- data access -----------------------------------------------------------------------------
    @property
    def row_ptr(self):
        return self._row_ptr

    @property
    def pt_idx(self):
        return self._pt_idx

    def cameras(self):
        """in-memory SnavelyCamera records [n,15]"""
        out = np.empty((self.num_cameras(), 15))
        L.check(L.lib().c2b_problem_download(self._h, _ptr(out), None, None))
        return out

    def cameras_bal(self):
        """SnavelyCamera::to_vec for every camera [n,9] (src/baproblem.rs:189-202)"""
        out = np.empty((self.num_cameras(), 9))
        L.check(L.lib().c2b_problem_download_bal(self._h, _ptr(out)))
        return out

    def points(self):
        out = np.empty((self.num_points(), 3))
        L.check(L.lib().c2b_problem_download(self._h, None, _ptr(out), None))
        return out

    def observations(self):
        out = np.empty((self.num_observations(), 2))
        L.check(L.lib().c2b_problem_download(self._h, None, None, _ptr(out)))
        return out

    # ---- hot path ------------------------------------------------------------------------------------
    def project(self):
        """camera.project(camera.project_world(point)) for every observation (src/baproblem.rs:272)."""
        out = np.empty((self.num_observations(), 2))
        L.check(L.lib().c2b_problem_project(self._h, _ptr(out)))
        return out

    def total_reprojection_error(self, norm):
        """src/baproblem.rs:265-279"""
        out = C.c_double()
        L.check(L.lib().c2b_problem_total_reprojection_error(self._h, float(norm), C.byref(out)))
        return out.value

    def total_reprojection_errors_l1_l2(self):
        """(total_reprojection_error(1.), total_reprojection_error(2.)) from ONE pass over the observations -- the pair
        run_noise prints before and after the noise (src/bin/city2ba.rs:283-287, 350-354)"""
        l1, l2 = C.c_double(), C.c_double()
        L.check(L.lib().c2b_problem_total_reprojection_errors_l1_l2(self._h, C.byref(l1), C.byref(l2)))
        return l1.value, l2.value

    def residual_jacobian(self, out=None, pinned=False):
        """r [n,2], Jc [n,2,9] (w t f k1 k2 columns), Jp [n,2,3].  Build-defined: the reference has no Jacobian.
        out = (r, Jc, Jp) reuses the caller's arrays; pinned=True returns arrays in page-locked memory
        (pinned_empty), which receive the results at PCIe speed -- keep them and pass them back as `out` when
        calling repeatedly."""
        n = self.num_observations()
        if out is not None:
            r, Jc, Jp = out
            for a, shape in ((r, (n, 2)), (Jc, (n, 2, 9)), (Jp, (n, 2, 3))):
                if a.dtype != np.float64 or a.shape != shape or not a.flags.c_contiguous:
                    raise L.City2baError(L.ERR_INVALID_ARGUMENT, "residual_jacobian: out arrays must be C-contiguous float64 of shapes (n,2), (n,2,9), (n,2,3)")
        elif pinned:
            r, Jc, Jp = pinned_empty((n, 2)), pinned_empty((n, 2, 9)), pinned_empty((n, 2, 3))
        else:
            r, Jc, Jp = np.empty((n, 2)), np.empty((n, 2, 9)), np.empty((n, 2, 3))
        L.check(L.lib().c2b_problem_residual_jacobian(self._h, _ptr(r), _ptr(Jc), _ptr(Jp)))
        return r, Jc, Jp

    def residual_jacobian_device(self, outputs=None, max_attempts=8):
        """residual + Jacobian of every observation in ONE launch with the results left on the device
        (c2b_problem_residual_jacobian_device): returns (outputs, sum_sq) -- outputs.r [n,2], .Jc [n,18], .Jp [n,6] are
        torch views of device arrays placed for streaming stores (device.JacobianOutputs; pass the object back in to
        reuse it in a loop), sum_sq = sum of squared residuals (total_reprojection_error(2.) ** 2).  The BAProblem-level
        route to the Level-0 headline rate: nothing but the 8-byte sum crosses PCIe."""
        from . import device as D
        h = C.c_void_p(outputs.handle.value) if outputs is not None else C.c_void_p()
        s = C.c_double()
        L.check(L.lib().c2b_problem_residual_jacobian_device(self._h, int(max_attempts), C.byref(h), C.byref(s)))
        if outputs is None:
            outputs = D.JacobianOutputs(self.num_observations(), self._device, max_attempts, _handle=h)
        return outputs, s.value

    def _stats(self):
        s = np.empty(L.STATS_DOUBLES)
        L.check(L.lib().c2b_problem_stats(self._h, _ptr(s)))
        return s

    def mean(self):          # src/baproblem.rs:282-289
        return self._stats()[0:3].copy()

    def std(self):           # src/baproblem.rs:292-304
        return self._stats()[3:6].copy()

    def extent(self):        # src/baproblem.rs:307-331
        s = self._stats()
        return s[6:9].copy(), s[9:12].copy()

    def dimensions(self):    # src/baproblem.rs:334-337
        return self._stats()[12:15].copy()

    def drift_origin(self):
        """element of centers ++ points closest to the world origin (src/noise.rs:75-87)"""
        s = self._stats()
        return s[15:18].copy(), int(s[18])

    def generate_world_points(self, triangles, num_points, max_dist, seed=0):
        """generate_world_points_uniform (src/generate.rs:356-420) for this problem's cameras, on the device: its points
        are replaced by `num_points` points sampled on the mesh by area, each within max_dist of some camera (the same
        points as generate.generate_world_points_uniform with the same seed).  The problem must hold no observations."""
        tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
        n = C.c_int64()
        L.check(L.lib().c2b_problem_generate_world_points(self._h, _ptr(tri), len(tri), int(num_points), float(max_dist),
                                                          int(seed), C.byref(n)))
        return n.value

    def visibility_graph(self, max_dist, triangles=None, fetch=True, prebuilt_hierarchy=False, dense=False):
        """generate::visibility_graph (src/generate.rs:424-481): for every camera the points within max_dist of its centre
        (`locate_within_distance`) that pass the predicate -- through the cell list of c2b_problem_visibility_within_distance,
        or, with dense=True, by the brute-force sweep of every camera against every point (the same lists: the sweep is
        what BASELINE's 1e10-pair configuration times); with `triangles` ([n,9] f32 mesh) the survivors also pass the
        occlusion rays of :455-476 (a hierarchy on the device in place of Embree).  Returns the CSR graph (row_ptr u64,
        pt_idx u64, uv) with points in ascending order per camera, like the reference's push order."""
        n_cam = self.num_cameras()
        row_ptr = np.zeros(n_cam + 1, dtype=np.uint64)
        if dense:
            L.check(L.lib().c2b_problem_visibility_dense(self._h, float(max_dist), _ptr(row_ptr)))
        else:
            L.check(L.lib().c2b_problem_visibility_within_distance(self._h, float(max_dist), 0, 0.0, 0.0, _ptr(row_ptr)))
        if triangles is not None:
            tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
            if prebuilt_hierarchy and len(tri):
                # the caller-built form (c2b_bvh_build needs only the triangles: cli/main.cpp builds it on a second
                # thread while cameras are placed and points sampled)
                h = C.c_void_p()
                L.check(L.lib().c2b_bvh_build(_ptr(tri), len(tri), C.byref(h)))
                try:
                    L.check(L.lib().c2b_problem_visibility_dense_occlude_bvh(self._h, h, _ptr(row_ptr)))
                finally:
                    L.lib().c2b_bvh_free(h)
            else:
                L.check(L.lib().c2b_problem_visibility_dense_occlude(self._h, _ptr(tri), len(tri), _ptr(row_ptr)))
        if not fetch:                     # lists stay on the device for adopt_visibility()
            return row_ptr
        n = int(row_ptr[-1])
        pt_idx = np.empty(n, dtype=np.uint64)
        uv = np.empty((n, 2))
        L.check(L.lib().c2b_problem_visibility_dense_fetch(self._h, _ptr(pt_idx), _ptr(uv)))
        return row_ptr, pt_idx, uv

    def _cameras_from_position_direction(self, pos, dirs):
        """Camera::from_position_direction (src/baproblem.rs:153-159), batched on this problem's device."""
        pos, dirs = _f64(pos, (-1, 3)), _f64(dirs, (-1, 9))
        out = np.empty((len(pos), 15))
        L.check(L.lib().c2b_problem_from_position_direction(self._h, len(pos), _ptr(pos), _ptr(dirs), _ptr(out)))
        return out

    def _camera_centers(self):
        """Camera::center (src/baproblem.rs:161-163) of every camera [n,3]"""
        out = np.empty((self.num_cameras(), 3))
        L.check(L.lib().c2b_problem_centers(self._h, _ptr(out)))
        return out

    # ---- host-side rows: cull and file IO (C++ host code behind the same ABI) -----------------------
    def cull(self, faithful=True):
        """BAProblem::cull (src/baproblem.rs:538-549): largest connected component + cameras with > 3
        observations / points with > 1, to a fixed point -- on the device, in place (the reference consumes self and
        returns the culled problem; this returns self)."""
        L.check(L.lib().c2b_problem_cull(self._h, int(bool(faithful))))
        return self._refresh_graph()

    def largest_connected_component(self, faithful=True):
        """BAProblem::largest_connected_component (src/baproblem.rs:456-534), one application, on the device"""
        L.check(L.lib().c2b_problem_largest_connected_component(self._h, int(bool(faithful))))
        return self._refresh_graph()

    def remove_singletons(self):
        """BAProblem::remove_singletons (src/baproblem.rs:426-453), one application, on the device"""
        L.check(L.lib().c2b_problem_remove_singletons(self._h))
        return self._refresh_graph()

    def subset(self, ci, pi):
        """BAProblem::subset (src/baproblem.rs:394-423): cameras ci and points pi in the given order; observations of
        dropped points disappear.  Host-side index shuffling; returns a NEW device problem."""
        ci = np.asarray(ci, dtype=np.int64).reshape(-1)
        pi = np.asarray(pi, dtype=np.int64).reshape(-1)
        n_cam, n_pts = self.num_cameras(), self.num_points()
        if (len(ci) and (ci.min() < 0 or ci.max() >= n_cam)) or (len(pi) and (pi.min() < 0 or pi.max() >= n_pts)):
            raise L.City2baError(L.ERR_INDEX_OUT_OF_RANGE, "subset: index out of range")
        cams, pts, uv = self.cameras(), self.points(), self.observations()
        new_of = np.full(n_pts, -1, dtype=np.int64)
        new_of[pi] = np.arange(len(pi))                       # HashMap::from_iter: a repeated point keeps its last slot
        rows, cols, vals = [0], [], []
        for c in ci:
            a, b = int(self._row_ptr[c]), int(self._row_ptr[c + 1])
            p_new = new_of[self._pt_idx[a:b].astype(np.int64)]
            k = p_new >= 0
            cols.append(p_new[k])
            vals.append(uv[a:b][k])
            rows.append(rows[-1] + int(k.sum()))
        cols = np.concatenate(cols).astype(np.uint64) if cols else np.zeros(0, np.uint64)
        vals = np.concatenate(vals) if vals else np.zeros((0, 2))
        return BAProblem.from_visibility(cams[ci], pts[pi], np.array(rows, dtype=np.uint64), cols, vals.reshape(-1, 2),
                                         self._device)

    def _refresh_graph(self):
        """host mirrors of the graph (row_ptr, pt_idx) after a device-side change"""
        n_cam, _, n_obs = self._sizes()
        row_ptr = np.zeros(n_cam + 1, dtype=np.uint64)
        pt_idx = np.zeros(n_obs, dtype=np.uint64)
        L.check(L.lib().c2b_problem_download_graph(self._h, _ptr(row_ptr), _ptr(pt_idx)))
        self._row_ptr, self._pt_idx = row_ptr, pt_idx
        return self

    def adopt_visibility(self, mirror=True):
        """BAProblem::from_visibility (src/baproblem.rs:360-376) on the device: the pending result of
        visibility_pairs_compact(fetch=False) / visibility_graph(fetch=False) becomes this problem's vis_graph.
        mirror = False leaves the host copies of the graph (row_ptr(), pt_idx()) stale: for callers that stay on the
        device (export_device) and do not want 12 bytes per observation to cross PCIe"""
        L.check(L.lib().c2b_problem_adopt_visibility(self._h))
        return self._refresh_graph() if mirror else self

    def export_device(self, cam_lo=0, cam_hi=None):
        """c2b_problem_export_device: the resident problem -- or the shard of the camera range [cam_lo, cam_hi) -- as
        torch tensors on the problem's device, copied device to device (Level 1 -> Level 0 without PCIe):
        dict(cam15 [nc,15], pts4 [n_pts,4], row_ptr int64 [nc+1] rebased to 0, pt_idx int32 [n], uv [n,2], obs_lo, n_obs)."""
        import torch
        n_cam, n_pts, _ = self._sizes()
        cam_hi = n_cam if cam_hi is None else int(cam_hi)
        lo, n = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().c2b_problem_export_device(self._h, int(cam_lo), cam_hi, None, None, None, None, None, C.byref(lo), C.byref(n)))
        dev = torch.device("cuda", self._device)
        nc = cam_hi - int(cam_lo)
        out = dict(cam15=torch.empty((nc, 15), dtype=torch.float64, device=dev),
                   pts4=torch.empty((n_pts, 4), dtype=torch.float64, device=dev),
                   row_ptr=torch.empty(nc + 1, dtype=torch.int64, device=dev),
                   pt_idx=torch.empty(n.value, dtype=torch.int32, device=dev),
                   uv=torch.empty((n.value, 2), dtype=torch.float64, device=dev))
        torch.cuda.synchronize(dev)                          # the buffers exist before the problem's own stream writes them
        p = lambda t: C.c_void_p(t.data_ptr()) if t.numel() else None
        L.check(L.lib().c2b_problem_export_device(self._h, int(cam_lo), cam_hi, p(out["cam15"]), p(out["pts4"]), C.c_void_p(out["row_ptr"].data_ptr()),
                                                  p(out["pt_idx"]), p(out["uv"]), C.byref(lo), C.byref(n)))
        out["obs_lo"], out["n_obs"] = int(lo.value), int(n.value)
        return out

    def cull_host(self, faithful=True):
        """the same through the host implementation (c2b_cull on downloaded arrays); returns a NEW device problem"""
        cams = self.cameras()
        pts = self.points()
        uv = self.observations()
        row_ptr = self._row_ptr.copy()
        pt_idx = self._pt_idx.copy()
        cams, pts, row_ptr, pt_idx, uv = cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful)
        return BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv, self._device)

    @classmethod
    def from_file(cls, path, device=0, fmt=None):
        """BAProblem::from_file (src/baproblem.rs:697-706) straight into the resident problem (c2b_problem_read): a .bbal is
        streamed to the device and decoded there (byte order, index / uv split, range checks, from_vec); .bal text is
        tokenised and parsed on the device too (every number correctly rounded), except files smaller than the option
        text_device_min_bytes (default 64 KiB), files the device parser declines (NaN, more than 19 digits, glued
        numbers) and everything under the option host_text, which go through the host parser and an upload -- the same
        resident state either way.  fmt None = by extension, "text", "binary"."""
        self = cls(device)
        L.check(L.lib().c2b_problem_read(self._h, str(path).encode(), _FORMATS[fmt]))
        return self._refresh_graph()

    @classmethod
    def from_file_text(cls, path, device=0):        # src/baproblem.rs:580-630
        return cls.from_file(path, device, "text")

    @classmethod
    def from_file_binary(cls, path, device=0):      # src/baproblem.rs:632-695
        return cls.from_file(path, device, "binary")

    def write(self, path, fmt=None):
        """BAProblem::write (src/baproblem.rs:768-785) straight from the resident problem: a .bbal image is assembled on
        the device (to_vec, counts, byte order) and only its bytes cross PCIe; so is .bal text (shortest round-trip
        decimals formatted on the device); the option host_text selects the host formatter over a download: same bytes"""
        L.check(L.lib().c2b_problem_write(self._h, str(path).encode(), _FORMATS[fmt]))

    def write_text(self, path):                     # src/baproblem.rs:709-733
        self.write(path, "text")

    def write_binary(self, path):                   # src/baproblem.rs:736-764
        self.write(path, "binary")

    def visibility_pairs(self, cam_idx, pt_idx, max_dist):
        """predicate of the generator loops (src/synthetic.rs:285-291; src/generate.rs:448-454)"""
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.uint32)
        pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint32)
        n = len(cam_idx)
        if len(pt_idx) != n:
            raise L.City2baError(L.ERR_INVALID_ARGUMENT, "cam_idx and pt_idx differ in length")
        uv = np.empty((n, 2))
        keep = np.empty(n, dtype=np.uint8)
        L.check(L.lib().c2b_problem_visibility_pairs(self._h, n, _ptr(cam_idx), _ptr(pt_idx), float(max_dist),
                                                     _ptr(uv), _ptr(keep)))
        return uv, keep

    def visibility_within_distance(self, max_dist, occlusion=False, block_length=20.0, block_inset=1.0, fetch=True):
        """the synthetic generators' whole visibility loop on the device (src/synthetic.rs:268-297, :353-378): candidates
        within max_dist of each camera centre (a cell list in place of rstar's locate_within_distance), hits_building when
        `occlusion`, the predicate, kept lists per camera in ascending point index.  Returns (row_ptr, pt_idx, uv), or
        the row pointer only with fetch=False (the lists stay on the device for adopt_visibility)."""
        row_ptr = np.zeros(self.num_cameras() + 1, dtype=np.uint64)
        L.check(L.lib().c2b_problem_visibility_within_distance(self._h, float(max_dist), int(bool(occlusion)), float(block_length),
                                                               float(block_inset), _ptr(row_ptr)))
        if not fetch:
            return row_ptr
        n = int(row_ptr[-1])
        kept = np.empty(n, dtype=np.uint64)
        uv = np.empty((n, 2))
        L.check(L.lib().c2b_problem_visibility_dense_fetch(self._h, _ptr(kept), _ptr(uv)))
        return row_ptr, kept, uv

    def visibility_pairs_compact(self, cam_idx, pt_idx, max_dist, fetch=True):
        """the same predicate with the kept pairs compacted on the device (cam_idx non-decreasing): returns the CSR
        graph (row_ptr u64, pt_idx u64, uv) of the survivors in candidate order; fetch=False leaves the lists on the
        device (for adopt_visibility) and returns the row pointer only"""
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.uint32)
        pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint32)
        if len(pt_idx) != len(cam_idx):
            raise L.City2baError(L.ERR_INVALID_ARGUMENT, "cam_idx and pt_idx differ in length")
        row_ptr = np.zeros(self.num_cameras() + 1, dtype=np.uint64)
        L.check(L.lib().c2b_problem_visibility_pairs_compact(self._h, len(cam_idx), _ptr(cam_idx), _ptr(pt_idx),
                                                             float(max_dist), _ptr(row_ptr)))
        if not fetch:
            return row_ptr
        n = int(row_ptr[-1])
        kept = np.empty(n, dtype=np.uint64)
        uv = np.empty((n, 2))
        L.check(L.lib().c2b_problem_visibility_dense_fetch(self._h, _ptr(kept), _ptr(uv)))
        return row_ptr, kept, uv


class _PinnedBlock:
    """owner of one c2b_host_alloc block; arrays made from it keep it alive through their .base chain"""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        L.check(L.lib().c2b_host_alloc(C.byref(self.ptr), self.nbytes))
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr.value or 0, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr:
                L.lib().c2b_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """np.empty in page-locked host memory (c2b_host_alloc): the destination that takes device results at link speed"""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    if n == 0:
        return np.empty(shape, dtype=dtype)
    return np.asarray(_PinnedBlock(n)).view(dtype).reshape(shape)


# ---- host-array helpers over the C ABI (no GPU involved) ------------------------------------------------
def cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful=True, step="cull"):
    """c2b_cull on numpy arrays; cams is [n, stride] (cam15 or bal9 rows).  Returns the culled copies.
    step = "cull" | "lcc" (largest_connected_component once) | "singletons" (remove_singletons once)"""
    cams = np.array(cams, dtype=np.float64, order="C", copy=True)
    cams = cams.reshape(len(cams), -1) if cams.size else cams.reshape(0, cams.shape[-1] if cams.ndim == 2 else 15)
    pts = np.array(pts, dtype=np.float64, order="C", copy=True).reshape(-1, 3)
    row_ptr = np.array(row_ptr, dtype=np.uint64, order="C", copy=True)
    pt_idx = np.array(pt_idx, dtype=np.uint64, order="C", copy=True)
    uv = np.array(uv, dtype=np.float64, order="C", copy=True).reshape(-1, 2)
    if len(row_ptr) != len(cams) + 1:
        raise L.City2baError(L.ERR_INVALID_ARGUMENT, "row_ptr must have n_cameras + 1 entries")
    stride = cams.shape[1] if cams.ndim == 2 and len(cams) else 0
    nc, npts = C.c_int64(len(cams)), C.c_int64(len(pts))
    args = (C.byref(nc), _ptr(cams), int(stride), C.byref(npts), _ptr(pts), _ptr(row_ptr), _ptr(pt_idx), _ptr(uv))
    if step == "cull":
        L.check(L.lib().c2b_cull(*args, int(bool(faithful))))
    elif step == "lcc":
        L.check(L.lib().c2b_largest_connected_component(*args, int(bool(faithful))))
    elif step == "singletons":
        L.check(L.lib().c2b_remove_singletons(*args))
    else:
        raise L.City2baError(L.ERR_INVALID_ARGUMENT, "step must be 'cull', 'lcc' or 'singletons'")
    n_obs = int(row_ptr[nc.value])
    return (cams[:nc.value].copy(), pts[:npts.value].copy(), row_ptr[:nc.value + 1].copy(), pt_idx[:n_obs].copy(),
            uv[:n_obs].copy())


_FORMATS = {None: -1, "text": 0, "binary": 1}


def read_bal(path, fmt=None):
    """(bal9 [n,9], pts [m,3], row_ptr, pt_idx, uv) from a .bal / .bbal file; fmt None = by extension (from_file),
    "text" = from_file_text, "binary" = from_file_binary (src/baproblem.rs:580, :632, :697)."""
    h = C.c_void_p()
    L.check(L.lib().c2b_bal_read_as(str(path).encode(), _FORMATS[fmt], C.byref(h)))
    try:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().c2b_bal_sizes(h, C.byref(a), C.byref(b), C.byref(c)))
        bal9, pts = np.empty((a.value, 9)), np.empty((b.value, 3))
        row_ptr = np.empty(a.value + 1, dtype=np.uint64)
        pt_idx = np.empty(c.value, dtype=np.uint64)
        uv = np.empty((c.value, 2))
        L.check(L.lib().c2b_bal_copy(h, _ptr(bal9), _ptr(pts), _ptr(row_ptr), _ptr(pt_idx), _ptr(uv)))
    finally:
        L.lib().c2b_bal_close(h)
    return bal9, pts, row_ptr, pt_idx, uv


def format_f64(values):
    """the texts write_text gives these doubles (Rust's `{}`: shortest round-trip digits, no exponent) -- c2b_format_f64"""
    v = np.ascontiguousarray(values, dtype=np.float64).ravel()
    buf = C.create_string_buffer(max(1, 331 * len(v)))
    n = C.c_int64()
    L.check(L.lib().c2b_format_f64(len(v), _ptr(v), buf, len(buf), C.byref(n)))
    return buf.raw[:n.value].decode("ascii").split("\n")[:-1]


def parse_f64(tokens):
    """(values, status) of decimal tokens as from_file_text reads them -- c2b_parse_f64: status 0 parsed (correctly
    rounded), 1 a spelling left to strtod, 2 too many digits / an undecided rounding"""
    text = " ".join(tokens).encode("ascii")
    vals, st = np.empty(len(tokens)), np.empty(len(tokens), dtype=np.int32)
    L.check(L.lib().c2b_parse_f64(text, len(text), len(tokens), _ptr(vals), _ptr(st)))
    return vals, st


def write_bal(path, bal9, pts, row_ptr, pt_idx, uv, fmt=None):
    """write (by extension), write_text or write_binary (src/baproblem.rs:709, :736, :768)"""
    bal9 = _f64(bal9, (-1, 9))
    pts = _f64(pts, (-1, 3))
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint64)
    uv = _f64(uv, (-1, 2))
    L.check(L.lib().c2b_bal_write_as(str(path).encode(), _FORMATS[fmt], len(bal9), _ptr(bal9), len(pts), _ptr(pts),
                                     _ptr(row_ptr), _ptr(pt_idx), _ptr(uv)))
