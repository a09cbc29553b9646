"""Level-0 launchers over torch CUDA(=HIP) tensors.

torch is plumbing here: it owns device memory, the current stream and torch.distributed (RCCL).
Every function forwards raw device pointers + the current stream handle to the C ABI
(include/city2ba_hip.h); nothing is computed by torch ops."""
import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _pn(t, n_rows):
    """pointer of an optional [n_rows][4] f64 table (None -> NULL)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64 and tuple(t.shape) == (n_rows, 4), "centers"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, name


_live_workspaces = None     # weak references to every workspace made here (tests check their counters after each test)


def workspace(n_obs, device):
    """Scratch of the in-kernel folds for launches of up to n_obs observations, initialised on the current stream
    (c2b_workspace_init: arrival counters zeroed, magic written).  One workspace serves one launch at a time."""
    global _live_workspaces
    import weakref
    nbytes = L.lib().c2b_workspace_bytes(int(n_obs))
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)      # torch allocations are 512-B aligned
    with torch.cuda.device(ws.device):
        L.check(L.lib().c2b_workspace_init(_p(ws), _stream()))
    if _live_workspaces is None:
        _live_workspaces = weakref.WeakSet()
    _live_workspaces.add(ws)
    return ws


def workspace_selfcheck(ws):
    """non-zero arrival counters left in ws (0 = clean; -1 = never initialised); synchronises the current stream"""
    n = C.c_int64(-2)
    with torch.cuda.device(ws.device):
        L.check(L.lib().c2b_workspace_selfcheck(_p(ws), _stream(), C.byref(n)))
    return n.value


def live_workspaces():
    return list(_live_workspaces) if _live_workspaces is not None else []


def cameras_from_bal(bal9):
    _chk(bal9, torch.float64, "bal9")
    n = bal9.shape[0]
    cam15 = torch.empty((n, 15), dtype=torch.float64, device=bal9.device)
    L.check(L.lib().c2b_cameras_from_bal(_p(bal9), n, _p(cam15), _stream()))
    return cam15


def cameras_to_bal(cam15):
    _chk(cam15, torch.float64, "cam15")
    n = cam15.shape[0]
    bal9 = torch.empty((n, 9), dtype=torch.float64, device=cam15.device)
    L.check(L.lib().c2b_cameras_to_bal(_p(cam15), n, _p(bal9), _stream()))
    return bal9


def centers_table(n_cam, device):
    """cen4[n_cam][4]: the compact table of camera centres cameras_prepare_* fill next to camblk and the statistics
    read (32 bytes per camera instead of a 128-byte line of the 256-byte record)"""
    return torch.empty((n_cam, 4), dtype=torch.float64, device=device)


class CameraTable:
    """A camblk table for n_cam cameras, OPAQUE on purpose: the table is blocked in groups of 8 cameras (a group's eight
    light lines, then its J_l tails, centres and pad -- include/city2ba_hip.h) and holds whole groups, so its rows are not records:
    slicing, indexing, .to(), .contiguous() or .clone() of a plain [n, 32] view would silently give truncated or scrambled data.
    What exists instead: data_ptr() for the launchers, clone() (whole groups, storage of its own), records() (a COPY of the logical
    [n_cam][32] records, for looking at), flat (the storage: capacity_doubles() doubles, 1-D) and shape == (n_cam, 32) so that
    callers can read the camera count the way they did."""
    __slots__ = ("flat", "n_cam")

    def __init__(self, n_cam, device, flat=None):
        self.n_cam = int(n_cam)
        need = ((self.n_cam + 7) // 8 * 8) * L.CAMBLK_DOUBLES
        if flat is None:
            flat = torch.empty(need, dtype=torch.float64, device=device)
        if not (flat.is_cuda and flat.dtype == torch.float64 and flat.dim() == 1 and flat.is_contiguous() and flat.numel() >= need
                and flat.data_ptr() % 256 == 0):
            raise ValueError("a camera table for %d cameras needs a contiguous, 256-byte aligned 1-D float64 CUDA tensor of >= %d doubles "
                             "(whole groups of 8 cameras)" % (self.n_cam, need))
        self.flat = flat

    shape = property(lambda self: (self.n_cam, L.CAMBLK_DOUBLES))
    device = property(lambda self: self.flat.device)
    dtype = property(lambda self: self.flat.dtype)
    is_cuda = property(lambda self: True)

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.flat.data_ptr()

    def capacity_doubles(self):
        return self.flat.numel()

    def clone(self):
        return CameraTable(self.n_cam, self.flat.device, self.flat.clone())

    def prefix(self, n_cam):
        """the same storage seen as a table of its first n_cam cameras (a camera's place depends on its index only, so a PREFIX
        is a table; any other range is not -- a shard prepares a table of its own)"""
        if not 0 <= int(n_cam) <= self.n_cam:
            raise ValueError("prefix of %d cameras of a table of %d" % (int(n_cam), self.n_cam))
        return CameraTable(n_cam, self.flat.device, self.flat)

    def records(self):
        g = (self.n_cam + 7) // 8
        flat = self.flat[: g * 256].view(g, 256)
        parts = (flat[:, :128].reshape(g, 8, 16), flat[:, 128:192].reshape(g, 8, 8), flat[:, 192:224].reshape(g, 8, 4), flat[:, 224:].reshape(g, 8, 4))
        return torch.cat(parts, dim=-1).reshape(g * 8, L.CAMBLK_DOUBLES)[: self.n_cam].clone()

    def __getitem__(self, _):
        raise TypeError("a CameraTable is not a tensor of records (it is blocked in groups of 8 cameras): use .records() to look at it")

    def __len__(self):
        return self.n_cam


def camblk_table(n_cam, device):
    """an uninitialised camera table for n_cam cameras (CameraTable: storage for whole groups of 8)"""
    return CameraTable(n_cam, device)


def camblk_records(camblk):
    """the table's logical records, [n_cam][CAMBLK_DOUBLES] (R 9, t 3, intrinsics 3, J_l 9, centre 3, pad), as a COPY taken out of
    the blocked layout -- for looking at; the kernels take the table itself"""
    return _as_table(camblk).records()


def camblk_clone(camblk):
    """a copy of a prepared table in storage of its own"""
    return _as_table(camblk).clone()


def _as_table(x, n_cam=None):
    """`out=` arguments and hand-made tables: a CameraTable passes; a tensor is accepted only if its STORAGE from its first
    element on holds whole groups for its row count (what the kernel will write), and is wrapped -- never written out of bounds"""
    if isinstance(x, CameraTable):
        if n_cam is not None and x.n_cam != int(n_cam):
            raise ValueError("the camera table was made for %d cameras, not %d" % (x.n_cam, int(n_cam)))
        return x
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()):
        raise ValueError("camblk: a CameraTable (device.camblk_table) or a contiguous float64 CUDA tensor")
    n = int(n_cam) if n_cam is not None else (x.shape[0] if x.dim() == 2 else x.numel() // L.CAMBLK_DOUBLES)
    need = ((n + 7) // 8 * 8) * L.CAMBLK_DOUBLES
    have = (x.untyped_storage().nbytes() - 8 * x.storage_offset()) // 8
    if have < need:
        raise ValueError("camblk for %d cameras needs %d doubles of storage (whole groups of 8 cameras), this tensor has %d: "
                         "allocate with device.camblk_table()" % (n, need, have))
    return CameraTable(n, x.device, torch.as_strided(x, (need,), (1,)))


def cameras_prepare_state(cam15, out=None, centers=None):
    """the camera table from the in-memory cameras; `centers` (centers_table) also receives every camera's centre"""
    _chk(cam15, torch.float64, "cam15")
    n = cam15.shape[0]
    blk = _as_table(out, n) if out is not None else CameraTable(n, cam15.device)
    L.check(L.lib().c2b_camblk_from_state(_p(cam15), n, _p(blk), blk.capacity_doubles(), _pn(centers, n), _stream()))
    return blk


def cameras_prepare_bal(bal9, out=None, centers=None):
    _chk(bal9, torch.float64, "bal9")
    n = bal9.shape[0]
    blk = _as_table(out, n) if out is not None else CameraTable(n, bal9.device)
    L.check(L.lib().c2b_camblk_from_bal(_p(bal9), n, _p(blk), blk.capacity_doubles(), _pn(centers, n), _stream()))
    return blk


def points_pad(pts3):
    _chk(pts3, torch.float64, "pts3")
    n = pts3.shape[0]
    pts4 = torch.empty((n, 4), dtype=torch.float64, device=pts3.device)
    L.check(L.lib().c2b_points_pad(_p(pts3), n, _p(pts4), _stream()))
    return pts4


def expand_rows(row_ptr, n_obs, obs_base=0):
    _chk(row_ptr, torch.int64, "row_ptr")       # u64 values < 2^63 share the int64 bit pattern
    n_cam = row_ptr.shape[0] - 1
    cam_idx = torch.empty(n_obs, dtype=torch.int32, device=row_ptr.device)
    L.check(L.lib().c2b_expand_rows(_p(row_ptr), n_cam, int(obs_base), int(n_obs), _p(cam_idx), _stream()))
    return cam_idx


def project(camblk, pts4, cam_idx, pt_idx, uv_out):
    L.check(L.lib().c2b_project(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), cam_idx.shape[0], _p(uv_out),
                                _stream()))
    return uv_out


def reprojection_error_sum(camblk, pts4, cam_idx, pt_idx, uv, norm, ws, out_sum):
    L.check(L.lib().c2b_reprojection_error_sum(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), _p(uv),
                                               cam_idx.shape[0], float(norm), _p(ws), _p(out_sum), _stream()))
    return out_sum


class Rows:
    """A camera-major observation list addressed through its row structure -- the reference's one list per camera
    (vis_graph, src/baproblem.rs:256-260) -- instead of a 4-byte camera index per observation: row_ptr [n_cam + 1]
    (int64 tensor holding the u64 values) plus the tile records c2b_rows_pack derives from it (16 bytes per 64
    observations).  The *_rows launchers give the same bits as their cam_idx forms and read ~3.7 bytes per observation
    less."""

    def __init__(self, row_ptr, n_obs=None):
        _chk(row_ptr, torch.int64, "row_ptr")
        self.row_ptr = row_ptr
        self.n_cam = row_ptr.shape[0] - 1
        self.n_obs = int(row_ptr[-1].item()) if n_obs is None else int(n_obs)
        self.tiles = torch.empty((L.lib().c2b_rows_tiles_bytes(self.n_obs) // 16, 4), dtype=torch.int32, device=row_ptr.device)
        L.check(L.lib().c2b_rows_pack(_p(row_ptr), self.n_cam, self.n_obs, _p(self.tiles), _stream()))


def project_rows(camblk, pts4, rows, pt_idx, uv_out):
    L.check(L.lib().c2b_project_rows(_p(camblk), _p(pts4), _p(rows.row_ptr), rows.n_cam, _p(rows.tiles), _p(pt_idx),
                                     rows.n_obs, _p(uv_out), _stream()))
    return uv_out


def reprojection_error_sum_rows(camblk, pts4, rows, pt_idx, uv, norm, ws, out_sum):
    L.check(L.lib().c2b_reprojection_error_sum_rows(_p(camblk), _p(pts4), _p(rows.row_ptr), rows.n_cam, _p(rows.tiles),
                                                    _p(pt_idx), _p(uv), rows.n_obs, float(norm), _p(ws), _p(out_sum),
                                                    _stream()))
    return out_sum


def reprojection_error_sums2_rows(camblk, pts4, rows, pt_idx, uv, ws, out_sums):
    """out_sums[0] = the L1 sum, out_sums[1] = the L2 sum, ONE pass (what run_noise evaluates back to back,
    src/bin/city2ba.rs:283-287, 350-354); each bit-identical to reprojection_error_sum_rows with that norm"""
    L.check(L.lib().c2b_reprojection_error_sums2_rows(_p(camblk), _p(pts4), _p(rows.row_ptr), rows.n_cam, _p(rows.tiles),
                                                      _p(pt_idx), _p(uv), rows.n_obs, _p(ws), _p(out_sums), _stream()))
    return out_sums


def add_noise_observations_error_sums2_rows(camblk, pts4, rows, pt_idx, uv, obs_base, observations_std, seed, ws, out_sums):
    """add_noise's observation pass (src/noise.rs:152-170) fused with the L1 / L2 error sums of the perturbed
    observations (src/bin/city2ba.rs:350-354): uv is perturbed in place exactly as add_noise_observations would"""
    L.check(L.lib().c2b_add_noise_observations_error_sums2_rows(_p(camblk), _p(pts4), _p(rows.row_ptr), rows.n_cam,
                                                                _p(rows.tiles), _p(pt_idx), _p(uv), rows.n_obs, int(obs_base),
                                                                float(observations_std), int(seed), _p(ws), _p(out_sums),
                                                                _stream()))
    return out_sums


def visibility_rows(camblk, pts4, rows, pt_idx, max_dist, uv_out, keep):
    L.check(L.lib().c2b_visibility_rows(_p(camblk), _p(pts4), _p(rows.row_ptr), rows.n_cam, _p(rows.tiles), _p(pt_idx),
                                        rows.n_obs, float(max_dist), _p(uv_out), _p(keep), _stream()))


def visibility_rows_bits(camblk, pts4, rows, pt_idx, max_dist, uv_out, keep_bits):
    """visibility_rows with the mask as one 64-bit word per 64 pairs (keep_bits: int64 tensor of ceil(n / 64) words)"""
    L.check(L.lib().c2b_visibility_rows_bits(_p(camblk), _p(pts4), _p(rows.row_ptr), rows.n_cam, _p(rows.tiles), _p(pt_idx),
                                             rows.n_obs, float(max_dist), _p(uv_out), _p(keep_bits), _stream()))


def residual_jacobian_rows(camblk, pts4, rows, pt_idx, uv, r, Jc, Jp, norm=2.0, ws=None, out_sum=None, obs_base=0,
                           n_obs=None):
    """residual + Jacobian (+ sum |r|^norm when ws is given: into out_sum, or into ws for error_sum_finish) of the
    observations [obs_base, obs_base + n_obs) of the list `rows` describes; pt_idx / uv / r / Jc / Jp are that slice."""
    n = rows.n_obs - int(obs_base) if n_obs is None else int(n_obs)
    tiles = rows.tiles[int(obs_base) // 64:]
    L.check(L.lib().c2b_residual_jacobian_rows(_p(camblk), _p(pts4), pts4.shape[0], _p(rows.row_ptr), rows.n_cam, _p(tiles),
                                               int(obs_base), _p(pt_idx), _p(uv), n, _p(r), _p(Jc), _p(Jp),
                                               float(norm), _p(ws), _p(out_sum), _stream()))
    return out_sum


def jacobian_stream_policy(n_obs, n_cam, n_pts):
    """0 / 2 / 3: which once-read streams of a residual_jacobian_rows launch of this size bypass the caches"""
    return int(L.lib().c2b_jacobian_stream_policy(int(n_obs), int(n_cam), int(n_pts)))


def jacobian_tiles_per_wave(n_obs):
    return int(L.lib().c2b_jacobian_tiles_per_wave(int(n_obs)))


def jacobian_launch_shape(n_obs, store_GBs=0.0):
    """(waves per workgroup, tiles of 64 observations per wave) of a residual_jacobian_rows launch of this size into an
    output set that takes streaming stores at store_GBs (0 = unknown)"""
    w, t = C.c_int(0), C.c_int(0)
    L.check(L.lib().c2b_jacobian_launch_shape(int(n_obs), float(store_GBs), C.byref(w), C.byref(t)))
    return w.value, t.value


def residual_jacobian_rows_placed(camblk, pts4, rows, pt_idx, uv, outputs, norm=2.0, ws=None, out_sum=None):
    """residual_jacobian_rows over the whole list into a placed output set (JacobianOutputs): the launch takes the
    workgroup shape that suits the store rate measured for that set"""
    L.check(L.lib().c2b_residual_jacobian_rows_placed(_p(camblk), _p(pts4), pts4.shape[0], _p(rows.row_ptr), rows.n_cam, _p(rows.tiles),
                                                      _p(pt_idx), _p(uv), rows.n_obs, outputs.handle, float(norm), _p(ws), _p(out_sum),
                                                      _stream()))
    return out_sum


def residual_jacobian(camblk, pts4, cam_idx, pt_idx, uv, r, Jc, Jp, norm=2.0, ws=None):
    """ws != None -> the same launch also folds sum |r|^norm into ws (see error_sum_finish)."""
    L.check(L.lib().c2b_residual_jacobian(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), _p(uv), cam_idx.shape[0],
                                          _p(r), _p(Jc), _p(Jp), float(norm), _p(ws), _stream()))


def residual_jacobian_sum(camblk, pts4, cam_idx, pt_idx, uv, r, Jc, Jp, norm, ws, out_sum):
    """residual + Jacobian + sum |r|^norm -> out_sum[0], ONE launch."""
    L.check(L.lib().c2b_residual_jacobian_sum(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), _p(uv), cam_idx.shape[0],
                                              _p(r), _p(Jc), _p(Jp), float(norm), _p(ws), _p(out_sum), _stream()))
    return out_sum


def calib_store_pattern(r, Jc, Jp):
    """Calibration: the residual+Jacobian kernel's store geometry alone (fills r, Jc, Jp with a pattern)."""
    L.check(L.lib().c2b_calib_store_pattern(r.shape[0], _p(r), _p(Jc), _p(Jp), _stream()))


def calib_copy(src, dst):
    """Calibration: 16-bytes-per-lane streaming copy of src into dst (same byte size, multiple of 16)."""
    nbytes = src.numel() * src.element_size()
    assert nbytes == dst.numel() * dst.element_size()
    L.check(L.lib().c2b_calib_copy(_p(src), _p(dst), nbytes, _stream()))


class _DeviceArray:
    """memory owned by a C-ABI handle, exposed to torch through __cuda_array_interface__ (zero copy; the tensor made
    from it keeps this object, and through it the handle, alive)"""

    def __init__(self, ptr, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2,
                                         "strides": None}
        self._owner = owner


class _OutputsHandle:
    """owns one c2b_jacobian_outputs; freed when the last tensor viewing it (and the JacobianOutputs) is gone.  Kept
    apart from JacobianOutputs so that tensor -> array -> handle holds no reference cycle through torch."""

    def __init__(self):
        self.h = C.c_void_p()

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and L is not None:                              # (at interpreter shutdown the module globals may be gone already)
            L.lib().c2b_jacobian_outputs_free(h)


class JacobianOutputs:
    """c2b_jacobian_outputs: r [n,2], Jc [n,18], Jp [n,6] in device allocations chosen for streaming-store speed
    (include/city2ba_hip.h).  .r / .Jc / .Jp are torch views of the handle's memory; .log = store
    GB/s of every attempt, .chosen = the attempt kept."""

    def __init__(self, n_obs, device, max_attempts=8, fast_store_GBs=7000.0, _handle=None):
        own = self._own = _OutputsHandle()
        dev = torch.device(device)
        if _handle is not None:                                  # a set the library allocated (c2b_problem_residual_jacobian_device)
            own.h = _handle
        else:
            with torch.cuda.device(dev):
                L.check(L.lib().c2b_jacobian_outputs_alloc(int(n_obs), int(max_attempts), float(fast_store_GBs), _stream(),
                                                           C.byref(own.h)))
        r, jc, jp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib().c2b_jacobian_outputs_pointers(own.h, C.byref(r), C.byref(jc), C.byref(jp)))
        rates = (C.c_double * 64)()
        na, ch = C.c_int(0), C.c_int(-1)
        L.check(L.lib().c2b_jacobian_outputs_log(own.h, rates, 64, C.byref(na), C.byref(ch)))
        self.log = [round(rates[i], 1) for i in range(na.value)] if n_obs >= 1_000_000 else []
        self.chosen = ch.value
        sr = C.c_double(0.0)
        L.check(L.lib().c2b_jacobian_outputs_store_rate(own.h, C.byref(sr)))
        self.store_GBs = sr.value                                # of the kept set (0: not measured)
        n = max(int(n_obs), 0)
        if n == 0:
            self.r, self.Jc, self.Jp = (torch.empty((0, k), dtype=torch.float64, device=dev) for k in (2, 18, 6))
        else:
            self.r = torch.as_tensor(_DeviceArray(r.value, (n, 2), own), device=dev)
            self.Jc = torch.as_tensor(_DeviceArray(jc.value, (n, 18), own), device=dev)
            self.Jp = torch.as_tensor(_DeviceArray(jp.value, (n, 6), own), device=dev)

    @property
    def handle(self):
        return self._own.h

    def set_store_rate(self, store_GBs):
        """replace the rate the library measured (residual_jacobian_rows_placed chooses its launch shape by it)"""
        L.check(L.lib().c2b_jacobian_outputs_set_store_rate(self._own.h, float(store_GBs)))
        self.store_GBs = max(float(store_GBs), 0.0)


def alloc_jacobian_outputs(n_obs, device, max_attempts=8, fast_store_GBs=7000.0):
    """((r, Jc, Jp), log): the output arrays of residual_jacobian*, placed for streaming stores by the library
    (c2b_jacobian_outputs_alloc -- every caller of the C ABI gets the same placement, not just this wrapper)."""
    out = JacobianOutputs(n_obs, device, max_attempts, fast_store_GBs)
    return (out.r, out.Jc, out.Jp), out.log


def error_sum_finish(ws, n_obs, out_sum):
    L.check(L.lib().c2b_error_sum_finish(_p(ws), int(n_obs), _p(out_sum), _stream()))
    return out_sum


def cameras_from_position_direction(pos3, dir9):
    _chk(pos3, torch.float64, "pos3")
    _chk(dir9, torch.float64, "dir9")
    n = pos3.shape[0]
    cam15 = torch.empty((n, 15), dtype=torch.float64, device=pos3.device)
    L.check(L.lib().c2b_cameras_from_position_direction(_p(pos3), _p(dir9), n, _p(cam15), _stream()))
    return cam15


def visibility_pairs(camblk, pts4, cam_idx, pt_idx, max_dist, uv_out, keep):
    L.check(L.lib().c2b_visibility_pairs(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), cam_idx.shape[0],
                                         float(max_dist), _p(uv_out), _p(keep), _stream()))


def occlusion_filter(camblk, pts4, cam_idx, pt_idx, tri9, keep):
    """occlusion rays of visibility_graph (src/generate.rs:455-476) against f32 triangles [n,9], brute force"""
    _chk(tri9, torch.float32, "tri9")
    L.check(L.lib().c2b_occlusion_filter(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), cam_idx.shape[0], _p(tri9),
                                         tri9.shape[0], _p(keep), _stream()))


class OcclusionBVH:
    """host-built hierarchy over f32 triangles [n,9] placed in device memory; filter() = occlusion_filter through it"""

    def __init__(self, tri9, device):
        import numpy as np
        tri = np.ascontiguousarray(tri9, dtype=np.float32).reshape(-1, 9)
        h = C.c_void_p()
        L.check(L.lib().c2b_bvh_build(tri.ctypes.data_as(C.c_void_p), len(tri), C.byref(h)))
        try:
            nn, ns, depth = C.c_int64(), C.c_int64(), C.c_int()
            L.check(L.lib().c2b_bvh_sizes(h, C.byref(nn), C.byref(ns), C.byref(depth)))
            self.n_nodes, self.n_slots, self.depth = nn.value, ns.value, depth.value
            nodes = np.empty((self.n_nodes, 16), dtype=np.float32)
            tris = np.empty((max(self.n_slots, 1), 12), dtype=np.float32)
            self.order = np.empty(self.n_slots, dtype=np.uint32)
            L.check(L.lib().c2b_bvh_copy(h, nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p),
                                         self.order.ctypes.data_as(C.c_void_p)))
        finally:
            L.lib().c2b_bvh_free(h)
        self.nodes_host = nodes
        self.nodes = torch.from_numpy(nodes).to(device)
        self.tris = torch.from_numpy(tris).to(device)

    def filter(self, camblk, pts4, cam_idx, pt_idx, keep):
        """keep[i] = 1 iff observation i's ray reaches its point unoccluded.  Raises if a traversal overflowed its
        stack (only possible for node arrays not made by c2b_bvh_build): the mask would be wrong."""
        overflow = torch.zeros(1, dtype=torch.int32, device=keep.device)
        L.check(L.lib().c2b_occlusion_filter_bvh(_p(camblk), _p(pts4), _p(cam_idx), _p(pt_idx), cam_idx.shape[0],
                                                 _p(self.nodes), self.n_nodes, _p(self.tris), self.n_slots, _p(keep),
                                                 _p(overflow), _stream()))
        if int(overflow.item()):
            raise L.City2baError(L.ERR_INVALID_ARGUMENT, "occlusion_filter_bvh: hierarchy deeper than the traversal stack")


def stats(camblk, pts4, ws, out=None, centers=None):
    """centers = the table cameras_prepare_* filled for these cameras (None: the centres are read out of camblk, one
    128-byte line per camera)"""
    out = out if out is not None else torch.empty(L.STATS_DOUBLES, dtype=torch.float64, device=camblk.device)
    L.check(L.lib().c2b_stats(_p(camblk), _pn(centers, camblk.shape[0]), camblk.shape[0], _p(pts4), pts4.shape[0], _p(ws),
                              _p(out), _stream()))
    return out


def stats_partial_pass1(camblk, cam_base, n_cam_global, pts4_slice, pt_base, n_entities_global, ws, part=None, centers=None):
    """this shard's share of the statistics (c2b_stats_partial_pass1); pts4_slice = the rows of the point table this
    rank reduces"""
    part = part if part is not None else torch.empty(L.STATS_DOUBLES, dtype=torch.float64, device=camblk.device)
    L.check(L.lib().c2b_stats_partial_pass1(_p(camblk), _pn(centers, camblk.shape[0]), camblk.shape[0], int(cam_base), int(n_cam_global), _p(pts4_slice),
                                            pts4_slice.shape[0], int(pt_base), int(n_entities_global), _p(ws), _p(part),
                                            _stream()))
    return part


def stats_partial_pass2(camblk, pts4_slice, mean3, ws, out=None, centers=None):
    out = out if out is not None else torch.empty(3, dtype=torch.float64, device=camblk.device)
    L.check(L.lib().c2b_stats_partial_pass2(_p(camblk), _pn(centers, camblk.shape[0]), camblk.shape[0], _p(pts4_slice), pts4_slice.shape[0], _p(mean3),
                                            _p(ws), _p(out), _stream()))
    return out


def add_drift_sharded(cam15, cam_base, pts4, stats_, strength, angle_strength, std, seed, direction=None):
    """direction None = add_drift_normalized (src/noise.rs:47-56), else add_drift with that direction"""
    d = (0.0, 0.0, 0.0) if direction is None else [float(x) for x in direction]
    L.check(L.lib().c2b_add_drift_sharded(_p(cam15), cam15.shape[0], int(cam_base), _p(pts4), pts4.shape[0], _p(stats_),
                                          1 if direction is None else 0, float(strength), float(angle_strength),
                                          float(std), d[0], d[1], d[2], int(seed), _stream()))


def add_noise_entities_sharded(cam15, cam_base, pts4, stats_, translation_std, rotation_std, point_std, seed):
    L.check(L.lib().c2b_add_noise_entities_sharded(_p(cam15), cam15.shape[0], int(cam_base), _p(pts4), pts4.shape[0],
                                                   _p(stats_), float(translation_std), float(rotation_std),
                                                   float(point_std), int(seed), _stream()))


def add_drift_normalized(cam15, pts4, stats_, strength, angle_strength, std, seed):
    L.check(L.lib().c2b_add_drift_normalized(_p(cam15), cam15.shape[0], _p(pts4), pts4.shape[0], _p(stats_),
                                             float(strength), float(angle_strength), float(std), int(seed), _stream()))


def add_noise_entities(cam15, pts4, stats_, translation_std, rotation_std, point_std, seed):
    L.check(L.lib().c2b_add_noise_entities(_p(cam15), cam15.shape[0], _p(pts4), pts4.shape[0], _p(stats_),
                                           float(translation_std), float(rotation_std), float(point_std), int(seed),
                                           _stream()))


def add_noise_observations(uv, obs_base, observations_std, seed):
    L.check(L.lib().c2b_add_noise_observations(_p(uv), uv.shape[0], int(obs_base), float(observations_std),
                                               int(seed), _stream()))


# ---- f32 extension (BASELINE configs[4]): the same entity kernels over a float state ---------------------
def to_f32(t):
    _chk(t, torch.float64, "to_f32 input")
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    L.check(L.lib().c2b_convert_f64_to_f32(_p(t), t.numel(), _p(out), _stream()))
    return out


def to_f64(t):
    _chk(t, torch.float32, "to_f64 input")
    out = torch.empty(t.shape, dtype=torch.float64, device=t.device)
    L.check(L.lib().c2b_convert_f32_to_f64(_p(t), t.numel(), _p(out), _stream()))
    return out


def stats_f32(cam15_f32, pts4_f32, ws, out=None):
    out = out if out is not None else torch.empty(L.STATS_DOUBLES, dtype=torch.float64, device=cam15_f32.device)
    L.check(L.lib().c2b_stats_f32(_p(cam15_f32), cam15_f32.shape[0], _p(pts4_f32), pts4_f32.shape[0], _p(ws), _p(out),
                                  _stream()))
    return out


def add_drift_f32(cam15, pts4, stats_, strength, angle_strength, std, dir_, seed):
    L.check(L.lib().c2b_add_drift_f32(_p(cam15), cam15.shape[0], _p(pts4), pts4.shape[0],
                                      C.c_void_p(stats_.data_ptr() + 15 * 8), float(strength), float(angle_strength),
                                      float(std), float(dir_[0]), float(dir_[1]), float(dir_[2]), int(seed), _stream()))


def add_drift_normalized_f32(cam15, pts4, stats_, strength, angle_strength, std, seed):
    L.check(L.lib().c2b_add_drift_normalized_f32(_p(cam15), cam15.shape[0], _p(pts4), pts4.shape[0], _p(stats_),
                                                 float(strength), float(angle_strength), float(std), int(seed),
                                                 _stream()))


def add_noise_entities_f32(cam15, pts4, stats_, translation_std, rotation_std, point_std, seed):
    L.check(L.lib().c2b_add_noise_entities_f32(_p(cam15), cam15.shape[0], _p(pts4), pts4.shape[0], _p(stats_),
                                               float(translation_std), float(rotation_std), float(point_std),
                                               int(seed), _stream()))


def add_sin_noise_f32(cam15, pts4, stats_, dir_, noise_dir, strength, frequency):
    L.check(L.lib().c2b_add_sin_noise_f32(_p(cam15), cam15.shape[0], _p(pts4), pts4.shape[0], _p(stats_),
                                          float(dir_[0]), float(dir_[1]), float(dir_[2]), float(noise_dir[0]),
                                          float(noise_dir[1]), float(noise_dir[2]), float(strength), float(frequency),
                                          _stream()))


# ---- the remaining Camera trait methods, batched (src/baproblem.rs:141-143, 165-175) ------------------------
def project_world(cam15, cam_idx, p3):
    out = torch.empty_like(p3)
    L.check(L.lib().c2b_project_world(_p(cam15), _p(cam_idx), _p(p3), p3.shape[0], _p(out), _stream()))
    return out


def to_world(cam15, cam_idx, p3):
    out = torch.empty_like(p3)
    L.check(L.lib().c2b_to_world(_p(cam15), _p(cam_idx), _p(p3), p3.shape[0], _p(out), _stream()))
    return out


def cameras_transform(cam15, delta_dir9, delta_loc3):
    L.check(L.lib().c2b_cameras_transform(_p(cam15), _p(delta_dir9), _p(delta_loc3), cam15.shape[0], _stream()))
    return cam15
