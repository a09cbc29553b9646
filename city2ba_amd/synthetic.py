"""Host-side mirror of synthetic.rs over the C ABI: the layout runs in the native host code (csrc/host_synthetic.hpp);
from_position_direction, the whole visibility loop (src/synthetic.rs:268-297: candidates within max_dist, hits_building,
the predicate -- csrc/cell_kernels.hpp) and cull() (src/synthetic.rs:299) on the GPU.  The host candidate search
(candidate_pairs) stays as the second implementation the device path is tested against."""
import ctypes as C
import os

import numpy as np

from . import _lib as L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def grid_sizes(num_blocks, cameras_per_block=10, points_per_block=10):
    a, b = C.c_int64(), C.c_int64()
    L.check(L.lib().c2b_synthetic_grid_sizes(cameras_per_block, points_per_block, num_blocks, C.byref(a), C.byref(b)))
    return a.value, b.value


def grid_layout(num_blocks, cameras_per_block=10, points_per_block=10, block_length=20.0, block_inset=1.0,
                camera_height=1.0, point_height=1.0):
    """camera positions [n,3], directions [n,9] (col-major), points [n,3]: src/synthetic.rs:178-258."""
    n_cam, n_pts = grid_sizes(num_blocks, cameras_per_block, points_per_block)
    pos, dirs, pts = np.empty((n_cam, 3)), np.empty((n_cam, 9)), np.empty((n_pts, 3))
    L.check(L.lib().c2b_synthetic_grid_layout(cameras_per_block, points_per_block, num_blocks, float(block_length),
                                              float(block_inset), float(camera_height), float(point_height),
                                              _ptr(pos), _ptr(dirs), _ptr(pts)))
    return pos, dirs, pts


def line_layout(num_cameras, num_points, length, point_offset, camera_height, point_height):
    """src/synthetic.rs:323-344"""
    pos, dirs, pts = np.empty((num_cameras, 3)), np.empty((num_cameras, 9)), np.empty((num_points, 3))
    L.check(L.lib().c2b_synthetic_line_layout(num_cameras, num_points, float(length), float(point_offset),
                                              float(camera_height), float(point_height), _ptr(pos), _ptr(dirs),
                                              _ptr(pts)))
    return pos, dirs, pts


def candidate_pairs(centers, pts, max_dist, cam_lo=0, cam_hi=None, occlusion=False, block_length=20.0,
                    block_inset=1.0, n_threads=None):
    """(cam_idx u32, pt_idx u32): rstar's locate_within_distance (src/synthetic.rs:277-280) replaced by
    cell binning, optionally filtered by hits_building (:52-124).  Camera-major, ascending point index."""
    centers = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    cam_hi = len(centers) if cam_hi is None else cam_hi
    n_threads = n_threads or max(1, min(16, os.cpu_count() or 1))
    h = C.c_void_p()
    L.check(L.lib().c2b_candidate_pairs(_ptr(centers), len(centers), _ptr(pts), len(pts), float(max_dist), int(cam_lo),
                                        int(cam_hi), int(bool(occlusion)), float(block_length), float(block_inset),
                                        int(n_threads), C.byref(h)))
    try:
        n = L.lib().c2b_pairs_count(h)
        if n == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        ci = np.ctypeslib.as_array(C.cast(L.lib().c2b_pairs_cam_idx(h), C.POINTER(C.c_uint32)), shape=(n,)).copy()
        pi = np.ctypeslib.as_array(C.cast(L.lib().c2b_pairs_pt_idx(h), C.POINTER(C.c_uint32)), shape=(n,)).copy()
    finally:
        L.lib().c2b_pairs_free(h)
    return ci, pi


def _visibility_problem(layout, max_dist, occlusion, block_length, block_inset, cull, device, host_candidates=False, mirror=True):
    """Shared tail of synthetic_grid / synthetic_line (src/synthetic.rs:260-299, 346-380).  `layout` = ("grid", cpb, ppb,
    blocks, L, inset, cam_h, pt_h) or ("line", n_cam, n_pts, length, point_offset, cam_h, pt_h).  By default everything
    -- the layout loops, the candidate search, hits_building, the predicate, cull -- runs on the resident problem;
    host_candidates = True takes rounds 1-3's route (layout, candidate search and hits_building on the host, predicate +
    compaction on the device): the same problem, element for element -- kept for the tests that compare the two."""
    from .baproblem import BAProblem
    ba = BAProblem(device)
    rows = None
    if host_candidates:
        pos, dirs, pts = grid_layout(*layout[3:4], *layout[1:3], *layout[4:]) if layout[0] == "grid" else line_layout(*layout[1:])
        cam15 = ba._cameras_from_position_direction(pos, dirs)
        ba._upload(cam15, False, pts, np.zeros(len(pos) + 1, dtype=np.uint64), [], np.zeros((0, 2)))
        s = ba._camera_centers()
        ci, pi = candidate_pairs(s, pts, max_dist, occlusion=occlusion, block_length=block_length,
                                 block_inset=block_inset)
        ba.visibility_pairs_compact(ci, pi, max_dist, fetch=False)         # survivors compacted on the device ...
    else:
        if layout[0] == "grid":
            L.check(L.lib().c2b_problem_synthetic_grid_layout(ba._h, *[int(v) for v in layout[1:4]], *[float(v) for v in layout[4:]]))
        else:
            L.check(L.lib().c2b_problem_synthetic_line_layout(ba._h, int(layout[1]), int(layout[2]), *[float(v) for v in layout[3:]]))
        ba._row_ptr = np.zeros(ba._sizes()[0] + 1, dtype=np.uint64)
        ba._pt_idx = np.zeros(0, dtype=np.uint64)
        rows = ba.visibility_within_distance(max_dist, occlusion, block_length, block_inset, fetch=False)
    ba.adopt_visibility(mirror=mirror or cull or rows is None)             # ... where they become the vis_graph
    if not (mirror or cull) and rows is not None:
        ba._row_ptr = rows                                                 # the row pointer came back with the loop; pt_idx() stays stale
    return ba.cull() if cull else ba


def synthetic_grid(num_cameras_per_block, num_points_per_block, num_blocks, block_length, block_inset,
                   camera_height, point_height, max_dist, verbose=False, cull=True, device=0, host_candidates=False,
                   mirror=True):
    """synthetic_grid (src/synthetic.rs:163-300), same argument order.  In-camera observation order is
    ascending point index (the reference's is rstar's traversal order).  mirror = False (with cull = False): the graph
    stays on the device only (BAProblem.export_device hands it to the Level-0 launchers)."""
    layout = ("grid", num_cameras_per_block, num_points_per_block, num_blocks, block_length, block_inset, camera_height, point_height)
    return _visibility_problem(layout, max_dist, True, block_length, block_inset, cull, device, host_candidates, mirror)


def synthetic_line(num_cameras, num_points, length, point_offset, camera_height, point_height, max_dist,
                   verbose=False, cull=True, device=0, host_candidates=False):
    """synthetic_line (src/synthetic.rs:313-381)"""
    layout = ("line", num_cameras, num_points, length, point_offset, camera_height, point_height)
    return _visibility_problem(layout, max_dist, False, 1.0, 0.0, cull, device, host_candidates)
