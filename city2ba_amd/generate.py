"""Host-side mirror of generate.rs over the C ABI.  Mesh loading, camera placement and point sampling run in the
native host code (csrc/host_generate.hpp); from_position_direction, the visibility sweep and the occlusion rays of
visibility_graph (src/generate.rs:424-481) run on the GPU.  Rays are cast by brute force over the mesh's triangles in
place of Embree, and the samplers draw from seeded std::mt19937_64 streams where the reference uses thread_rng()."""
import ctypes as C
import os

import numpy as np

from . import _lib as L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class ObjFile:
    """tobj::load_obj (src/bin/city2ba.rs:481-491): the models of a .obj file, by name."""

    def __init__(self, path):
        self._h = C.c_void_p()
        L.check(L.lib().c2b_obj_load(os.fsencode(str(path)), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().c2b_obj_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def names(self):
        lib = L.lib()
        return [lib.c2b_obj_model_name(self._h, m).decode() for m in range(lib.c2b_obj_model_count(self._h))]

    def index(self, name):
        """position of the model called `name` or -1 (src/bin/city2ba.rs:494)"""
        names = self.names()
        return names.index(name) if name in names else -1

    def model(self, m):
        """(positions [n,3] f32, indices u32, is_lines) of model m"""
        npos, nidx, lines = C.c_int64(), C.c_int64(), C.c_int()
        L.check(L.lib().c2b_obj_model_sizes(self._h, m, C.byref(npos), C.byref(nidx), C.byref(lines)))
        pos = np.empty((npos.value, 3), dtype=np.float32)
        idx = np.empty(nidx.value, dtype=np.uint32)
        L.check(L.lib().c2b_obj_model_copy(self._h, m, _ptr(pos), _ptr(idx)))
        return pos, idx, bool(lines.value)

    def move_to_origin(self, skip_model=-1):
        """move_to_origin (src/generate.rs:484-527), in place"""
        L.check(L.lib().c2b_obj_move_to_origin(self._h, int(skip_model)))

    def triangles(self, skip_model=-1):
        """[n,9] f32 triangles of every mesh model except `skip_model`"""
        n = C.c_int64()
        L.check(L.lib().c2b_obj_triangles(self._h, int(skip_model), None, C.byref(n)))
        tri = np.empty((n.value, 9), dtype=np.float32)
        L.check(L.lib().c2b_obj_triangles(self._h, int(skip_model), _ptr(tri), C.byref(n)))
        return tri


def generate_cameras_path(obj, path_model, num_cameras, seed=0):
    """generate_cameras_path (src/generate.rs:109-148): (positions [n,3], directions [n,9] col-major)"""
    pos, dirs = np.empty((num_cameras, 3)), np.empty((num_cameras, 9))
    L.check(L.lib().c2b_generate_cameras_path(obj._h, int(path_model), int(num_cameras), 0.0, int(seed), _ptr(pos), _ptr(dirs)))
    return pos, dirs


def generate_cameras_path_step(obj, path_model, num_cameras, step_size):
    """generate_cameras_path_step (src/generate.rs:152-213)"""
    if not step_size > 0:
        raise L.City2baError(L.ERR_INVALID_ARGUMENT, "step_size must be > 0")
    pos, dirs = np.empty((num_cameras, 3)), np.empty((num_cameras, 9))
    L.check(L.lib().c2b_generate_cameras_path(obj._h, int(path_model), int(num_cameras), float(step_size), 0, _ptr(pos), _ptr(dirs)))
    return pos, dirs


def generate_cameras_poisson(triangles, num_points, height, ground, seed=0):
    """generate_cameras_poisson (src/generate.rs:217-280)"""
    tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    n = C.c_int64()
    args = (_ptr(tri), len(tri), int(num_points), float(height), float(ground), int(seed))
    cap = 2 * int(num_points) + 8 * int((2 * int(num_points)) ** 0.5) + 64      # densest packing bounds the count
    pos, dirs = np.empty((cap, 3)), np.empty((cap, 9))
    L.check(L.lib().c2b_generate_cameras_poisson(*args, cap, _ptr(pos), _ptr(dirs), C.byref(n)))
    if n.value > cap:                                                          # same seed => same cameras
        pos, dirs = np.empty((n.value, 3)), np.empty((n.value, 9))
        L.check(L.lib().c2b_generate_cameras_poisson(*args, n.value, _ptr(pos), _ptr(dirs), C.byref(n)))
    return pos[:n.value].copy(), dirs[:n.value].copy()


def modify_intrinsics(cams15, intrinsic_start, intrinsic_end, seed=0):
    """modify_intrinsics (src/generate.rs:530-544) on [n,15] camera rows; returns the modified copy"""
    cams = np.array(cams15, dtype=np.float64, order="C", copy=True).reshape(-1, 15)
    a = np.ascontiguousarray(intrinsic_start, dtype=np.float64).reshape(3)
    b = np.ascontiguousarray(intrinsic_end, dtype=np.float64).reshape(3)
    L.check(L.lib().c2b_modify_intrinsics(_ptr(cams), len(cams), _ptr(a), _ptr(b), int(seed)))
    return cams


def generate_world_points_uniform(triangles, centers, num_points, max_dist, seed=0):
    """generate_world_points_uniform (src/generate.rs:356-420): [n,3] points on the mesh near the cameras"""
    tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    centers = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
    pts = np.empty((num_points, 3))
    n = C.c_int64()
    L.check(L.lib().c2b_generate_world_points(_ptr(tri), len(tri), _ptr(centers), len(centers), int(num_points),
                                              float(max_dist), int(seed), _ptr(pts), C.byref(n)))
    return pts[:n.value].copy()


def generate(path, num_cameras=100, num_world_points=1000, max_dist=100.0, intrinsics_start=(1.0, 0.0, 0.0),
             intrinsics_end=(1.0, 0.0, 0.0), ground=0.0, height=1.0, no_lcc=False, move_to_origin=False, path_name=None,
             step_size=0.0, seed=0, device=0, faithful=True):
    """run_generate (src/bin/city2ba.rs:480-573) as a library call; returns the BAProblem.  Seeds: cameras `seed`,
    intrinsics `seed + 1`, points `seed + 2` (the C++ CLI uses the same assignment).  faithful=False culls without
    the reference's observation-filter quirk (src/baproblem.rs:523; the CLI's --exact-lcc)."""
    from .baproblem import BAProblem
    obj = ObjFile(path)
    pm = -1
    if path_name is not None:
        pm = obj.index(path_name)
        if pm < 0:
            raise L.City2baError(L.ERR_INVALID_ARGUMENT, "Could not find a path named %s. Available model names are %s"
                                 % (path_name, ", ".join(obj.names())))
    if move_to_origin:
        obj.move_to_origin(pm)
    tri = obj.triangles(pm)
    if pm >= 0:
        pos, dirs = (generate_cameras_path(obj, pm, num_cameras, seed) if step_size <= 0.0
                     else generate_cameras_path_step(obj, pm, num_cameras, step_size))
    else:
        pos, dirs = generate_cameras_poisson(tri, num_cameras, height, ground, seed)
    obj.close()
    stage = BAProblem(device)
    cams = stage._cameras_from_position_direction(pos, dirs)
    cams = modify_intrinsics(cams, intrinsics_start, intrinsics_end, seed + 1)
    n_cam = len(cams)
    empty = np.zeros(n_cam + 1, dtype=np.uint64)
    ba = BAProblem.from_visibility(cams, np.zeros((0, 3)), empty, [], np.zeros((0, 2)), device)
    ba.generate_world_points(tri, num_world_points, max_dist, seed + 2)   # on the device: the points of the host sampler
    ba.visibility_graph(max_dist, triangles=tri, fetch=False)             # sweep + occlusion rays, lists stay on the device
    ba.adopt_visibility()
    if not no_lcc:
        ba = ba.cull(faithful)
    if ba.num_cameras() == 0 or ba.num_points() == 0:
        raise L.City2baError(L.ERR_INVALID_ARGUMENT, "EmptyProblem: No cameras remain")
    return ba
