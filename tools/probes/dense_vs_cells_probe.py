import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import __graft_entry__ as e; e.build()
import city2ba_amd as c2b
from city2ba_amd import generate as G
import subprocess, tempfile, os
d = tempfile.mkdtemp()
subprocess.check_call([sys.executable, 'tools/make_city_obj.py', d + '/c.obj', '--blocks', '48', '--detail', '4'], stdout=subprocess.DEVNULL)
o = G.ObjFile(d + '/c.obj'); tri = o.triangles(o.index('street') if 'street' in o.names() else -1)
pm = o.index('street')
pos, dirs = G.generate_cameras_path_step(o, pm, 30000, 1.5)
empty = np.zeros(len(pos) + 1, dtype=np.uint64)
b0 = c2b.BAProblem.from_visibility(np.zeros((0, 15)), np.zeros((0, 3)), np.zeros(1, dtype=np.uint64), [], np.zeros((0, 2)))
cams = b0._cameras_from_position_direction(pos, dirs)
centers = c2b.BAProblem.from_visibility(cams, np.zeros((0, 3)), empty, [], np.zeros((0, 2)))._camera_centers()
pts = G.generate_world_points_uniform(tri, centers, 300000, 40.0, seed=3)
ba = c2b.BAProblem.from_visibility(cams, pts, empty, [], np.zeros((0, 2)))
for rep in range(2):
    t = time.time(); r0 = ba.visibility_graph(40.0, fetch=False); t0 = time.time() - t
    t = time.time(); r1 = ba.visibility_within_distance(40.0, False, fetch=False); t1 = time.time() - t
    print('dense %.1f ms, cells %.1f ms, edges %d %d' % (1e3 * t0, 1e3 * t1, r0[-1], r1[-1]))
row0, pi0, uv0 = ba.visibility_graph(40.0)
row1, pi1, uv1 = ba.visibility_within_distance(40.0, False)
print('equal:', np.array_equal(row0, row1), np.array_equal(pi0, pi1), np.array_equal(uv0.view(np.uint64), uv1.view(np.uint64)), len(cams), len(pts))
