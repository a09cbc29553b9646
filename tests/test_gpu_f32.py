"""f32 extension (BASELINE.json configs[4]: "noise.rs drift+rotation kernels, f32 path").  The reference has no
f32 compute path; the float kernels are judged against the f64 oracle with an f32 tolerance:
same Philox draws, state rounded to float, so |f32 - f64| <~ 1e-6 * scale of the perturbed quantity."""
import numpy as np
import pytest

import oracle as O
from _problems import grid_cameras_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as entry
    entry.build()
    import torch
    import city2ba_amd
    from city2ba_amd import device as D
    assert city2ba_amd.device_count() > 0
    dev = torch.device("cuda", 0)
    cams, pts = grid_cameras_points(3, cpb=10, ppb=20, L=5.0)
    cam15 = torch.from_numpy(cams).to(dev)
    pts4 = D.points_pad(torch.from_numpy(pts).to(dev))
    ws = D.workspace(0, dev)
    return dict(torch=torch, D=D, dev=dev, cams=cams, pts=pts, cam15=cam15, pts4=pts4, ws=ws)


def _f32_state(env):
    D = env["D"]
    return D.to_f32(env["cam15"]), D.to_f32(env["pts4"])


def _back(env, c32, p32):
    D = env["D"]
    return D.to_f64(c32).cpu().numpy(), D.to_f64(p32).cpu().numpy()[:, :3]


def test_conversion_and_stats_f32(env):
    D = env["D"]
    c32, p32 = _f32_state(env)
    c_back, p_back = _back(env, c32, p32)
    assert np.array_equal(c_back, env["cams"].astype(np.float32).astype(np.float64))
    assert np.array_equal(p_back, env["pts"].astype(np.float32).astype(np.float64))
    st = D.stats_f32(c32, p32, env["ws"]).cpu().numpy()
    assert np.allclose(st[0:3], O.mean(env["cams"], env["pts"]), rtol=1e-6, atol=1e-6)
    assert np.allclose(st[3:6], O.std(env["cams"], env["pts"]), rtol=1e-6)
    mn, mx = O.extent(env["cams"], env["pts"])
    assert np.allclose(st[6:9], mn, atol=1e-6) and np.allclose(st[9:12], mx, atol=1e-5)
    _, idx = O.drift_origin(env["cams"], env["pts"])
    assert int(st[18]) == idx


def test_add_drift_f32_tracks_f64_oracle(env):
    D = env["D"]
    c32, p32 = _f32_state(env)
    st = D.stats_f32(c32, p32, env["ws"])
    d = np.array([0.3, -0.5, 0.8])
    D.add_drift_f32(c32, p32, st, 1e-3, 2e-3, 0.2, d, seed=42)
    got_c, got_p = _back(env, c32, p32)
    want_c, want_p = O.add_drift(env["cams"], env["pts"], 1e-3, 2e-3, 0.2, d, seed=42)
    scale = np.abs(want_p).max()
    assert np.max(np.abs(got_p - want_p)) < 4e-6 * scale
    assert np.max(np.abs(got_c - want_c)) < 4e-6 * max(1.0, np.abs(want_c).max())
    # and it really moved things (not a no-op)
    assert np.max(np.abs(want_p - env["pts"])) > 1e-3


def test_add_noise_entities_f32_tracks_f64_oracle(env):
    D = env["D"]
    c32, p32 = _f32_state(env)
    st = D.stats_f32(c32, p32, env["ws"])
    D.add_noise_entities_f32(c32, p32, st, 0.1, 0.1, 0.1, seed=99)
    got_c, got_p = _back(env, c32, p32)
    want_c, want_p, _ = O.add_noise(env["cams"], env["pts"], np.zeros((0, 2)), 0.1, 0.1, 0.1, 0.0, seed=99)
    assert np.max(np.abs(got_p - want_p)) < 4e-6 * np.abs(want_p).max()
    assert np.max(np.abs(got_c - want_c)) < 1e-5 * max(1.0, np.abs(want_c).max())
    R = got_c[:, :9].reshape(-1, 3, 3)
    assert np.max(np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3))) < 1e-5       # still rotations


def test_normalized_drift_and_sin_f32(env):
    D = env["D"]
    c32, p32 = _f32_state(env)
    st = D.stats_f32(c32, p32, env["ws"])
    D.add_drift_normalized_f32(c32, p32, st, 0.01, 0.01, 0.1, seed=7)
    got_c, got_p = _back(env, c32, p32)
    want_c, want_p = O.add_drift_normalized(env["cams"], env["pts"], 0.01, 0.01, 0.1, seed=7)
    assert np.max(np.abs(got_p - want_p)) < 2e-5 * np.abs(want_p).max()
    c32, p32 = _f32_state(env)
    D.add_sin_noise_f32(c32, p32, st, [1.0, 1.0, 0.0], [0.0, 1.0, 0.0], 1.0, 2.0)
    got_c, got_p = _back(env, c32, p32)
    want_c, want_p = O.add_sin_noise(env["cams"], env["pts"], [1.0, 1.0, 0.0], [0.0, 1.0, 0.0], 1.0, 2.0)
    assert np.max(np.abs(got_p - want_p)) < 1e-5 * np.abs(want_p).max()
    assert np.max(np.abs(got_c - want_c)) < 1e-5 * max(1.0, np.abs(want_c).max())
