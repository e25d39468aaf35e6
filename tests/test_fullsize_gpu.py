"""Full-size frames against the oracle, pixel by pixel (the only full-size check of the bench workload compared
every 6th pixel with a thumbnail).  The GPU box's host has enough threads for the scalar oracle to render BASELINE configs
2, 3 and 4 at their full resolution in seconds:

  config 2  scenes/balls.scene (through the loader) 1920x1080, 4 reflection bounces      — the bench workload itself
  config 3  crytek_sponza stand-in 1920x1080, 1 light                                    — the bench line's secondary block
  config 4  crytek_sponza stand-in 3840x2160, 8 lights (whole frame on one GPU; its 8-way tiling is bit-identical to this
            frame by tests/test_configs_gpu.py)

Every channel of every pixel within 1e-4 of the oracle, ray classes exactly equal."""
import ctypes as C
import os

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi, scenefile
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _full_frame(scene, p):
    import torch
    out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    abi.check(abi.load_hip_lib().nrays_render_device(scene.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(scene)
    ref, ost = oracle.render(scene.descriptor, p, os.cpu_count() or 8)
    img = out.cpu().numpy()
    err = np.abs(img - ref)
    assert err.max() <= TOL, "max err %g at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))
    for k in CLASSES:
        assert getattr(st, k) == getattr(ost, k), (k, st.as_dict(), ost.as_dict())
    return float(err.max()), st


def test_config2_balls_scene_file_full_1080p(gpu):
    from tools import gen_assets
    gen_assets.gen_globe()
    fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "balls.scene"))
    cam = fs.camera_dict()
    w, h = cam["resolution"]
    assert (w, h) == (1920, 1080)
    p = nr.make_params((w, h), 1, 0.0, cam["eye"], fs.inverse_projection(0, w, h))
    err, st = _full_frame(fs, p)
    assert st.total_rays() == 2272878 and st.generations == 4  # the bench line's rays_per_frame


def test_config3_sponza_standin_full_1080p(gpu):
    sc, cam = standins.sponza_scene()
    p, _ = su.camera_params(cam, 1920, 1080)
    err, st = _full_frame(sc, p)
    assert st.rays_primary == 1920 * 1080 and st.rays_refraction > 0


def test_config4_sponza_standin_8_lights_full_4k(gpu):
    sc, cam = standins.sponza_scene(n_lights=8)
    p, _ = su.camera_params(cam, 3840, 2160)
    err, st = _full_frame(sc, p)
    assert st.rays_primary == 3840 * 2160 and st.rays_shadow >= 8 * 0.9 * 3840 * 2160


def test_hairball_standin_full_1080p_and_aa(gpu):
    """The 2.88 M-triangle stand-in at 1920x1080: one sample per pixel, then `aa 4 1.0` (sample-major lane mapping, 4 lanes
    per pixel) — every pixel against the oracle."""
    sc, cam = standins.hairball_scene()
    p, _ = su.camera_params(cam, 1920, 1080)
    _full_frame(sc, p)
    p4, _ = su.camera_params(cam, 960, 540, spp=4, window=1.0, seed=1)
    _, st = _full_frame(sc, p4)
    assert st.rays_primary == 960 * 540 * 4


def test_primitives_scene_file_full_1080p_area_light(gpu):
    """BASELINE config 1's file as shipped (area light: 9 jittered shadow rays per hit, transparent shapes, reflecting
    plane) at 1920x1080 with the counter-based RNG."""
    from tools import gen_assets
    gen_assets.gen_globe()
    fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "primitives.scene"))
    cam = fs.camera_dict()
    p = nr.make_params((1920, 1080), 1, 0.0, cam["eye"], fs.inverse_projection(0, 1920, 1080), seed=5)
    _, st = _full_frame(fs, p)
    assert st.rays_shadow > 9 * 0.3 * 1920 * 1080
