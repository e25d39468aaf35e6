"""Work lists of the mesh kernels (k_primary): a wave's first entry is static — workgroup b owns entry (b / 8) * 4 + wave of list
b mod 8 — the eight counters count on from the number of static owners, and a wave whose list runs dry looks at all counters before it
steals.  None of that may change a pixel: grids smaller and larger than the lists, frames with fewer tiles than waves, ragged frames,
cost-ordered and image-order lists (reference: the thread partition of src/scene.rs:49-66 never changes a pixel either)."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _frames(make, w, h, n=3):
    sc, cam = make()
    p, _ = su.camera_params(cam, w, h)
    lib = abi.load_hip_lib()
    out = []
    for _ in range(n):
        img = np.empty((h, w, 3), np.float32)
        abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
        st = nr.get_stats(sc)
        out.append((img, tuple(getattr(st, k) for k in CLASSES)))
    return out


@pytest.mark.parametrize("res", [(24, 16), (101, 67), (320, 180)])
def test_grid_size_and_list_order_do_not_change_a_pixel(gpu, monkeypatch, res):
    make = lambda: standins.sponza_scene(detail=0.2)
    w, h = res
    monkeypatch.setenv("NRAYS_LPT", "0")  # image-order lists, default grid: the reference frame
    ref = _frames(make, w, h, 1)[0]
    monkeypatch.delenv("NRAYS_LPT")
    for wg_per_cu in ("0", "1", "3"):  # default / fewer workgroups than lists entries / an odd number per CU
        monkeypatch.setenv("NRAYS_GRID_WG_PER_CU", wg_per_cu)
        for img, rays in _frames(make, w, h, 3):  # first frame image order, then cost order
            assert rays == ref[1], (wg_per_cu, rays, ref[1])
            assert np.array_equal(img, ref[0]), (wg_per_cu, float(np.abs(img - ref[0]).max()))


def test_hair_and_many_lights(gpu, monkeypatch):
    for make, (w, h) in ((lambda: standins.hairball_scene(strands=300), (200, 120)), (lambda: standins.sponza_scene(detail=0.2, n_lights=8), (192, 108))):
        monkeypatch.setenv("NRAYS_GRID_WG_PER_CU", "0")
        ref = _frames(make, w, h, 1)[0]
        monkeypatch.setenv("NRAYS_GRID_WG_PER_CU", "1")
        for img, rays in _frames(make, w, h, 3):
            assert rays == ref[1]
            assert np.array_equal(img, ref[0])
