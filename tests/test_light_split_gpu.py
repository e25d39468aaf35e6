"""Light-parallel wave tiles (DRender::light_lsl, k_tile_order): in multi-light mesh frames the tiles above a cost threshold are
rendered as 2^k parts, 2^k lanes per pixel, one light per lane in the shadow phase, the per-light sums folded in light order by
`__shfl`; in ONE-light alpha-mapped mesh frames the same parts are 64 >> k pixels whose lanes trace the same rays — scheduling only: the frame and the ray counts must be IDENTICAL to the one-lane-per-pixel render, whatever is split
(reference: the light loop of src/phong_material.rs:106-147)."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _frames(make, w, h, n, **kw):
    """n frames of one fresh handle (the first records the tile costs, the second sorts and may split, the later ones reuse the order)."""
    sc, cam = make()
    p, _ = su.camera_params(cam, w, h, **kw)
    lib = abi.load_hip_lib()
    out = []
    for _ in range(n):
        img = np.empty((h, w, 3), np.float32)
        abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
        st = nr.get_stats(sc)
        out.append((img, tuple(getattr(st, k) for k in CLASSES)))
    return sc, p, out


@pytest.mark.parametrize("lights", [8, 5, 3, 2, 1])  # (1: the pixel-split parts of one-light alpha-mapped frames, NR_PIXEL_SPLIT)
def test_split_tiles_do_not_change_a_pixel(gpu, monkeypatch, lights):
    from tools import standins
    make = lambda: standins.sponza_scene(detail=0.2, n_lights=lights)
    monkeypatch.setenv("NRAYS_LIGHT_SPLIT", "0")
    sc, p, ref = _frames(make, 192, 108, 2)
    for mode in ("-1", "0.05", "1"):  # every tile split / most tiles / the default threshold
        monkeypatch.setenv("NRAYS_LIGHT_SPLIT", mode)
        _, _, got = _frames(make, 192, 108, 4)
        for k, (img, rays) in enumerate(got):
            assert rays == ref[0][1], (mode, k, rays, ref[0][1])
            assert np.array_equal(img, ref[0][0]), (mode, k, np.abs(img - ref[0][0]).max())
    want, ost = oracle.render(sc.descriptor, p, 32)
    assert np.abs(ref[0][0] - want).max() <= 1e-4


def test_split_tiles_with_area_lights_bands_and_ragged_frames(gpu, monkeypatch):
    """Area lights (several samples per light, keyed RNG), a frame that is not a multiple of the tile size, band tiling."""
    def make():
        sc, cam = su.mesh_scene(n_lights=2)
        sc._lights = [nr.Light(l.pos, 0.3, 4, l.color) for l in sc._lights] + [nr.Light((1.0, 6.0, -2.0), 0.0, 1, (0.3, 0.3, 0.3))]
        sc._descriptor = None
        return sc, cam
    monkeypatch.setenv("NRAYS_LIGHT_SPLIT", "0")
    _, _, ref = _frames(make, 101, 67, 1, seed=5)
    monkeypatch.setenv("NRAYS_LIGHT_SPLIT", "-1")
    _, _, got = _frames(make, 101, 67, 3, seed=5)
    for img, rays in got:
        assert rays == ref[0][1] and np.array_equal(img, ref[0][0])
    # a band tile of the same frame (3 owners): rows of the compact buffer equal the full frame's
    lib = abi.load_hip_lib()
    sc, cam = make()
    full, _ = su.camera_params(cam, 101, 67, seed=5)
    from nrays_amd import tiling
    for owner in range(3):
        tp = tiling.tile_params(full, owner, 3, 16)
        rows = lib.nrays_tile_rows(C.byref(tp))
        for _ in range(3):
            tile = np.empty((rows, 101, 3), np.float32)
            abi.check(lib.nrays_render(sc.device_handle(), C.byref(tp), tile.ctypes.data_as(C.POINTER(C.c_float))))
        own = tiling.owned_rows(67, 16, owner, 3)
        assert np.array_equal(tile[:len(own)], ref[0][0][own])


def test_every_tile_split_without_the_preallocated_order_buffer(gpu, monkeypatch):
    """101 x 67 pixels = 140 wave tiles = 4 (mod 8): with every tile split, the lists 0..3 hold one tile more than the lists 4..7 and
    their last entries lie beyond `tiles << k` words (the order buffer must hold 8 ceil(tiles / 8) << k).  The 4K
    preallocation of a handle hides the difference, so the handles of this test allocate on their first frame (NRAYS_PREALLOC=0);
    several handles in a row, so that an out-of-bounds write would land in a neighbouring allocation that is in use."""
    def make():
        sc, cam = su.mesh_scene(n_lights=2)
        sc._lights = list(sc._lights) * 4  # 8 lights: 8 parts per tile
        sc._descriptor = None
        return sc, cam
    monkeypatch.setenv("NRAYS_PREALLOC", "0")
    monkeypatch.setenv("NRAYS_LIGHT_SPLIT", "0")
    _, _, ref = _frames(make, 101, 67, 1)
    monkeypatch.setenv("NRAYS_LIGHT_SPLIT", "-1")
    keep = []
    for _ in range(3):
        sc, _, got = _frames(make, 101, 67, 3)
        keep.append(sc)
        for img, rays in got:
            assert rays == ref[0][1] and np.array_equal(img, ref[0][0])
