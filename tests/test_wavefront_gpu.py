"""The staged ("wavefront") form of the trace loop (nrays_amd/csrc/wavefront.hip): primary stage -> compacted (ray, hit) queue ->
closest / shadow / shade stages generation after generation.  It replaces the megakernel for frames the library's rule selects
(NRAYS_WAVEFRONT=1 forces it, =0 forbids it); arithmetic and summation orders are the megakernel's, so frames and ray classes must be
IDENTICAL between the two forms — and within the north star's 1e-4 of the oracle (reference: Scene::trace, src/scene.rs:163-252,
the per-pixel sample sum src/scene.rs:72-94)."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "generations")


@pytest.fixture(params=["2", "0"], autouse=True, ids=["refill", "norefill"])
def refill(request, monkeypatch):
    """Both forms of the traversal stages: lanes refilled from the wave's ray stream (stream_device.h: traverse_stream, the default)
    and 64 rays run from start to end together (traverse())."""
    monkeypatch.setenv("NRAYS_WF_REFILL", request.param)


def _render(make, w, h, frames=1, **kw):
    sc, cam = make()
    p, _ = su.camera_params(cam, w, h, **kw)
    lib = abi.load_hip_lib()
    out = []
    for _ in range(frames):
        img = np.empty((p.height if not p.band_owners > 1 else lib.nrays_tile_rows(C.byref(p)), w, 3), np.float32)
        abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
        st = nr.get_stats(sc)
        out.append((img, tuple(getattr(st, k) for k in CLASSES)))
    return sc, p, out


def _both(monkeypatch, make, w, h, frames=2, **kw):
    monkeypatch.setenv("NRAYS_WAVEFRONT", "0")
    sc, p, ref = _render(make, w, h, **kw)
    monkeypatch.setenv("NRAYS_WAVEFRONT", "1")
    _, _, got = _render(make, w, h, frames=frames, **kw)
    for img, counts in got:
        assert counts == ref[0][1], (counts, ref[0][1])
        assert np.array_equal(img, ref[0][0]), np.abs(img - ref[0][0]).max()
    return sc, p, got[0][0]


def _area(make, radius):
    def f():
        sc, cam = make()
        lights = [nr.Light(l.pos, radius, 1, l.color) for l in sc._lights]
        return nr.Scene(list(sc._nodes), lights, (1, 1, 1)), cam
    return f


@pytest.mark.parametrize("lights", [1, 8])
def test_sponza_frames_are_the_megakernels(gpu, monkeypatch, lights):
    """Alpha-mapped layers: up to nine generations of refraction continuations, shadow rays with colour filters; one light (the shadow
    ray is traced inside the shade stage) and eight (the (chunk, light) shadow stage)."""
    make = lambda: standins.sponza_scene(detail=0.2, n_lights=lights)
    sc, p, img = _both(monkeypatch, make, 320, 180)
    want, _ = oracle.render(sc.descriptor, p, 32)
    assert np.abs(img - want).max() <= 1e-4  # north_star tolerance (BASELINE.json)


@pytest.mark.parametrize("spp", [1, 3, 4, 64])
def test_hair_frames_are_the_megakernels(gpu, monkeypatch, spp):
    make = lambda: standins.hairball_scene(strands=400)
    kw = dict(spp=spp, window=1.0, seed=3) if spp > 1 else {}
    w, h = (96, 54) if spp == 64 else (240, 136)
    sc, p, img = _both(monkeypatch, make, w, h, **kw)
    if spp <= 4:
        want, _ = oracle.render(sc.descriptor, p, 32)
        assert np.abs(img - want).max() <= 1e-4


def test_anti_aliased_sponza_with_an_area_light(gpu, monkeypatch):
    """RNG keys travel through the queues: AA jitter, a jittered light position per hit, key hashes per continuation."""
    make = _area(lambda: standins.sponza_scene(detail=0.15), 3.0)
    _both(monkeypatch, make, 160, 90, spp=4, window=1.0, seed=11)
    make8 = _area(lambda: standins.sponza_scene(detail=0.15, n_lights=8), 2.0)
    _both(monkeypatch, make8, 160, 90, spp=2, window=0.5, seed=5)


def test_several_passes_over_tile_ranges(gpu, monkeypatch):
    """NRAYS_WF_MAX_PATHS bounds the (pixel, sample) paths of one pass: the frame is rendered range after range of wave tiles."""
    monkeypatch.setenv("NRAYS_WF_MAX_PATHS", "4096")
    make = lambda: standins.sponza_scene(detail=0.2, n_lights=2)
    _both(monkeypatch, make, 200, 120)
    _both(monkeypatch, lambda: standins.hairball_scene(strands=300), 120, 80, spp=16, window=1.0, seed=2)


def test_band_tiles_of_a_multi_gpu_frame(gpu, monkeypatch):
    """An owner's compact tile (16-row bands dealt round-robin, padding rows of the last band) through the staged path."""
    make = lambda: standins.sponza_scene(detail=0.2, n_lights=8)
    for owner in (0, 2):
        _both(monkeypatch, make, 200, 136, band_rows=16, band_owner=owner, band_owners=3)


def test_max_depth_cuts_the_generations(gpu, monkeypatch):
    make = lambda: standins.sponza_scene(detail=0.2)
    _both(monkeypatch, make, 200, 120, max_depth=2)
