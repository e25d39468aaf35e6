"""Shadow rays whose result is multiplied by exactly zero are counted, not traced, by plain renders (trace_device.h: shade_hit, light_is_dark).

(1) Hits that contribute nothing of their own to the pixel — a fully transparent point (opacity-map texel 0, node alpha 0: the holes of alpha-tested
foliage) or a perfect mirror (refl_mix 1): obj.rgb * (weight * alpha * (1 - mix)) = 0, scene.rs:179-190 — are not shaded by plain renders
(trace_device.h: shade_hit): their shadow rays are COUNTED (NraysStats::rays_shadow stays the reference's number, rays_shadow_elided says how many
were not traced) and the pixel must be the bit-identical one of the instrumented render, which traces and shades everything, and the oracle's.

(2) Light samples behind the surface: diffuse = kd * max(l.n, 0) = 0 and, where the mirrored light direction also points away from the eye, no specular
term either (phong_material.rs:109-141): the sample adds light.color * (filter * 0) whatever its shadow ray returns.  Every scene kind: same frames, same counts."""
import ctypes as C
import math

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su

pytestmark = pytest.mark.gpu


def _scene(n_lights, light_radius=0.0, nsample=1):
    """An alpha-mapped wall (holes: alpha 0), a node of alpha 0 behind it, a perfect mirror ball, a half mirror, a floor."""
    wall = nr.PhongMaterial((0.1, 0.3, 0.1), (0.2, 0.9, 0.3), (1, 1, 1), su.checker_texture(32, 4), su.checker_texture(32, 6, alpha_holes=True), 60.0)
    plain = nr.PhongMaterial((0.2, 0.2, 0.25), (0.7, 0.7, 0.8), (1, 1, 1), None, None, 80.0)
    quad = su.f32_exact([[-2.5, -1.0, 0.0], [2.5, -1.0, 0.0], [2.5, 2.0, 0.0], [-2.5, 2.0, 0.0]])
    quv = su.f32_exact([[0, 0], [2, 0], [2, 1], [0, 1]])
    qidx = np.asarray([[0, 2, 1], [0, 3, 2]], dtype=np.uint32)
    fl = su.f32_exact([[-8, -1.25, -8], [8, -1.25, -8], [8, -1.25, 8], [-8, -1.25, 8]])
    nodes = [
        nr.SceneNode(wall, 0.0, 0.0, 1.0, 1.1, nr.Isometry3((0.0, 0.0, -2.0), (0.0, math.radians(10.0), 0.0)), nr.TriMesh(quad, qidx, quv)),
        nr.SceneNode(plain, 0.0, 0.0, 0.0, 1.0, nr.Isometry3((0.0, 0.0, -1.0)), nr.TriMesh(quad, qidx, None)),      # alpha 0: invisible, still refracts
        nr.SceneNode(plain, 1.0, 0.3, 1.0, 1.0, nr.Isometry3((-1.6, 0.0, 1.5)), nr.Ball(0.9)),                          # perfect mirror
        nr.SceneNode(plain, 0.5, 0.3, 1.0, 1.0, nr.Isometry3((1.6, 0.0, 1.5)), nr.Ball(0.9)),                           # half mirror: shaded
        nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0)), nr.TriMesh(fl, qidx, quv)),
    ]
    lights = [nr.Light((3.0, 6.0, -6.0), light_radius, nsample, (0.7, 0.7, 0.7)), nr.Light((-4.0, 5.0, -2.0), 0.0, 1, (0.5, 0.5, 0.6)),
              nr.Light((0.0, 7.0, 3.0), 0.0, 1, (0.3, 0.3, 0.3))][:n_lights]
    per_hit = sum(l.racsample ** 2 for l in lights)  # shadow rays of one shaded hit (light.rs:20, scene.rs:262-299)
    return nr.Scene(nodes, lights, (0.6, 0.7, 0.9)), dict(eye=(0.3, 2.0, -9.0), at=(0.0, 0.2, 0.0), fovy=40.0), per_hit


@pytest.mark.parametrize("n_lights,radius,nsample,spp", [(1, 0.0, 1, 1), (3, 0.0, 1, 1), (2, 0.3, 4, 2)])
def test_hits_without_a_term_of_their_own_are_counted_not_traced(gpu, n_lights, radius, nsample, spp):
    import torch
    lib = abi.load_hip_lib()
    sc, cam, per_hit = _scene(n_lights, radius, nsample)
    p, _ = su.camera_params(cam, 208, 120, **(dict(spp=spp, window=1.0, seed=5) if spp > 1 else {}))
    ref, ost = oracle.render(sc.descriptor, p, 8)
    out = torch.empty((120, 208, 3), dtype=torch.float32, device="cuda")
    frames, stats = [], []
    for fn in (lib.nrays_render_device, lib.nrays_render_device, lib.nrays_render_device_instrumented):  # (second plain frame: cost-ordered lists, split tiles)
        abi.check(fn(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        frames.append(out.cpu().numpy().copy()); stats.append(nr.get_stats(sc))
    assert np.array_equal(frames[0], frames[2]) and np.array_equal(frames[1], frames[2])      # not shading them changes no bit
    assert float(np.abs(frames[0] - ref).max()) <= 1e-4
    for st in stats:
        for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
            assert getattr(st, k) == getattr(ost, k), (k, getattr(st, k), getattr(ost, k))   # the reference's counts, traced or not
    assert stats[2].rays_shadow_elided == 0                                                    # the instrumented render traces everything
    assert 0 < stats[0].rays_shadow_elided == stats[1].rays_shadow_elided < stats[0].rays_shadow
    assert stats[0].rays_shadow_elided >= per_hit                                              # (+ the samples of lights behind their surface)


def _balls():
    return su.balls_scene(tex_size=(64, 32))


def _prims():
    return su.primitives_scene(light_radius=0.1, nsample=10)


def _mesh3():
    return su.mesh_scene(alpha_mapped=False, n_lights=2)


def _shapes():
    return su.random_shapes_scene(3)


def _matte():
    """No specular colour (Ks 0 0 0): every light behind the surface is dark, whatever the mirrored direction."""
    pts, idx, uvs = su.torus_mesh()
    matte = nr.PhongMaterial((0.1, 0.1, 0.1), (0.9, 0.8, 0.7), (0.0, 0.0, 0.0), su.checker_texture(64, 8), None, 40.0)
    shiny = nr.PhongMaterial((0.1, 0.1, 0.1), (0.5, 0.6, 0.9), (1.0, 1.0, 1.0), None, None, 20.0)
    fl = su.f32_exact([[-6, -1.25, -6], [6, -1.25, -6], [6, -1.25, 6], [-6, -1.25, 6]])
    nodes = [nr.SceneNode(matte, 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0), (0.3, 0.2, 0.0)), nr.TriMesh(pts, idx, uvs)),
             nr.SceneNode(shiny, 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0)), nr.TriMesh(fl, np.asarray([[0, 2, 1], [0, 3, 2]], dtype=np.uint32), None)),
             nr.SceneNode(matte, 0.0, 0.0, 1.0, 1.0, nr.Isometry3((2.8, 0.2, 0.5)), nr.Ball(0.7))]
    lights = [nr.Light((3.0, 6.0, -6.0), 0.0, 1, (0.7, 0.7, 0.7)), nr.Light((-4.0, -0.5, 4.0), 0.0, 1, (0.5, 0.5, 0.6)), nr.Light((0.0, 0.5, 8.0), 0.2, 4, (0.4, 0.4, 0.4))]
    return nr.Scene(nodes, lights, (0.2, 0.3, 0.5)), dict(eye=(0.5, 3.0, -9.0), at=(0.0, 0.0, 0.0), fovy=40.0)


@pytest.mark.parametrize("make,kw", [(_balls, {}), (_prims, {}), (_prims, dict(spp=2, window=1.0, seed=3)), (_mesh3, {}), (su.mesh_scene, {}), (_shapes, {}), (_matte, {}), (_matte, dict(spp=2, window=1.0, seed=9))])
def test_lights_behind_the_surface_are_counted_not_traced(gpu, make, kw):
    import torch
    lib = abi.load_hip_lib()
    sc, cam = make()
    p, _ = su.camera_params(cam, 200, 120, **kw)
    ref, ost = oracle.render(sc.descriptor, p, 8)
    out = torch.empty((120, 200, 3), dtype=torch.float32, device="cuda")
    frames, stats = [], []
    for fn in (lib.nrays_render_device, lib.nrays_render_device, lib.nrays_render_device_instrumented):
        abi.check(fn(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        frames.append(out.cpu().numpy().copy()); stats.append(nr.get_stats(sc))
    assert np.array_equal(frames[0], frames[2]) and np.array_equal(frames[1], frames[2])
    assert float(np.abs(frames[0] - ref).max()) <= 1e-4
    for st in stats:
        for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
            assert getattr(st, k) == getattr(ost, k), (k, getattr(st, k), getattr(ost, k))
    assert stats[2].rays_shadow_elided == 0
    assert 0 < stats[0].rays_shadow_elided == stats[1].rays_shadow_elided < stats[0].rays_shadow


def _nonfinite_scene(kind):
    """The alpha-mapped scene above with ONE non-finite input: the reference multiplies it by 0 and gets NaN (phong_material.rs:131-141, scene.rs:179-190);
    skipping the multiplication would give a finite pixel, so nrays_scene_create switches the elisions off for such a scene (DScene::no_elide)."""
    sc, cam, per_hit = _scene(3)
    nodes, lights = list(sc._nodes), list(sc._lights)
    if kind == "light":      # an infinitely bright light behind half of the surfaces
        lights[1] = nr.Light((-4.0, 5.0, -2.0), 0.0, 1, (float("inf"), 0.5, 0.6))
    elif kind == "texel":    # one +inf texel in an RGBA32F colour texture of the alpha-mapped wall (its holes have alpha 0: inf * 0)
        tex = np.full((8, 8, 4), 0.5, np.float32); tex[..., 3] = 1.0; tex[3, 4, 1] = np.inf
        nodes[0].material.texture = nr.Texture2d(nr.ImageData(tex), nr.Interpolation.Nearest, nr.Overflow.Wrap)
    elif kind == "shininess":  # a negative exponent: scoeff^n -> inf as scoeff -> 0, on the node of alpha 0
        nodes[1].material = nr.PhongMaterial((0.2, 0.2, 0.25), (0.7, 0.7, 0.8), (1, 1, 1), None, None, -3.0)
    return nr.Scene(nodes, lights, (0.6, 0.7, 0.9)), cam


@pytest.mark.parametrize("kind", ["light", "texel", "shininess"])
def test_non_finite_inputs_switch_the_elisions_off(gpu, kind):
    import torch
    lib = abi.load_hip_lib()
    sc, cam = _nonfinite_scene(kind)
    p, _ = su.camera_params(cam, 208, 120)
    ref, ost = oracle.render(sc.descriptor, p, 8)
    out = torch.empty((120, 208, 3), dtype=torch.float32, device="cuda")
    frames, stats = [], []
    for fn in (lib.nrays_render_device, lib.nrays_render_device, lib.nrays_render_device_instrumented):
        abi.check(fn(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        frames.append(out.cpu().numpy().copy()); stats.append(nr.get_stats(sc))
    assert all(st.rays_shadow_elided == 0 for st in stats)                          # nothing is skipped in this scene
    assert np.array_equal(frames[0], frames[2], equal_nan=True) and np.array_equal(frames[1], frames[2], equal_nan=True)
    # the reference's pixels: NaN / inf where it produces them, the usual tolerance elsewhere
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(frames[0]), np.isnan(ref)) and np.array_equal(np.isinf(frames[0]), np.isinf(ref))
    assert float(np.abs(frames[0][fin] - ref[fin]).max()) <= 1e-4
    if kind != "shininess":
        assert (~fin).any()                                                          # the case really produces non-finite pixels
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
        assert getattr(stats[0], k) == getattr(ost, k), (k, getattr(stats[0], k), getattr(ost, k))
