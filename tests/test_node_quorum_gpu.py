"""Quorum-ended node phases (trace_device.h: traverse(), DScene::incoherent): in scenes with hair-like meshes a node phase ends once
fewer than a third of the query's lanes are still on internal nodes; the parked lanes resume after the leaf phase of the others.
Scheduling only: frames, ray classes and hit records must be IDENTICAL to the run in which every node phase lasts until its last
lane holds a leaf (NRAYS_NODE_QUORUM=0), and both must agree with the oracle (reference: the BVT queries of src/scene.rs:147-161,
262-339 do not depend on the visiting order)."""
import ctypes as C
import os

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _render(make, w, h, frames=2, instrumented=False, **kw):
    sc, cam = make()
    p, _ = su.camera_params(cam, w, h, **kw)
    lib = abi.load_hip_lib()
    out = []
    for _ in range(frames):
        if instrumented:
            import torch
            dev = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
            abi.check(lib.nrays_render_device_instrumented(sc.device_handle(), C.byref(p), C.c_void_p(dev.data_ptr()), None))
            img = dev.cpu().numpy()
        else:
            img = np.empty((h, w, 3), np.float32)
            abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
        st = nr.get_stats(sc)
        flags = (C.c_uint32 * 2)()
        abi.check(lib.nrays_debug_scene_flags(sc.device_handle(), flags))
        assert flags[0] & 2 and not flags[0] & 5, flags[0]  # the opaque-mesh kernels: the ones that hold the quorum code
        assert flags[1] == (0 if os.environ.get("NRAYS_NODE_QUORUM") == "0" else 1)
        out.append((img, tuple(getattr(st, k) for k in CLASSES + (("hit_records",) if instrumented else ()))))
    return sc, p, out


@pytest.mark.parametrize("spp", [1, 4])
def test_hair_frames_do_not_depend_on_the_quorum(gpu, monkeypatch, spp):
    make = lambda: standins.hairball_scene(strands=400)
    kw = dict(spp=spp, window=1.0, seed=3) if spp > 1 else {}
    monkeypatch.setenv("NRAYS_NODE_QUORUM", "0")
    sc, p, ref = _render(make, 240, 136, **kw)
    monkeypatch.setenv("NRAYS_NODE_QUORUM", "1")
    _, _, got = _render(make, 240, 136, frames=3, **kw)
    for img, counts in got:
        assert counts == ref[0][1]
        assert np.array_equal(img, ref[0][0]), np.abs(img - ref[0][0]).max()
    assert np.array_equal(ref[1][0], ref[0][0])
    want, _ = oracle.render(sc.descriptor, p, 32)
    assert np.abs(got[0][0] - want).max() <= 1e-4  # north_star tolerance (BASELINE.json)


def test_hair_beside_a_coherent_mesh_and_reflections(gpu, monkeypatch):
    """A hair-like mesh and an ordinary one in one opaque scene (two BLASes: different isometries), a reflective floor, two lights:
    rays leave the hair towards the torus and back, lanes of one wave sit in different BLASes."""
    def make():
        import math
        sc, cam = standins.hairball_scene(strands=150)
        pts, idx, uvs = su.torus_mesh()
        mat = nr.PhongMaterial((0.2, 0.2, 0.2), (1, 1, 1), (0.5, 0.5, 0.5), su.checker_texture(64, 8), None, 60.0)
        iso = nr.Isometry3((0.4, -0.3, 1.5), (0.0, math.radians(20.0), 0.0))
        fl = su.f32_exact([[-6, -1.25, -6], [6, -1.25, -6], [6, -1.25, 6], [-6, -1.25, 6]])
        fl_uv = su.f32_exact([[0, 0], [3, 0], [3, 3], [0, 3]])
        fl_idx = np.asarray([[0, 2, 1], [0, 3, 2]], dtype=np.uint32)
        nodes = list(sc._nodes) + [nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(_scaled(pts, 0.5), idx, uvs)),
                                   nr.SceneNode(mat, 0.3, 0.4, 1.0, 1.0, iso, nr.TriMesh(fl, fl_idx, fl_uv))]
        lights = list(sc._lights) + [nr.Light((3.0, 6.0, -6.0), 0.0, 1, (0.4, 0.4, 0.4))]
        return nr.Scene(nodes, lights, (1, 1, 1)), cam
    monkeypatch.setenv("NRAYS_NODE_QUORUM", "0")
    sc, p, ref = _render(make, 160, 120)  # (the instrumented kernel is the full-feature one, which has no quorum: plain renders)
    monkeypatch.setenv("NRAYS_NODE_QUORUM", "1")
    _, _, got = _render(make, 160, 120)
    for (img, counts), (rimg, rcounts) in zip(got, ref):
        assert counts == rcounts
        assert np.array_equal(img, rimg)
    assert ref[0][1][1] > 0  # the floor reflects
    want, ost = oracle.render(sc.descriptor, p, 32)
    assert np.abs(got[0][0] - want).max() <= 1e-4
    assert got[0][1][:4] == (ost.rays_primary, ost.rays_reflection, ost.rays_refraction, ost.rays_shadow)


def _scaled(pts, s):
    return su.f32_exact(np.asarray(pts, dtype=np.float64) * s)
