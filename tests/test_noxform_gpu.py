"""kFeatNoXform (device_types.h): scenes of TriMesh nodes whose BLASes all sit in world space (identity rotation, zero translation — the
reference's Isometry3 of an untransformed `obj` node, examples/loader3d.rs:546-552) are rendered by permutations of the mesh kernels that
keep ONE ray instead of a world and an instance-local one.  o - 0 and R = I leave every coordinate bit for bit, so frames and ray classes
must equal the general permutations' (NRAYS_NOXFORM=0) and the oracle's."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "generations")


def _render(make, w, h, **kw):
    sc, cam = make()
    p, _ = su.camera_params(cam, w, h, **kw)
    img = np.empty((h, w, 3), np.float32)
    abi.check(abi.load_hip_lib().nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
    st = nr.get_stats(sc)
    return sc, p, img, tuple(getattr(st, k) for k in CLASSES)


@pytest.mark.parametrize("lights,spp", [(1, 1), (8, 1), (1, 4), (2, 3)])
def test_untransformed_mesh_scenes_render_the_same_frame(gpu, monkeypatch, lights, spp):
    make = lambda: standins.sponza_scene(detail=0.2, n_lights=lights)
    kw = dict(spp=spp, window=1.0, seed=7) if spp > 1 else {}
    monkeypatch.setenv("NRAYS_NOXFORM", "0")
    sc, p, ref, ref_counts = _render(make, 320, 180, **kw)
    monkeypatch.delenv("NRAYS_NOXFORM")
    _, _, img, counts = _render(make, 320, 180, **kw)
    assert counts == ref_counts
    assert np.array_equal(img, ref), np.abs(img - ref).max()
    if spp == 1:
        want, _ = oracle.render(sc.descriptor, p, 32)
        assert np.abs(img - want).max() <= 1e-4  # north_star tolerance (BASELINE.json)
