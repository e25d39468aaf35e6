"""The HIP intersectors and traversals ALONE (nrays_debug_cast_batch: no raygen, no shading) against the fixtures derived
independently of this code base (tests/golden/make_kat_independent.py: 60-digit mpmath bisection on point membership, exact
plane / barycentric solutions) — the same certification through the support maps as tests/test_kat_independent.py applies to
the oracle, so the device code meets the independent fixtures directly and not only through rendered frames
(reference call sites: src/scene_node.rs:41,51-54, examples/loader3d.rs:601-656,695)."""
import numpy as np
import pytest

import nrays_amd as nr
import oracle
from tests import test_kat_independent as kat
from tools import scenes_util as su

pytestmark = pytest.mark.gpu


def hip_cast(scene, o, d):
    return nr.cast_rays(scene, [o], [d])


@pytest.mark.parametrize("kind", [kat.BALL, kat.CUBOID, kat.CYLINDER, kat.CAPSULE, kat.CONE])
def test_hip_shape_casts_against_independent_fixtures(gpu, kind):
    kat.check_shape_cases(kind, hip_cast, "HIP")


def test_hip_triangle_casts_against_independent_fixtures(gpu):
    kat.check_triangle_cases(hip_cast, "HIP")


def _same_casts(sc, o, d):
    """hit / miss, scene node and toi IDENTICAL; normals to 1e-15; uv equal up to libm (u modulo the seam of atan2) wherever the
    device computes it — a ball whose material never reads the values skips atan2 / asin (kInstNoUvValues: u = v = 0)."""
    hit, out = nr.cast_rays(sc, o, d)
    ohit, oout = oracle.cast(sc.descriptor, o, d)
    assert np.array_equal(hit, ohit)
    assert np.array_equal(out[hit, 0], oout[hit, 0]) and np.array_equal(out[hit, 7], oout[hit, 7])
    assert np.abs(out[hit, 1:4] - oout[hit, 1:4]).max() <= 1e-15
    m = hit & ((out[:, 5] != 0.0) | (out[:, 6] != 0.0))
    if m.any():
        du = np.abs(out[m, 5] - oout[m, 5]); du = np.minimum(du, np.abs(1.0 - du))
        assert du.max() <= 1e-12 and np.abs(out[m, 6] - oout[m, 6]).max() <= 1e-12, (du.max(), np.abs(out[m, 6] - oout[m, 6]).max())
    return int(hit.sum()), int(m.sum())


def test_hip_casts_equal_the_oracle_bit_for_bit_on_random_scenes(gpu):
    """Closest hits of random rays through mixed scenes (every analytic shape under random isometries + meshes, then the
    rotated alpha-mapped mesh scene): f64 in the reference's operation order, -ffp-contract=off on both sides."""
    rng = np.random.default_rng(11)
    n_hit = n_uv = 0
    for trial in range(4):
        sc, _ = su.random_shapes_scene(seed=100 + trial)
        o = rng.uniform(-6, 6, (4000, 3)); d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        h, u = _same_casts(sc, o, d)
        n_hit += h; n_uv += u
    assert n_hit > 2000 and n_uv > 300
    sc, _ = su.mesh_scene(alpha_mapped=True, rotate=True)
    o = rng.uniform(-4, 4, (4000, 3)); d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    h, u = _same_casts(sc, o, d)
    assert h > 100 and u > 50


def test_hip_casts_through_a_hair_mesh_equal_the_oracle(gpu):
    """Scenes of opaque TriMesh nodes only are probed with the opaque-mesh permutation of the traversal (k_cast_batch<kFeatMesh>): in a
    hair-like mesh its node phases end by quorum and the parked lanes resume later (trace_device.h: traverse, DScene::incoherent) —
    hit / miss, node and toi of random rays through the hairball stand-in must still be the oracle's, bit for bit."""
    import ctypes as C
    from nrays_amd import abi
    from tools import standins
    sc, _ = standins.hairball_scene(strands=400)
    flags = (C.c_uint32 * 2)()
    abi.check(abi.load_hip_lib().nrays_debug_scene_flags(sc.device_handle(), flags))
    assert flags[0] == 2 and flags[1] == 1, tuple(flags)  # opaque meshes only, hair-like: the quorum path
    rng = np.random.default_rng(23)
    o = rng.normal(size=(6000, 3)); o *= 3.0 / np.linalg.norm(o, axis=1, keepdims=True)
    t = rng.normal(size=(6000, 3)) * 0.12 + np.array([0.0, 0.1, 0.0])
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    h, _ = _same_casts(sc, o, d)
    assert h > 1500


def test_hip_shadow_queries_equal_the_oracle(gpu):
    """Scene::intersects_ray on the device alone: blocked / lit identical, the colour filter of the transparent nodes
    crossed equal up to the order of its f32 products."""
    rng = np.random.default_rng(5)
    sc, _ = su.mesh_scene(alpha_mapped=True, rotate=True)
    o = rng.uniform(-4, 4, (1500, 3)); d = rng.normal(size=(1500, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = rng.uniform(0.5, 12.0, 1500)
    blocked, filt = nr.shadow_rays(sc, o, d, t)
    n_filtered = 0
    for i in range(len(o)):
        f = oracle.shadow(sc.descriptor, o[i], d[i], t[i])
        assert (f is None) == bool(blocked[i])
        if f is not None:
            assert np.abs(filt[i] - f).max() <= 1e-6
            n_filtered += int((f != 1.0).any())
    assert 0 < blocked.sum() < len(o)


# ---- second fixture set (tests/golden/make_kat_independent2.py): planes, shape AABBs, rotated meshes, coincident triangles
def hip_aabb(scene, i):
    import ctypes as C
    from nrays_amd import abi
    out = (C.c_double * 6)()
    abi.check(abi.load_hip_lib().nrays_debug_node_aabb(scene.device_handle(), i, out))
    return np.array(out[:])


def test_hip_planes_against_independent_fixtures(gpu):
    from tests import test_kat_independent2 as kat2
    kat2.check_planes(hip_cast, "HIP")


def test_device_aabbs_against_support_function_extremes(gpu):
    from tests import test_kat_independent2 as kat2
    kat2.check_aabbs(hip_aabb, "HIP")


def test_hip_rotated_meshes_and_ties_against_independent_fixtures(gpu):
    from tests import test_kat_independent2 as kat2
    kat2.check_meshes(hip_cast, "HIP")
    kat2.check_ties(hip_cast, "HIP")
