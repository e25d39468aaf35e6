"""Planes, shape AABBs under isometries, rotated several-triangle meshes and coincident triangles against fixtures derived
independently of the oracle and of the kernels (tests/golden/make_kat_independent2.py: 60-digit mpmath, support functions,
exact plane / barycentric solutions).  The checkers take the caster, so tests/test_kat_independent_gpu.py runs the same
fixtures through the HIP intersectors (nrays_debug_cast_batch / nrays_debug_node_aabb)."""
import os

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from tools import scenes_util as su

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_independent2.npz")
BALL, CUBOID, CYLINDER, CAPSULE, CONE = 0, 1, 2, 3, 4


def oracle_cast(scene, o, d):
    return oracle.cast(scene.descriptor, [o], [d])


def oracle_aabb(scene, i):
    return np.array(oracle.node_aabb(scene.descriptor, i))


def geometry(kind, prm):
    return {BALL: lambda: nr.Ball(prm[0]), CUBOID: lambda: nr.Cuboid(tuple(prm)), CYLINDER: lambda: nr.Cylinder(prm[0], prm[1]),
            CAPSULE: lambda: nr.Capsule(prm[0], prm[1]), CONE: lambda: nr.Cone(prm[0], prm[1])}[int(kind)]()


def check_planes(cast, label="oracle"):
    """ncollide Plane (examples/loader3d.rs:656, SURVEY B-6): toi, the normal facing the ray, the solid half-space."""
    rows = np.load(FIXTURE)["planes"]
    worst_t = worst_n = 0.0
    hits = 0
    for c in rows:
        n, t, w, solid, o, d, hit, toi, nrm = c[0:3], c[3:6], c[6:9], bool(c[9]), c[10:13], c[13:16], bool(c[16]), c[17], c[18:21]
        node = nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3(tuple(t), tuple(w)), nr.Plane(tuple(n)), None, solid)
        got_hit, out = cast(nr.Scene([node], []), o, d)
        assert bool(got_hit[0]) == hit, c
        if not hit:
            continue
        hits += 1
        worst_t = max(worst_t, abs(out[0, 0] - toi) / max(1.0, toi))
        if toi > 0.0:
            worst_n = max(worst_n, float(np.abs(out[0, 1:4] - nrm).max()))
        assert out[0, 4] == 0  # a plane carries no uv (uv_material.rs falls back to the origin)
    print("%s plane: %d hits of %d, max |toi - exact| / max(1, toi) = %.2e, |n - exact| = %.2e" % (label, hits, len(rows), worst_t, worst_n))
    assert hits >= 40 and worst_t <= 1e-12 and worst_n <= 1e-14


def check_aabbs(aabb, label="oracle"):
    """geometry.bounding_volume(&transform) (src/scene_node.rs:41): the support-function extremes to a few ulps of the box's scale."""
    rows = np.load(FIXTURE)["aabbs"]
    worst = 0.0
    for c in rows:
        kind, prm, t, w, lo, hi = c[0], c[1:4], c[4:7], c[7:10], c[10:13], c[13:16]
        node = nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3(tuple(t), tuple(w)), geometry(kind, prm))
        box = aabb(nr.Scene([node], []), 0)
        scale = max(1.0, float(np.abs(np.concatenate([lo, hi])).max()))
        worst = max(worst, float(np.abs(box - np.concatenate([lo, hi])).max()) / scale)
    print("%s aabb: %d boxes, max |box - exact support extremes| / scale = %.2e" % (label, len(rows), worst))
    assert worst <= 4e-15


def _mesh_scene(two_rows):
    nodes = []
    for r in two_rows:
        V, UV, F, t, w = r[0:24].reshape(8, 3), r[24:40].reshape(8, 2), r[40:58].reshape(6, 3).astype(np.uint32), r[58:61], r[61:64]
        nodes.append(nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3(tuple(t), tuple(w)), nr.TriMesh(V, F, UV)))
    return nr.Scene(nodes, [])


def check_meshes(cast, label="oracle"):
    """TriMesh nodes under rotations (examples/loader3d.rs:695; SURVEY B-8 / B-9): the closest hit over all triangles of both
    nodes — hit / miss, toi, WHICH node, the flat normal towards the ray origin, the interpolated uv."""
    z = np.load(FIXTURE)
    rays, scenes = z["mesh_rays"], z["mesh_scenes"]
    worst = {"toi": 0.0, "n": 0.0, "uv": 0.0}
    hits = 0
    for s in np.unique(rays[:, 0]).astype(int):
        sc = _mesh_scene(scenes[2 * s:2 * s + 2])
        for c in rays[rays[:, 0] == s]:
            o, d, hit, toi, node, n, u, v = c[1:4], c[4:7], bool(c[7]), c[8], int(c[9]), c[10:13], c[13], c[14]
            got_hit, out = cast(sc, o, d)
            assert bool(got_hit[0]) == hit, c
            if not hit:
                continue
            hits += 1
            assert int(out[0, 7]) == node
            worst["toi"] = max(worst["toi"], abs(out[0, 0] - toi) / max(1.0, toi))
            worst["n"] = max(worst["n"], float(np.abs(out[0, 1:4] - n).max()))
            worst["uv"] = max(worst["uv"], abs(out[0, 5] - u), abs(out[0, 6] - v))
    print("%s meshes: %d hits of %d rays, |toi - exact| / max(1, toi) = %.2e, |n - exact| = %.2e, |uv - exact| = %.2e"
          % (label, hits, len(rays), worst["toi"], worst["n"], worst["uv"]))
    assert hits >= 100 and worst["toi"] <= 1e-11 and worst["n"] <= 1e-10 and worst["uv"] <= 1e-10


def check_ties(cast, label="oracle"):
    """Coincident surfaces: the same triangle in node 0 and node 1 — equal toi, DESIGN D-2 returns the smaller node index
    whatever the order in which the BVH reaches them (both orders of the node list are cast)."""
    rows = np.load(FIXTURE)["ties"]
    for c in rows:
        A, B, C_, t, w, o, d, toi = c[0:3], c[3:6], c[6:9], c[9:12], c[12:15], c[15:18], c[18:21], c[21]
        mesh = lambda: nr.TriMesh(np.stack([A, B, C_]), np.array([[0, 1, 2]], dtype=np.uint32), np.zeros((3, 2)))
        mk = lambda m: nr.SceneNode(m, 0.0, 0.0, 1.0, 1.0, nr.Isometry3(tuple(t), tuple(w)), mesh())
        for mats in ((su.default_material(), nr.NormalMaterial()), (nr.NormalMaterial(), su.default_material())):
            got_hit, out = cast(nr.Scene([mk(mats[0]), mk(mats[1])], []), o, d)
            assert bool(got_hit[0]) and int(out[0, 7]) == 0
            assert abs(out[0, 0] - toi) <= 1e-11 * max(1.0, toi)
    print("%s ties: %d coincident pairs, node 0 wins in both list orders" % (label, len(rows)))


def test_planes_against_independent_fixtures():
    check_planes(oracle_cast)


def test_shape_aabbs_against_support_function_extremes():
    check_aabbs(oracle_aabb)


def test_rotated_meshes_against_independent_fixtures():
    check_meshes(oracle_cast)


def test_coincident_triangles_follow_the_stated_tie_rule():
    check_ties(oracle_cast)
