"""The oracle's analytic ray casts against fixtures derived INDEPENDENTLY of it (tests/golden/make_kat_independent.py:
60-digit mpmath bisection on point membership, no formula shared with oracle/nrays_oracle.c), and certified through the
shapes' SUPPORT MAPS — the representation ncollide3d's GJK ray cast works on for cone / cylinder / capsule, where the
oracle and the HIP kernels use closed forms instead (DESIGN D-3):

  toi      equals the fixture's (the first / last parameter at which o + t d belongs to the shape);
  normal   is a unit vector whose plane through the hit point SUPPORTS the shape, h(n) = n . x — on faces and smooth
           sides that makes it THE outward normal, on rims it puts it inside the normal cone, which is all a GJK ray
           cast guarantees there; for outside origins it faces the ray and the support-plane bound
           (n.o - h(n)) / (-n.d), the quantity a support-map ray cast maximises over n, reproduces the toi.

The printed maxima are the D-3 deviation as numbers (DESIGN.md 2 quotes them)."""
import os

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from tools import scenes_util as su

BALL, CUBOID, CYLINDER, CAPSULE, CONE = 0, 1, 2, 3, 4
NAMES = {BALL: "ball", CUBOID: "cuboid", CYLINDER: "cylinder", CAPSULE: "capsule", CONE: "cone"}
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_independent.npz")


def support(kind, prm, n):
    """h(n) = max over the shape of n . p (local frame)."""
    if kind == BALL:
        return prm[0] * np.linalg.norm(n)
    if kind == CUBOID:
        return float(np.sum(np.asarray(prm) * np.abs(n)))
    hh, r = prm[0], prm[1]
    rad = np.hypot(n[0], n[2])
    if kind == CYLINDER:
        return hh * abs(n[1]) + r * rad
    if kind == CAPSULE:
        return hh * abs(n[1]) + r * np.linalg.norm(n)
    return max(hh * n[1], -hh * n[1] + r * rad)  # cone: apex (0, hh, 0) or the base circle


def rotation(w):
    th = np.linalg.norm(w)
    if th == 0.0:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def geometry(kind, prm):
    return {BALL: lambda: nr.Ball(prm[0]), CUBOID: lambda: nr.Cuboid(tuple(prm)), CYLINDER: lambda: nr.Cylinder(prm[0], prm[1]),
            CAPSULE: lambda: nr.Capsule(prm[0], prm[1]), CONE: lambda: nr.Cone(prm[0], prm[1])}[kind]()


def oracle_cast(scene, o, d):
    return oracle.cast(scene.descriptor, [o], [d])


def check_shape_cases(kind, cast, label="oracle"):
    """`cast(scene, o, d)` -> (hit mask, (1, 8) record): the oracle here, the HIP intersectors in test_kat_independent_gpu.py."""
    cases = np.load(FIXTURE)["cases"]
    cases = cases[cases[:, 0] == kind]
    assert len(cases) == 120
    worst = {"toi_rel": 0.0, "plane_residual": 0.0, "bound_rel": 0.0, "unit": 0.0}
    n_hit = n_rim = 0
    for c in cases:
        prm, t, w, solid = c[1:4], c[4:7], c[7:10], bool(c[10])
        o, d, hit, toi, was_inside = c[11:14], c[14:17], bool(c[17]), c[18], bool(c[19])
        node = nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3(tuple(t), tuple(w)), geometry(kind, prm), None, solid)
        got_hit, out = cast(nr.Scene([node], []), o, d)
        assert bool(got_hit[0]) == hit, (NAMES[kind], c)
        if not hit:
            continue
        n_hit += 1
        size = float(max(prm))
        got_toi, n = out[0, 0], out[0, 1:4]
        worst["toi_rel"] = max(worst["toi_rel"], abs(got_toi - toi) / max(1.0, toi))
        if was_inside and solid:
            assert got_toi == 0.0
            continue
        R = rotation(w)
        x = R.T @ (o + d * got_toi - t)
        nl = R.T @ n
        ol, dl = R.T @ (o - t), R.T @ d
        worst["unit"] = max(worst["unit"], abs(np.linalg.norm(n) - 1.0))
        # ball and cuboid report the RAY-FACING normal for interior origins (SURVEY B-4 / B-5); the convex closed forms
        # the exit point's outward normal (DESIGN D-3)
        outward = -nl if (was_inside and kind in (BALL, CUBOID)) else nl
        worst["plane_residual"] = max(worst["plane_residual"], abs(support(kind, prm, outward) - outward @ x) / size)
        if was_inside:
            assert (nl @ dl < 0) if kind in (BALL, CUBOID) else (nl @ dl > 0)
        else:
            assert nl @ dl < 0
            bound = (nl @ ol - support(kind, prm, nl)) / (-(nl @ dl))
            worst["bound_rel"] = max(worst["bound_rel"], abs(bound - toi) / max(1.0, toi))
            if kind in (CYLINDER, CONE) and abs(abs(x[1]) - prm[0]) < 1e-9 * size:
                n_rim += 1
    print("%s %s: %d hits, max |toi - exact| / max(1, toi) = %.2e, supporting-plane residual / size = %.2e, "
          "support-plane bound vs toi = %.2e, | |n| - 1 | = %.2e" % (label, NAMES[kind], n_hit, worst["toi_rel"], worst["plane_residual"], worst["bound_rel"], worst["unit"]))
    assert n_hit >= 80
    assert worst["toi_rel"] <= 1e-11 and worst["plane_residual"] <= 1e-9 and worst["bound_rel"] <= 1e-9 and worst["unit"] <= 1e-12


@pytest.mark.parametrize("kind", [BALL, CUBOID, CYLINDER, CAPSULE, CONE])
def test_shape_casts_against_independent_fixtures(kind):
    check_shape_cases(kind, oracle_cast)


def test_triangle_casts_against_independent_fixtures():
    check_triangle_cases(oracle_cast)


def check_triangle_cases(cast, label="oracle"):
    """ncollide triangle_ray_intersection + TriMesh uv interpolation (SURVEY B-8 / B-9, reference call site
    examples/loader3d.rs:695) against exact plane / barycentric solutions: toi, the flat normal facing the ray origin,
    the interpolated uv."""
    tris = np.load(FIXTURE)["triangles"]
    worst = {"toi_rel": 0.0, "normal": 0.0, "uv": 0.0}
    hits = 0
    for c in tris:
        A, B, C_, uv = c[0:3], c[3:6], c[6:9], c[9:15].reshape(3, 2)
        t, w, o, d = c[15:18], c[18:21], c[21:24], c[24:27]
        hit, toi, n, u, v = bool(c[27]), c[28], c[29:32], c[32], c[33]
        mesh = nr.TriMesh(np.stack([A, B, C_]), np.array([[0, 1, 2]], dtype=np.uint32), uv)
        node = nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3(tuple(t), tuple(w)), mesh)
        got_hit, out = cast(nr.Scene([node], []), o, d)
        assert bool(got_hit[0]) == hit
        if not hit:
            continue
        hits += 1
        worst["toi_rel"] = max(worst["toi_rel"], abs(out[0, 0] - toi) / max(1.0, toi))
        worst["normal"] = max(worst["normal"], float(np.abs(out[0, 1:4] - n).max()))
        assert out[0, 4] == 1
        worst["uv"] = max(worst["uv"], abs(out[0, 5] - u), abs(out[0, 6] - v))
    print(label + " triangle: %d hits of %d, max |toi - exact| / max(1, toi) = %.2e, |n - exact| = %.2e, |uv - exact| = %.2e"
          % (hits, len(tris), worst["toi_rel"], worst["normal"], worst["uv"]))
    assert hits >= 50 and worst["toi_rel"] <= 1e-11 and worst["normal"] <= 1e-10 and worst["uv"] <= 1e-10
