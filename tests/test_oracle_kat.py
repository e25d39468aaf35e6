"""Known-answer tests pinning the CPU oracle (oracle/nrays_oracle.c) to closed-form geometry.

The reference holds no tests, golden vectors or images for this path and cannot be
run here, so parity is "unpinned"; these analytic KATs are what pins the oracle:
every expected value below is derived by hand from the reference's formulas
(/root/reference/src/*.rs and the ncollide3d behaviour restated in SURVEY Appendix B).
"""
import math

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from tools import scenes_util as su

ISO = nr.Isometry3


def one(geom, material=None, iso=None, solid=False, alpha=1.0, refl=(0.0, 0.0), refr=1.0):
    return nr.SceneNode(material or su.default_material(), refl[0], refl[1], alpha, refr, iso or ISO(), geom, None, solid)


def cast1(nodes, o, d):
    sc = nr.Scene(nodes, [])
    hit, out = oracle.cast(sc.descriptor, [o], [d])
    return (out[0] if hit[0] else None)


# ------------------------------------------------------------------ ball (SURVEY B-4) --------
def test_ball_outside_hit_uv():
    r = cast1([one(nr.Ball(1.0))], (0, 0, -5), (0, 0, 1))
    assert r[0] == pytest.approx(4.0, abs=1e-15)
    assert np.allclose(r[1:4], (0, 0, -1), atol=1e-15)
    assert r[4] == 1 and r[5] == pytest.approx(0.25) and r[6] == pytest.approx(0.5)


def test_ball_translation_and_rotation_ignored():
    r = cast1([one(nr.Ball(0.5), iso=ISO((1, 2, 3), (0.3, 0.2, 0.1)))], (1, 2, -5), (0, 0, 1))
    assert r[0] == pytest.approx(7.5) and np.allclose(r[1:4], (0, 0, -1))


def test_ball_inside_non_solid_exit_point_normal_faces_origin():
    r = cast1([one(nr.Ball(1.0))], (0, 0, 0), (0, 0, 1))
    assert r[0] == pytest.approx(1.0) and np.allclose(r[1:4], (0, 0, -1))
    assert r[5] == pytest.approx(0.75)  # uv of the outward normal (0,0,1)


def test_ball_inside_solid_toi_zero():
    r = cast1([one(nr.Ball(1.0), solid=True)], (0.2, 0, 0), (0, 0, 1))
    assert r[0] == 0.0


def test_ball_behind_and_grazing_miss():
    assert cast1([one(nr.Ball(1.0))], (0, 0, 5), (0, 0, 1)) is None
    assert cast1([one(nr.Ball(1.0))], (1.0000001, 0, -5), (0, 0, 1)) is None


# ------------------------------------------------------------------ cuboid (B-5) -------------
def test_cuboid_face_normal_and_uv():
    r = cast1([one(nr.Cuboid((1, 2, 3)))], (-5, 0.5, 1), (1, 0, 0))
    assert r[0] == pytest.approx(4.0) and np.allclose(r[1:4], (-1, 0, 0))
    assert r[5] == pytest.approx(2.5 / 4.0) and r[6] == pytest.approx(4.0 / 6.0)
    r = cast1([one(nr.Cuboid((1, 2, 3)))], (0.5, 9, -1), (0, -1, 0))
    assert r[0] == pytest.approx(7.0) and np.allclose(r[1:4], (0, 1, 0))
    assert r[5] == pytest.approx(2.0 / 6.0) and r[6] == pytest.approx(1.5 / 2.0)  # face y: (z, x)


def test_cuboid_inside_exit_normal_faces_origin():
    r = cast1([one(nr.Cuboid((1, 1, 1)))], (0, 0, 0), (1, 0, 0))
    assert r[0] == pytest.approx(1.0) and np.allclose(r[1:4], (-1, 0, 0))
    r = cast1([one(nr.Cuboid((1, 1, 1)), solid=True)], (0, 0, 0), (1, 0, 0))
    assert r[0] == 0.0 and np.allclose(r[1:4], 0)


def test_cuboid_rotated_90_about_z():
    # Isometry3::new takes a scaled AXIS-ANGLE (SURVEY B-10): half extents (1,2,3) turned by 90 deg about z
    r = cast1([one(nr.Cuboid((1, 2, 3)), iso=ISO((0, 0, 0), (0, 0, math.pi / 2)))], (-5, 0, 0), (1, 0, 0))
    assert r[0] == pytest.approx(3.0) and np.allclose(r[1:4], (-1, 0, 0), atol=1e-12)


def test_axis_angle_is_not_euler():
    # rotate by 120 deg about (1,1,1)/sqrt3: x->y->z->x, so the long axis (z, he=3) ends up along x
    w = np.array([1.0, 1.0, 1.0]) / math.sqrt(3) * (2 * math.pi / 3)
    r = cast1([one(nr.Cuboid((1, 2, 3)), iso=ISO((0, 0, 0), w))], (-9, 0, 0), (1, 0, 0))
    assert r[0] == pytest.approx(6.0, abs=1e-12)


# ------------------------------------------------------------------ plane (B-6) --------------
def test_plane_both_sides_and_parallel():
    pl = [one(nr.Plane((0, 1, 0)), iso=ISO((0, -3, 0)))]
    r = cast1(pl, (0, 0, 0), (0, -1, 0))
    assert r[0] == pytest.approx(3.0) and np.allclose(r[1:4], (0, 1, 0)) and r[4] == 0
    r = cast1(pl, (0, -5, 0), (0, 1, 0))
    assert r[0] == pytest.approx(2.0) and np.allclose(r[1:4], (0, -1, 0))
    assert cast1(pl, (0, 0, 0), (1, 0, 0)) is None
    assert cast1(pl, (0, 0, 0), (0, 1, 0)) is None


# ------------------------------------------------------------------ triangle / trimesh (B-8,9)
def tri_mesh(flip=False):
    pts = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    idx = [[0, 2, 1]] if flip else [[0, 1, 2]]
    uvs = [[0, 0], [1, 0], [0, 1]]
    return nr.TriMesh(pts, idx, uvs)


@pytest.mark.parametrize("flip", [False, True])
def test_triangle_two_sided_normal_faces_origin_and_barycentric_uv(flip):
    m = [one(tri_mesh(flip))]
    r = cast1(m, (0.25, 0.5, -2), (0, 0, 1))
    assert r[0] == pytest.approx(2.0) and np.allclose(r[1:4], (0, 0, -1))
    assert r[5] == pytest.approx(0.25) and r[6] == pytest.approx(0.5)
    r = cast1(m, (0.25, 0.5, 3), (0, 0, -1))
    assert r[0] == pytest.approx(3.0) and np.allclose(r[1:4], (0, 0, 1))
    assert cast1(m, (0.75, 0.75, -2), (0, 0, 1)) is None  # outside the hypotenuse
    assert cast1(m, (0.25, 0.5, -2), (1, 0, 0)) is None  # parallel


def test_trimesh_under_isometry():
    r = cast1([one(tri_mesh(), iso=ISO((0, 0, 4), (0, math.pi, 0)))], (-0.25, 0.5, -2), (0, 0, 1))
    assert r[0] == pytest.approx(6.0) and np.allclose(r[1:4], (0, 0, -1), atol=1e-12)
    assert r[5] == pytest.approx(0.25) and r[6] == pytest.approx(0.5)


# ------------------------------------------------------------------ cylinder / cone / capsule (D-3)
def test_cylinder_side_cap_inside():
    cyl = [one(nr.Cylinder(1.0, 0.5))]
    r = cast1(cyl, (-5, 0.3, 0), (1, 0, 0))
    assert r[0] == pytest.approx(4.5) and np.allclose(r[1:4], (-1, 0, 0))
    r = cast1(cyl, (0.2, 5, 0.1), (0, -1, 0))
    assert r[0] == pytest.approx(4.0) and np.allclose(r[1:4], (0, 1, 0))
    r = cast1(cyl, (0, 0, 0), (1, 0, 0))  # inside, non-solid: exit point, OUTWARD normal
    assert r[0] == pytest.approx(0.5) and np.allclose(r[1:4], (1, 0, 0))
    r = cast1(cyl, (0, 0, 0), (0, 1, 0))
    assert r[0] == pytest.approx(1.0) and np.allclose(r[1:4], (0, 1, 0))
    assert cast1(cyl, (-5, 1.2, 0), (1, 0, 0)) is None
    assert cast1(cyl, (0.6, 5, 0), (0, -1, 0)) is None
    r = cast1([one(nr.Cylinder(1.0, 0.5), solid=True)], (0, 0, 0), (1, 0, 0))
    assert r[0] == 0.0


def test_cylinder_oblique_through_cap_then_side():
    d = np.array([1.0, -1.0, 0.0]) / math.sqrt(2)
    r = cast1([one(nr.Cylinder(1.0, 0.5))], (-1.2, 2.0, 0), d)  # enters the top cap at x=-0.2
    assert r[0] == pytest.approx(math.sqrt(2)) and np.allclose(r[1:4], (0, 1, 0))


def test_cone_side_base_apex_inside():
    cone = [one(nr.Cone(1.0, 1.0))]  # apex (0,1,0), base disc r=1 at y=-1; radius(y) = (1-y)/2
    r = cast1(cone, (-5, 0, 0), (1, 0, 0))
    assert r[0] == pytest.approx(4.5)
    assert np.allclose(r[1:4], np.array([-0.5, 0.25, 0]) / np.linalg.norm([-0.5, 0.25, 0]))
    r = cast1(cone, (0.2, -5, 0), (0, 1, 0))
    assert r[0] == pytest.approx(4.0) and np.allclose(r[1:4], (0, -1, 0))
    r = cast1(cone, (0.25, 5, 0), (0, -1, 0))  # steep ray (A < 0 branch): hits the side at y = 0.5
    assert r[0] == pytest.approx(4.5)
    r = cast1(cone, (0, -0.5, 0), (1, 0, 0))  # inside: exit at x = 0.75, outward normal
    assert r[0] == pytest.approx(0.75) and r[1] > 0 and r[2] > 0
    assert cast1(cone, (-5, 1.5, 0), (1, 0, 0)) is None
    assert cast1(cone, (0.9, 5, 0), (0, -1, 0))[0] == pytest.approx(5.8)  # radius .9 at y=-.8
    assert cast1(cone, (1.1, 5, 0), (0, -1, 0)) is None


def test_capsule_side_and_caps():
    cap = [one(nr.Capsule(1.0, 0.5))]
    r = cast1(cap, (-5, 0, 0), (1, 0, 0))
    assert r[0] == pytest.approx(4.5) and np.allclose(r[1:4], (-1, 0, 0))
    r = cast1(cap, (0, 5, 0), (0, -1, 0))
    assert r[0] == pytest.approx(3.5) and np.allclose(r[1:4], (0, 1, 0))
    r = cast1(cap, (-5, 1.2, 0), (1, 0, 0))
    assert r[0] == pytest.approx(5 - math.sqrt(0.25 - 0.04))
    r = cast1(cap, (0, 0, 0), (0, 1, 0))  # inside: exit through the top ball
    assert r[0] == pytest.approx(1.5) and np.allclose(r[1:4], (0, 1, 0))
    assert cast1(cap, (-5, 1.6, 0), (1, 0, 0)) is None


# ------------------------------------------------------------------ node AABBs ---------------
def test_world_aabbs():
    sc = nr.Scene([one(nr.Ball(2.0), iso=ISO((1, 2, 3), (0.5, 0, 0))),
                   one(nr.Cuboid((1, 2, 3)), iso=ISO((0, 0, 0), (0, 0, math.pi / 2))),
                   one(nr.Cylinder(1.0, 0.5), iso=ISO((0, 1, 0))),
                   one(nr.Cone(1.0, 0.5)), one(nr.Capsule(1.0, 0.5)), one(tri_mesh(), iso=ISO((0, 0, 4)))], [])
    d = sc.descriptor
    assert np.allclose(oracle.node_aabb(d, 0), (-1, 0, 1, 3, 4, 5))
    assert np.allclose(oracle.node_aabb(d, 1), (-2, -1, -3, 2, 1, 3))
    assert np.allclose(oracle.node_aabb(d, 2), (-0.5, 0, -0.5, 0.5, 2, 0.5))
    assert np.allclose(oracle.node_aabb(d, 3), (-0.5, -1, -0.5, 0.5, 1, 0.5))
    assert np.allclose(oracle.node_aabb(d, 4), (-0.5, -1.5, -0.5, 0.5, 1.5, 0.5))
    assert np.allclose(oracle.node_aabb(d, 5), (0, 0, 4, 1, 1, 4))


# ------------------------------------------------------------------ BVT == brute force -------
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bvt_equals_bruteforce(seed):
    sc, _ = su.random_shapes_scene(seed, n=30)
    rng = np.random.default_rng(100 + seed)
    o = rng.uniform(-12, 12, (1500, 3))
    d = rng.normal(size=(1500, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    h1, r1 = oracle.cast(sc.descriptor, o, d)
    h2, r2 = oracle.cast(sc.descriptor, o, d, bruteforce=True)
    assert (h1 == h2).all() and h1.sum() > 200
    assert np.array_equal(r1[h1], r2[h1])


# ------------------------------------------------------------------ textures (texture2d.rs:207-256)
def tex2x2(interp=nr.Interpolation.Bilinear, overflow=nr.Overflow.Wrap):
    px = np.array([[[0, 0, 0, 255], [255, 0, 0, 255]], [[0, 255, 0, 255], [255, 255, 255, 51]]], dtype=np.uint8)
    return nr.Texture2d(nr.ImageData(px), interp, overflow)


def test_texture_bilinear_taps_and_wrap():
    t = tex2x2()
    assert np.allclose(oracle.tex_sample(t, 0.0, 0.0), (0, 0, 0, 1))
    assert np.allclose(oracle.tex_sample(t, 0.5, 0.0), (0.5, 0, 0, 1))
    assert np.allclose(oracle.tex_sample(t, 0.5, 0.5), (0.5, 0.5, 0.25, 0.8))
    assert np.allclose(oracle.tex_sample(t, 1.25, -0.75), oracle.tex_sample(t, 0.25, 0.25))  # wrap: % 1, +1 if < 0
    assert np.allclose(oracle.tex_sample(t, 1.0, 1.0), (0, 0, 0, 1))  # 1.0 % 1.0 == 0


def test_texture_clamp_and_nearest():
    t = tex2x2(overflow=nr.Overflow.ClampToEdges)
    assert np.allclose(oracle.tex_sample(t, 7.0, 9.0), (1, 1, 1, 0.2))  # taps clamped (D-6)
    assert np.allclose(oracle.tex_sample(t, -3.0, 0.0), (0, 0, 0, 1))
    n = tex2x2(interp=nr.Interpolation.Nearest)
    assert np.allclose(oracle.tex_sample(n, 0.6, 0.4), (1, 0, 0, 1))


def test_image_decode_conventions():
    rows = np.array([[[10, 20, 30]], [[40, 50, 60]]], dtype=np.uint8)  # 2 rows x 1 col RGB, top row first
    img = nr.ImageData.from_image_rows(rows)
    assert (img.pixels[0, 0] == (40, 50, 60, 255)).all()  # row 0 = bottom (texture2d.rs:99-107)
    op = nr.ImageData.from_image_rows(rows, opacity=True)
    assert (op.pixels[1, 0] == (255, 255, 255, 10)).all()  # depth 3 opacity uses r (texture2d.rs:146-148)


# ------------------------------------------------------------------ shading KATs -------------
def render1(scene, cam, w=8, h=8, **kw):
    p, _ = su.camera_params(cam, w, h, **kw)
    return oracle.render(scene.descriptor, p, 1)


def test_phong_normal_incidence():
    # white plane y=0 seen straight down from the light position: Ka + lc*(Kd*1 + Ks*1^Ns) = .1 + 1 + 1
    sc = nr.Scene([one(nr.Plane((0, 1, 0)))], [nr.Light((0, 5, 0), 0.0, 1, (1, 1, 1))])
    cam = dict(eye=(0, 5, 0), at=(0, 0, 1e-9), fovy=30.0)
    img, st = render1(sc, cam, 2, 2)
    # pixel corner (1,1) of a 2x2 image is the exact image centre (NDC 0,0)
    assert np.allclose(img[1, 1], 2.1, atol=1e-5)
    assert st.rays_primary == 4 and st.rays_shadow == 4 and st.rays_reflection == 0


def test_background_and_camera_directions():
    # inside a huge non-solid NormalMaterial ball centred on the eye: colour = (1 - d)/2 (normal faces the origin)
    eye = (1.0, 2.0, 3.0)
    sc = nr.Scene([one(nr.Ball(100.0), material=nr.NormalMaterial(), iso=ISO(eye))], [])
    cam = dict(eye=eye, at=(1.0, 2.0, 13.0), fovy=60.0)
    w, h = 6, 4
    img, _ = render1(sc, cam, w, h)
    d = 1.0 - 2.0 * img.astype(np.float64)
    t = math.tan(math.radians(30.0))
    for j in range(h):
        for i in range(w):
            dx, dy = (i / w - 0.5) * 2, -(j / h - 0.5) * 2  # pixel CORNER, scene.rs:81-82
            # look_at_rh towards +z with up +y: camera x axis = up x z_cam = -x_world
            v = np.array([-dx * t * (w / h), dy * t, 1.0])
            assert np.allclose(d[j, i], v / np.linalg.norm(v), atol=2e-6)
    empty = nr.Scene([], [], (0.25, 0.5, 0.75))
    img, _ = render1(empty, cam, 3, 2)
    assert np.allclose(img, (0.25, 0.5, 0.75))


@pytest.mark.parametrize("att,gens", [(0.25, 4), (0.2, 5), (0.5, 2), (1.0, 1)])
def test_reflection_energy_ladder(att, gens):
    # two facing mirrors: every reflection hits again, so the count is set by the energy rule (scene.rs:204)
    mat = nr.NormalMaterial()
    nodes = [one(nr.Plane((0, 1, 0)), material=mat, iso=ISO((0, -1, 0)), refl=(0.5, att)),
             one(nr.Plane((0, -1, 0)), material=mat, iso=ISO((0, 1, 0)), refl=(0.5, att))]
    sc = nr.Scene(nodes, [])
    cam = dict(eye=(0, 0, 0), at=(0, -1, 1), fovy=20.0)
    img, st = render1(sc, cam, 2, 2)
    assert st.rays_reflection == 4 * gens


def test_max_depth_caps_generations():
    mat = nr.NormalMaterial()
    nodes = [one(nr.Plane((0, 1, 0)), material=mat, iso=ISO((0, -1, 0)), refl=(0.5, 0.0)),
             one(nr.Plane((0, -1, 0)), material=mat, iso=ISO((0, 1, 0)), refl=(0.5, 0.0))]
    sc = nr.Scene(nodes, [])
    cam = dict(eye=(0, 0, 0), at=(0, -1, 1), fovy=20.0)
    _, st = render1(sc, cam, 2, 2, max_depth=3)
    assert st.rays_reflection == 12
    _, st = render1(sc, cam, 2, 2)  # attenuation 0: only the hard cap of 64 generations stops it
    assert st.rays_reflection == 4 * 64


def test_transparency_blend_and_straight_refraction():
    # alpha .25 normal-material plane, refr 1.0: ray continues undeviated into the white background
    sc = nr.Scene([one(nr.Plane((0, 0, -1)), material=nr.NormalMaterial(), iso=ISO((0, 0, 5)), alpha=0.25)], [])
    cam = dict(eye=(0, 0, 0), at=(0, 0, 1), fovy=20.0)
    img, st = render1(sc, cam, 2, 2)
    obj = (1.0 + np.array([0, 0, -1.0])) / 2
    assert np.allclose(img[1, 1], obj * 0.25 + 1.0 * 0.75, atol=1e-6)
    assert st.rays_refraction == 4


def test_transparent_shadow_filter():
    # a semi-transparent quad between the light and the floor filters the light: filter = Ka*(1-alpha)
    quad = nr.TriMesh([[-1, 2, -1], [1, 2, -1], [1, 2, 1], [-1, 2, 1]], [[0, 1, 2], [0, 2, 3]], [[0, 0], [1, 0], [1, 1], [0, 1]])
    glass = nr.PhongMaterial((0.5, 0.25, 1.0), (0, 0, 0), (0, 0, 0), None, None, 10.0)
    sc = nr.Scene([one(quad, material=glass, alpha=0.4)], [])
    assert np.allclose(oracle.shadow(sc.descriptor, (0, 0, 0), (0, 1, 0), 5.0), np.float32([0.5, 0.25, 1.0]) * np.float32(0.6))
    assert np.allclose(oracle.shadow(sc.descriptor, (0, 0, 0), (0, 1, 0), 1.5), (1, 1, 1))  # beyond maxtoi
    opaque = nr.Scene([one(quad, material=glass, alpha=1.0)], [])
    assert oracle.shadow(opaque.descriptor, (0, 0, 0), (0, 1, 0), 5.0) is None
    assert oracle.shadow(opaque.descriptor, (0, 0, 0), (0, 1, 0), 2.0) is None  # toi <= maxtoi is inclusive
    assert oracle.shadow(opaque.descriptor, (0, 0, 0), (0, 1, 0), 1.999) is not None
    # UVMaterial on a shape without uvs has w = 0 (uv_material.rs:18): fully transparent, black filter
    ghost = nr.Scene([one(nr.Plane((0, -1, 0)), material=nr.UVMaterial(), iso=ISO((0, 2, 0)))], [])
    assert np.allclose(oracle.shadow(ghost.descriptor, (0, 0, 0), (0, 1, 0), 5.0), (0, 0, 0))


# ------------------------------------------------------------------ RNG & tiling -------------
def test_rng_reproducible_uniform():
    vals = np.array([oracle.rng_u01(1, p, s, d) for p in range(40) for s in range(4) for d in range(3)])
    assert (vals >= 0).all() and (vals < 1).all() and abs(vals.mean() - 0.5) < 0.06
    assert oracle.rng_u01(1, 7, 2, 1) == oracle.rng_u01(1, 7, 2, 1)
    assert oracle.rng_u01(1, 7, 2, 1) != oracle.rng_u01(2, 7, 2, 1)


def test_area_light_and_aa_are_deterministic():
    sc, cam = su.primitives_scene(light_radius=0.1, nsample=10)
    p, _ = su.camera_params(cam, 24, 18, spp=2, window=1.0, seed=5)
    a, st = oracle.render(sc.descriptor, p, 1)
    b, _ = oracle.render(sc.descriptor, p, 4)  # thread count must not matter (counter-based RNG)
    assert np.array_equal(a, b)
    assert st.rays_primary == 24 * 18 * 2
    p2, _ = su.camera_params(cam, 24, 18, spp=2, window=1.0, seed=6)
    c, _ = oracle.render(sc.descriptor, p2, 1)
    assert not np.array_equal(a, c)


def test_tiled_render_equals_full_frame_rows():
    sc, cam = su.balls_scene(tex_size=(64, 32))
    w, h = 40, 37
    full, _ = oracle.render(sc.descriptor, su.camera_params(cam, w, h)[0], 2)
    for owner in range(3):
        p, _ = su.camera_params(cam, w, h, band_rows=8, band_owner=owner, band_owners=3)
        tile, _ = oracle.render(sc.descriptor, p, 2)
        for j in range(h):
            band = j // 8
            if band % 3 == owner:
                assert np.array_equal(tile[(band // 3) * 8 + j % 8], full[j])
