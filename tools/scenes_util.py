"""Scene builders shared by the tests, __graft_entry__.smoke() and bench.py.

They build the BASELINE.json configurations (and small variants of them) through the host-side
mirror of the reference API (nrays_amd.scene), so the same descriptor feeds the HIP path and the
oracle.  Scene contents follow /root/reference/scenes/{balls,primitives}.scene; assets the
reference downloads (media/globe.png, OBJ meshes — SURVEY F7) are procedural stand-ins.
"""
import math

import numpy as np

import nrays_amd as nr


def f32_exact(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def globe_texture(w=1024, h=512):
    """Stand-in for scenes/media/globe.png (absent from the reference tree): lat/long checker plus
    gradients, RGB8, row 0 = bottom."""
    yy, xx = np.mgrid[0:h, 0:w]
    tex = np.empty((h, w, 4), dtype=np.uint8)
    chk = ((xx * 24 // w) + (yy * 12 // h)) % 2
    tex[..., 0] = (40 + 180 * chk).astype(np.uint8)
    tex[..., 1] = (xx * 255 // (w - 1)).astype(np.uint8)
    tex[..., 2] = (yy * 255 // (h - 1)).astype(np.uint8)
    tex[..., 3] = 255
    return nr.Texture2d(nr.ImageData(tex), nr.Interpolation.Bilinear, nr.Overflow.Wrap)


def default_material():
    """`white` of loader3d.rs:226-233: Ka .1, Kd 1, Ks 1, Ns 100."""
    return nr.PhongMaterial((0.1, 0.1, 0.1), (1, 1, 1), (1, 1, 1), None, None, 100.0)


def balls_scene(refl=(0.2, 0.25), tex_size=(1024, 512)):
    """scenes/balls.scene with the BASELINE "4 bounces" variant `refl 0.2 0.25` (energy 1 -> .75 -> .5 ->
    .25 -> 0: exactly four reflection generations; the file as shipped, refl 0.2 0.2, gives five)."""
    globe = nr.PhongMaterial((1, 1, 1), (1, 1, 1), (1, 1, 1), globe_texture(*tex_size), None, 100.0)  # mtl.rs:160 Ks=1
    iso = nr.Isometry3
    nodes = [
        nr.SceneNode(nr.UVMaterial(), refl[0], refl[1], 1.0, 1.0, iso((-2.1, 0, 0)), nr.Ball(1.0)),
        nr.SceneNode(nr.NormalMaterial(), refl[0], refl[1], 1.0, 1.0, iso((2.1, 0, 0)), nr.Ball(1.0)),
        nr.SceneNode(globe, refl[0], refl[1], 1.0, 1.0, iso((0, 0, 0)), nr.Ball(1.0)),
    ]
    lights = [nr.Light((0.0, 10.0, 0.0), 0.0, 1, (1, 1, 1))]
    cam = dict(eye=(0.0, 5.0, -10.0), at=(0.0, 0.0, 0.0), fovy=45.0)
    return nr.Scene(nodes, lights, (1, 1, 1)), cam


def primitives_scene(light_radius=0.1, nsample=10):
    """scenes/primitives.scene: ball, transparent box / cone / cylinder, reflective plane."""
    def mtl(ka, kd, d=1.0):
        return nr.PhongMaterial(ka, kd, (1, 1, 1), None, None, 100.0), d
    t_red, a_red = mtl((0.0, 0.0, 0.1), (0.0, 0.0, 1.0), 0.2)      # basic_materials.mtl "transparent_red"
    t_blue, a_blue = mtl((0.1, 0.0, 0.0), (1.0, 0.0, 0.0), 0.2)    # "transparent_blue"
    t_green, a_green = mtl((0.0, 0.1, 0.0), (0.0, 1.0, 0.0), 0.2)  # "transparent_green"
    white = default_material()
    iso = nr.Isometry3
    nodes = [
        nr.SceneNode(white, 0.0, 0.0, 1.0, 1.5, iso((-2.1, 0, 0)), nr.Ball(1.0)),
        nr.SceneNode(t_red, 0.0, 0.0, a_red, 1.5, iso((2.1, 0, 0)), nr.Cuboid((1, 1, 1))),
        nr.SceneNode(t_blue, 0.0, 0.0, a_blue, 1.5, iso((0, -2.1, 0)), nr.Cone(1.0, 1.0)),
        nr.SceneNode(t_green, 0.0, 0.0, a_green, 1.5, iso((0, 2.1, 1.0)), nr.Cylinder(1.0, 1.0)),
        nr.SceneNode(white, 0.2, 0.5, 1.0, 1.0, iso((0, -3.0, 0)), nr.Plane((0, 1, 0))),
    ]
    lights = [nr.Light((0.0, 0.0, 0.0), light_radius, nsample, (1, 1, 1))]
    cam = dict(eye=(0.0, 5.0, -20.0), at=(0.0, 0.0, 0.0), fovy=45.0)
    return nr.Scene(nodes, lights, (1, 1, 1)), cam


def torus_mesh(nu=48, nv=24, R=1.5, r=0.6):
    """Closed triangle mesh with uvs; vertices rounded to f32 like obj.rs:197-205."""
    u = np.linspace(0, 2 * math.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * math.pi, nv, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    x = (R + r * np.cos(vv)) * np.cos(uu)
    y = r * np.sin(vv)
    z = (R + r * np.cos(vv)) * np.sin(uu)
    pts = f32_exact(np.stack([x, y, z], -1).reshape(-1, 3))
    uvs = f32_exact(np.stack([uu / (2 * math.pi) * 4.0, vv / (2 * math.pi) * 2.0], -1).reshape(-1, 2))
    idx = []
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv
            idx += [(a, b, c), (a, c, d)]
    return pts, np.asarray(idx, dtype=np.uint32), uvs


def checker_texture(n=64, cells=8, alpha_holes=False):
    yy, xx = np.mgrid[0:n, 0:n]
    chk = ((xx * cells // n) + (yy * cells // n)) % 2
    tex = np.empty((n, n, 4), dtype=np.uint8)
    if alpha_holes:  # opacity map: (1,1,1,a) — texture2d.rs:117-119
        tex[..., :3] = 255
        tex[..., 3] = (255 * chk).astype(np.uint8)
    else:
        tex[..., 0] = (60 + 195 * chk).astype(np.uint8)
        tex[..., 1] = (xx * 255 // (n - 1)).astype(np.uint8)
        tex[..., 2] = 200
        tex[..., 3] = 255
    return nr.Texture2d(nr.ImageData(tex), nr.Interpolation.Bilinear, nr.Overflow.Wrap)


def mesh_scene(alpha_mapped=True, rotate=True, n_lights=2):
    """Small TriMesh scene: textured torus, an alpha-mapped quad wall in front of it (transparent
    shadows + refraction continuations), a floor mesh sharing the torus' isometry, two lights."""
    pts, idx, uvs = torus_mesh()
    tex_mat = nr.PhongMaterial((0.2, 0.2, 0.2), (1, 1, 1), (0.5, 0.5, 0.5), checker_texture(64, 8), None, 60.0)
    floor_mat = nr.PhongMaterial((0.1, 0.1, 0.1), (0.8, 0.8, 0.7), (1, 1, 1), None, None, 100.0)
    holes = nr.PhongMaterial((0.1, 0.3, 0.1), (0.2, 0.9, 0.3), (1, 1, 1), checker_texture(32, 4),
                             checker_texture(32, 6, alpha_holes=True) if alpha_mapped else None, 60.0)
    iso = nr.Isometry3((0.0, 0.0, 0.0), (0.0, math.radians(20.0), 0.0) if rotate else (0, 0, 0))
    fl = f32_exact([[-6, -1.25, -6], [6, -1.25, -6], [6, -1.25, 6], [-6, -1.25, 6]])
    fl_uv = f32_exact([[0, 0], [3, 0], [3, 3], [0, 3]])
    fl_idx = np.asarray([[0, 2, 1], [0, 3, 2]], dtype=np.uint32)
    wl = f32_exact([[-2.5, -1.0, -3.0], [2.5, -1.0, -3.0], [2.5, 2.0, -3.0], [-2.5, 2.0, -3.0]])
    wl_uv = f32_exact([[0, 0], [2, 0], [2, 1], [0, 1]])
    nodes = [
        nr.SceneNode(tex_mat, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(pts, idx, uvs)),
        nr.SceneNode(floor_mat, 0.3, 0.4, 1.0, 1.0, iso, nr.TriMesh(fl, fl_idx, fl_uv)),
        nr.SceneNode(holes, 0.0, 0.0, 0.9 if alpha_mapped else 1.0, 1.2, iso, nr.TriMesh(wl, fl_idx, wl_uv)),
        nr.SceneNode(default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3((2.8, 0.2, 0.5)), nr.Ball(0.7)),
    ]
    lights = [nr.Light((3.0, 6.0, -6.0), 0.0, 1, (0.7, 0.7, 0.7)), nr.Light((-4.0, 5.0, -2.0), 0.0, 1, (0.5, 0.5, 0.6))][:n_lights]
    cam = dict(eye=(0.5, 3.0, -9.0), at=(0.0, 0.0, 0.0), fovy=40.0)
    return nr.Scene(nodes, lights, (1, 1, 1)), cam


def random_shapes_scene(seed, n=24, with_mesh=True):
    """Random mix of every shape kind with random rotations (property tests: BVH vs brute force)."""
    rng = np.random.default_rng(seed)
    mats = [default_material(), nr.NormalMaterial(), nr.UVMaterial(),
            nr.PhongMaterial((0.1, 0.1, 0.2), (0.3, 0.5, 1.0), (1, 1, 1), checker_texture(32, 4), None, 30.0)]
    nodes = []
    for k in range(n):
        pos = rng.uniform(-6, 6, 3)
        ang = rng.uniform(-math.pi, math.pi, 3) * (rng.random() < 0.7)
        kind = k % 6
        if kind == 0:
            g = nr.Ball(rng.uniform(0.3, 1.2))
        elif kind == 1:
            g = nr.Cuboid(rng.uniform(0.3, 1.0, 3))
        elif kind == 2:
            g = nr.Cylinder(rng.uniform(0.3, 1.0), rng.uniform(0.3, 0.9))
        elif kind == 3:
            g = nr.Capsule(rng.uniform(0.3, 1.0), rng.uniform(0.2, 0.6))
        elif kind == 4:
            g = nr.Cone(rng.uniform(0.4, 1.0), rng.uniform(0.3, 0.9))
        else:
            if with_mesh:
                p, i, u = torus_mesh(12, 8, rng.uniform(0.6, 1.0), rng.uniform(0.2, 0.4))
                g = nr.TriMesh(p, i, u)
            else:
                g = nr.Ball(0.5)
        nodes.append(nr.SceneNode(mats[k % len(mats)], 0.0, 0.0, 1.0, 1.0, nr.Isometry3(pos, ang), g,
                                  None, bool(rng.random() < 0.2)))
    nodes.append(nr.SceneNode(default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0, -7.5, 0)), nr.Plane((0.1, 1, 0.05))))
    lights = [nr.Light((0.0, 12.0, -3.0), 0.0, 1, (1, 1, 1))]
    cam = dict(eye=(0.0, 4.0, -18.0), at=(0.0, 0.0, 0.0), fovy=50.0)
    return nr.Scene(nodes, lights, (1, 1, 1)), cam


def camera_params(cam, w, h, **kw):
    from nrays_amd import math3d
    proj = math3d.inverse_projection(cam["eye"], cam["at"], cam["fovy"], w, h)
    return nr.make_params((w, h), kw.pop("spp", 1), kw.pop("window", 0.0), cam["eye"], proj, **kw), proj
