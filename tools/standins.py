"""Procedural stand-ins for the OBJ assets the reference downloads (scenes/media/ is
git-ignored and absent): crytek_sponza (~262k triangles, ~300 groups, 25 materials, 8 textures incl.
2 alpha maps) and hairball (~2.88M triangles).  Deterministic (fixed seeds 0x5EED5A / 0x4A1B), every
vertex rounded to f32 BEFORE use (obj.rs:197-205 parses f32) and scaled by 1/4 exactly like
loader3d.rs:669.  Scene files: scenes/crytek_sponza.scene, scenes/hairball.scene (camera / light /
node parameters below are the ones of those files).
"""
import math

import numpy as np

import nrays_amd as nr

SPONZA_SEED = 0x5EED5A
HAIRBALL_SEED = 0x4A1B
SPONZA_TRIS = 0  # filled by sponza_scene()


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _grid(nu, nv):
    """Triangle indices of an (nu+1) x (nv+1) vertex grid."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    return np.concatenate([np.stack([a, b, b + 1], 1), np.stack([a, b + 1, a + 1], 1)]).astype(np.uint32)


def _surface(fn, nu, nv, uv_scale=(1.0, 1.0)):
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    p = fn(u, v).reshape(-1, 3)
    uv = np.stack([u * uv_scale[0], v * uv_scale[1]], -1).reshape(-1, 2)
    return p, uv, _grid(nu, nv)


class _MeshBuilder:
    """Accumulates groups that share one vertex / uv array, like an OBJ file does."""

    def __init__(self):
        self.pts, self.uvs, self.groups, self.nv = [], [], [], 0

    def add(self, material, p, uv, idx, join=False):
        """join=True appends the faces to the previous group when it has the same material."""
        self.pts.append(p)
        self.uvs.append(uv)
        if join and self.groups and self.groups[-1][0] == material:
            self.groups[-1] = (material, np.concatenate([self.groups[-1][1], idx + self.nv]))
        else:
            self.groups.append((material, idx + self.nv))
        self.nv += len(p)

    def finish(self, scale):
        pts = _f32(_f32(np.concatenate(self.pts)) * scale)  # f32 parse, then the exact /4
        uvs = _f32(np.concatenate(self.uvs))
        return pts, uvs, self.groups


def _procedural_texture(rng, kind, n=512):
    yy, xx = np.mgrid[0:n, 0:n]
    t = np.empty((n, n, 4), dtype=np.uint8)
    t[..., 3] = 255
    if kind == "brick":
        row = yy // 32
        mortar = ((yy % 32) < 3) | (((xx + 32 * (row % 2)) % 64) < 3)
        base = rng.integers(120, 200, (n // 32 + 1, n // 64 + 2))
        tone = base[row, (xx + 32 * (row % 2)) // 64]
        t[..., 0] = np.where(mortar, 200, tone)
        t[..., 1] = np.where(mortar, 195, tone * 0.55)
        t[..., 2] = np.where(mortar, 185, tone * 0.4)
    elif kind == "floor":
        chk = ((xx // 64) + (yy // 64)) % 2
        t[..., 0] = 90 + 120 * chk
        t[..., 1] = 80 + 110 * chk
        t[..., 2] = 70 + 90 * chk
    elif kind == "marble":
        v = (np.sin(xx * 0.05 + 4 * np.sin(yy * 0.021)) * 0.5 + 0.5)
        t[..., 0] = (190 + 60 * v).astype(np.uint8)
        t[..., 1] = (185 + 55 * v).astype(np.uint8)
        t[..., 2] = (175 + 60 * v).astype(np.uint8)
    elif kind == "fabric":
        s = ((xx // 16) % 2)
        t[..., 0] = 150 + 90 * s
        t[..., 1] = 30 + 20 * s
        t[..., 2] = 40
    elif kind == "leaf":
        t[..., 0] = 40 + (xx % 32)
        t[..., 1] = 120 + (yy % 64)
        t[..., 2] = 40
    else:  # noise
        t[..., :3] = rng.integers(60, 220, (n, n, 3))
    return nr.Texture2d(nr.ImageData(t), nr.Interpolation.Bilinear, nr.Overflow.Wrap)


def _alpha_texture(kind, n=512):
    """Opacity maps decode to (1,1,1,a) (texture2d.rs:117-119)."""
    yy, xx = np.mgrid[0:n, 0:n]
    t = np.full((n, n, 4), 255, dtype=np.uint8)
    if kind == "lace":
        a = (((xx // 24) + (yy // 24)) % 2) * 255
    else:  # leaves: discs on a grid
        cx, cy = (xx % 64) - 32, (yy % 64) - 32
        a = ((cx * cx + cy * cy) < 26 * 26) * 255
    t[..., 3] = a.astype(np.uint8)
    return nr.Texture2d(nr.ImageData(t), nr.Interpolation.Bilinear, nr.Overflow.Wrap)


def sponza_geometry(detail=1.0):
    """Atrium in OBJ units (x along the nave, y up; the loader scales by 1/4).  detail=1.0 gives ~262k
    triangles.  Returns (points f32-exact in OBJ units, uvs, groups [(material name, faces)],
    material definitions {name: (ka, kd, ks, tex name, alpha tex name, ns, d)}, textures {name: Texture2d})."""
    rng = np.random.default_rng(SPONZA_SEED)
    tex = {k: _procedural_texture(rng, k) for k in ("brick", "floor", "marble", "fabric", "leaf", "noise")}
    tex["lace"], tex["leaves"] = _alpha_texture("lace"), _alpha_texture("leaves")
    lace, leaves = "lace", "leaves"
    tex_names = {id(v): k for k, v in tex.items()}

    def phong(ka, kd, ks, t=None, a=None, ns=60.0):
        return (ka, kd, ks, tex_names.get(id(t)) if t is not None and not isinstance(t, str) else t, a, ns)
    mats = {
        "floor": phong((.12, .12, .12), (1, 1, 1), (.3, .3, .3), tex["floor"]),
        "bricks": phong((.1, .1, .1), (1, 1, 1), (.1, .1, .1), tex["brick"]),
        "ceiling": phong((.15, .15, .15), (.9, .9, .85), (0, 0, 0), tex["noise"]),
        "column_a": phong((.1, .1, .1), (1, 1, 1), (.5, .5, .5), tex["marble"], None, 100.0),
        "column_b": phong((.1, .1, .1), (.9, .85, .8), (.5, .5, .5), tex["marble"], None, 80.0),
        "column_c": phong((.1, .1, .1), (.8, .8, .9), (.4, .4, .4), tex["marble"], None, 40.0),
        "arch": phong((.1, .1, .1), (.95, .9, .85), (.2, .2, .2), tex["brick"]),
        "fabric_a": phong((.15, .05, .05), (1, 1, 1), (.1, .1, .1), tex["fabric"]),
        "fabric_c": phong((.05, .05, .2), (.5, .6, 1), (.1, .1, .1), tex["fabric"]),
        "fabric_d": phong((.05, .15, .05), (.6, 1, .6), (.1, .1, .1), tex["fabric"]),
        "fabric_e": phong((.1, .1, .1), (1, 1, 1), (.1, .1, .1), tex["fabric"], lace),
        "chain": phong((.2, .2, .1), (.9, .8, .3), (1, 1, 1), None, lace, 100.0),
        "leaf": phong((.05, .15, .05), (1, 1, 1), (.1, .1, .1), tex["leaf"], leaves),
        "vase": phong((.1, .1, .1), (.7, .5, .3), (.8, .8, .8), None, None, 100.0),
        "vase_round": phong((.1, .1, .1), (.5, .5, .7), (.9, .9, .9), None, None, 100.0),
        "vase_hanging": phong((.1, .1, .1), (.4, .4, .4), (1, 1, 1), None, None, 100.0),
        "flagpole": phong((.1, .1, .1), (.3, .3, .3), (.9, .9, .9), None, None, 100.0),
        "details": phong((.1, .1, .1), (.85, .8, .7), (.2, .2, .2), tex["noise"]),
        "lion": phong((.12, .1, .08), (.9, .8, .6), (.6, .6, .6), tex["marble"], None, 60.0),
        "roof": phong((.1, .1, .1), (.8, .4, .3), (.1, .1, .1), tex["brick"]),
        "plinth": phong((.1, .1, .1), (.7, .7, .7), (.3, .3, .3), tex["marble"]),
        "glass": phong((.05, .05, .08), (.6, .7, .9), (1, 1, 1), None, None, 100.0),
        "default": phong((.1, .1, .1), (1, 1, 1), (1, 1, 1), None, None, 100.0),
        "trim": phong((.1, .1, .1), (.6, .55, .5), (.2, .2, .2), tex["noise"]),
        "banner": phong((.1, .1, .1), (1, .9, .5), (.1, .1, .1), tex["fabric"]),
    }
    alphas = {"glass": 0.35}
    mb = _MeshBuilder()
    d = lambda n: max(2, int(round(n * math.sqrt(detail))))  # noqa: E731
    L, Wd, Hh = 1800.0, 600.0, 1200.0  # half length, half width, height (OBJ units)

    def quad(mat, o, eu, ev, nu, nv, uvs=(8.0, 8.0), join=False):
        o, eu, ev = map(np.asarray, (o, eu, ev))
        p, uv, idx = _surface(lambda u, v: o + u[..., None] * eu + v[..., None] * ev, nu, nv, uvs)
        mb.add(mat, p, uv, idx, join)

    # floor (4 groups), ceiling, walls
    for k in range(4):
        quad("floor", (-L + k * L / 2, 0, -Wd), (L / 2, 0, 0), (0, 0, 2 * Wd), d(40), d(48), (6, 8))
    quad("ceiling", (-L, Hh, -Wd), (0, 0, 2 * Wd), (2 * L, 0, 0), d(24), d(64), (4, 12))
    for s in (-1, 1):
        for k in range(6):
            quad("bricks", (-L + k * L / 3, 0, s * Wd), (L / 3, 0, 0), (0, Hh, 0), d(24), d(40), (4, 8))
    for s in (-1, 1):
        quad("bricks", (s * L, 0, -Wd), (0, 0, 2 * Wd), (0, Hh, 0), d(32), d(40), (6, 8))
    # gallery floors over the aisles (two storeys)
    for s in (-1, 1):
        for level in (420.0, 800.0):
            quad("details", (-L, level, s * 300.0), (2 * L, 0, 0), (0, 0, s * 300.0), d(96), d(10), (24, 2))
            quad("trim", (-L, level - 30, s * 300.0), (2 * L, 0, 0), (0, 30, 0), d(96), 2, (24, 1))

    # two arcades x two storeys of columns + arches
    col_mats = ["column_a", "column_b", "column_c"]
    xs = np.linspace(-L + 150, L - 150, 12)
    for s in (-1, 1):
        for level, h, r in ((0.0, 390.0, 42.0), (420.0, 350.0, 30.0)):
            for ci, x in enumerate(xs):
                def col(u, v, x=x, h=h, r=r, level=level, s=s):
                    rr = r * (1.0 + 0.08 * np.cos(v * 2 * math.pi * 3))  # slight entasis / fluting along the shaft
                    return np.stack([x + rr * np.cos(u * 2 * math.pi), level + v * h, s * 300.0 + rr * np.sin(u * 2 * math.pi)], -1)
                p, uv, idx = _surface(col, d(42), d(27), (2, 4))
                mb.add(col_mats[ci % 3], p, uv, idx)
                # plinth + capital boxes (one group per column)
                for bi, (y0, hh, w) in enumerate(((level, 24.0, r * 1.5), (level + h - 24.0, 24.0, r * 1.4))):
                    for ax in range(4):
                        a0 = ax * math.pi / 2
                        c0 = np.array([x + w * math.cos(a0 + math.pi / 4) * math.sqrt(2), y0, s * 300.0 + w * math.sin(a0 + math.pi / 4) * math.sqrt(2)])
                        c1 = np.array([x + w * math.cos(a0 + 3 * math.pi / 4) * math.sqrt(2), y0, s * 300.0 + w * math.sin(a0 + 3 * math.pi / 4) * math.sqrt(2)])
                        quad("plinth", c0, c1 - c0, (0, hh, 0), 2, 2, (1, 1), join=(bi + ax) > 0)
            for ci in range(len(xs) - 1):  # arches between neighbouring columns
                x0, x1 = xs[ci], xs[ci + 1]
                def arch(u, v, x0=x0, x1=x1, level=level, h=h, s=s):
                    ang = u * math.pi
                    cx, rad = (x0 + x1) / 2, (x1 - x0) / 2 - 20.0
                    return np.stack([cx - rad * np.cos(ang), level + h + 0.45 * rad * np.sin(ang), s * 300.0 + (v - 0.5) * 60.0], -1)
                p, uv, idx = _surface(arch, d(28), d(5), (4, 1))
                mb.add("arch", p, uv, idx)

    # hanging fabrics (curved), some lace (alpha mapped)
    fab = ["fabric_a", "fabric_c", "fabric_d", "fabric_e", "banner"]
    for k in range(10):
        x = -L + 300 + k * 330.0
        s = -1 if k % 2 else 1
        def drape(u, v, x=x, s=s, k=k):
            sag = 60.0 * np.sin(u * math.pi) * (0.5 + 0.5 * v)
            return np.stack([x + u * 220.0, 760.0 - v * 330.0 - 0.2 * sag, s * (230.0 - sag) + 8 * np.sin(v * 9 + k)], -1)
        p, uv, idx = _surface(drape, d(36), d(36), (2, 2))
        mb.add(fab[k % 5], p, uv, idx)
    # chains + hanging vases
    for k in range(8):
        x = -L + 400 + k * 400.0
        for s in (-1, 1):
            quad("chain", (x - 6, 560.0, s * 180.0), (12, 0, 0), (0, 240.0, 0), 2, d(20), (1, 12))
            def vase(u, v, x=x, s=s):
                th, ph = u * 2 * math.pi, v * math.pi
                r = 34.0 * (0.6 + 0.4 * np.sin(ph))
                return np.stack([x + r * np.sin(ph) * np.cos(th), 520.0 - 40.0 * np.cos(ph), s * 180.0 + r * np.sin(ph) * np.sin(th)], -1)
            p, uv, idx = _surface(vase, d(20), d(12))
            mb.add("vase_hanging", p, uv, idx)
    # plants in vases along the nave: vase (opaque) + leaf quads (alpha mapped)
    for k in range(10):
        x = -L + 250 + k * 340.0
        for s in (-1, 1):
            def pot(u, v, x=x, s=s):
                th = u * 2 * math.pi
                r = 40.0 + 18.0 * np.sin(v * math.pi)
                return np.stack([x + r * np.cos(th), v * 90.0, s * 140.0 + r * np.sin(th)], -1)
            p, uv, idx = _surface(pot, d(20), d(10))
            mb.add("vase" if k % 2 else "vase_round", p, uv, idx)
            for lf in range(6):
                a = lf * math.pi / 3 + 0.3 * k
                o = np.array([x, 90.0, s * 140.0])
                eu = np.array([70.0 * math.cos(a), 50.0, 70.0 * math.sin(a)])
                ev = np.array([-25.0 * math.sin(a), 60.0, 25.0 * math.cos(a)])
                quad("leaf", o, eu, ev, d(4), d(4), (2, 2), join=lf > 0)  # one group per plant
    # flagpoles
    for k in range(6):
        x = -L + 500 + k * 520.0
        for s in (-1, 1):
            def pole(u, v, x=x, s=s):
                th = u * 2 * math.pi
                return np.stack([x + 5.0 * np.cos(th), 640.0 + 5.0 * np.sin(th), s * (300.0 - v * 190.0)], -1)
            p, uv, idx = _surface(pole, d(8), d(12))
            mb.add("flagpole", p, uv, idx)
    # a lion-head-like displaced sphere on the end wall + roof tiles + glass panes
    def lion(u, v):
        th, ph = u * 2 * math.pi, v * math.pi
        r = 150.0 * (1.0 + 0.12 * np.sin(7 * th) * np.sin(5 * ph) + 0.05 * np.cos(13 * th + 3 * ph))
        return np.stack([L - 40.0 + 0.5 * r * np.sin(ph) * np.cos(th), 420.0 - r * np.cos(ph), r * np.sin(ph) * np.sin(th)], -1)
    p, uv, idx = _surface(lion, d(160), d(96), (4, 2))
    mb.add("lion", p, uv, idx)
    for s in (-1, 1):
        quad("roof", (-L, Hh, s * Wd), (2 * L, 0, 0), (0, 140.0, -s * 200.0), d(64), d(6), (16, 2))
        for k in range(6):
            quad("glass", (-L + 300 + k * 560.0, 500.0, s * (Wd - 4.0)), (240.0, 0, 0), (0, 300.0, 0), 2, 2, (1, 1))

    pts, uvs, groups = mb.finish(1.0)
    defs = {k: v + (alphas.get(k, 1.0),) for k, v in mats.items()}
    return pts, uvs, groups, defs, tex


def sponza_scene(detail=1.0, n_lights=1):
    """The stand-in as a Scene, built exactly as the loader would build it from the OBJ written by
    tools/gen_assets.py: vertices / 4 (loader3d.rs:669), one SceneNode per group, node alpha = mtl d."""
    global SPONZA_TRIS
    pts, uvs, groups, defs, tex = sponza_geometry(detail)
    pts = _f32(pts * 0.25)
    mats = {}
    for name, (ka, kd, ks, t, a, ns, d) in defs.items():
        mats[name] = nr.PhongMaterial(ka, kd, ks, tex[t] if t else None, tex[a] if a else None, ns)
    alphas = {name: v[6] for name, v in defs.items()}
    iso = nr.Isometry3((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    nodes, ntri = [], 0
    for mat, idx in groups:
        # one SceneNode per group: TriMesh over the WHOLE vertex array with the group's faces (loader3d.rs:690-695)
        nodes.append(nr.SceneNode(mats[mat], 0.0, 0.0, alphas.get(mat, 1.0), 1.0, iso, nr.TriMesh(pts, idx, uvs)))
        ntri += len(idx)
    SPONZA_TRIS = ntri
    eye = (-250.0, 50.0, 0.0)
    lights = [nr.Light(eye, 0.0, 1, (1.0, 1.0, 1.0))]
    for k in range(1, n_lights):  # BASELINE config 4: 7 more point lights on a ring y=50, r=100
        a = 2 * math.pi * k / 8.0
        lights.append(nr.Light((100.0 * math.cos(a), 50.0, 100.0 * math.sin(a)), 0.0, 1, (1.0, 1.0, 1.0)))
    if n_lights > 1:
        lights = [nr.Light(l.pos, 0.0, 1, (1.0 / n_lights,) * 3) for l in lights]
    cam = dict(eye=eye, at=(0.0, 50.0, 0.0), fovy=45.0)
    return nr.Scene(nodes, lights, (1, 1, 1)), cam


def hairball_geometry(strands=3000, sides=8, segments=60):
    """~strands*sides*segments*2 triangles of thin tubes around a unit ball, in OBJ units (x4)."""
    rng = np.random.default_rng(HAIRBALL_SEED)
    # strand centre lines: start on a sphere of radius .55, wander outwards with curl
    dirs = rng.normal(size=(strands, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    t = np.linspace(0.0, 1.0, segments + 1)
    curl_axis = rng.normal(size=(strands, 3))
    curl_axis /= np.linalg.norm(curl_axis, axis=1, keepdims=True)
    side = np.cross(dirs, curl_axis)
    side /= np.linalg.norm(side, axis=1, keepdims=True) + 1e-12
    amp = rng.uniform(0.05, 0.35, (strands, 1, 1))
    freq = rng.uniform(1.0, 3.5, (strands, 1, 1))
    phase = rng.uniform(0, 2 * math.pi, (strands, 1, 1))
    rad = 0.55 + 0.75 * t[None, :, None]
    wob = amp * np.sin(freq * 2 * math.pi * t[None, :, None] + phase)
    centre = dirs[:, None, :] * rad + side[:, None, :] * wob + np.cross(dirs, side)[:, None, :] * (amp * np.cos(freq * 2 * math.pi * t[None, :, None] + phase) - amp)
    tang = np.gradient(centre, axis=1)
    tang /= np.linalg.norm(tang, axis=2, keepdims=True) + 1e-12
    ref = np.where(np.abs(tang[..., :1]) < 0.9, np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
    n1 = np.cross(tang, ref)
    n1 /= np.linalg.norm(n1, axis=2, keepdims=True) + 1e-12
    n2 = np.cross(tang, n1)
    ang = np.arange(sides) * 2 * math.pi / sides
    w = 0.006 * (1.0 - 0.6 * t)[None, :, None, None]
    ring = centre[:, :, None, :] + w * (np.cos(ang)[None, None, :, None] * n1[:, :, None, :] + np.sin(ang)[None, None, :, None] * n2[:, :, None, :])
    pts = _f32(ring.reshape(-1, 3))
    s_i, g_i, k_i = np.meshgrid(np.arange(strands), np.arange(segments), np.arange(sides), indexing="ij")
    base = s_i * (segments + 1) * sides
    a = base + g_i * sides + k_i
    b = base + g_i * sides + (k_i + 1) % sides
    c = a + sides
    dd = b + sides
    idx = np.concatenate([np.stack([a, b, dd], -1).reshape(-1, 3), np.stack([a, dd, c], -1).reshape(-1, 3)]).astype(np.uint32)
    uvs = _f32(np.stack([np.tile(np.repeat(t, sides), strands), np.tile(np.tile(np.arange(sides) / sides, segments + 1), strands)], -1))
    return _f32(pts * 4.0), idx, uvs


def hairball_scene(strands=3000, sides=8, segments=60):
    """hairball.scene: eye (0,.2,-5), fovy 25, node pos (0,.1,0) angle (0,.1 deg,0) as an axis-angle."""
    pts, idx, uvs = hairball_geometry(strands, sides, segments)
    pts = _f32(pts * 0.25)
    mat = nr.PhongMaterial((0.1, 0.1, 0.1), (1, 1, 1), (1, 1, 1), None, None, 100.0)  # `material default`
    iso = nr.Isometry3((0.0, 0.1, 0.0), (0.0, math.radians(0.1), 0.0))
    node = nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(pts, idx, uvs))
    eye = (0.0, 0.2, -5.0)
    cam = dict(eye=eye, at=(0.0, 0.2, 0.0), fovy=25.0)
    return nr.Scene([node], [nr.Light(eye, 0.0, 1, (1, 1, 1))], (1, 1, 1)), cam
