#!/usr/bin/env python
"""Randomised GPU-vs-oracle parity sweep (GPU box): random scenes mixing every shape kind, meshes, alpha,
reflection / refraction coefficients, light counts / radii, AA windows; same tolerance as tests/ (1e-4 per
channel) and exact ray-class counts.  Prints one line per case and a summary; exit code 1 on any mismatch.

  python tools/fuzz_parity.py [first_seed] [count]
"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su

lib = abi.load_hip_lib()


def random_scene(seed):
    rng = np.random.default_rng(seed)
    sc, cam = su.random_shapes_scene(seed, n=int(rng.integers(4, 40)), with_mesh=bool(rng.random() < 0.7))
    d = sc  # rebuild with random coefficients and materials
    tex = [su.checker_texture(32, 4), su.checker_texture(16, 3), None]
    amap = [su.checker_texture(32, 6, alpha_holes=True), None, None]
    def material():
        k = int(rng.integers(0, 6))
        if k == 0: return nr.NormalMaterial()
        if k == 1: return nr.UVMaterial()
        return nr.PhongMaterial(tuple(rng.uniform(0, 0.3, 3)), tuple(rng.uniform(0.2, 1, 3)), tuple(rng.uniform(0, 1, 3)),
                                tex[int(rng.integers(0, 3))], amap[int(rng.integers(0, 3))], float(rng.choice([5.0, 30.0, 100.0])))
    nodes = []
    for nd in d._nodes:
        nd.material = material() if rng.random() < 0.8 else nd.material
        refl_mix = float(rng.choice([0.0, 0.0, 0.2, 0.5]))
        refl_att = float(rng.choice([0.2, 0.3, 0.5]))
        alpha = float(rng.choice([1.0, 1.0, 1.0, 0.4])) if (refl_mix == 0.0 or rng.random() < 0.15) else 1.0  # double branching is rare
        refr = float(rng.choice([1.0, 1.3]))
        nodes.append(nr.SceneNode(nd.material, refl_mix, refl_att, alpha, refr, nd.transform, nd.geometry, None, nd.solid))
    nl = int(rng.choice([1, 1, 2, 3]))
    lights = []
    for _ in range(nl):
        rad = float(rng.choice([0.0, 0.0, 0.3]))
        lights.append(nr.Light(tuple(rng.uniform(-10, 10, 3) + np.array([0, 14, 0])), rad, int(rng.choice([1, 4])) if rad > 0 else 1,
                               tuple(rng.uniform(0.3, 1.0, 3))))
    return nr.Scene(nodes, lights, tuple(rng.uniform(0, 1, 3))), cam, rng


def random_mesh_scene(seed):
    """Triangle soups: several TriMesh nodes, some sharing one isometry (merged into one BLAS), some rotated, some
    alpha-mapped / transparent / reflective, exact duplicates of triangles across nodes (ties broken by node and
    triangle index), degenerate slivers, plus a few analytic shapes."""
    rng = np.random.default_rng(seed)
    isos = [nr.Isometry3((0.0, 0.0, 0.0)), nr.Isometry3(tuple(rng.uniform(-1, 1, 3)), tuple(rng.uniform(-1.5, 1.5, 3))),
            nr.Isometry3(tuple(rng.uniform(-2, 2, 3)))]
    tex = [su.checker_texture(32, 4), su.checker_texture(16, 3), None]
    amap = [su.checker_texture(32, 6, alpha_holes=True), None, None]
    nodes, prev = [], None
    for k in range(int(rng.integers(2, 7))):
        nt = int(rng.integers(4, 500))  # >= 64 triangles in a BLAS: the pre-splitting pass is active
        ctr = rng.uniform(-4, 4, (nt, 1, 3)) * np.array([1.0, 0.6, 1.0])
        tri = su.f32_exact((ctr + rng.normal(0, float(rng.choice([0.2, 0.8, 2.0])), (nt, 3, 3))).reshape(-1, 3))
        if rng.random() < 0.2:
            tri[3:6] = tri[0:3]                      # an exact duplicate inside the node
            tri[8] = tri[7]                          # a degenerate triangle
        iso = isos[int(rng.integers(0, 3))]
        if prev is not None and rng.random() < 0.4:  # copy a few triangles of the previous node: exact ties across nodes
            m = min(len(prev[0]), len(tri), 30) // 3 * 3
            tri[:m] = prev[0][:m]
            iso = prev[1]
        idx = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        uvs = su.f32_exact(rng.uniform(-1, 2, (3 * nt, 2))) if rng.random() < 0.8 else None
        mat = nr.PhongMaterial(tuple(rng.uniform(0, 0.3, 3)), tuple(rng.uniform(0.2, 1, 3)), tuple(rng.uniform(0, 1, 3)),
                               tex[int(rng.integers(0, 3))] if uvs is not None else None,
                               amap[int(rng.integers(0, 3))] if uvs is not None else None, float(rng.choice([5.0, 30.0, 100.0])))
        refl = float(rng.choice([0.0, 0.0, 0.3]))
        alpha = float(rng.choice([1.0, 1.0, 0.5])) if refl == 0.0 else 1.0
        nodes.append(nr.SceneNode(mat, refl, 0.3, alpha, float(rng.choice([1.0, 1.2])), iso, nr.TriMesh(tri, idx, uvs)))
        prev = (tri, iso)
    for _ in range(int(rng.integers(0, 3))):
        nodes.append(nr.SceneNode(su.default_material(), float(rng.choice([0.0, 0.3])), 0.4, 1.0, 1.0,
                                  nr.Isometry3(tuple(rng.uniform(-4, 4, 3)), tuple(rng.uniform(-1, 1, 3))),
                                  [nr.Ball(0.8), nr.Cuboid((0.6, 0.4, 0.9)), nr.Cone(0.7, 0.5)][int(rng.integers(0, 3))]))
    if rng.random() < 0.5:
        nodes.append(nr.SceneNode(su.default_material(), 0.2, 0.5, 1.0, 1.0, nr.Isometry3((0, -5.0, 0)), nr.Plane((0, 1, 0))))
    lights = []
    for _ in range(int(rng.choice([1, 1, 2, 3]))):
        rad = float(rng.choice([0.0, 0.0, 0.4]))
        lights.append(nr.Light(tuple(rng.uniform(-8, 8, 3) + np.array([0, 10, 0])), rad, int(rng.choice([1, 3])) if rad > 0 else 1,
                               tuple(rng.uniform(0.3, 1.0, 3))))
    cam = dict(eye=tuple(rng.uniform(-3, 3, 3) + np.array([0.0, 2.0, -14.0])), at=(0.0, 0.0, 0.0), fovy=float(rng.choice([35.0, 50.0, 70.0])))
    return nr.Scene(nodes, lights, tuple(rng.uniform(0, 1, 3))), cam, rng


def random_hair_scene(seed):
    """Opaque scenes of thin strands (the opaque-mesh kernels with quorum-ended node phases, DScene::incoherent): one to three
    strand meshes, some sharing an isometry, some rotated, reflective ones among them, sometimes an ordinary mesh beside them;
    one to three lights, one of them sometimes an area light."""
    rng = np.random.default_rng(seed)
    nodes = []
    isos = [nr.Isometry3((0.0, 0.0, 0.0)), nr.Isometry3(tuple(rng.uniform(-1, 1, 3)), tuple(rng.uniform(-1.0, 1.0, 3)))]
    for k in range(int(rng.integers(1, 4))):
        ns, seg = int(rng.integers(20, 160)), int(rng.integers(3, 12))
        width = float(rng.choice([0.004, 0.01, 0.03]))
        start = rng.normal(0, 1, (ns, 3)); start /= np.linalg.norm(start, axis=1, keepdims=True)
        pts = [start * float(rng.uniform(0.3, 1.0))]
        dirn = start + rng.normal(0, 0.6, (ns, 3))
        for _ in range(seg):
            dirn = dirn + rng.normal(0, 0.5, (ns, 3)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
            pts.append(pts[-1] + dirn * float(rng.uniform(0.1, 0.4)))
        pts = np.stack(pts, 1)                                  # ns x (seg + 1) x 3: the strands' centre lines
        side = np.cross(dirn, rng.normal(0, 1, (ns, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
        a, b = pts - side[:, None, :] * width, pts + side[:, None, :] * width
        tri = np.concatenate([np.stack([a[:, :-1], b[:, :-1], a[:, 1:]], 2), np.stack([b[:, :-1], b[:, 1:], a[:, 1:]], 2)], 1)
        tri = su.f32_exact(tri.reshape(-1, 3))
        nt = len(tri) // 3
        idx = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        mat = nr.PhongMaterial(tuple(rng.uniform(0, 0.3, 3)), tuple(rng.uniform(0.2, 1, 3)), tuple(rng.uniform(0, 1, 3)), None, None, float(rng.choice([5.0, 30.0, 100.0])))
        nodes.append(nr.SceneNode(mat, float(rng.choice([0.0, 0.0, 0.3])), 0.4, 1.0, 1.0, isos[int(rng.integers(0, 2))], nr.TriMesh(tri, idx, None)))
    if rng.random() < 0.5:
        p3, i3, uv3 = su.torus_mesh(24, 12)
        nodes.append(nr.SceneNode(su.default_material(), float(rng.choice([0.0, 0.4])), 0.4, 1.0, 1.0,
                                  nr.Isometry3(tuple(rng.uniform(-1, 1, 3) + np.array([0.0, -1.5, 0.0]))), nr.TriMesh(p3, i3, uv3)))
    lights = []
    for _ in range(int(rng.choice([1, 1, 2, 3]))):
        rad = float(rng.choice([0.0, 0.0, 0.4]))
        lights.append(nr.Light(tuple(rng.uniform(-6, 6, 3) + np.array([0, 6, -4])), rad, int(rng.choice([1, 2])) if rad > 0 else 1, tuple(rng.uniform(0.3, 1.0, 3))))
    cam = dict(eye=tuple(rng.uniform(-1, 1, 3) + np.array([0.0, 0.5, -7.0])), at=(0.0, 0.0, 0.0), fovy=float(rng.choice([30.0, 45.0])))
    return nr.Scene(nodes, lights, tuple(rng.uniform(0, 1, 3))), cam, rng


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    worst, bad, quorum_scenes = 0.0, 0, 0
    for seed in range(first, first + count):
        try:
            sc, cam, rng = random_hair_scene(seed) if seed % 3 == 2 else (random_mesh_scene(seed) if seed % 2 else random_scene(seed))
        except AttributeError as e:
            print("scene construction needs attribute:", e); return 2
        w, h = int(rng.integers(40, 200)), int(rng.integers(30, 140))
        spp = int(rng.choice([1, 1, 2]))
        kw = dict(spp=spp, window=float(rng.choice([0.0, 1.0])) if spp > 1 else 0.0, seed=int(seed),
                  max_depth=int(rng.choice([3, 5, 8])))  # bounds the 2^depth ray trees of double-branching nodes
        p, _ = su.camera_params(cam, w, h, **kw)
        ref, ost = oracle.render(sc.descriptor, p, 32)
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        flags = (C.c_uint32 * 2)()
        abi.check(lib.nrays_debug_scene_flags(sc.device_handle(), flags))
        quorum_scenes += 1 if (flags[1] and (flags[0] & 7) == 2) else 0
        for rep in range(2):  # second frame runs through the cost-ordered work lists
            abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
            st = nr.get_stats(sc)
            err = float(np.abs(out.cpu().numpy() - ref).max())
            same = all(getattr(st, k) == getattr(ost, k) for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"))
            worst = max(worst, err)
            ok = err <= 1e-4 and same
            bad += 0 if ok else 1
            print("seed %d frame %d %dx%d spp %d nodes %d lights %d rays %d: max err %.3g counts %s %s" % (
                seed, rep, w, h, spp, len(sc._nodes), len(sc._lights), st.total_rays(), err, "equal" if same else "DIFFER", "" if ok else "<-- MISMATCH"), flush=True)
    print("worst error %.3g over %d cases, %d mismatches; %d scenes rendered with quorum-ended node phases" % (worst, 2 * count, bad, quorum_scenes))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
