#!/usr/bin/env python
"""More fixtures derived independently of oracle/nrays_oracle.c and of the HIP kernels (VERDICT r2, next #4c): what the first
set (make_kat_independent.py) left without an independent answer.

  planes     ncollide Plane(n) under an isometry (examples/loader3d.rs:656; SURVEY B-6): the half-space {x : n.x <= 0}; the
             ray meets its boundary at t = n.(-o) / n.d (60-digit mpmath), hit iff t >= 0; a solid plane whose half-space
             contains the origin answers toi 0; the normal faces the ray.
  aabbs      geometry.bounding_volume(&transform) (src/scene_node.rs:41) of the five convex shapes: the EXACT axis-aligned box
             of a convex body under an isometry is [t_i - h(-R^T e_i), t_i + h(R^T e_i)] with h the support function — the
             definition ncollide's support-map AABB evaluates.  The kernels use that box as an exact gate (every accepted hit
             must pass ncollide's ray / AABB test against it), so its last bits matter.
  meshes     several-triangle TriMesh nodes under rotations, two of them per scene, rays aimed at them: the closest hit over
             ALL triangles of all nodes by exact plane / barycentric solutions (60 digits) — toi, node, flat normal towards
             the ray origin, interpolated uv; cases whose two smallest distances differ by less than 1e-9 are dropped.
  ties       the SAME triangle in two nodes (coincident surfaces): equal toi; DESIGN D-2 chooses the smaller node index (the
             reference's choice depends on its heap order) — pinned here as the stated rule, not as ncollide's.

  python tests/golden/make_kat_independent2.py      # rewrites tests/golden/kat_independent2.npz (seed fixed)
"""
import os
import sys

import numpy as np
from mpmath import mp, mpf, sqrt

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_kat_independent import rodrigues, to_local, BALL, CUBOID, CYLINDER, CAPSULE, CONE  # noqa: E402  (60-digit Rodrigues rotation)

mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))
M = lambda v: [mpf(float(c)) for c in v]
f32 = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)


def support(kind, prm, n):
    """h(n) = max over the shape of n . p, 60 digits (local frame; cylinder / capsule / cone along Y, cone apex at +hh)."""
    nx, ny, nz = n
    if kind == BALL:
        return prm[0] * sqrt(nx * nx + ny * ny + nz * nz)
    if kind == CUBOID:
        return prm[0] * abs(nx) + prm[1] * abs(ny) + prm[2] * abs(nz)
    hh, r = prm[0], prm[1]
    rad = sqrt(nx * nx + nz * nz)
    if kind == CYLINDER:
        return hh * abs(ny) + r * rad
    if kind == CAPSULE:
        return hh * abs(ny) + r * sqrt(nx * nx + ny * ny + nz * nz)
    return max(hh * ny, -hh * ny + r * rad)


def planes(rng, count=120):
    rows = []
    for case in range(count):
        n = rng.normal(size=3); n /= np.linalg.norm(n)   # loader3d.rs:656 normalises the file's vector
        t = rng.uniform(-3, 3, 3)
        w = rng.normal(size=3); w = w / np.linalg.norm(w) * rng.uniform(0, 3.0) if case % 4 else np.zeros(3)
        solid = bool(case % 2)
        o = rng.uniform(-6, 6, 3); d = rng.normal(size=3); d /= np.linalg.norm(d)
        R = rodrigues(M(w))
        ol, dl = to_local(R, M(t), M(o), M(d))
        nm = M(n)
        s = -sum(nm[i] * ol[i] for i in range(3))      # n . (-o)
        den = sum(nm[i] * dl[i] for i in range(3))
        if abs(den) < mpf("1e-6") or abs(s) < mpf("1e-6"):
            continue                                     # grazing ray / origin on the plane: no unambiguous fixture
        hit, toi, nw = False, mpf(0), [mpf(0)] * 3
        if solid and s > 0:                               # origin inside the half-space of a solid plane
            hit, toi = True, mpf(0)
        else:
            tt = s / den
            if tt >= 0:
                hit, toi = True, tt
                nl = [-c for c in nm] if s > 0 else nm    # faces the ray origin
                nw = [sum(R[i][j] * nl[j] for j in range(3)) for i in range(3)]
        rows.append(list(n) + list(t) + list(w) + [float(solid)] + list(o) + list(d) + [float(hit), float(toi)] + [float(c) for c in nw] + [float(s > 0)])
    print("planes: %d cases, %d hits" % (len(rows), int(sum(r[16] for r in rows))))
    return np.array(rows)


def aabbs(rng, per_kind=40):
    rows = []
    mk = {BALL: lambda: [rng.uniform(0.3, 2.0), 0, 0], CUBOID: lambda: list(rng.uniform(0.3, 2.0, 3)),
          CYLINDER: lambda: [rng.uniform(0.3, 2.0), rng.uniform(0.3, 1.5), 0], CAPSULE: lambda: [rng.uniform(0.3, 2.0), rng.uniform(0.3, 1.5), 0],
          CONE: lambda: [rng.uniform(0.3, 2.0), rng.uniform(0.3, 1.5), 0]}
    for kind in (BALL, CUBOID, CYLINDER, CAPSULE, CONE):
        for case in range(per_kind):
            prm = [float(v) for v in mk[kind]()]
            t = rng.uniform(-5, 5, 3)
            w = rng.normal(size=3); w = w / np.linalg.norm(w) * rng.uniform(0, 3.1) if case % 5 else np.zeros(3)
            R = rodrigues(M(w))
            lo, hi = [], []
            for i in range(3):
                col = [R[i][j] for j in range(3)]          # R^T e_i = row i of R
                hi.append(mpf(float(t[i])) + support(kind, M(prm), col))
                lo.append(mpf(float(t[i])) - support(kind, M(prm), [-c for c in col]))
            rows.append([kind] + prm + list(t) + list(w) + [float(c) for c in lo] + [float(c) for c in hi])
    print("aabbs: %d cases" % len(rows))
    return np.array(rows)


def tri_hit(a, b, c, ol, dl):
    """(t, wa, wb, wc, n, dn) of the ray with the triangle's plane, exact; None if parallel."""
    sub = lambda p, q: [p[i] - q[i] for i in range(3)]
    cross = lambda p, q: [p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]]
    dot = lambda p, q: sum(p[i] * q[i] for i in range(3))
    n = cross(sub(b, a), sub(c, a))
    dn = dot(n, dl)
    if dn == 0:
        return None
    tt = dot(sub(a, ol), n) / dn
    pt = [ol[i] + tt * dl[i] for i in range(3)]
    area = dot(n, n)
    wa = dot(cross(sub(b, pt), sub(c, pt)), n) / area
    wb = dot(cross(sub(c, pt), sub(a, pt)), n) / area
    return tt, wa, wb, 1 - wa - wb, n, dn


def meshes(rng, scenes=30, rays_per_scene=8):
    """Each scene: two TriMesh nodes (6 triangles each, shared-vertex strips) under different isometries."""
    rows, scene_rows = [], []
    for s in range(scenes):
        nodes = []
        for k in range(2):
            V = f32(rng.uniform(-1.5, 1.5, (8, 3)))
            UV = f32(rng.uniform(0, 1, (8, 2)))
            F = np.array([[0, 1, 2], [2, 1, 3], [2, 3, 4], [4, 3, 5], [4, 5, 6], [6, 5, 7]], dtype=np.uint32)
            t = rng.uniform(-2, 2, 3)
            w = rng.normal(size=3); w = w / np.linalg.norm(w) * rng.uniform(0.2, 3.0) if (s + k) % 3 else np.zeros(3)
            nodes.append((V, UV, F, t, w))
        for r in range(rays_per_scene):
            k = r % 2
            V, UV, F, t, w = nodes[k]
            f = F[rng.integers(0, len(F))]
            bu, bv = rng.uniform(0.1, 0.4), rng.uniform(0.1, 0.4)
            tgt_l = V[f[0]] + bu * (V[f[1]] - V[f[0]]) + bv * (V[f[2]] - V[f[0]])
            R64 = np.array([[float(v) for v in row] for row in rodrigues(M(w))])
            tgt = R64 @ tgt_l + t
            o = tgt + rng.normal(size=3) * rng.uniform(3, 7)
            d = tgt - o; d /= np.linalg.norm(d)
            if r % 4 == 3:
                d = rng.normal(size=3); d /= np.linalg.norm(d)  # an arbitrary direction: mostly misses
            cands = []
            for ni, (V2, UV2, F2, t2, w2) in enumerate(nodes):
                Rm = rodrigues(M(w2))
                ol, dl = to_local(Rm, M(t2), M(o), M(d))
                for fi, ff in enumerate(F2):
                    h = tri_hit(M(V2[ff[0]]), M(V2[ff[1]]), M(V2[ff[2]]), ol, dl)
                    if h is None:
                        continue
                    tt, wa, wb, wc, n, dn = h
                    if tt < 0:
                        continue
                    if min(wa, wb, wc) > mpf("1e-9"):
                        ln = sqrt(sum(c * c for c in n)); sg = -1 if dn > 0 else 1
                        nl = [sg * c / ln for c in n]
                        nw = [sum(Rm[i][j] * nl[j] for j in range(3)) for i in range(3)]
                        u = wa * mpf(float(UV2[ff[0], 0])) + wb * mpf(float(UV2[ff[1], 0])) + wc * mpf(float(UV2[ff[2], 0]))
                        v = wa * mpf(float(UV2[ff[0], 1])) + wb * mpf(float(UV2[ff[1], 1])) + wc * mpf(float(UV2[ff[2], 1]))
                        cands.append((tt, ni, nw, u, v, False))
                    elif min(wa, wb, wc) > -mpf("1e-9"):
                        cands.append((tt, ni, None, 0, 0, True))  # on an edge: ambiguous
            cands.sort(key=lambda c: c[0])
            if cands and (cands[0][5] or (len(cands) > 1 and cands[1][0] - cands[0][0] < mpf("1e-9"))):
                continue
            if cands:
                tt, ni, nw, u, v, _ = cands[0]
                rows.append([s] + list(o) + list(d) + [1.0, float(tt), float(ni)] + [float(c) for c in nw] + [float(u), float(v)])
            else:
                rows.append([s] + list(o) + list(d) + [0.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        for (V, UV, F, t, w) in nodes:
            scene_rows.append(list(V.reshape(-1)) + list(UV.reshape(-1)) + list(F.reshape(-1).astype(np.float64)) + list(t) + list(w))
    print("meshes: %d scenes, %d rays, %d hits" % (scenes, len(rows), int(sum(r[7] for r in rows))))
    return np.array(rows), np.array(scene_rows)


def ties(rng, count=40):
    """One triangle present in TWO nodes with the same isometry: both hits have the same toi; D-2: the smaller node index wins."""
    rows = []
    for case in range(count):
        A, B, C = (f32(rng.uniform(-2, 2, 3)) for _ in range(3))
        t = rng.uniform(-2, 2, 3)
        w = rng.normal(size=3); w = w / np.linalg.norm(w) * rng.uniform(0, 3.0) if case % 3 else np.zeros(3)
        tgt_l = A + 0.3 * (B - A) + 0.3 * (C - A)
        R64 = np.array([[float(v) for v in row] for row in rodrigues(M(w))])
        tgt = R64 @ tgt_l + t
        o = tgt + rng.normal(size=3) * rng.uniform(3, 6); d = tgt - o; d /= np.linalg.norm(d)
        Rm = rodrigues(M(w))
        ol, dl = to_local(Rm, M(t), M(o), M(d))
        h = tri_hit(M(A), M(B), M(C), ol, dl)
        tt, wa, wb, wc = h[:4]
        if min(wa, wb, wc) <= mpf("1e-6") or tt < 0:
            continue
        rows.append(list(A) + list(B) + list(C) + list(t) + list(w) + list(o) + list(d) + [float(tt)])
    print("ties: %d cases" % len(rows))
    return np.array(rows)


def main():
    rng = np.random.default_rng(0x4B415432)  # "KAT2"
    p = planes(rng); a = aabbs(rng); m, ms = meshes(rng); t = ties(rng)
    np.savez_compressed(os.path.join(HERE, "kat_independent2.npz"), planes=p, aabbs=a, mesh_rays=m, mesh_scenes=ms, ties=t,
                        plane_columns=np.array(["n(3)", "t(3)", "w(3)", "solid", "o(3)", "d(3)", "hit", "toi", "normal(3)", "origin_in_halfspace"]),
                        aabb_columns=np.array(["kind", "p(3)", "t(3)", "w(3)", "min(3)", "max(3)"]),
                        mesh_ray_columns=np.array(["scene", "o(3)", "d(3)", "hit", "toi", "node", "n(3)", "u", "v"]),
                        mesh_scene_columns=np.array(["2 rows per scene: V(24)", "UV(16)", "F(18)", "t(3)", "w(3)"]),
                        tie_columns=np.array(["a(3)", "b(3)", "c(3)", "t(3)", "w(3)", "o(3)", "d(3)", "toi"]))
    print("wrote kat_independent2.npz")


if __name__ == "__main__":
    main()
