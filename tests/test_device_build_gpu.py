"""The device BLAS builder (nrays_amd/csrc/bvh_device.hip) against the host builder (bvh_build.cpp / scene_build.cpp) — both stand for
ncollide's BVT::new_balanced inside TriMesh::new (reference: examples/loader3d.rs:695, src/scene.rs:119-133).  Same split rule, same
f32 arithmetic: from the same references the two must return the SAME 4-wide nodes (planes, child refs, depth-first order) and the
same triangles behind every leaf; a scene whose BLASes were built on the device must render the frame of the host-built scene bit
for bit (a BVT query's answer does not depend on the tree, D-2) with the same ray classes AND the same number of AABB tests."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu


def _mesh(pts, idx):
    pts = np.ascontiguousarray(np.asarray(pts, np.float32).astype(np.float64))
    idx = np.ascontiguousarray(np.asarray(idx, np.uint32))
    m = abi.NraysMesh(len(pts), len(idx), pts.ctypes.data_as(C.POINTER(C.c_double)), None, idx.ctypes.data_as(C.POINTER(C.c_uint32)))
    return m, (pts, idx)


def _build(pts, idx, device, presplit=False):
    m, keep = _mesh(pts, idx)
    cap_r = 12 * len(idx) + 64
    nodes = np.zeros((cap_r, 32), np.float32)
    tri = np.zeros(cap_r, np.uint32)
    d = abi.NraysBlasDump()
    d.node_capacity = cap_r; d.ref_capacity = cap_r
    d.nodes = nodes.ctypes.data_as(C.POINTER(C.c_float)); d.tri_ids = tri.ctypes.data_as(C.POINTER(C.c_uint32))
    abi.check(abi.load_hip_lib().nrays_debug_blas_build(C.byref(m), (1 if device else 0) | (0 if presplit else 2), C.byref(d)))
    return dict(nodes=nodes[:d.num_nodes].copy(), tri=tri[:d.num_refs].copy(), root=d.root, depth=d.max_depth, hairy=d.hairy)


def _leaves(b):
    """{(first, count): sorted triangle ids} over all leaf refs of the tree."""
    refs = b["nodes"][:, 8:12].view(np.int32).ravel() if len(b["nodes"]) else np.asarray([b["root"]], np.int32)
    out = {}
    for r in refs:
        if r < 0 and r != -2**31:
            v = int(~r) & 0xffffffff
            out[(v >> 3, (v & 7) + 1)] = tuple(sorted(b["tri"][(v >> 3):(v >> 3) + (v & 7) + 1].tolist()))
    return out


def _same_tree(h, d):
    assert d["root"] == h["root"] and d["depth"] == h["depth"], (d["root"], h["root"], d["depth"], h["depth"])
    assert d["nodes"].shape == h["nodes"].shape, (d["nodes"].shape, h["nodes"].shape)
    if len(h["nodes"]):
        hp, dp = np.delete(h["nodes"], np.s_[8:12], axis=1), np.delete(d["nodes"], np.s_[8:12], axis=1)
        bad = np.nonzero((hp != dp).any(axis=1))[0]
        assert bad.size == 0, ("first differing node", int(bad[0]), hp[bad[0]], dp[bad[0]])
        hr, dr = h["nodes"][:, 8:12].view(np.int32), d["nodes"][:, 8:12].view(np.int32)
        bad = np.nonzero((hr != dr).any(axis=1))[0]
        assert bad.size == 0, ("first node with other child refs", int(bad[0]), hr[bad[0]], dr[bad[0]])
    assert len(d["tri"]) == len(h["tri"])
    assert _leaves(d) == _leaves(h)


def _soup(n, seed, spread=1.0, size=0.05):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3))
    pts = (c + rng.uniform(-size, size, (n, 3, 3))).reshape(-1, 3)
    return pts, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


@pytest.mark.parametrize("n", [1, 2, 3, 8, 9, 63, 256, 257, 300, 1000, 5000, 70000])
def test_random_soups_give_the_host_tree(gpu, n):
    pts, idx = _soup(n, 100 + n)
    _same_tree(_build(pts, idx, False), _build(pts, idx, True))


def test_clustered_and_flat_meshes_give_the_host_tree(gpu):
    # clusters of very different sizes (uneven SAH splits), a flat sheet (one axis unusable), a line of triangles
    rng = np.random.default_rng(7)
    parts = [_soup(20000, 1, 0.01, 0.001)[0] + [5, 0, 0], _soup(3000, 2, 3.0, 0.2)[0], _soup(40000, 3, 0.5, 0.002)[0] - [0, 4, 0]]
    pts = np.concatenate(parts); idx = np.arange(len(pts), dtype=np.uint32).reshape(-1, 3)
    _same_tree(_build(pts, idx, False), _build(pts, idx, True))
    sheet, si = _soup(6000, 4, 2.0, 0.03); sheet[:, 1] = 0.25
    _same_tree(_build(sheet, si, False), _build(sheet, si, True))
    line, li = _soup(900, 5, 2.0, 0.01); line[:, 1] = 0.0; line[:, 2] = 0.0
    _same_tree(_build(line, li, False), _build(line, li, True))


def test_coinciding_centroids_split_by_index(gpu):
    """Copies of one triangle: every centroid coincides, the builders split by index (bvh_build.cpp: best_axis < 0).  Which copy lands in
    which leaf depends on the order inside the range, so only the shape is compared: same leaf sizes in the same places."""
    tri = np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    for n in (2, 9, 40, 700):
        pts = np.tile(tri, (n, 1)); idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        h, d = _build(pts, idx, False), _build(pts, idx, True)
        assert d["nodes"].shape == h["nodes"].shape and d["depth"] == h["depth"]
        assert sorted(_leaves(d).keys()) == sorted(_leaves(h).keys())
        assert sorted(d["tri"].tolist()) == sorted(h["tri"].tolist()) == list(range(n))
        assert np.array_equal(np.delete(h["nodes"], np.s_[8:12], axis=1), np.delete(d["nodes"], np.s_[8:12], axis=1))


def test_presplit_on_the_device_matches_the_host_rule(gpu):
    """Hair (thin diagonal triangles): both builders classify it hair-like; the device splits against the threshold its budget amounts to
    over ALL pieces (a histogram), the host against the one a sample gives — same rule, nearly the same references; every triangle must
    still be reachable, and a mesh the rule leaves alone must come out exactly like the host's."""
    pts, idx, _ = standins.hairball_geometry(strands=120, sides=6, segments=30)
    h, d = _build(pts, idx, False, presplit=True), _build(pts, idx, True, presplit=True)
    assert h["hairy"] == 1 and d["hairy"] == 1
    assert len(d["tri"]) > 2 * len(idx) and 0.85 * len(h["tri"]) <= len(d["tri"]) <= 1.01 * len(h["tri"]), (len(d["tri"]), len(h["tri"]))
    assert set(d["tri"].tolist()) == set(range(len(idx)))
    assert abs(len(d["nodes"]) - len(h["nodes"])) <= 0.15 * len(h["nodes"])
    pts, idx = _soup(4000, 9, 1.0, 0.2)  # fat random triangles: few qualify, the budget never binds -> the heap and the threshold rule agree
    h, d = _build(pts, idx, False, presplit=True), _build(pts, idx, True, presplit=True)
    assert h["hairy"] == 0 and d["hairy"] == 0 and len(d["tri"]) == len(h["tri"])
    assert np.array_equal(np.delete(h["nodes"], np.s_[8:12], axis=1), np.delete(d["nodes"], np.s_[8:12], axis=1))


def test_presplit_in_one_walk_two_walks_and_after_an_overflow_is_one_tree(gpu, monkeypatch, capfd):
    """Round 6: the counting pass of the pre-splitting keeps its pieces and k_place_extra moves them where the second walk (kModeEmit) would have written them.  The same
    mesh built with the one walk, with NRAYS_PRESPLIT_ONE_WALK=0 (two walks), and with piece regions too small (NRAYS_DEBUG_PIECE_CAP: overflow -> the second walk takes over)
    must give the same nodes and the same triangle behind every reference."""
    pts, idx, _ = standins.hairball_geometry(strands=200, sides=6, segments=40)
    one = _build(pts, idx, True, presplit=True)
    assert one["hairy"] == 1 and len(one["tri"]) > 2 * len(idx)
    monkeypatch.setenv("NRAYS_PRESPLIT_ONE_WALK", "0")
    two = _build(pts, idx, True, presplit=True)
    monkeypatch.delenv("NRAYS_PRESPLIT_ONE_WALK")
    monkeypatch.setenv("NRAYS_DEBUG_PIECE_CAP", "3")
    monkeypatch.setenv("NRAYS_BUILD_TIMES", "1")
    over = _build(pts, idx, True, presplit=True)
    assert "piece list overflow" in capfd.readouterr().err
    for other in (two, over):
        _same_tree(one, other)
        assert np.array_equal(one["tri"], other["tri"])


def test_bad_meshes_are_refused_like_on_the_host(gpu):
    pts, idx = _soup(100, 1)
    bad = idx.copy(); bad[50, 1] = 10**6
    m, keep = _mesh(pts, bad)
    d = abi.NraysBlasDump()
    lib = abi.load_hip_lib()
    assert lib.nrays_debug_blas_build(C.byref(m), 1 | 2, C.byref(d)) == lib.nrays_debug_blas_build(C.byref(m), 2, C.byref(d)) != 0
    p64 = np.asarray(pts, np.float32).astype(np.float64); p64[7, 2] += 1e-12  # not f32-exact
    m = abi.NraysMesh(len(p64), len(idx), p64.ctypes.data_as(C.POINTER(C.c_double)), None, np.ascontiguousarray(idx).ctypes.data_as(C.POINTER(C.c_uint32)))
    assert lib.nrays_debug_blas_build(C.byref(m), 1 | 2, C.byref(d)) == lib.nrays_debug_blas_build(C.byref(m), 2, C.byref(d)) != 0
    assert b"f32" in lib.nrays_last_error()


def _frame(make, w, h, **kw):
    import torch
    sc, cam = make()
    p, _ = su.camera_params(cam, w, h, **kw)
    dev = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    abi.check(abi.load_hip_lib().nrays_render_device_instrumented(sc.device_handle(), C.byref(p), C.c_void_p(dev.data_ptr()), None))
    st = nr.get_stats(sc)
    return dev.cpu().numpy(), st.as_dict()


@pytest.mark.parametrize("scene,res,min_tris", [("hair", (240, 136), 1), ("sponza", (320, 180), 1), ("sponza", (320, 180), 3000)])
def test_device_built_scenes_render_the_host_built_frame(gpu, monkeypatch, scene, res, min_tris):
    """min_tris = 1: every BLAS on the device; 3000: the sponza stand-in's large BLASes on the device and its small ones on the host —
    the mixed layout (device segments first, host refs shifted behind them)."""
    make = (lambda: standins.hairball_scene(strands=300)) if scene == "hair" else (lambda: standins.sponza_scene())
    monkeypatch.setenv("NRAYS_GPU_BUILD", "0")
    want, wst = _frame(make, *res)
    monkeypatch.delenv("NRAYS_GPU_BUILD")
    monkeypatch.setenv("NRAYS_GPU_BUILD_MIN", str(min_tris))
    got, gst = _frame(make, *res)
    assert np.array_equal(got, want), np.abs(got - want).max()
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "hit_records"):
        assert gst[k] == wst[k], (k, gst[k], wst[k])
    if scene == "sponza":  # no hair: pre-splitting is mild and never budget-bound -> the very same trees, the very same visits
        assert gst["node_tests"] == wst["node_tests"], (gst, wst)
        assert abs(gst["tri_tests"] - wst["tri_tests"]) <= 0.01 * wst["tri_tests"]  # (any-hit shadow rays: the order inside a leaf is free)
    else:
        assert abs(gst["node_tests"] - wst["node_tests"]) <= 0.05 * wst["node_tests"], (gst["node_tests"], wst["node_tests"])


def test_full_size_hairball_tree_is_the_host_tree(gpu):
    """BASELINE config 5's mesh at full size (the 2.88 M-triangle stand-in, pre-split into 23 M references): the device builder's 4.9 M
    4-wide nodes equal the host builder's bit for bit, and every leaf holds the same triangles (compared through per-leaf sums of the
    triangle ids and of their squares: the order inside a leaf is free)."""
    pts, idx, _ = standins.hairball_geometry()
    pts = (np.asarray(pts, np.float32) * np.float32(0.25)).astype(np.float32)
    d, h = _build(pts, idx, True, presplit=True), _build(pts, idx, False, presplit=True)
    assert d["hairy"] == h["hairy"] == 1 and d["root"] == h["root"] and d["depth"] == h["depth"]
    assert d["nodes"].shape == h["nodes"].shape and len(d["nodes"]) > 4_000_000 and len(d["tri"]) == len(h["tri"]) > 20_000_000
    assert np.array_equal(np.delete(h["nodes"], np.s_[8:12], axis=1), np.delete(d["nodes"], np.s_[8:12], axis=1))
    refs = h["nodes"][:, 8:12].view(np.int32)
    assert np.array_equal(refs, d["nodes"][:, 8:12].view(np.int32))
    leaf = refs[(refs < 0) & (refs != -2**31)]
    starts = np.sort((~leaf).astype(np.uint32) >> 3).astype(np.int64)
    assert starts[0] == 0 and len(np.unique(starts)) == len(starts)
    hs, ds = h["tri"].astype(np.int64), d["tri"].astype(np.int64)
    assert np.array_equal(np.add.reduceat(hs, starts), np.add.reduceat(ds, starts))
    assert np.array_equal(np.add.reduceat(hs * hs, starts), np.add.reduceat(ds * ds, starts))


def test_degenerate_and_huge_triangles_do_not_derail_the_build(gpu, monkeypatch):
    """A soup of small triangles with a few enormous thin diagonal ones (and points, lines): pre-splitting rates pieces by their EMPTY box
    area, so each of the enormous ones alone could be cut 2^20 times against the small average; the device builder's survey pass is depth-
    capped and its threshold iterates onto the budget (the host builder is budget-bound by its heap).  The build must stay within the budget,
    every triangle must stay reachable, and the scene must render the host-built frame."""
    import time
    import torch
    rng = np.random.default_rng(11)
    pts, idx = _soup(6000, 21, 1.0, 0.01)
    big = []
    for k in range(24):  # thin slivers across the whole scene, two of them degenerate (a line, a point)
        a = rng.uniform(-40, 40, 3); b = a + rng.uniform(-80, 80, 3); c = a + (b - a) * 0.5 + rng.uniform(-1e-3, 1e-3, 3)
        if k == 0: c = b
        if k == 1: b = a; c = a
        big += [a, b, c]
    pts = np.concatenate([pts, np.asarray(big)]).astype(np.float32); idx = np.arange(len(pts), dtype=np.uint32).reshape(-1, 3)
    t0 = time.time(); d = _build(pts, idx, True, presplit=True); dt = time.time() - t0
    h = _build(pts, idx, False, presplit=True)
    assert dt < 2.0, dt
    assert len(d["tri"]) <= 2 * len(idx) + 1024 + len(idx), (len(d["tri"]), len(idx))  # budget 1.0 per triangle (+ the bin the threshold falls in)
    assert set(d["tri"].tolist()) == set(range(len(idx))) == set(h["tri"].tolist())

    def make():
        mat = nr.PhongMaterial((0.1, 0.1, 0.1), (1, 1, 1), (1, 1, 1), None, None, 50.0)
        node = nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), nr.TriMesh(pts.astype(np.float64), idx, None))
        return nr.Scene([node], [nr.Light((0.0, 3.0, -6.0), 0.0, 1, (1, 1, 1))], (1, 1, 1)), dict(eye=(0.0, 0.5, -6.0), at=(0.0, 0.0, 0.0), fovy=40.0)
    monkeypatch.setenv("NRAYS_GPU_BUILD", "0")
    want, wst = _frame(make, 200, 120)
    monkeypatch.delenv("NRAYS_GPU_BUILD")
    monkeypatch.setenv("NRAYS_GPU_BUILD_MIN", "1")
    got, gst = _frame(make, 200, 120)
    assert np.array_equal(got, want), np.abs(got - want).max()
    for k in ("rays_primary", "rays_shadow", "hit_records"):
        assert gst[k] == wst[k], (k, gst[k], wst[k])


def test_merged_group_with_and_without_uvs(gpu, monkeypatch):
    """Two TriMesh nodes under ONE isometry are merged into one BLAS (scene_build.cpp); one carries uvs and a texture, the other none.
    The device builder takes the parts with their own vertex arrays (null uvs -> zero uvs, as the host builder stores them)."""
    import math
    def make():
        pts, idx, uvs = su.torus_mesh()
        iso = nr.Isometry3((0.2, -0.1, 0.5), (0.0, math.radians(25.0), 0.0))
        tex = nr.PhongMaterial((0.2, 0.2, 0.2), (1, 1, 1), (0.5, 0.5, 0.5), su.checker_texture(64, 8), None, 40.0)
        plain = nr.PhongMaterial((0.1, 0.3, 0.1), (0.4, 1, 0.4), (1, 1, 1), None, None, 80.0)
        a = nr.SceneNode(tex, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(pts, idx, uvs))
        b = nr.SceneNode(plain, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(su.f32_exact(np.asarray(pts) * 0.5 + [0.0, 1.2, 0.0]), idx, None))
        return nr.Scene([a, b], [nr.Light((2.0, 4.0, -6.0), 0.0, 1, (1, 1, 1))], (0.2, 0.3, 0.4)), dict(eye=(0.0, 1.0, -7.0), at=(0.0, 0.3, 0.0), fovy=35.0)
    monkeypatch.setenv("NRAYS_GPU_BUILD", "0")
    want, wst = _frame(make, 192, 128)
    monkeypatch.delenv("NRAYS_GPU_BUILD")
    monkeypatch.setenv("NRAYS_GPU_BUILD_MIN", "1")
    got, gst = _frame(make, 192, 128)
    assert np.array_equal(got, want), np.abs(got - want).max()
    # (same tree, so the same AABB tests; the triangle tests of any-hit shadow rays depend on the ORDER inside a leaf, which the host's
    # unstable partition and the device's stable one leave different)
    for k in ("rays_primary", "rays_shadow", "hit_records", "tex_samples", "node_tests"):
        assert gst[k] == wst[k], (k, gst[k], wst[k])
    assert abs(gst["tri_tests"] - wst["tri_tests"]) <= 0.01 * wst["tri_tests"]


def test_list_overflow_falls_back_to_the_host_builder(gpu, monkeypatch, capfd):
    """An internal limit of the device builder (task / small-subtree list overflow) is not the caller's problem: append_blas() reports it on
    stderr and builds that BLAS on the host.  NRAYS_DEBUG_BUILD_CAPS shrinks the lists to 1 / n of their capacity so that the path runs; the
    frame must equal both the host-built and the (unconstrained) device-built one."""
    def make():
        pts, idx = _soup(30000, 5, 2.0, 0.05)
        pts = pts.astype(np.float32)
        mat = nr.PhongMaterial((0.1, 0.1, 0.1), (1, 1, 1), (1, 1, 1), None, None, 50.0)
        node = nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), nr.TriMesh(pts.astype(np.float64), idx, None))
        return nr.Scene([node], [nr.Light((0.0, 3.0, -6.0), 0.0, 1, (1, 1, 1))], (1, 1, 1)), dict(eye=(0.0, 0.5, -6.0), at=(0.0, 0.0, 0.0), fovy=40.0)
    monkeypatch.setenv("NRAYS_GPU_BUILD", "0")
    host, hst = _frame(make, 200, 120)
    monkeypatch.delenv("NRAYS_GPU_BUILD")
    monkeypatch.setenv("NRAYS_GPU_BUILD_MIN", "1")
    dev, dst = _frame(make, 200, 120)
    capfd.readouterr()
    monkeypatch.setenv("NRAYS_DEBUG_BUILD_CAPS", "100000")
    fb, fst = _frame(make, 200, 120)
    msg = capfd.readouterr().err
    assert "overflow" in msg and "building this BLAS on the host" in msg, msg
    assert np.array_equal(fb, host) and np.array_equal(fb, dev)
    for k in ("rays_primary", "rays_shadow", "hit_records", "node_tests"):
        assert fst[k] == hst[k] == dst[k], (k, fst[k], hst[k], dst[k])


def test_device_built_random_scenes_against_the_oracle(gpu, monkeypatch):
    """50 random scenes of tools/fuzz_parity.py (mixed shapes + meshes, triangle soups with ties / duplicates / merged groups, hair) with
    EVERY mesh BLAS built on the device, two frames each (the second runs from the cost order), against the CPU oracle: 1e-4 per channel and
    equal ray classes — the sweep of profiles/r04_fuzz_device_build.log as a test."""
    import torch
    import oracle
    from tools import fuzz_parity as fz
    monkeypatch.setenv("NRAYS_GPU_BUILD_MIN", "1")
    lib = abi.load_hip_lib()
    worst = 0.0
    for seed in range(31000, 31050):
        sc, cam, rng = fz.random_hair_scene(seed) if seed % 3 == 2 else (fz.random_mesh_scene(seed) if seed % 2 else fz.random_scene(seed))
        w, h = int(rng.integers(40, 120)), int(rng.integers(30, 90))
        spp = int(rng.choice([1, 1, 2]))
        p, _ = su.camera_params(cam, w, h, spp=spp, window=float(rng.choice([0.0, 1.0])) if spp > 1 else 0.0, seed=int(seed), max_depth=int(rng.choice([3, 5])))
        ref, ost = oracle.render(sc.descriptor, p, 16)
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        for rep in range(2):
            abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
            st = nr.get_stats(sc)
            err = float(np.abs(out.cpu().numpy() - ref).max())
            worst = max(worst, err)
            assert err <= 1e-4, (seed, rep, err)
            for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
                assert getattr(st, k) == getattr(ost, k), (seed, rep, k, getattr(st, k), getattr(ost, k))
    assert worst <= 1e-4
