"""GPU parity tests: the HIP path (through the C ABI, libnrays_hip.so) against the CPU oracle on the
same descriptors.  Tolerance: 1e-4 per channel on the float image before 8-bit quantisation
(BASELINE.json north_star); ray counts per class must agree exactly."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su

pytestmark = pytest.mark.gpu
TOL = 1e-4


def hip_render(scene, params, instrumented=False):
    import torch
    lib = abi.load_hip_lib()
    rows = lib.nrays_tile_rows(C.byref(params))
    out = torch.empty((rows, params.width, 3), dtype=torch.float32, device="cuda")
    fn = lib.nrays_render_device_instrumented if instrumented else lib.nrays_render_device
    abi.check(fn(scene.device_handle(), C.byref(params), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(scene)
    return out.cpu().numpy(), st


def compare(scene, cam, w, h, threads=8, **kw):
    p, _ = su.camera_params(cam, w, h, **kw)
    ref, ost = oracle.render(scene.descriptor, p, threads)
    img, st = hip_render(scene, p)
    err = np.abs(img - ref)
    assert err.max() <= TOL, "max err %g at %s (mean %g)" % (err.max(), np.unravel_index(err.argmax(), err.shape), err.mean())
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
        assert getattr(st, k) == getattr(ost, k), (k, st.as_dict(), ost.as_dict())
    return img, ref, st, ost


def test_balls_4_bounces(gpu):
    sc, cam = su.balls_scene()
    img, ref, st, _ = compare(sc, cam, 320, 180)
    assert st.rays_reflection > 0 and st.rays_shadow > 0 and st.generations == 4


def test_balls_shipped_refl_gives_five_generations(gpu):
    sc, cam = su.balls_scene(refl=(0.2, 0.2))
    _, _, st, ost = compare(sc, cam, 160, 90)
    assert 4 <= st.generations <= 5  # deepest trace depth actually reached (energy rule allows 5)


def test_primitives_point_light(gpu):
    sc, cam = su.primitives_scene(light_radius=0.0, nsample=1)
    _, _, st, _ = compare(sc, cam, 320, 240)
    assert st.rays_refraction > 0


def test_primitives_area_light_and_aa_share_the_rng(gpu):
    sc, cam = su.primitives_scene(light_radius=0.1, nsample=10)
    _, _, st, _ = compare(sc, cam, 96, 72, spp=3, window=1.0, seed=11)
    assert st.rays_shadow > 9 * 96 * 72


def test_mesh_scene_alpha_mapped_rotated(gpu):
    sc, cam = su.mesh_scene(alpha_mapped=True, rotate=True)
    compare(sc, cam, 256, 192)


def test_mesh_scene_opaque_identity(gpu):
    sc, cam = su.mesh_scene(alpha_mapped=False, rotate=False)
    compare(sc, cam, 200, 150)


@pytest.mark.parametrize("seed", [3, 4])
def test_random_shapes(gpu, seed):
    sc, cam = su.random_shapes_scene(seed, n=30)
    compare(sc, cam, 192, 144)


def test_max_depth_cap(gpu):
    sc, cam = su.balls_scene(refl=(0.3, 0.0))
    _, _, st, _ = compare(sc, cam, 96, 54, max_depth=3)
    assert 1 <= st.generations <= 3


def test_empty_scene_is_background(gpu):
    sc = nr.Scene([], [], (0.25, 0.5, 0.75))
    p, _ = su.camera_params(dict(eye=(0, 0, -5), at=(0, 0, 0), fovy=45.0), 33, 17)
    img, st = hip_render(sc, p)
    assert np.allclose(img, (0.25, 0.5, 0.75)) and st.rays_primary == 33 * 17


def test_ragged_resolutions(gpu):
    sc, cam = su.balls_scene(tex_size=(64, 32))
    for (w, h) in [(1, 1), (17, 3), (130, 67)]:
        compare(sc, cam, w, h, threads=1)


def test_tiled_render_is_bit_identical_to_full_frame(gpu):
    import torch
    sc, cam = su.balls_scene()
    w, h = 200, 117
    full, _ = hip_render(sc, su.camera_params(cam, w, h)[0])
    owners, band = 3, 16
    lib = abi.load_hip_lib()
    tiles = []
    for o in range(owners):
        p, _ = su.camera_params(cam, w, h, band_rows=band, band_owner=o, band_owners=owners)
        t, _ = hip_render(sc, p)
        tiles.append(torch.from_numpy(t).cuda())
    gathered = torch.stack(tiles).contiguous()
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    abi.check(lib.nrays_untile_device(C.c_void_p(gathered.data_ptr()), C.c_void_p(out.data_ptr()), w, h, band, owners, None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full)


def test_host_buffer_entry_point_and_repeatability(gpu):
    sc, cam = su.mesh_scene()
    w, h = 128, 96
    _, proj = su.camera_params(cam, w, h)
    a = nr.render(sc, (w, h), 1, 0.0, cam["eye"], proj)
    b = nr.render(sc, (w, h), 1, 0.0, cam["eye"], proj)
    assert np.array_equal(a, b)
    c, _ = hip_render(sc, su.camera_params(cam, w, h)[0])
    assert np.array_equal(a, c)


def test_instrumented_counters_and_algorithmic_bytes(gpu):
    sc, cam = su.mesh_scene()
    p, _ = su.camera_params(cam, 128, 96)
    img, st = hip_render(sc, p, instrumented=True)
    plain, st2 = hip_render(sc, p)
    assert np.array_equal(img, plain)
    assert st.instrumented == 1 and st2.instrumented == 0
    assert st.node_tests > st.total_rays() and st.tri_tests > 0 and st.hit_records > 0 and st.tex_samples > 0
    assert st.algorithmic_bytes(128, 96) > 64 * st.total_rays()
    assert st.kernel_ms_total > 0 and st.kernel_ms_primary > 0


def test_errors_are_reported_not_fatal(gpu):
    sc, cam = su.balls_scene(tex_size=(64, 32))
    p, _ = su.camera_params(cam, 16, 16)
    p.ray_per_pixel = 0  # assert!(ray_per_pixel > 0), scene.rs:37
    lib = abi.load_hip_lib()
    buf = np.zeros((16, 16, 3), np.float32)
    rc = lib.nrays_render(sc.device_handle(), C.byref(p), buf.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == abi.ERR_BAD_ARG and b"ray_per_pixel" in lib.nrays_last_error()
    bad = nr.TriMesh([[0.1, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], None)  # 0.1 is not f32-exact
    s2 = nr.Scene([nr.SceneNode(su.default_material(), 0, 0, 1, 1, nr.Isometry3(), bad)], [])
    with pytest.raises(abi.NraysError) as e:
        s2.device_handle()
    assert e.value.status == abi.ERR_UNSUPPORTED


def test_full_size_balls_properties(gpu):
    """BASELINE config 2 at full size (1920x1080, 4 bounces): too slow for the scalar oracle in a unit
    test, so check size-independent properties: the 320x180 oracle frame equals the full frame on
    the pixels whose corner rays coincide (every 6th pixel), and tiling does not change a bit."""
    sc, cam = su.balls_scene()
    W, H = 1920, 1080
    full, st = hip_render(sc, su.camera_params(cam, W, H)[0])
    assert st.rays_primary == W * H and st.generations == 4
    small, _ = oracle.render(sc.descriptor, su.camera_params(cam, 320, 180)[0], 8)
    assert np.abs(full[::6, ::6] - small).max() <= TOL
    p, _ = su.camera_params(cam, W, H, band_rows=16, band_owner=1, band_owners=2)
    tile, _ = hip_render(sc, p)
    rows = [j for j in range(H) if (j // 16) % 2 == 1]
    assert np.array_equal(tile[: len(rows)], full[rows])


def test_sponza_standin_including_the_symmetry_plane_ties(gpu):
    """The stand-in has a column of coincident pole vertices exactly in the camera's symmetry plane:
    the centre pixel column (d.z == 0) hits ~160 triangles at the same toi.  Reference semantics =
    lexicographic min (toi, node, triangle) over hits whose node AABB and triangle AABB pass the exact
    ncollide slab test; the GPU culls conservatively in f32 and gates accepted hits with those tests."""
    from tools import standins
    sc, cam = standins.sponza_scene()
    compare(sc, cam, 160, 90, threads=32)
    compare(sc, cam, 96, 54, threads=32, max_depth=2)


def test_hairball_standin_small(gpu):
    from tools import standins
    sc, cam = standins.hairball_scene(strands=400)
    compare(sc, cam, 128, 128, threads=32)


def test_double_branching_uses_the_compacted_queue(gpu):
    """A node with refl_mix != 0 AND alpha != 1 spawns a reflection and a refraction at one hit
    (scene.rs:175-181): the reflection continues in registers, the refraction goes through the
    ballot-compacted HBM queue and k_bounce."""
    glass = nr.PhongMaterial((0.1, 0.1, 0.15), (0.6, 0.7, 0.9), (1, 1, 1), None, None, 80.0)
    iso = nr.Isometry3
    nodes = [nr.SceneNode(glass, 0.3, 0.4, 0.5, 1.3, iso((-1.2, 0, 0)), nr.Ball(1.0)),
             nr.SceneNode(glass, 0.3, 0.4, 0.5, 1.3, iso((1.2, 0, 0.5)), nr.Cuboid((0.7, 0.7, 0.7))),
             nr.SceneNode(su.default_material(), 0.25, 0.5, 1.0, 1.0, iso((0, -1.2, 0)), nr.Plane((0, 1, 0))),
             nr.SceneNode(nr.NormalMaterial(), 0.0, 0.0, 1.0, 1.0, iso((0, 0.3, 3.0)), nr.Ball(0.8))]
    sc = nr.Scene(nodes, [nr.Light((2.0, 6.0, -4.0), 0.0, 1, (1, 1, 1))])
    cam = dict(eye=(0.0, 2.0, -7.0), at=(0.0, 0.0, 0.0), fovy=45.0)
    img, _, st, _ = compare(sc, cam, 160, 120)
    assert st.rays_reflection > 0 and st.rays_refraction > 0 and st.generations >= 3
    # the queued chains of a pixel are summed in fixed point (k_bounce / k_fold_fixed): the frame does not depend on the
    # order in which they finish
    p, _ = su.camera_params(cam, 160, 120)
    for _ in range(3):
        again, _ = hip_render(sc, p)
        assert np.array_equal(again, img)


def test_sample_batching_multiple_primary_launches(gpu, monkeypatch):
    """ray_per_pixel > 1 with the per-launch primary-ray budget forced low: several k_primary launches
    accumulate into the frame, then k_resolve divides by ray_per_pixel (scene.rs:91-94)."""
    monkeypatch.setenv("NRAYS_MAX_PRIMARY", str(96 * 72 * 2))  # 2 samples per launch -> 3 launches for spp 5
    sc, cam = su.primitives_scene(light_radius=0.1, nsample=10)
    _, _, st, _ = compare(sc, cam, 96, 72, spp=5, window=1.0, seed=3)
    assert st.rays_primary == 96 * 72 * 5
    sc2, cam2 = su.mesh_scene()
    compare(sc2, cam2, 96, 72, spp=5, window=0.7, seed=9)


def test_area_lights_on_meshes_and_eight_lights(gpu):
    from tools import standins
    sc, cam = standins.sponza_scene(detail=0.1, n_lights=8)
    compare(sc, cam, 96, 54, threads=32)
    sc, cam = su.mesh_scene(n_lights=2)
    sc._lights = [nr.Light(l.pos, 0.3, 4, l.color) for l in sc._lights]
    sc._descriptor = None
    _, _, st, _ = compare(sc, cam, 80, 60, seed=5)
    assert st.rays_shadow > 4 * 2 * 80 * 60 * 0.3


def test_one_scene_handle_many_cameras_and_resolutions(gpu):
    """The raygen tables are cached per (camera, resolution) on the scene handle: changing either between
    renders of the same handle must rebuild them (and jittered frames must bypass them)."""
    sc, cam = su.balls_scene(tex_size=(64, 32))
    cams = [cam, dict(cam, eye=(3.0, 2.0, -7.0)), dict(cam, at=(0.5, 0.2, 0.0), fovy=60.0), cam]
    for c, (w, h) in zip(cams, [(160, 90), (160, 90), (96, 120), (200, 64)]):
        compare(sc, c, w, h)
    compare(sc, cam, 64, 48, spp=2, window=1.0, seed=5)
    compare(sc, cam, 64, 48)


def test_empty_screen_tiles_and_single_leaf_roots(gpu):
    """Wave tiles whose rays all miss the root of the BVT are finished without entering the trace loop: a camera
    that looks away (every tile), a scene whose root is a single leaf (no box above it), a scene with a plane
    (never skipped), banded tiles with padding rows."""
    sc, cam = su.balls_scene(tex_size=(64, 32))
    away = dict(cam, at=(0.0, 5.0, -30.0))
    img, _, st, _ = compare(sc, away, 150, 70)
    assert st.rays_shadow == 0 and np.all(img == 1.0)
    one = nr.Scene([nr.SceneNode(su.default_material(), 0.3, 0.5, 1.0, 1.0, nr.Isometry3((0.5, 0.0, 0.0)), nr.Ball(1.0))],
                   [nr.Light((0.0, 10.0, 0.0), 0.0, 1, (1, 1, 1))], (0.1, 0.2, 0.3))
    compare(one, cam, 120, 80)
    prim, pcam = su.primitives_scene(light_radius=0.0, nsample=1)
    compare(prim, dict(pcam, at=(0.0, 8.0, 0.0)), 96, 64)  # half of the frame sees only the sky... and the plane's horizon
    import torch
    from nrays_amd import tiling
    lib = abi.load_hip_lib()
    full, _ = su.camera_params(cam, 100, 52)
    ref, _ = oracle.render(sc.descriptor, full, 4)
    for rank in range(3):
        p = tiling.tile_params(full, rank, 3, 16)
        rows = lib.nrays_tile_rows(C.byref(p))
        out = torch.full((rows, 100, 3), -7.0, dtype=torch.float32, device="cuda")
        abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        got = out.cpu().numpy()
        own = tiling.owned_rows(52, 16, rank, 3)
        assert np.abs(got[:len(own)] - ref[own]).max() <= TOL
        assert np.all(got[len(own):] == 0.0)  # padding rows of the last band are zero-filled, also by skipped tiles


def test_mesh_frames_scheduled_from_last_frames_tile_costs(gpu):
    """Mesh scenes order their wave tiles by the previous frame's per-tile cost (k_tile_order): frames 2.. of one
    handle and geometry run through the sorted work lists and must stay identical to frame 1 and to the oracle,
    also after a geometry change and with band tiling."""
    import torch
    from nrays_amd import tiling
    lib = abi.load_hip_lib()
    sc, cam = su.mesh_scene()
    p, _ = su.camera_params(cam, 200, 120)
    ref, _ = oracle.render(sc.descriptor, p, 8)
    frames = [hip_render(sc, p)[0] for _ in range(4)]
    assert np.abs(frames[0] - ref).max() <= TOL
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])
    p2, _ = su.camera_params(cam, 97, 61)  # geometry change: the history of the 200x120 frames must not be used
    ref2, _ = oracle.render(sc.descriptor, p2, 8)
    for _ in range(3):
        assert np.abs(hip_render(sc, p2)[0] - ref2).max() <= TOL
    for _ in range(2):
        assert np.array_equal(hip_render(sc, p)[0], frames[0])
    bp = tiling.tile_params(p, 1, 2, 16)
    own = tiling.owned_rows(120, 16, 1, 2)
    for _ in range(3):
        got = hip_render(sc, bp)[0]
        assert np.array_equal(got[:len(own)], frames[0][own])


def test_randomised_scenes_sweep(gpu):
    """tools/fuzz_parity.py on a fixed seed range: random shape mixes, rotations, reflection / refraction / alpha
    coefficients, light sets, AA windows, depth caps; every case twice (the second frame of a mesh scene runs
    through the cost-ordered work lists).  Pixels within 1e-4, ray-class counts exactly equal."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "5000", "60"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 mismatches" in r.stdout


def test_distinct_scenes_render_concurrently_from_two_threads(gpu):
    """SURVEY 8b threading contract: the library is re-entrant on distinct scene handles (each call on its own
    HIP stream, from its own host thread); the frames must equal the ones rendered alone."""
    import threading
    import torch
    lib = abi.load_hip_lib()
    sa, ca = su.balls_scene(tex_size=(64, 32))
    sb, cb = su.mesh_scene()
    pa, _ = su.camera_params(ca, 320, 200)
    pb, _ = su.camera_params(cb, 240, 160)
    alone_a, _ = hip_render(sa, pa)
    alone_b, _ = hip_render(sb, pb)
    results, errors = {}, []

    def worker(name, scene, params, shape):
        try:
            stream = torch.cuda.Stream()
            out = torch.empty(shape, dtype=torch.float32, device="cuda")
            for _ in range(40):
                abi.check(lib.nrays_render_device(scene.device_handle(), C.byref(params), C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(stream.cuda_stream)))
            stream.synchronize()
            results[name] = out.cpu().numpy()
        except Exception as e:  # pragma: no cover - reported below
            errors.append((name, repr(e)))

    ta = threading.Thread(target=worker, args=("a", sa, pa, (200, 320, 3)))
    tb = threading.Thread(target=worker, args=("b", sb, pb, (160, 240, 3)))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errors, errors
    assert np.array_equal(results["a"], alone_a) and np.array_equal(results["b"], alone_b)


def test_long_thin_diagonal_triangles(gpu):
    """Needle field: hundreds of long, thin, randomly oriented triangles — the references the BLAS builder pre-splits
    (their boxes are almost empty).  Every piece box must still lead to the whole triangle: pixels and ray counts
    equal to the oracle, for a rotated and an unrotated node, with reflections bouncing between the needles."""
    rng = np.random.default_rng(77)
    nt = 600
    a = rng.uniform(-4, 4, (nt, 3))
    d = rng.normal(size=(nt, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    q = np.cross(d, rng.normal(size=(nt, 3))); q /= np.linalg.norm(q, axis=1, keepdims=True)
    length = rng.uniform(2.0, 7.0, (nt, 1)); width = rng.uniform(0.01, 0.15, (nt, 1))
    tri = su.f32_exact(np.stack([a, a + d * length, a + d * length * 0.5 + q * width], axis=1).reshape(-1, 3))
    idx = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
    uvs = su.f32_exact(rng.uniform(0, 1, (3 * nt, 2)))
    mat = nr.PhongMaterial((0.1, 0.1, 0.1), (0.9, 0.6, 0.3), (1, 1, 1), su.checker_texture(32, 4), None, 40.0)
    half = 3 * (nt // 2)
    nodes = [nr.SceneNode(mat, 0.3, 0.4, 1.0, 1.0, nr.Isometry3((0, 0, 0)), nr.TriMesh(tri[:half], idx[:nt // 2], uvs[:half])),
             nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.3, -0.2, 0.1), (0.2, 0.7, -0.4)),
                          nr.TriMesh(tri[half:], idx[:nt - nt // 2], uvs[half:]))]
    sc = nr.Scene(nodes, [nr.Light((3.0, 9.0, -6.0), 0.0, 1, (1, 1, 1)), nr.Light((-5.0, 4.0, -8.0), 0.0, 1, (0.4, 0.4, 0.6))], (0.2, 0.3, 0.4))
    cam = dict(eye=(1.0, 2.0, -13.0), at=(0.0, 0.0, 0.0), fovy=45.0)
    _, _, st, _ = compare(sc, cam, 400, 300)
    assert st.rays_shadow > 0 and st.rays_reflection > 0
    compare(sc, dict(cam, eye=(-9.0, -3.0, 6.0)), 233, 171)


def test_rays_in_a_coordinate_plane_with_box_faces_in_that_plane(gpu):
    """Regression for the round-2 culling fix: a fan of triangles around the camera axis has vertices exactly in
    the planes x = 0 and y = 0, so many leaf boxes have a FACE in those planes, and with an even resolution the centre
    column / row of pixels have rays with d.x == 0 / d.y == 0 exactly and an origin coordinate of exactly 0 — the case in
    which a zero margin used to cull boxes the reference's `o < mn || o > mx` rule accepts.  Every pixel against the
    oracle, for an identity and a translated node (the translated one moves the planes off the f32 grid's zero)."""
    n = 48
    ang = np.arange(n) * (2 * np.pi / n)
    ring = np.stack([2.0 * np.cos(ang), 2.0 * np.sin(ang), np.full(n, 8.0)], 1)
    ring[np.abs(ring) < 1e-12] = 0.0  # the four axis points exactly on x = 0 / y = 0
    pts = su.f32_exact(np.concatenate([[[0.0, 0.0, 5.0]], ring, [[0.0, 0.0, 8.0]]]))
    tris = []
    for k in range(n):
        a, b = 1 + k, 1 + (k + 1) % n
        tris.append([0, a, b])          # cone side: apex on the camera axis
        tris.append([n + 1, b, a])      # base disc: centre on the camera axis
    idx = np.array(tris, dtype=np.uint32)
    uvs = su.f32_exact(np.random.default_rng(5).uniform(0, 1, (len(pts), 2)))
    mat = nr.PhongMaterial((0.2, 0.2, 0.2), (0.8, 0.7, 0.6), (1, 1, 1), su.checker_texture(32, 4), None, 30.0)
    for trans in [(0.0, 0.0, 0.0), (0.0, 0.0, 1.5)]:
        node = nr.SceneNode(mat, 0.3, 0.5, 1.0, 1.0, nr.Isometry3(trans), nr.TriMesh(pts, idx, uvs))
        floor = nr.SceneNode(su.default_material(), 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0)),
                             nr.TriMesh(su.f32_exact([[-6, -6, 12], [6, -6, 12], [6, 6, 12], [-6, 6, 12], [0, 0, 12]]),
                                        np.array([[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]], dtype=np.uint32), None))
        sc = nr.Scene([node, floor], [nr.Light((0.0, 0.0, -5.0), 0.0, 1, (1, 1, 1)), nr.Light((3.0, 4.0, 0.0), 0.0, 1, (0.5, 0.5, 0.5))], (0.1, 0.2, 0.3))
        cam = dict(eye=(0.0, 0.0, -5.0), at=(0.0, 0.0, 0.0), fovy=50.0)
        compare(sc, cam, 256, 192)   # centre column i = 128: d.x == 0; centre row j = 96: d.y == 0
        compare(sc, cam, 64, 64)
