"""GPU tests of the rows SURVEY 8f marks next-1..next-3, end to end on the HIP path:
`.scene` / `.mtl` / `.obj` / PNG files -> libnrays_host.so (src/obj.rs:51-397, src/mtl.rs:18-189,
src/texture2d.rs:78-201, examples/loader3d.rs:214-790) -> nrays_render -> float frame within 1e-4 of the oracle on the
same descriptor, ray classes exactly equal, and the PNG written with the reference's quantisation (src/image.rs:60-90)
equal to the oracle's frame quantised the same way."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi, scenefile
from tests import test_frontend as tf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _quantise(img):
    return np.clip(img * np.float32(255.0), 0, 255).astype(np.uint8)  # image.rs:66-76: x255, clamp, truncate


def _hip_vs_oracle(fs, w, h, tmp_path, spp=None, window=None, seed=0, exact_png=True):
    cam = fs.camera_dict()
    spp = cam["aa"][0] if spp is None else spp
    window = cam["aa"][1] if window is None else window
    p = nr.make_params((w, h), spp, window, cam["eye"], fs.inverse_projection(0, w, h), seed=seed)
    img = np.empty((h, w, 3), np.float32)
    abi.check(abi.load_hip_lib().nrays_render(fs.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
    st = nr.get_stats(fs)
    ref, ost = oracle.render(fs.descriptor, p, 32)
    assert np.abs(img - ref).max() <= TOL, np.abs(img - ref).max()
    for k in CLASSES:
        assert getattr(st, k) == getattr(ost, k), (k, st.as_dict(), ost.as_dict())
    out = str(tmp_path / "frame.png")
    scenefile.write_png(out, img)
    got, want = scenefile.read_png(out).astype(np.int16), _quantise(ref).astype(np.int16)
    d = np.abs(got - want)
    # the float frames agree to ~1e-7, so a channel can only differ where c*255 sits within 1e-4 of an integer
    assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (d.max(), (d != 0).sum())
    if exact_png:
        assert d.max() == 0
    return img, st


@pytest.fixture(scope="module")
def globe(gpu):
    from tools import gen_assets
    return gen_assets.gen_globe()


def test_balls_scene_file_on_the_hip_path(globe, tmp_path):
    fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "balls.scene"))
    assert fs.camera_dict()["resolution"] == (1920, 1080)
    _, st = _hip_vs_oracle(fs, 320, 180, tmp_path)
    assert st.generations == 4 and st.rays_reflection > 0


def test_primitives_scene_file_on_the_hip_path(globe, tmp_path):
    """BASELINE config 1's file as shipped: area light (radius 0.1, nsample 10 -> 9 samples), transparent box / cone /
    cylinder, reflecting plane; counter-based RNG seed 7."""
    fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "primitives.scene"))
    _, st = _hip_vs_oracle(fs, 320, 240, tmp_path, seed=7, exact_png=False)
    assert st.rays_refraction > 0 and st.rays_shadow > 9 * 0.3 * 320 * 240


def test_sponza_scene_file_with_generated_assets_on_the_hip_path(gpu, tmp_path):
    """scenes/crytek_sponza.scene over the OBJ + MTL + PNG files written by tools/gen_assets.py (the stand-in for the
    asset upstream does not ship): `usemtl` group splitting, (v,t) de-duplication, the /4 scale, map_Kd / map_d decode."""
    import tools.gen_assets as ga
    old = ga.MEDIA
    ga.MEDIA = str(tmp_path / "media")
    try:
        ga.gen_sponza(0.15)
    finally:
        ga.MEDIA = old
    text = open(os.path.join(ROOT, "scenes", "crytek_sponza.scene")).read()
    fs = scenefile.FileScene(tf._scene(tmp_path, text))
    assert fs.descriptor.desc.num_nodes > 100 and fs.descriptor.desc.num_textures >= 8
    _, st = _hip_vs_oracle(fs, 160, 90, tmp_path, exact_png=False)
    assert st.rays_refraction > 0  # alpha-mapped foliage


def test_sponza_textures_as_tga_give_the_same_frame_as_png(gpu, tmp_path):
    """The Crytek Sponza distribution ships its textures as TGA; the reference decodes them through stb_image
    (src/texture2d.rs:95).  The stand-in's textures written as TGA (run-length coded colour maps, raw 8-bit opacity maps,
    bottom-up) must give the SAME texels and therefore the same frame, bit for bit, as the PNG files."""
    import tools.gen_assets as ga
    frames = {}
    for ext in ("png", "tga"):
        root = tmp_path / ext
        root.mkdir()
        old = ga.MEDIA
        ga.MEDIA = str(root / "media")
        try:
            ga.gen_sponza(0.15, ext)
        finally:
            ga.MEDIA = old
        assert any(f.endswith("." + ext) for f in os.listdir(str(root / "media" / "crytek-sponza" / "textures")))
        text = open(os.path.join(ROOT, "scenes", "crytek_sponza.scene")).read()
        fs = scenefile.FileScene(tf._scene(root, text))
        assert fs.descriptor.desc.num_textures >= 8
        cam = fs.camera_dict()
        p = nr.make_params((160, 90), 1, 0.0, cam["eye"], fs.inverse_projection(0, 160, 90))
        img = np.empty((90, 160, 3), np.float32)
        abi.check(abi.load_hip_lib().nrays_render(fs.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
        frames[ext] = (img, nr.get_stats(fs).total_rays())
    assert frames["png"][1] == frames["tga"][1] and np.array_equal(frames["png"][0], frames["tga"][0])


def test_obj_quirks_on_the_hip_path(gpu, tmp_path):
    """src/obj.rs:232-273,334-366: the on-the-fly fan (v0,v1,v2),(v0,v2,v3),(v2,v3,v4), negative indices, a second
    `usemtl` splitting a group, `d 0.5` as node alpha — rendered, not just parsed."""
    scene = tf.SCENE.replace("resolution 8 8", "resolution 96 96") + "light\n pos 2 2 -10\n color 1 1 1\n"
    fs = scenefile.FileScene(tf._scene(tmp_path, scene, {"o.obj": tf.OBJ, "m.mtl": tf.MTL}))
    img, st = _hip_vs_oracle(fs, 96, 96, tmp_path)
    assert st.rays_refraction > 0 and st.rays_shadow > st.rays_refraction  # the half-transparent quad continues behind itself; every hit is lit


def test_loader3d_cli_end_to_end(globe, tmp_path):
    """The native CLI (examples/loader3d.rs:34-101 over the C ABI, no Python, no torch in the process): parse, render,
    quantise, write the PNG — also with the frame tiled over 3 band owners through nrays_render_multi."""
    exe = os.path.join(ROOT, "nrays_amd", "lib", "loader3d")
    scene = os.path.join(ROOT, "scenes", "balls.scene")
    fs = scenefile.FileScene(scene)
    cam = fs.camera_dict()
    p = nr.make_params((160, 90), 1, 0.0, cam["eye"], fs.inverse_projection(0, 160, 90))
    ref, _ = oracle.render(fs.descriptor, p, 8)
    pngs = []
    for extra in (["--times"], ["--gpus", "3"]):
        r = subprocess.run([exe, scene, "--width", "160", "--height", "90"] + extra, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        assert "Rays cast." in r.stdout and "Image saved." in r.stdout
        if "--times" in extra:  # the stage times behind bench.py's drop_in_end_to_end block: one JSON line, every stage present, the stages within the total
            import json
            line = [l for l in r.stdout.splitlines() if l.startswith('{"loader3d_times_ms"')]
            assert len(line) == 1
            t = json.loads(line[0])
            ms = t["loader3d_times_ms"]
            assert set(ms) == {"parse_scene_obj_mtl_textures", "dlopen_libnrays_hip", "nrays_scene_create_incl_hip_init", "render_cold_incl_d2h", "render_gpu_events", "quantise_encode_write_image", "total"}
            assert t["cameras"] == 1 and t["rays"] > 160 * 90 and ms["render_gpu_events"] > 0.0
            assert sum(v for k, v in ms.items() if k not in ("total", "render_gpu_events")) <= ms["total"] * 1.001
        pngs.append(open(str(tmp_path / "out.png"), "rb").read())
        os.remove(str(tmp_path / "out.png"))
    assert pngs[0] == pngs[1]  # tiling does not change a byte
    (tmp_path / "cli.png").write_bytes(pngs[0])
    got = scenefile.read_png(str(tmp_path / "cli.png"))
    assert np.array_equal(got, _quantise(ref))


def test_render_rgb8_is_the_host_quantisation_of_the_float_frame(gpu):
    """nrays_render_rgb8 quantises on the device with Image::to_png's rule (src/image.rs:66-76: c * 255, clamped, truncated, NaN -> 0):
    byte for byte what the host front-end's quantize_rgb8 makes of nrays_render's float frame — also for a tiled render."""
    import ctypes as C
    import numpy as np
    import nrays_amd as nr
    from nrays_amd import abi, tiling
    from tools import scenes_util as su
    lib = abi.load_hip_lib()
    for make in (su.balls_scene, lambda: su.mesh_scene()):
        sc, cam = make()
        for kw in (dict(), dict(spp=3, window=1.0, seed=5), dict(band_rows=16, band_owner=1, band_owners=2)):
            p, _ = su.camera_params(cam, 333, 190, **kw)
            rows = tiling.tile_rows(p.height, p.band_rows, p.band_owners) if p.band_owners > 1 else p.height
            f = np.empty((rows, p.width, 3), np.float32)
            q = np.empty((rows, p.width, 3), np.uint8)
            abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), f.ctypes.data_as(C.POINTER(C.c_float))))
            abi.check(lib.nrays_render_rgb8(sc.device_handle(), C.byref(p), q.ctypes.data_as(C.POINTER(C.c_uint8))))
            v = f * np.float32(255.0)
            v = np.where(v > 0, v, np.float32(0.0))
            v = np.minimum(v, np.float32(255.0))
            assert np.array_equal(q, v.astype(np.uint8)), kw
