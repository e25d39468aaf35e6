"""CPU tests of the C++ loader3d front-end (nrays_amd/host): `.scene` grammar, MTL, OBJ quirks,
texture decode, PNG codec, camera set-up — the rows SURVEY §8f marks next-1..next-3."""
import os
import subprocess

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi, math3d, scenefile
from tools import scenes_util as su

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(built):
    return scenefile.host_lib()


def test_balls_scene_file_equals_the_programmatic_scene(host):
    fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "balls.scene"), allow_standins=True)
    cam = fs.camera_dict()
    assert cam["resolution"] == (1920, 1080) and cam["aa"] == (1, 0.0) and cam["output"] == "out.png"
    m = fs.inverse_projection(0, 64, 36)
    assert np.abs(m - math3d.inverse_projection(cam["eye"], cam["at"], cam["fovy"], 64, 36)).max() < 1e-13
    p = nr.make_params((64, 36), 1, 0.0, cam["eye"], m)
    a, _ = oracle.render(fs.descriptor, p, 2)
    sc, _ = su.balls_scene()
    b, _ = oracle.render(sc.descriptor, p, 2)
    assert np.array_equal(a, b)


def test_primitives_scene_file(host):
    fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "primitives.scene"), allow_standins=True)
    d = fs.descriptor.desc
    assert d.num_nodes == 5 and d.num_lights == 1
    assert d.lights[0].racsample == 3 and d.lights[0].radius == pytest.approx(0.1)  # floor(sqrt(10)), light.rs:20
    kinds = [d.nodes[i].shape_kind for i in range(5)]
    assert kinds == [abi.SHAPE_BALL, abi.SHAPE_CUBOID, abi.SHAPE_CONE, abi.SHAPE_CYLINDER, abi.SHAPE_PLANE]
    assert d.nodes[1].alpha == pytest.approx(0.2) and d.nodes[1].refr_coeff == 1.5  # `d 0.2` of transparent_red
    assert d.nodes[4].refl_mix == pytest.approx(0.2) and d.nodes[4].refl_atenuation == pytest.approx(0.5)
    sc, cam = su.primitives_scene(0.1, 10)
    p, _ = su.camera_params(cam, 48, 36, seed=3)
    a, _ = oracle.render(fs.descriptor, p, 2)
    b, _ = oracle.render(sc.descriptor, p, 2)
    assert np.array_equal(a, b)


def test_missing_asset_and_attribute_errors(host, tmp_path):
    with pytest.raises(RuntimeError, match="Image not found"):
        scenefile.FileScene(os.path.join(ROOT, "scenes", "balls.scene") if not os.path.exists(os.path.join(ROOT, "scenes", "media", "globe.png"))
                            else _scene(tmp_path, "mtllib m.mtl\n", {"m.mtl": "newmtl a\nmap_Kd nope.png\n"}))
    with pytest.raises(RuntimeError, match="missing attribute: pos"):
        scenefile.FileScene(_scene(tmp_path, "light\n color 1 1 1\n"))
    with pytest.raises(RuntimeError, match="unknown material"):
        scenefile.FileScene(_scene(tmp_path, "geometry\n ball 1\n pos 0 0 0\n angle 0 0 0\n material nope\n"))
    with pytest.raises(RuntimeError, match="failed to parse"):
        scenefile.FileScene(_scene(tmp_path, "light\n pos 0 x 0\n color 1 1 1\n"))


def _scene(tmp_path, text, extra=None):
    for name, content in (extra or {}).items():
        (tmp_path / name).write_text(content)
    p = tmp_path / "t.scene"
    p.write_text(text)
    return str(p)


OBJ = """# quad + pentagon + negative indices + two usemtl in one group
mtllib m.mtl
v 0 0 0
v 4 0 0
v 4 4 0
v 0 4 0
v 2 6 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vt 0.5 1.5
g wall
usemtl red
f 1/1 2/2 3/3 4/4
usemtl blue
f -5/-5 -4/-4 -3/-3 -2/-2 -1/-1
"""
MTL = "newmtl red\nKd 1 0 0\nd 0.5\n\nnewmtl blue\nKd 0 0 1\nNs 10\n"
SCENE = "camera\n output o.png\n resolution 8 8\n eye 2 2 -10\n at 2 2 0\n fovy 45\n" \
        "geometry\n obj o.obj .\n pos 0 0 0\n angle 0 0 0\n material default\n"


def test_obj_loader_quirks(host, tmp_path):
    fs = scenefile.FileScene(_scene(tmp_path, SCENE, {"o.obj": OBJ, "m.mtl": MTL}))
    d = fs.descriptor.desc
    assert d.num_nodes == 2  # a second usemtl inside one group splits it (obj.rs:146-158)
    m0, m1 = d.meshes[d.nodes[0].mesh_id], d.meshes[d.nodes[1].mesh_id]
    assert m0.num_triangles == 2 and m1.num_triangles == 3
    v = np.ctypeslib.as_array(m0.vertices, (m0.num_vertices * 3,)).reshape(-1, 3)
    assert np.allclose(v.max(0), (1.0, 1.5, 0.0))  # coordinates divided by 4 (loader3d.rs:669)
    i0 = np.ctypeslib.as_array(m0.indices, (6,)).reshape(2, 3)
    assert (v[i0[0]] * 4).tolist() == [[0, 0, 0], [4, 0, 0], [4, 4, 0]] and (v[i0[1]] * 4).tolist() == [[0, 0, 0], [4, 4, 0], [0, 4, 0]]
    i1 = np.ctypeslib.as_array(m1.indices, (9,)).reshape(3, 3)
    # the reference's on-the-fly "fan" (obj.rs:232-239): (v0,v1,v2), (v0,v2,v3), then (v2,v3,v4) — not (v0,v3,v4)
    assert (v[i1[2]] * 4).tolist() == [[4, 4, 0], [0, 4, 0], [2, 6, 0]]
    assert d.nodes[0].alpha == pytest.approx(0.5) and d.nodes[1].alpha == pytest.approx(1.0)
    mats = [d.materials[d.nodes[k].material_id] for k in range(2)]
    assert tuple(mats[0].diffuse) == (1, 0, 0) and mats[0].shininess == 60.0  # mtl.rs:149-161 defaults
    assert tuple(mats[1].diffuse) == (0, 0, 1) and mats[1].shininess == 10.0 and tuple(mats[1].ambiant) == (1, 1, 1)
    uv = np.ctypeslib.as_array(m0.uvs, (m0.num_vertices * 2,)).reshape(-1, 2)
    assert uv.max() == 1.5
    # special materials keep the scene material but still take the mtl alpha (loader3d.rs:753-762)
    fs2 = scenefile.FileScene(_scene(tmp_path, SCENE.replace("material default", "material normals"), {"o.obj": OBJ, "m.mtl": MTL}))
    d2 = fs2.descriptor.desc
    assert d2.materials[d2.nodes[0].material_id].kind == abi.MAT_NORMAL and d2.nodes[0].alpha == pytest.approx(0.5)


def test_obj_without_uvs_gets_zero_uvs(host, tmp_path):
    obj = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"
    fs = scenefile.FileScene(_scene(tmp_path, SCENE, {"o.obj": obj}))
    d = fs.descriptor.desc
    m = d.meshes[d.nodes[0].mesh_id]
    assert d.num_nodes == 1 and bool(m.uvs) and np.ctypeslib.as_array(m.uvs, (6,)).max() == 0.0  # obj.rs:377
    assert d.materials[d.nodes[0].material_id].kind == abi.MAT_PHONG


def test_png_codec_against_pillow(host, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.random((37, 53, 3)).astype(np.float32) * 1.2 - 0.1
    p = str(tmp_path / "a.png")
    scenefile.write_png(p, img)
    want = np.clip(img * np.float32(255.0), 0, 255).astype(np.uint8)  # image.rs:66-76: clamp then truncate
    assert np.array_equal(np.asarray(Image.open(p)), want)
    assert np.array_equal(scenefile.read_png(p), want)
    for mode, ch in (("RGB", 3), ("RGBA", 4), ("L", 1), ("LA", 2)):
        q = str(tmp_path / (mode + ".png"))
        src = Image.fromarray(want).convert(mode)
        src.save(q, optimize=True)  # dynamic-Huffman zlib + PNG filters
        got = scenefile.read_png(q)
        assert got.shape[2] == ch and np.array_equal(got.reshape(np.asarray(src).shape), np.asarray(src))


def test_texture_decode_matches_python_mirror(host, tmp_path):
    """map_Kd / map_d textures decoded by the C++ path equal nrays_amd.scene.ImageData.from_image_rows."""
    from PIL import Image
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (6, 5), dtype=np.uint8)
    Image.fromarray(rgb).save(str(tmp_path / "kd.png"))
    Image.fromarray(gray).save(str(tmp_path / "d.png"))
    mtl = "newmtl m\nmap_Kd kd.png\nmap_d d.png\n"
    fs = scenefile.FileScene(_scene(tmp_path, "mtllib m.mtl\n", {"m.mtl": mtl}))
    d = fs.descriptor.desc
    assert d.num_textures == 2
    t0, t1 = d.textures[0], d.textures[1]
    a = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (6 * 5 * 4)).from_address(t0.texels)).reshape(6, 5, 4)
    b = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (6 * 5 * 4)).from_address(t1.texels)).reshape(6, 5, 4)
    assert np.array_equal(a, nr.ImageData.from_image_rows(rgb).pixels)
    assert np.array_equal(b, nr.ImageData.from_image_rows(gray, opacity=True).pixels)
    assert t0.interp == abi.INTERP_BILINEAR and t0.overflow == abi.OVERFLOW_WRAP


def test_sponza_standin_through_the_file_front_end(host, tmp_path):
    """OBJ + MTL + PNG written by tools/gen_assets.py and parsed by the C++ loader give the same frame as
    the scene built in Python (low detail for speed)."""
    import tools.gen_assets as ga
    from tools import standins
    old = ga.MEDIA
    ga.MEDIA = str(tmp_path / "media")
    try:
        ga.gen_sponza(0.05)
    finally:
        ga.MEDIA = old
    text = open(os.path.join(ROOT, "scenes", "crytek_sponza.scene")).read()
    fs = scenefile.FileScene(_scene(tmp_path, text))
    cam = fs.camera_dict()
    p = nr.make_params((48, 27), 1, 0.0, cam["eye"], fs.inverse_projection(0, 48, 27))
    a, sa = oracle.render(fs.descriptor, p, 4)
    pts, uvs, groups, defs, tex = standins.sponza_geometry(0.05)
    assert fs.descriptor.desc.num_nodes == len(groups)
    sc = _python_sponza(0.05)
    b, sb = oracle.render(sc.descriptor, p, 4)
    assert np.array_equal(a, b) and sa.total_rays() == sb.total_rays()


def _python_sponza(detail):
    from tools import standins
    return standins.sponza_scene(detail)[0]


def test_loader3d_cli_usage(built):
    exe = os.path.join(ROOT, "nrays_amd", "lib", "loader3d")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "Usage" in r.stderr


REFERENCE_SCENES = "/root/reference/scenes"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SCENES), reason="the reference checkout is only mounted in the build container")
def test_every_shipped_reference_scene_file_parses(host, tmp_path):
    """Grammar coverage: all .scene / .mtl files the reference ships go through the loader unchanged (read from the
    read-only reference mount at test time, never copied into the repo).  Their OBJ / image assets are not
    distributed, so every file the loader asks for is replaced by a one-triangle OBJ or a 2x2 PNG until the
    whole scene loads; a syntax error or an unknown directive would surface as a different message."""
    import glob
    import re
    import shutil
    work = tmp_path / "scenes"
    work.mkdir()
    for f in glob.glob(os.path.join(REFERENCE_SCENES, "*")):
        if os.path.isfile(f):
            shutil.copy(f, work / os.path.basename(f))
    tiny_png = np.zeros((2, 2, 3), dtype=np.float32)
    scenes = sorted(glob.glob(str(work / "*.scene")))
    assert len(scenes) >= 20
    for sf in scenes:
        for _ in range(200):
            try:
                fs = scenefile.FileScene(sf)
                break
            except RuntimeError as e:
                m = re.match(r"(Unable to find the file|Image not found): (.*)$", str(e))
                assert m, "%s: %s" % (os.path.basename(sf), e)
                path = m.group(2)
                assert path.startswith(str(tmp_path)), path
                os.makedirs(os.path.dirname(path), exist_ok=True)
                if m.group(1).startswith("Image"):
                    scenefile.write_png(path, tiny_png)
                else:
                    with open(path, "w") as o:
                        o.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n")
        else:
            raise AssertionError("asset loop did not converge for " + sf)
        assert len(fs.cameras) >= 1 and fs.descriptor.pointer() is not None, sf
        fs.close()


def _png_bytes(ihdr_payload, idat=b"\x00\x00"):
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr_payload) + chunk(b"IDAT", zlib.compress(idat)) + chunk(b"IEND", b"")


def test_malformed_png_headers_are_errors_not_crashes(host, tmp_path):
    """a short IHDR, bit depth 0 (division by zero in the stride) or an illegal (colour type, depth) pair
    must be reported, as must the loader's `Image not found`-style failures — never a crash."""
    import struct
    cases = {"short_ihdr.png": struct.pack(">II", 1, 1) + b"\x08\x02",                      # 10 bytes instead of 13
             "depth0.png": struct.pack(">IIBBBBB", 1, 1, 0, 2, 0, 0, 0),
             "depth3.png": struct.pack(">IIBBBBB", 1, 1, 3, 0, 0, 0, 0),
             "rgb_depth4.png": struct.pack(">IIBBBBB", 1, 1, 4, 2, 0, 0, 0),
             "palette16.png": struct.pack(">IIBBBBB", 1, 1, 16, 3, 0, 0, 0)}
    for name, ihdr in cases.items():
        p = tmp_path / name
        p.write_bytes(_png_bytes(ihdr))
        with pytest.raises(RuntimeError, match="png"):
            scenefile.read_png(str(p))
    ok = tmp_path / "ok.png"
    ok.write_bytes(_png_bytes(struct.pack(">IIBBBBB", 1, 1, 8, 0, 0, 0, 0), b"\x00\x7f"))
    assert scenefile.read_png(str(ok)).tolist() == [[[0x7f]]]


def test_saturating_casts_of_the_reference(host, tmp_path):
    """`nsample as usize` (loader3d.rs:456) and the u8 quantisation of image.rs:66-76 saturate in Rust: a negative
    nsample gives racsample 0, a NaN / negative channel gives 0, anything above 1 gives 255."""
    fs = scenefile.FileScene(_scene(tmp_path, "light\n pos 0 1 0\n color 1 1 1\n radius 0.5\n nsample -7\n"))
    assert fs.descriptor.desc.lights[0].racsample == 0
    img = np.array([[[np.nan, -1.0, 0.5], [2.0, 1.0, 0.0039]]], dtype=np.float32)
    out = tmp_path / "q.png"
    scenefile.write_png(str(out), img)
    assert scenefile.read_png(str(out)).tolist() == [[[0, 0, 127], [255, 255, 0]]]
