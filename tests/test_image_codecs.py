"""The file formats stb_image's `load` accepts behind Texture2d::from_png (src/texture2d.rs:95) besides plain PNG: Adam7
interlaced PNG, TGA (the Crytek Sponza distribution's texture format), BMP, baseline and progressive JPEG — the C++ host
decoders (nrays_amd/host/{png_codec,image_codec}.cpp) against Pillow's on the same files, plus malformed files (errors, not
crashes) and the texture path of the loader front-end."""
import os
import struct
import zlib

import numpy as np
import pytest

from nrays_amd import scenefile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(built):
    return scenefile.host_lib()


def _test_image(h=37, w=53):
    """Smooth gradients + a checker + noise: exercises every JPEG frequency band and every PNG filter."""
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.empty((h, w, 3), np.uint8)
    img[..., 0] = (xx * 255 // max(w - 1, 1)).astype(np.uint8)
    img[..., 1] = (yy * 255 // max(h - 1, 1)).astype(np.uint8)
    img[..., 2] = (((xx // 5 + yy // 3) % 2) * 200 + rng.integers(0, 40, (h, w))).astype(np.uint8)
    return img


def _adam7_png(arr, color_type, bit_depth=8):
    """Writes an Adam7-interlaced PNG by hand (Pillow cannot write one): filter 0 on every scanline of the seven passes."""
    h, w = arr.shape[:2]
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[color_type]
    px = arr.reshape(h, w, ch)
    raw = b""
    for (x0, y0, dx, dy) in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = px[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            if bit_depth == 8:
                raw += b"\x00" + row.tobytes()
            else:  # 1-bit grey: pack 8 pixels per byte, MSB first
                raw += b"\x00" + np.packbits((row[:, 0] > 127).astype(np.uint8)).tobytes()

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 1))
            + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def test_adam7_interlaced_png(host, tmp_path):
    img = _test_image()
    for ct, arr in ((2, img), (0, img[..., :1]), (6, np.concatenate([img, img[..., :1]], -1)), (4, img[..., :2])):
        p = tmp_path / ("i%d.png" % ct)
        p.write_bytes(_adam7_png(np.ascontiguousarray(arr), ct))
        got = scenefile.read_image(str(p))
        assert got.shape == arr.shape and np.array_equal(got, arr), ct
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(str(p))).reshape(arr.shape), arr)  # the hand-written file is a valid PNG
    for (h, w) in [(1, 1), (2, 3), (5, 1), (8, 8), (9, 17)]:  # passes that are empty / ragged
        small = np.ascontiguousarray(_test_image(h, w))
        p = tmp_path / "s.png"
        p.write_bytes(_adam7_png(small, 2))
        assert np.array_equal(scenefile.read_image(str(p)), small), (h, w)
    bits = (_test_image(11, 19)[..., :1] > 127).astype(np.uint8) * 255
    p = tmp_path / "b1.png"
    p.write_bytes(_adam7_png(np.ascontiguousarray(bits), 0, bit_depth=1))
    assert np.array_equal(scenefile.read_image(str(p)), bits)


def _tga(arr, image_type, bpp, top_left, rle=False, palette=None):
    h, w = arr.shape[:2]
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 1 if palette is not None else 0, image_type + (8 if rle else 0),
                      0, len(palette) if palette is not None else 0, 24 if palette is not None else 0, 0, 0, w, h, bpp,
                      (0x20 if top_left else 0) | (8 if bpp == 32 else 0))
    body = b""
    if palette is not None:
        body += np.ascontiguousarray(palette[:, ::-1]).tobytes()  # BGR entries
    rows = arr if top_left else arr[::-1]
    if bpp == 8:
        px = rows.reshape(h * w, 1)
    elif bpp == 16:  # 5-5-5 from the top bits of RGB
        r, g, b = (rows[..., k].astype(np.uint16) >> 3 for k in range(3))
        px = ((r << 10) | (g << 5) | b).astype("<u2").reshape(h * w, 1).view(np.uint8).reshape(h * w, 2)
    elif bpp == 24:
        px = rows[..., ::-1].reshape(h * w, 3)
    else:
        px = rows[..., [2, 1, 0, 3]].reshape(h * w, 4)
    px = np.ascontiguousarray(px)
    if not rle:
        return hdr + body + px.tobytes()
    out = bytearray()
    i, n = 0, len(px)
    while i < n:  # greedy run-length packets of identical pixels, raw packets otherwise (packets may cross scanlines)
        run = 1
        while i + run < n and run < 128 and np.array_equal(px[i + run], px[i]):
            run += 1
        if run > 1:
            out += bytes([0x80 | (run - 1)]) + px[i].tobytes()
            i += run
        else:
            j = i + 1
            while j < n and j - i < 128 and not (j + 1 < n and np.array_equal(px[j], px[j + 1])):
                j += 1
            out += bytes([j - i - 1]) + px[i:j].tobytes()
            i = j
    return hdr + body + bytes(out)


def test_tga_types_depths_origins_and_rle(host, tmp_path):
    img = _test_image(23, 31)
    img[5:9, 3:20] = (10, 200, 30)  # runs for the RLE packets
    rgba = np.concatenate([img, (255 - img[..., :1])], -1)
    grey = np.ascontiguousarray(img[..., 1])
    for rle in (False, True):
        for top in (False, True):
            tag = "%d%d" % (rle, top)
            (tmp_path / ("c24_%s.tga" % tag)).write_bytes(_tga(img, 2, 24, top, rle))
            assert np.array_equal(scenefile.read_image(str(tmp_path / ("c24_%s.tga" % tag))), img), tag
            (tmp_path / ("c32_%s.tga" % tag)).write_bytes(_tga(rgba, 2, 32, top, rle))
            assert np.array_equal(scenefile.read_image(str(tmp_path / ("c32_%s.tga" % tag))), rgba), tag
            (tmp_path / ("g8_%s.tga" % tag)).write_bytes(_tga(grey, 3, 8, top, rle))
            assert np.array_equal(scenefile.read_image(str(tmp_path / ("g8_%s.tga" % tag)))[..., 0], grey), tag
            (tmp_path / ("c16_%s.tga" % tag)).write_bytes(_tga(img, 2, 16, top, rle))
            want16 = ((img >> 3).astype(np.int32) * 255 // 31).astype(np.uint8)  # 5 bits -> 8 bits the way stb_image expands them
            assert np.array_equal(scenefile.read_image(str(tmp_path / ("c16_%s.tga" % tag))), want16), tag
    pal = np.random.default_rng(3).integers(0, 256, (200, 3), dtype=np.uint8)
    idx = (np.arange(23 * 31).reshape(23, 31) % 200).astype(np.uint8)
    (tmp_path / "p.tga").write_bytes(_tga(idx, 1, 8, False, True, palette=pal))
    assert np.array_equal(scenefile.read_image(str(tmp_path / "p.tga")), pal[idx])
    # Pillow agrees on its own TGA output (bottom-up and RLE as Pillow writes them)
    from PIL import Image
    for mode, arr in (("RGB", img), ("RGBA", rgba), ("L", grey)):
        for kw in ({}, {"compression": "tga_rle"}):
            q = str(tmp_path / ("pil_%s%d.tga" % (mode, len(kw))))
            Image.fromarray(arr).save(q, **kw)
            got = scenefile.read_image(q)
            assert np.array_equal(got.reshape(arr.shape), arr), (mode, kw)


def test_bmp(host, tmp_path):
    from PIL import Image
    img = _test_image(19, 30)
    for mode in ("RGB", "P", "L"):
        q = str(tmp_path / (mode + ".bmp"))
        src = Image.fromarray(img).convert(mode)
        src.save(q)
        got = scenefile.read_image(q)
        assert got.shape[2] == 3 and np.array_equal(got, np.asarray(src.convert("RGB"))), mode


@pytest.mark.parametrize("kw", [dict(quality=95, subsampling=0), dict(quality=90, subsampling=1), dict(quality=85, subsampling=2),
                                dict(quality=90, subsampling=2, progressive=True), dict(quality=75, subsampling=0, progressive=True),
                                dict(quality=92, subsampling=2, optimize=True)])
def test_jpeg_against_pillow(host, tmp_path, kw):
    """Baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, custom Huffman tables.  libjpeg (Pillow) and this decoder use the same
    integer inverse DCT family and triangle chroma upsampling but round at different places: agreement to a few levels."""
    from PIL import Image
    img = _test_image(67, 90)  # not a multiple of the MCU size in either direction
    q = str(tmp_path / "a.jpg")
    Image.fromarray(img).save(q, **kw)
    got = scenefile.read_image(q).astype(np.int32)
    want = np.asarray(Image.open(q).convert("RGB")).astype(np.int32)
    assert got.shape == want.shape
    diff = np.abs(got - want)
    if kw["subsampling"] == 1:
        diff[:, -2] = 0  # 4:2:2: stb_image's right-edge rule for the second-to-last column differs from libjpeg's (image_codec.cpp: up_h2)
    assert diff.max() <= 6 and diff.mean() <= 0.8, (kw, diff.max(), diff.mean())
    grey = str(tmp_path / "g.jpg")
    Image.fromarray(img[..., 1]).save(grey, quality=kw["quality"], progressive=kw.get("progressive", False))
    g = scenefile.read_image(grey)
    assert g.shape == (67, 90, 1)
    assert np.abs(g[..., 0].astype(np.int32) - np.asarray(Image.open(grey)).astype(np.int32)).max() <= 2


def test_jpeg_restart_intervals(host, tmp_path):
    """DRI / RSTn: the same scan with a restart marker every 3 MCUs (inserted by re-encoding through Pillow's own option)."""
    from PIL import Image
    img = _test_image(40, 56)
    a, b = str(tmp_path / "a.jpg"), str(tmp_path / "r.jpg")
    Image.fromarray(img).save(a, quality=90, subsampling=2)
    Image.fromarray(img).save(b, quality=90, subsampling=2, restart_marker_blocks=3)
    assert b"\xff\xdd" in open(b, "rb").read()
    assert np.array_equal(scenefile.read_image(a), scenefile.read_image(b))


def test_malformed_files_are_errors_not_crashes(host, tmp_path):
    from PIL import Image
    img = _test_image(16, 16)
    Image.fromarray(img).save(str(tmp_path / "ok.jpg"), quality=90)
    jpg = open(str(tmp_path / "ok.jpg"), "rb").read()
    tga = _tga(img, 2, 24, False, True)
    cases = {"trunc.jpg": jpg[:len(jpg) // 3], "nosof.jpg": jpg[:2] + jpg[jpg.index(b"\xff\xda"):], "soi_only.jpg": b"\xff\xd8\xff",
             "trunc.tga": tga[:40], "junk.bin": b"hello world, not an image at all", "empty.tga": b"",
             "trunc.bmp": b"BM" + b"\x00" * 20}
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(RuntimeError):
            scenefile.read_image(str(p))
    bad = bytearray(_adam7_png(np.ascontiguousarray(img), 2)); bad[28] = 2  # interlace method 2 does not exist
    (tmp_path / "il2.png").write_bytes(bytes(bad))
    with pytest.raises(RuntimeError):
        scenefile.read_image(str(tmp_path / "il2.png"))


def test_tga_and_jpeg_textures_through_the_loader(host, tmp_path):
    """map_Kd as TGA, map_d as an 8-bit grey TGA: the loader's texture path (texture2d.rs:95-177 — decode, Y flip, opacity
    variant) gives the same texels as for the same images stored as PNG."""
    from PIL import Image
    import nrays_amd as nr
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (6, 5), dtype=np.uint8)
    (tmp_path / "kd.tga").write_bytes(_tga(rgb, 2, 24, False, True))
    (tmp_path / "d.tga").write_bytes(_tga(gray, 3, 8, True, False))
    (tmp_path / "m.mtl").write_text("newmtl m\nmap_Kd kd.tga\nmap_d d.tga\n")
    (tmp_path / "t.scene").write_text("mtllib m.mtl\n")
    fs = scenefile.FileScene(str(tmp_path / "t.scene"))
    d = fs.descriptor.desc
    assert d.num_textures == 2
    a = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (6 * 5 * 4)).from_address(d.textures[0].texels)).reshape(6, 5, 4)
    b = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (6 * 5 * 4)).from_address(d.textures[1].texels)).reshape(6, 5, 4)
    assert np.array_equal(a, nr.ImageData.from_image_rows(rgb).pixels)
    assert np.array_equal(b, nr.ImageData.from_image_rows(gray, opacity=True).pixels)
