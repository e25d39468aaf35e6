#!/usr/bin/env python
"""Writes the procedural stand-ins for the assets the reference downloads as real files
under scenes/media/, so that the whole loader3d front-end (OBJ, MTL, PNG decode) is exercised:

  python tools/gen_assets.py [globe] [sponza] [hairball] [--hairball-strands N] [--detail D]

  scenes/media/globe.png
  scenes/media/crytek-sponza/{sponza.obj, sponza.mtl, textures/*.png}
  scenes/media/hairball/hairball.obj

scenes/media/ is git-ignored (as upstream); files are deterministic (fixed seeds).  If the real
assets are dropped in the same places they are used instead and nothing is overwritten.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MEDIA = os.path.join(ROOT, "scenes", "media")


def _fmt(a):
    return np.char.mod("%.9g", a)  # 9 significant digits round-trip an f32 exactly


def write_obj(path, pts, uvs, groups, mtllib=None):
    with open(path, "w") as f:
        f.write("# procedural stand-in written by tools/gen_assets.py\n")
        if mtllib:
            f.write("mtllib %s\n" % mtllib)
        v = _fmt(pts)
        f.write("\n".join("v " + " ".join(r) for r in v) + "\n")
        t = _fmt(uvs)
        f.write("\n".join("vt " + " ".join(r) for r in t) + "\n")
        for gi, (mat, idx) in enumerate(groups):
            f.write("g group_%04d\n" % gi)
            if mat:
                f.write("usemtl %s\n" % mat)
            i1 = (idx + 1).astype(str)
            f.write("\n".join("f " + " ".join(a + "/" + a for a in r) for r in i1) + "\n")


def save_png(path, rgba_bottom_first, kind):
    """Writes the texture in the format of the path's extension (.png, .tga — run-length coded for colour maps, raw for opacity
    maps — .bmp, .jpg): Texture2d::from_png decodes all of them through stb_image (texture2d.rs:95)."""
    from PIL import Image
    top_first = rgba_bottom_first[::-1]  # Texture2d::from_png flips Y on load (texture2d.rs:99-107)
    kw = {"compression": "tga_rle"} if path.endswith(".tga") and kind != "alpha" else {}
    if kind == "alpha":
        Image.fromarray(np.ascontiguousarray(top_first[..., 3])).save(path, **kw)  # depth 1 opacity map -> (1,1,1,g)
    else:
        Image.fromarray(np.ascontiguousarray(top_first[..., :3])).save(path, **kw)  # depth 3 -> (r,g,b,1)


def gen_globe():
    from tools import scenes_util as su
    p = os.path.join(MEDIA, "globe.png")
    if os.path.exists(p):
        return p
    os.makedirs(MEDIA, exist_ok=True)
    save_png(p, su.globe_texture().data.pixels, "rgb")
    return p


def gen_sponza(detail, ext="png"):
    from tools import standins
    d = os.path.join(MEDIA, "crytek-sponza")
    obj = os.path.join(d, "sponza.obj")
    if os.path.exists(obj):
        return obj
    os.makedirs(os.path.join(d, "textures"), exist_ok=True)
    pts, uvs, groups, defs, tex = standins.sponza_geometry(detail)
    for name, t in tex.items():
        save_png(os.path.join(d, "textures", name + "." + ext), t.data.pixels, "alpha" if name in ("lace", "leaves") else "rgb")
    with open(os.path.join(d, "sponza.mtl"), "w") as f:
        for name, (ka, kd, ks, t, a, ns, alpha) in defs.items():
            f.write("newmtl %s\nNs %.9g\nd %.9g\nKa %.9g %.9g %.9g\nKd %.9g %.9g %.9g\nKs %.9g %.9g %.9g\n" % ((name, ns, alpha) + tuple(ka) + tuple(kd) + tuple(ks)))
            if t:
                f.write("map_Kd textures/%s.%s\n" % (t, ext))
            if a:
                f.write("map_d textures/%s.%s\n" % (a, ext))
            f.write("\n")
    write_obj(obj, pts, uvs, groups, "sponza.mtl")
    return obj


def gen_hairball(strands):
    from tools import standins
    d = os.path.join(MEDIA, "hairball")
    obj = os.path.join(d, "hairball.obj")
    if os.path.exists(obj):
        return obj
    os.makedirs(d, exist_ok=True)
    pts, idx, uvs = standins.hairball_geometry(strands)
    write_obj(obj, pts, uvs, [(None, idx)])
    return obj


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["globe", "sponza"])
    ap.add_argument("--hairball-strands", type=int, default=3000)
    ap.add_argument("--detail", type=float, default=1.0)
    a = ap.parse_args()
    for w in a.what:
        p = gen_globe() if w == "globe" else gen_sponza(a.detail) if w == "sponza" else gen_hairball(a.hairball_strands)
        print(w, "->", p, "%.1f MB" % (os.path.getsize(p) / 1e6))
