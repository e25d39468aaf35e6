#!/usr/bin/env python
"""Generates the golden frames under tests/golden/ with the CPU oracle (oracle/nrays_oracle.c).

The reference ships no golden vectors and cannot run here, so these frames pin the
ORACLE against regressions and give the GPU tests committed expectations; they do not pin the
oracle to the Rust reference (parity unpinned).  Re-run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tools import scenes_util as su  # noqa: E402

CASES = {
    # name: (scene builder, width, height, camera_params kwargs)
    "balls_64x36": (lambda: su.balls_scene(tex_size=(128, 64)), 64, 36, {}),
    "balls_shipped_refl_48x27": (lambda: su.balls_scene(refl=(0.2, 0.2), tex_size=(128, 64)), 48, 27, {}),
    "primitives_point_64x48": (lambda: su.primitives_scene(0.0, 1), 64, 48, {}),
    "primitives_area_aa_32x24": (lambda: su.primitives_scene(0.1, 10), 32, 24, dict(spp=2, window=1.0, seed=7)),
    "mesh_alpha_rot_64x48": (lambda: su.mesh_scene(True, True), 64, 48, {}),
    "random_shapes_48x36": (lambda: su.random_shapes_scene(5, n=24), 48, 36, {}),
}


def render_case(name):
    build, w, h, kw = CASES[name]
    sc, cam = build()
    p, _ = su.camera_params(cam, w, h, **dict(kw))
    img, st = oracle.render(sc.descriptor, p, 4)
    rays = np.array([st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow], dtype=np.int64)
    return sc, p, img, rays


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        _, _, img, rays = render_case(name)
        np.savez_compressed(os.path.join(out, name + ".npz"), image=img, rays=rays)
        print(name, img.shape, rays.tolist(), float(img.mean()))
