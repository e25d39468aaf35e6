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
from tests import scenes_util as su

lib = abi.load_hip_lib()


def random_scene(seed):
    rng = np.random.default_rng(seed)
    sc, cam = su.random_shapes_scene(seed, n=int(rng.integers(4, 40)), with_mesh=bool(rng.random() < 0.7))
    d = sc  # rebuild with random coefficients
    nodes = []
    for nd in d._nodes:
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


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    worst, bad = 0.0, 0
    for seed in range(first, first + count):
        try:
            sc, cam, rng = random_scene(seed)
        except AttributeError as e:
            print("scene construction needs attribute:", e); return 2
        w, h = int(rng.integers(40, 200)), int(rng.integers(30, 140))
        spp = int(rng.choice([1, 1, 2]))
        kw = dict(spp=spp, window=float(rng.choice([0.0, 1.0])) if spp > 1 else 0.0, seed=int(seed),
                  max_depth=int(rng.choice([3, 5, 8])))  # bounds the 2^depth ray trees of double-branching nodes
        p, _ = su.camera_params(cam, w, h, **kw)
        ref, ost = oracle.render(sc.descriptor, p, 32)
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
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
    print("worst error %.3g over %d cases, %d mismatches" % (worst, 2 * count, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
