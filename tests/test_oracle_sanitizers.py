"""SURVEY 5 (race detection / sanitizers): the CPU restatement (oracle/nrays_oracle.c) compiled with -fsanitize=address,undefined renders the
frames the plain build renders — threaded render with the reference's static partition (scene.rs:49-66), meshes with ties, textures, area
lights with AA jitter, the cast / shadow probes — without a report.  The sanitizer build runs in a subprocess with libasan preloaded."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import oracle
from tools import scenes_util as su
frames = []
for make, kw in ((lambda: su.primitives_scene(light_radius=0.2, nsample=2), dict(spp=2, window=1.0, seed=3)), (su.mesh_scene, {}), (su.balls_scene, {})):
    sc, cam = make()
    p, _ = su.camera_params(cam, 56, 40, **kw)
    img, st = oracle.render(sc.descriptor, p, 3)
    frames.append(img)
np.savez(sys.argv[1], *frames)
"""


def _run(lib, out, preload=None):
    env = dict(os.environ)
    if lib:
        env["NRAYS_ORACLE_LIB"] = lib
    if preload:
        env["LD_PRELOAD"] = preload
        env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"  # (CPython leaks by design; everything else is fatal)
        env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    return subprocess.run([sys.executable, "-c", CHILD % ROOT, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_oracle_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    plain, san = str(tmp_path / "plain.npz"), str(tmp_path / "san.npz")
    r = _run(None, plain)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run(os.path.join(ROOT, "oracle", "_build", "libnrays_oracle_asan.so"), san, preload=libasan)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    a, b = np.load(plain), np.load(san)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k  # same source, same -ffp-contract=off: the sanitizer build must not change a pixel
