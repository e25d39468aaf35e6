"""Golden frames (tests/golden/*.npz, produced by tests/golden/make_golden.py with the oracle):
CPU: the oracle still reproduces them; GPU: the HIP path matches them within 1e-4 per channel."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.golden.make_golden import CASES, render_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    _, _, img, rays = render_case(name)
    assert np.array_equal(rays, g["rays"])
    assert np.abs(img - g["image"]).max() <= 1e-6  # libm-level slack only (atan2 / asin / powf)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_golden(gpu, name):
    import torch
    import nrays_amd as nr
    from nrays_amd import abi
    g = np.load(os.path.join(GOLD, name + ".npz"))
    build, w, h, kw = CASES[name]
    from tools import scenes_util as su
    sc, cam = build()
    p, _ = su.camera_params(cam, w, h, **dict(kw))
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    abi.check(abi.load_hip_lib().nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    assert np.abs(out.cpu().numpy() - g["image"]).max() <= 1e-4
    assert [st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow] == g["rays"].tolist()
