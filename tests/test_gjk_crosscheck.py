"""An independent GJK ray cast over the shapes' support maps (tools/gjk_raycast_report.py: the published algorithm ncollide3d runs
for cylinder / cone / capsule, written from the paper) against the closed forms of the oracle — a sample of what
profiles/r03_gjk_vs_closed_form.json reports in full: away from rims the two agree to GJK's own tolerance."""
import os

import numpy as np
import pytest

from tools import gjk_raycast_report as gjk

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_independent.npz")


@pytest.mark.parametrize("kind", [gjk.CYLINDER, gjk.CAPSULE, gjk.CONE, gjk.CUBOID])
def test_gjk_ray_cast_agrees_with_the_closed_forms_off_the_rims(kind):
    cases = np.load(FIXTURE)["cases"]
    cases = cases[(cases[:, 0] == kind) & (cases[:, 19] == 0)][:36]
    n = 0
    for c in cases:
        prm, t, w, solid, o, d, hit, toi = c[1:4], c[4:7], c[7:10], bool(c[10]), c[11:14], c[14:17], bool(c[17]), c[18]
        R = gjk.rotation(w)
        g_hit, g_t, g_n, _ = gjk.gjk_raycast(int(kind), prm, R.T @ (o - t), R.T @ d)
        assert g_hit == hit
        if not hit:
            continue
        n += 1
        c_hit, c_t, c_n = gjk.closed_form(int(kind), prm, t, w, solid, o, d)
        assert c_hit and abs(g_t - toi) <= 1e-6 * max(1.0, toi) and abs(c_t - toi) <= 1e-11 * max(1.0, toi)
        assert gjk.angle_deg(R @ g_n, c_n) <= 0.1
    assert n >= 15
