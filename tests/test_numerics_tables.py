"""Exhaustive checks of the small exact shortcuts the device code takes instead of an IEEE division."""
import numpy as np


def test_byte_over_255_through_an_f64_product_is_the_correctly_rounded_f32_quotient():
    """trace_device.h: tex_at — `u8 as f32 / 255.0` (texture2d.rs:111-162) is evaluated as (float)((double)x * (1.0 / 255.0))."""
    x = np.arange(256, dtype=np.float32)
    ref = x / np.float32(255.0)
    got = (x.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    assert np.array_equal(ref, got)
    # the tempting f32 shortcut is NOT exact: this is why the product is taken in f64
    assert int((x * (np.float32(1.0) / np.float32(255.0)) != ref).sum()) > 0


def test_pixel_over_resolution_by_reciprocal_and_one_fma_correction_is_the_ieee_quotient(tmp_path):
    """trace_device.h: generate_primary<PLAIN> — `ox / width` (scene.rs:81-82) is evaluated as q0 = a * RN(1 / b), q = fma(fma(-q0, b, a), RN(1 / b), q0).
    tools/probe/div_markstein.c compares it with a / b for EVERY pixel index of every resolution up to 16384 (the host launches no PLAIN kernel beyond)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "div_markstein")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(root, "tools", "probe", "div_markstein.c"), "-lm"])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    assert " 0 mismatches" in out.stdout
