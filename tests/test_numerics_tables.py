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
