"""bench.py's roofline bookkeeping on synthetic counters (no GPU): one kernel, one set of units — the contract bytes follow SURVEY 8(d) from the counters
of the render that counts what the TIMED kernel does, the unique-fetch figure counts a node record as fetched (128 B per fetch), the limiter's fractions are fractions of the
cost-recording launch's own duration at the measured clock, and every utilisation is <= 1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stats(**kw):
    from nrays_amd import abi
    s = abi.NraysStats()
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_roofline_block_units_and_fractions():
    import bench
    from nrays_amd import abi
    W, H = 1920, 1080
    pk = _stats(rays_primary=W * H, rays_shadow=1000, rays_shadow_elided=100, rays_primary_traced=400000, node_tests=4_000_000, node_fetches=300_000, tri_tests=50_000,
                prim_tests=10_000, hit_records=20_000, tex_samples=5_000)
    pk_ref = _stats(rays_primary=W * H, rays_shadow=1000, rays_primary_traced=400000, node_tests=4_400_000, node_fetches=330_000, tri_tests=55_000, prim_tests=10_000,
                    hit_records=22_000, tex_samples=6_000)
    tst = _stats(kernel_ms_primary=0.050, kernel_ms_total=0.050, frames_timed=50)
    tc = abi.NraysTileCosts()
    tc.tiles, tc.sum_cycles, tc.max_cycles, tc.resident_waves, tc.shader_clock_hz, tc.kernel_ms = 4000, 2048 * 30_000, 120_000, 2048, 2.3e9, 0.060
    pmc = {"hbm_bytes_per_launch": 28.7e6, "TCC_HIT_sum": 2.7e5, "TCC_MISS_sum": 2.5e5, "SQ_ACTIVE_INST_VALU": 7.7e6, "SQ_WAIT_ANY": 1.8e7, "SQ_WAVE_CYCLES": 3.4e7}
    r = bench.roofline_block(pk, tst, W, H, 2_000_000, pmc, "synthetic", tc, step_ms=0.048, pk_ref=pk_ref)
    fb = 12 * W * H
    other = 36 * 50_000 + 64 * 10_000 + 64 * 20_000 + 16 * 5_000 + fb
    assert r["algorithmic_bytes_per_launch"] == 32 * 4_000_000 + other            # SURVEY 8(d) on the timed kernel's own counts
    assert r["unique_fetch_bytes_per_launch"] == 128 * 300_000 + other            # a node record as fetched
    assert r["kernel_ms"] == 0.048 and r["kernel_ms_events"] == 0.050             # single launch: the step bounds the kernel
    assert abs(r["contract_frac"] - r["algorithmic_bytes_per_launch"] / 0.048e-3 / 1e9 / 8000.0) < 1e-4
    assert r["units_per_launch"]["rays_shadow_counted_not_traced"] == 100 and r["reference_units_per_launch"]["node_tests"] == 4_400_000
    lim = r["limiter"]
    assert abs(lim["longest_tile_frac"] - 120_000 / 2.3e9 / 0.060e-3) < 1e-3 and abs(lim["wave_throughput_frac"] - 30_000 / 2.3e9 / 0.060e-3) < 1e-3
    assert lim["longest_tile_frac"] <= 1.0 and lim["name"] == "latency/longest-tile"
    for c in r["ceilings"].values():
        assert 0.0 <= c["frac"] <= 1.0
    assert r["frac"] == max(c["frac"] for c in r["ceilings"].values()) and r["bound"] in r["ceilings"]
    assert r["shader_clock_ghz"] == 2.3
    # without counters and without a measured clock the block says so instead of inventing a utilisation
    tc.shader_clock_hz = 0.0
    r2 = bench.roofline_block(pk, tst, W, H, 2_000_000, None, None, tc, step_ms=0.048, pk_ref=pk_ref)
    assert r2["frac"] is None and "unmeasured" in r2["bound"] and "limiter" not in r2 and r2["contract_frac"] == r["contract_frac"]
