"""Short summary of a bench.py JSON line (tuning runs): python tools/_bench_show.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(n, m):
    r = m["roofline"]; l = r.get("limiter", {})
    print(n, "value", m["value"], "traced", m["value_traced"], "cold", m.get("value_cold"), "moving", m.get("value_moving"), "| ms", m["ms_per_step"], "cold_ms", m["cold_frame_ms"], "gpu", m.get("cold_frame_gpu_ms"),
          "sync", m.get("steady_frame_sync_ms"), "mov", m.get("moving_camera_ms_per_frame"), "rest", m.get("moving_path_last_camera_at_rest_ms"), "build_s", m.get("scene_build_s"))
    print("   frac", r["frac"], r["bound"], "contract", r["contract_frac"], "unique", r["unique_fetch_frac"], "clk", r["shader_clock_ghz"], "limiter", l.get("name"), l.get("longest_tile_frac"), l.get("wave_throughput_frac"),
          "rec_ms", l.get("recording_launch_ms"), "valu", r.get("valu_active_frac"), "dram", r.get("dram_frac"), "kernel_ms", r["kernel_ms"], "cpu", m.get("cpu_baseline", {}).get("value"), m.get("gpu_over_cpu"), m.get("gpu_over_cpu_at_full_host"))
    print("   units", r["units_per_launch"]); print("   ref  ", r.get("reference_units_per_launch"))
show("balls", d)
for k, v in d.get("secondary", {}).items():
    if "roofline" in v: show(k, v)
    else: print(k, json.dumps(v)[:1500])
print(d.get("north_star_sponza"))
