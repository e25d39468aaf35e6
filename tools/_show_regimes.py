import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d["scene"], "steady", d["steady_ms"], "sync", d["steady_sync_wall_ms"], "| cold wall", d["cold_wall_ms"], "gpu", d["cold_gpu_ms"], "call", [c["call_ms"] for c in d["cold"]], "| moving", d["moving_ms"], "rest-along", d.get("steady_along_the_path_ms"), "ident", d["cold_identical"], d["moving_last_frame_identical_to_a_settled_render"], d["after_a_jump_identical"])
