#!/usr/bin/env python
"""rocprofv3 hardware counters of the primary kernel, per launch.

Runs separate `rocprofv3 --pmc ... --kernel-trace` passes (counters only; never combined with API / sys traces)
over `tools/kbench.py --child <scene>` and averages every counter over the launches of the primary kernel.
FETCH_SIZE and WRITE_SIZE are collected in separate passes and corrected exactly as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: the counters are in KiB (bytes = value x 1024), and on gfx950
FETCH_SIZE reports half of the bytes of wide reads, so the read side is doubled; WRITE_SIZE is taken as is.

  python tools/pmc_collect.py balls profiles/r02_pmc_balls.json            # all passes
  from tools import pmc_collect; pmc_collect.collect("balls", pmc_collect.TRAFFIC_PASSES)   # bench.py's live leg
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Two passes give the HBM traffic plus the issue-side picture (SQ: 8 slots per pass, TCC: 4; FETCH_SIZE costs 3 TCC
# slots and WRITE_SIZE 2, so they cannot share a pass).
TRAFFIC_PASSES = [
    ["FETCH_SIZE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
     "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_WAIT_ANY"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS",
     "SQ_WAIT_INST_ANY", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_FLAT"],
]
MORE_PASSES = [
    ["SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
     "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
]


def rocprofv3_path():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def collect(scene, passes=None, steps=12, kernel_filter="k_primary<false", width=1920, height=1080, timeout=300, keep_dir=None):
    """Returns {counter: mean per launch of the primary kernel, ..., 'launches': n, 'errors': [...]}."""
    rp = rocprofv3_path()
    res = {"scene": scene, "kernel_filter": kernel_filter, "resolution": [width, height], "errors": []}
    if rp is None:
        res["errors"].append("rocprofv3 not found")
        return res
    env = dict(os.environ, TMPDIR="/tmp")
    for k, ctrs in enumerate(passes or (TRAFFIC_PASSES + MORE_PASSES)):
        d = os.path.join(keep_dir, "pass%d" % k) if keep_dir else tempfile.mkdtemp(prefix="nrays_pmc_", dir="/tmp")
        cmd = [rp, "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--child", scene, "--steps", str(steps),
               "--width", str(width), "--height", str(height)]
        try:
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
        except subprocess.TimeoutExpired:
            res["errors"].append("pass %d timed out" % k)
            continue
        if r.returncode != 0:
            res["errors"].append("pass %d rc %d: %s" % (k, r.returncode, r.stdout[-300:]))
            continue
        acc = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel_filter in row.get("Kernel_Name", ""):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for c in ctrs:
            if c in acc:
                res[c] = sum(acc[c]) / len(acc[c])
                res["launches"] = len(acc[c])
            else:
                res[c] = None
        if not keep_dir:
            shutil.rmtree(d, ignore_errors=True)
    if res.get("FETCH_SIZE") is not None:
        res["hbm_read_bytes_per_launch"] = res["FETCH_SIZE"] * 1024 * 2
    if res.get("WRITE_SIZE") is not None:
        res["hbm_write_bytes_per_launch"] = res["WRITE_SIZE"] * 1024
    if "hbm_read_bytes_per_launch" in res and "hbm_write_bytes_per_launch" in res:
        res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
    return res


if __name__ == "__main__":
    scene, out = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "k_primary<false"
    r = collect(scene, kernel_filter=kern)
    r["command"] = "python tools/pmc_collect.py %s %s" % (scene, out)
    try:
        r["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        pass
    json.dump(r, open(out, "w"), indent=1)
    print(json.dumps(r))
