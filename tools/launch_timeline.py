#!/usr/bin/env python
"""Kernel durations against the launch period (GPU box): runs tools/kbench.py --child <scene> under
rocprofv3 --kernel-trace and prints, per kernel name, the launches, the mean duration and the mean distance between the
starts of consecutive launches — what a frame costs beyond the time its kernel is executing.

  python tools/launch_timeline.py balls [width height [steps]]
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "balls"
    w, h = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("1920", "1080")
    steps = sys.argv[4] if len(sys.argv) > 4 else "50"
    d = tempfile.mkdtemp(prefix="nrays_tl_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.join(ROOT, "tools", "kbench.py"),
           "--child", scene, "--steps", steps, "--width", w, "--height", h]
    r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print(r.stdout[-2000:]); raise SystemExit(1)
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    by = {}
    for x in rows:
        by.setdefault(x["Kernel_Name"][:60], []).append((int(x["Start_Timestamp"]), int(x["End_Timestamp"])))
    for k, v in by.items():
        dur = [e - s for s, e in v]
        per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
        per = sorted(per)[: max(1, len(per) * 3 // 4)]  # drop the gaps between the timing loops
        gaps = sorted(v[i + 1][0] - v[i][1] for i in range(len(v) - 1))[: max(1, (len(v) - 1) * 3 // 4)]
        print(json.dumps({"kernel": k, "launches": len(v), "duration_us": round(sum(dur) / len(dur) / 1e3, 2),
                          "period_us": round(sum(per) / max(len(per), 1) / 1e3, 2), "gap_us": round(sum(gaps) / max(len(gaps), 1) / 1e3, 2)}))


if __name__ == "__main__":
    main()
