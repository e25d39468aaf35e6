#!/usr/bin/env python
"""Runs several separate rocprofv3 --pmc passes (counters only, with --kernel-trace; never combined with
API / sys traces) over `tools/kbench.py --child <scene>` and prints the per-launch mean of every
counter for the primary kernel as one JSON object.

  python tools/pmc_collect.py balls gpurun_out/pmc_balls.json
"""
import csv, glob, json, os, subprocess, sys

PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"],
    ["SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"],
    ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"],
    ["SQ_WAVES", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_FLAT"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
]

def main():
    scene, out = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "k_primary<false"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TMPDIR="/tmp")
    res = {"scene": scene, "kernel_filter": kern}
    for k, ctrs in enumerate(PASSES):
        d = os.path.join(root, "gpurun_out", "pmc_%s_%d" % (scene, k))
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(root, "tools", "kbench.py"), "--child", scene, "--steps", "5"]
        r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        if r.returncode != 0:
            res["pass%d_error" % k] = r.stdout[-400:]
            continue
        acc = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row.get("Kernel_Name", ""):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for c in ctrs:
            if c in acc:
                res[c] = sum(acc[c]) / len(acc[c]); res.setdefault("launches", len(acc[c]))
            else:
                res[c] = None
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))

if __name__ == "__main__":
    main()
