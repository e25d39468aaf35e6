#!/usr/bin/env python
"""Instruction-fetch side of the primary kernel (GPU box): instruction / scalar-data cache requests and misses, instruction
fetches and branches per launch, from two rocprofv3 --pmc passes over tools/kbench.py --child <scene>.

  python tools/pmc_icache.py balls [out.json [width height]]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import pmc_collect  # noqa: E402

PASSES = [
    ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "SQ_INSTS_BRANCH", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
    ["SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQ_INSTS_SMEM", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "SQ_WAVES"],
]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "balls"
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
    res = pmc_collect.collect(scene, PASSES + [["SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU"]], width=w, height=h)
    res["command"] = "python tools/pmc_icache.py " + " ".join(sys.argv[1:])
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 2:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
