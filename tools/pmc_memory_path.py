#!/usr/bin/env python
"""Counters of the vector-memory path (TA / TCP / TD / address translation) for the primary kernel of a scene.
  NRAYS_PREPASS=0 python tools/pmc_memory_path.py sponza gpurun_out/out.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import pmc_collect
PASSES = [
    ["TA_BUSY_avr", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INST_CYCLES_VMEM"],
    ["TCP_GATE_EN1_sum", "TCP_TOTAL_ACCESSES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
    ["TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCP_LATENCY_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum", "TD_TD_BUSY_sum"],
    ["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TAGRAM0_REQ_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_TD_TCP_STALL_CYCLES_sum"],
]
if __name__ == "__main__":
    r = pmc_collect.collect(sys.argv[1], PASSES, steps=4)
    json.dump(r, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(r))
