#!/usr/bin/env python
"""Post-processes two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes) into profiles/pmc_k_primary.json:
HBM bytes per k_primary launch.  Units/corrections per the guide: the counters are in KiB
(bytes = value * 1024) and on gfx950 FETCH_SIZE reports half of the bytes of wide reads, so the
read side is doubled; WRITE_SIZE is uncalibrated (reported as is).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_k_primary.json
"""
import csv
import glob
import json
import sys


def mean_counter(d, name, kernel_substr):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name and kernel_substr in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


if __name__ == "__main__":
    fetch_dir, write_dir, out = sys.argv[1:4]
    kern = sys.argv[4] if len(sys.argv) > 4 else "k_primary<false"
    f, nf = mean_counter(fetch_dir, "FETCH_SIZE", kern)
    w, nw = mean_counter(write_dir, "WRITE_SIZE", kern)
    res = {"kernel": kern, "launches": [nf, nw], "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
           "note": "bytes = KiB * 1024; read side doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B); WRITE_SIZE uncalibrated; "
                   "Infinity-Cache hits appear to be counted, so this is fabric traffic, an upper bound on DRAM traffic"}
    if f is not None and w is not None:
        res["hbm_read_bytes_per_launch"] = f * 1024 * 2
        res["hbm_write_bytes_per_launch"] = w * 1024
        res["hbm_bytes_per_launch"] = f * 1024 * 2 + w * 1024
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
