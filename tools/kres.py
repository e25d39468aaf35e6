#!/usr/bin/env python
"""Compiles libnrays_hip.so (optionally with extra -D flags, to another output path) and prints the register /
spill / scratch figures of every kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks, plus the
number of flat_load / global_load / scratch instructions of each k_primary instantiation from the disassembly.

  python tools/kres.py [-o out.so] [-DNAME=VALUE ...] [--isa]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    out = os.path.join(ROOT, "nrays_amd", "lib", "v", "kres.so")  # never the product library: a failed link must not clobber it
    extra, isa = [], False
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "-o":
            out = os.path.abspath(args.pop(0))
        elif a == "--isa":
            isa = True
        else:
            extra.append(a)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # the library's translation units (k_primary's permutations: one unit per group of primary_inst.hip), compiled side by side with the remarks on
    from concurrent.futures import ThreadPoolExecutor
    objdir = out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    units = [("primary_inst.hip", "g%d.o" % k, ["-DNR_PRIMARY_GROUP=%d" % k]) for k in range(g.PRIMARY_GROUPS)] + [(s_, s_ + ".o", []) for s_ in g.HIP_SOURCES]
    def compile_unit(u):
        return subprocess.run(["/opt/rocm/bin/hipcc"] + g.HIP_FLAGS + extra + u[2] + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.path.join(objdir, u[1]),
                               os.path.join(g.CSRC, u[0])], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        rs = list(ex.map(compile_unit, units))
    for r in rs:
        if r.returncode != 0:
            print(r.stdout[-4000:])
            raise SystemExit(r.returncode)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [os.path.join(objdir, u[1]) for u in units] + g.HIP_LINK)
    class R_: pass
    r = R_(); r.stdout = "\n".join(x.stdout for x in rs)
    rows, cur = [], None
    for line in r.stdout.splitlines():
        m = re.search(r"remark: (?:Function Name|Name): (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+) \[", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/workgroup\])?: (\S+)", line) or re.search(r":\s+([A-Za-z][A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/workgroup\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    demangled = subprocess.run(["c++filt"] + [r_["name"] for r_ in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines()
    print("%-46s %5s %5s %6s %6s %8s %4s" % ("kernel", "SGPR", "VGPR", "sSpill", "vSpill", "scratch", "occ"))
    for r_, d in zip(rows, demangled):
        short = re.sub(r"\(.*", "", d).replace("nrays::", "").replace("void ", "")
        print("%-46s %5s %5s %6s %6s %8s %4s" % (short[:46], r_.get("TotalSGPRs", r_.get("SGPRs", "?")), r_.get("VGPRs", "?"), r_.get("SGPRs Spill", "?"),
                                               r_.get("VGPRs Spill", "?"), r_.get("ScratchSize", "?"), r_.get("Occupancy", "?")))
    if isa:
        co = out + ".co"
        subprocess.check_call("/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=%s --output=%s --unbundle 2>/dev/null || "
                              "/opt/rocm/bin/roc-obj-extract -o %s %s" % (out, co, co, out), shell=True)


if __name__ == "__main__":
    main()
