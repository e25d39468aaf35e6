#!/usr/bin/env python
"""Where a kernel's scratch (spill) accesses sit: instructions and scratch loads / stores of one k_primary instantiation by loop depth (LLVM's "Depth=" annotations
in the assembly), and every block of depth >= 5 — the node loop, the triangle loop — with its instruction mix.  CPU only (hipcc -S of every group of
primary_inst.hip, ~30 s; REBUILD=1 refreshes /tmp/nrays_isa.s).

  python tools/isa_scratch.py "k_primaryILb0ELi2ELb1ELi0E" ["k_primaryILb0ELi214ELb1ELi3E" ...]      (mangled-name fragments)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as g

asm = "/tmp/nrays_isa.s"
if not os.path.exists(asm) or os.environ.get("REBUILD"):
    from concurrent.futures import ThreadPoolExecutor
    flags = [f for f in g.HIP_FLAGS if f != "-fPIC"]
    parts = ["/tmp/nrays_isa_g%d.s" % k for k in range(g.PRIMARY_GROUPS)]
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(lambda k: subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DNR_PRIMARY_GROUP=%d" % k, "-S", "--cuda-device-only", "-o", parts[k],
                                                     os.path.join(g.CSRC, "primary_inst.hip")], stderr=subprocess.DEVNULL), range(g.PRIMARY_GROUPS)))
    with open(asm, "w") as f:
        for q in parts: f.write(open(q).read() + "\n")
text = open(asm).read().split("\n")
for frag in sys.argv[1:] or ["k_primaryILb0ELi2ELb1ELi0E"]:
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and frag in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    blocks, cur = [], dict(name="entry", depth=0, ins=0, ld=0, st=0, vmem=0, smem=0, lds=0, f64=0)
    blocks.append(cur)
    for l in text[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = dict(name=m.group(1), depth=0, ins=0, ld=0, st=0, vmem=0, smem=0, lds=0, f64=0); blocks.append(cur)
            d = re.search(r"Depth=(\d+)", l); cur["depth"] = int(d.group(1)) if d else 0
            continue
        s = l.strip()
        if s.startswith(";"):
            d = re.search(r"Depth=(\d+)", s)
            if d: cur["depth"] = max(cur["depth"], int(d.group(1)))
            continue
        if not s or s.startswith("."): continue
        cur["ins"] += 1
        if s.startswith("scratch_load"): cur["ld"] += 1
        elif s.startswith("scratch_store"): cur["st"] += 1
        elif s.startswith(("global_", "flat_", "buffer_")): cur["vmem"] += 1
        elif s.startswith(("s_load", "s_buffer_load")): cur["smem"] += 1
        elif s.startswith("ds_"): cur["lds"] += 1
        if "_f64" in s: cur["f64"] += 1
    print("== %s: %d blocks, %d instructions, scratch loads %d, stores %d" % (frag, len(blocks), sum(b["ins"] for b in blocks), sum(b["ld"] for b in blocks), sum(b["st"] for b in blocks)))
    for d in sorted({b["depth"] for b in blocks}):
        bs = [b for b in blocks if b["depth"] == d]
        print("  loop depth %d: %3d blocks %5d instructions   scratch loads %3d stores %3d" % (d, len(bs), sum(b["ins"] for b in bs), sum(b["ld"] for b in bs), sum(b["st"] for b in bs)))
    print("  blocks of depth >= 5 with >= 20 instructions (name, depth, instructions, scratch ld/st, vector / scalar / LDS memory, f64 ops):")
    for b in blocks:
        if b["depth"] >= 5 and b["ins"] >= 20:
            print("    %-12s d%d %4d   scratch %2d/%-2d  vmem %2d smem %2d lds %2d  f64 %3d" % (b["name"], b["depth"], b["ins"], b["ld"], b["st"], b["vmem"], b["smem"], b["lds"], b["f64"]))
