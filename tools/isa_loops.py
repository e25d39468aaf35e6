#!/usr/bin/env python
"""Instruction mix of the BVH node loops of one kernel in a hipcc -S listing.

  python tools/isa_loops.py listing.s 'k_primaryILb0ELi6ELb1E' [--dump N]

A node loop is recognised by its header block holding >= 6 global_load_dwordx4 (the six plane fetches of a BvhNode);
the loop body is every block LLVM annotates with that header.  Prints VALU / SALU / branch / VMEM / LDS
counts per loop (static counts: both sides of the rare paths are included).
"""
import re
import sys


def kernel_lines(path, pat):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def classify(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else -1
    L = kernel_lines(path, pat)
    labels = {}
    for i, l in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    heads = []
    for lab, i in labels.items():
        j, n = i + 1, 0
        while j < len(L) and not re.match(r"^\.LBB", L[j]):
            n += "global_load_dwordx4" in L[j]; j += 1
        if n >= 6: heads.append((i, lab))
    for k, (i, lab) in enumerate(sorted(heads)):
        # LLVM annotates every block of a loop with "in Loop: Header=BBx_y"; the header itself carries "Parent Loop" / "Loop Header"
        tag = "Header=" + lab[2:]
        body, inside = [], False
        for l in L:
            m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
            if m:
                inside = m.group(1) == lab or (tag + " ") in (m.group(2) + " ")
                continue
            if inside:
                t = l.split(";")[0].strip()
                if t and not t.startswith("."): body.append(t)
        mix = {}
        for l in body: mix[classify(l.split()[0])] = mix.get(classify(l.split()[0]), 0) + 1
        print(f"loop {k} {lab}: {len(body)} instructions", dict(sorted(mix.items())))
        if k == dump: print("\n".join(body))


if __name__ == "__main__":
    main()
