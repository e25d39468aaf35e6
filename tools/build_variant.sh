#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> nrays_amd/lib/ab/NAME.so (tuning builds for A/B runs: tools/kbench.py --libs, tools/wf_ab.py --libs)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python - "$name" "$@" <<'PY'
import sys, os
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
g.build_hip(force=True, extra_flags=["-DNR_ONLY_MESH"] + flags, out=os.path.join(g.LIBDIR, "ab", name + ".so"), objdir=os.path.join(g.LIBDIR, "ab", "obj_" + name))
PY
