#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> nrays_amd/lib/v/NAME.so (tuning builds for A/B runs: tools/kbench.py --libs, tools/wf_ab.py --libs).
# NR_VARIANT_FULL=1 keeps the analytic permutations (default: -DNR_ONLY_MESH, a third of the build time).
# The objects (obj_NAME/) stay off the GPU box (.gpurunignore); delete the .so files when an experiment is over: every push carries them.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python - "$name" "$@" <<'PY'
import sys, os
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
base = [] if os.environ.get("NR_VARIANT_FULL") else ["-DNR_ONLY_MESH"]
g.build_hip(force=True, extra_flags=base + flags, out=os.path.join(g.LIBDIR, "v", name + ".so"), objdir=os.path.join(g.LIBDIR, "v", "obj_" + name))
PY
