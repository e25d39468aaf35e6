#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> nrays_amd/lib/v/NAME.so (tuning builds for A/B runs: tools/kbench.py --libs, tools/wf_ab.py --libs).
# NR_ONLY=33 (or NR_ONLY=6,70,134,198: the FEAT codes of the k_primary permutations the A/B touches, primary_kernel.h: NR_PRIMARY_PERMUTATIONS) compiles only those
# permutations + the two full kernels — every other frame renders with the full kernel, same pixels — in ~15 s instead of ~30 s for everything.
# Codes: balls 33, primitives 37/53, sponza stand-in 6/70/134/198, 8 lights 22/86/150/214, hairball stand-in 2, untransformed opaque meshes 66.
# The objects (obj_NAME/) stay off the GPU box (.gpurunignore); delete the .so files when an experiment is over: every push carries them.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python - "$name" "$@" <<'PY'
import sys, os
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
base = ["-DNR_ONLY=" + os.environ["NR_ONLY"]] if os.environ.get("NR_ONLY") else []
g.build_hip(force=True, extra_flags=base + flags, out=os.path.join(g.LIBDIR, "v", name + ".so"), objdir=os.path.join(g.LIBDIR, "v", "obj_" + name))
PY
