#!/bin/bash
# tools/wf_prof.sh OUTDIR WORK [env assignments...] — rocprofv3 kernel stats of one tools/wf_ab.py child (run through gpurun from the repo root)
out=$1; work=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_$work
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$work -o p -- python $GRAFT_REPO_ROOT/tools/wf_ab.py --child $work --steps 20 > /tmp/rp_$work.log 2>&1
tail -1 /tmp/rp_$work.log | cut -c1-200
mkdir -p $GRAFT_REPO_ROOT/$out
cp $(find /tmp/rp_$work -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$out/${work}_kernel_stats.csv
cut -d, -f1-7 $GRAFT_REPO_ROOT/$out/${work}_kernel_stats.csv | sed 's/void nrays:://' | cut -c1-150 | head -12
