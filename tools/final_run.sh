# tools/final_run.sh — the measurement pass behind profiles/r06_*final* (run through gpurun from the repo root):
#   the default bench line; one `rocprofv3 --kernel-trace --stats` per workload over a steady-only run (balls, sponza, hairball, config 4, config 5), so that every
#   roofline block's kernel time can be recomputed from profiles/r05_rocprofv3_kernel_stats_<scene>.csv; the counter passes; kbench / bigconfigs.
set -x
cd $GRAFT_REPO_ROOT
R=r06
mkdir -p gpurun_out/final
timeout 1200 python bench.py > gpurun_out/final/${R}_bench_final.json 2> gpurun_out/final/bench.err; tail -c 400 gpurun_out/final/bench.err
cd /tmp; export TMPDIR=/tmp
for s in balls sponza hairball config4 config5; do
  case $s in config5) st=3; wu=1;; config4) st=40; wu=5;; balls) st=400; wu=10;; *) st=100; wu=10;; esac
  rm -rf /tmp/rp_$s
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$s -o b -- python $GRAFT_REPO_ROOT/bench.py --scene $s --steps $st --warmup $wu --no-cpu-baseline --no-pmc --no-secondary --steady-only > $GRAFT_REPO_ROOT/gpurun_out/final/${R}_bench_under_rocprof_$s.json 2>/tmp/rp_$s.err
  cp $(find /tmp/rp_$s -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/${R}_rocprofv3_kernel_stats_$s.csv
done
# the device BLAS builder (bvh_device.hip): per-kernel times of two hairball scene creations + one sponza creation
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpb -o b -- python $GRAFT_REPO_ROOT/tools/build_times.py > /tmp/rpb.log 2>&1
cp $(find /tmp/rpb -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/${R}_rocprofv3_kernel_stats_scene_build.csv
cd $GRAFT_REPO_ROOT
NRAYS_BUILD_TIMES=1 timeout 200 python tools/build_times.py 2>&1 | grep -v "8 triangles\|80 triangles\|192 tri\|amdgpu.ids\|upload of 0.0" > gpurun_out/final/${R}_build_times.log
for s in balls sponza hairball; do timeout 500 python tools/pmc_collect.py $s gpurun_out/final/${R}_pmc_$s.json > /dev/null 2>&1; done
timeout 400 python tools/kbench.py --scenes balls,ballsaway,primitives,sponza,sponza8,hairball --steps 30 > gpurun_out/final/${R}_kbench_final.log 2>&1
timeout 600 python tools/bigconfigs.py > gpurun_out/final/${R}_bigconfigs.log 2>&1
timeout 600 python tools/tile_scaling.py sponza8_4k 2>&1 | grep -v amdgpu.ids > gpurun_out/final/${R}_tile_scaling.log
timeout 300 python tools/rank_bounds.py 2>&1 | grep world >> gpurun_out/final/${R}_tile_scaling.log
for s in balls primitives sponza sponza8 hairball; do timeout 300 python tools/regimes.py $s 2>&1 | grep scene >> gpurun_out/final/${R}_regimes_final.log; done
python - <<'PY'
import json, csv, glob
d=json.loads(open('gpurun_out/final/r06_bench_final.json').read().strip().splitlines()[-1])
def show(n, m):
    r=m['roofline']; print(n, m['value'], m['ms_per_step'], r.get('kernel_ms'), r.get('kernel_ms_events'), r['frac'], r['bound'], r.get('dram_frac'), r.get('valu_active_frac'), m.get('cpu_baseline',{}).get('value'), m.get('gpu_over_cpu'), m.get('gpu_over_cpu_at_full_host'))
show('balls', d)
for k,v in d.get('secondary',{}).items():
    if 'roofline' in v: show(k, v)
print(d.get('north_star_sponza'))
for f in sorted(glob.glob('gpurun_out/final/r06_rocprofv3_kernel_stats_*.csv')):
    rows=[r for r in csv.DictReader(open(f)) if 'k_primary' in r['Name'] or 'k_resolve' in r['Name']]
    print(f.split('stats_')[-1], [(r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e3,2)) for r in rows[:3]])
PY
