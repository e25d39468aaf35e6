# tools/final_run.sh — the measurement pass behind profiles/r04_*final* (run through gpurun from the repo root)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 900 python bench.py > gpurun_out/final/r04_bench_final.json 2> gpurun_out/final/bench.err; tail -c 400 gpurun_out/final/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-pmc --no-secondary --steady-only > /tmp/rp_bench.json 2>/tmp/rp.err
cp $(find /tmp/rp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/r04_rocprofv3_kernel_stats_bench.csv
# the device BLAS builder (bvh_device.hip): per-kernel times of two hairball scene creations + one sponza creation
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpb -o b -- python $GRAFT_REPO_ROOT/tools/build_times.py > /tmp/rpb.log 2>&1
cp $(find /tmp/rpb -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/r04_rocprofv3_kernel_stats_scene_build.csv
cd $GRAFT_REPO_ROOT
NRAYS_BUILD_TIMES=1 timeout 200 python tools/build_times.py 2>&1 | grep -v "8 triangles\|80 triangles\|192 tri\|amdgpu.ids" > gpurun_out/final/r04_build_times.log
for s in balls sponza hairball; do timeout 500 python tools/pmc_collect.py $s gpurun_out/final/r04_pmc_$s.json > /dev/null 2>&1; done
timeout 400 python tools/kbench.py --scenes balls,ballsaway,primitives,sponza,sponza8,hairball --steps 30 > gpurun_out/final/r04_kbench_final.log 2>&1
timeout 600 python tools/bigconfigs.py > gpurun_out/final/r04_bigconfigs.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/r04_bench_final.json').read().strip().splitlines()[-1])
def show(n, m):
    r=m['roofline']; print(n, m['value'], m['ms_per_step'], r.get('kernel_ms'), r['frac'], r['bound'], r.get('dram_frac'), r.get('valu_active_frac'), m.get('cpu_baseline',{}).get('value'), m.get('gpu_over_cpu'))
show('balls', d)
for k,v in d.get('secondary',{}).items(): show(k, v)
PY
head -8 gpurun_out/final/r04_rocprofv3_kernel_stats_bench.csv | cut -c1-200
grep -h '"scene"\|"config"' gpurun_out/final/r04_kbench_final.log gpurun_out/final/r04_bigconfigs.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d.get('scene',d.get('config','?'))[:30], d['ms'])
"
