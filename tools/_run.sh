cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/kbench_simdfirst.log
NRAYS_ROW_MIX=1 python -m pytest tests/test_parity_gpu.py tests/test_worklists_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "# nw8 row mix" > $L; NRAYS_ROW_MIX=1 python tools/kbench.py --scenes balls,primitives --steps 60 >> $L 2>&1
echo "# nw8" >> $L; python tools/kbench.py --scenes balls,primitives --steps 60 >> $L 2>&1
echo "# nw4 row mix" >> $L; NRAYS_ROW_MIX=1 NRAYS_WG_WAVES=4 python tools/kbench.py --scenes balls,primitives --steps 60 >> $L 2>&1
echo "# nw4" >> $L; NRAYS_WG_WAVES=4 python tools/kbench.py --scenes balls,primitives --steps 60 >> $L 2>&1
NRAYS_ROW_MIX=1 NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/wave_breakdown.py balls 6 > gpurun_out/r05/breakdown_nw8_mix_d4b.log 2>&1
NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/wave_timeline.py balls 4 > gpurun_out/r05/timeline_nw8.log 2>&1
grep -v amdgpu.ids $L | cut -c1-130
