cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/kbench_pretouch.log
: > $L
echo "# default" >> $L; python tools/kbench.py --scenes balls,ballsaway,primitives --steps 60 >> $L 2>&1
echo "# pretouch 64 KiB" >> $L; python tools/kbench.py --libs nrays_amd/lib/v/pt64.so --scenes balls,ballsaway,primitives --steps 60 >> $L 2>&1
echo "# pretouch 2 MiB" >> $L; python tools/kbench.py --libs nrays_amd/lib/v/pt2m.so --scenes balls,ballsaway,primitives --steps 60 >> $L 2>&1
echo "# pretouch 64 KiB, deal 1,3,0,1" >> $L; NRAYS_DEAL=1,3,0,1 python tools/kbench.py --libs nrays_amd/lib/v/pt64.so --scenes balls,primitives --steps 60 >> $L 2>&1
NRAYS_HIP_LIB=nrays_amd/lib/v/tcpt.so python tools/wave_breakdown.py balls 6 1 > gpurun_out/r05/breakdown_pt_d1.log 2>&1
NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/wave_breakdown.py balls 6 1 > gpurun_out/r05/breakdown_nopt_d1.log 2>&1
grep -v "amdgpu.ids\|== lib" $L | cut -c1-100
