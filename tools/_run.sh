cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_device_build_gpu.py tests/test_wavefront_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -5
NRAYS_BUILD_TIMES=1 timeout 200 python tools/build_times.py 2>&1 | grep -v "8 triangles\|80 triangles\|192 tri\|amdgpu.ids" | tail -30
