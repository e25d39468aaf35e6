// Shader clock actually delivered to a kernel (GPU box): s_memtime counts shader-engine cycles, s_memrealtime a constant
// 100 MHz reference, so their ratio over a kernel is the engine clock that kernel ran at.  Three cases: one wave per CU
// spinning on dependent FMAs (a latency-bound kernel like the analytic frames), every SIMD busy, and short launches back
// to back with idle gaps (DPM ramp).   hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long* out, int iters) {
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = __builtin_fmaf(a, b, 1e-7f); a = __builtin_fmaf(a, b, 1e-7f); a = __builtin_fmaf(a, b, 1e-7f); a = __builtin_fmaf(a, b, 1e-7f); }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a == 12345.0f) out[0] = 0;
}
static void run(const char* what, int blocks, int threads, int iters, int reps, int gap_us) {
    unsigned long long* d; hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks);
    std::vector<unsigned long long> h(2 * blocks);
    double ghz = 0; int n = 0;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        if (r >= reps / 2) { ghz += (double)h[0] / ((double)h[1] * 10.0); ++n; } // 100 MHz: 10 ns per tick
        if (gap_us) { timespec ts{0, gap_us * 1000L}; nanosleep(&ts, nullptr); }
    }
    printf("{\"case\": \"%s\", \"blocks\": %d, \"threads\": %d, \"kernel_us\": %.1f, \"shader_clock_ghz\": %.3f}\n", what, blocks, threads, (double)h[1] / 100.0, ghz / n);
    hipFree(d);
}
int main() {
    run("one wave per CU, 50 us kernels back to back", 256, 64, 6000, 200, 0);
    run("one wave per CU, 50 us kernels with 2 ms idle gaps", 256, 64, 6000, 40, 2000);
    run("every SIMD busy (8 waves per CU), 50 us kernels back to back", 1024, 512, 6000, 200, 0);
    run("every SIMD busy, 5 ms kernels", 1024, 512, 600000, 10, 0);
    return 0;
}
