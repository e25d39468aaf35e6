// Probe: issue cost of VALU wave64 instructions on gfx950 — SIMD cycles per wave-instruction with the SIMD saturated
// (8 waves per SIMD, 8 independent dependency chains per wave) and with ONE wave per SIMD (what a latency-bound
// traversal wave sees).  The guide (MI355X_MICROARCH.md) gives 2 cycles for v_fma_f32 and no figure for f64 or packed
// f32; bench.py's fp64_issue_frac and DESIGN.md's issue accounting use the numbers printed here.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/valu_rate tools/probe/valu_rate.hip && tools/probe/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

constexpr int kIters = 2048, kChains = 8;

#define CHAIN_KERNEL(name, T, INIT, ASM, CONSTRAINT)                                                            \
    __global__ void __launch_bounds__(256) name(T* out, unsigned long long* cyc) {                              \
        T a[kChains];                                                                                           \
        for (int c = 0; c < kChains; ++c) a[c] = INIT + (T)(threadIdx.x + c);                                   \
        T m = (T)1.0000001, b = (T)0.5;                                                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kIters; ++i) {                                                                      \
            _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile(ASM : "+" CONSTRAINT(a[c]) : CONSTRAINT(m), CONSTRAINT(b)); \
        }                                                                                                       \
        unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        T s = 0; for (int c = 0; c < kChains; ++c) s += a[c];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                         \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                        \
    }

CHAIN_KERNEL(k_fma_f32, float, 1.0f, "v_fma_f32 %0, %0, %1, %2", "v")
CHAIN_KERNEL(k_mul_f32, float, 1.0f, "v_mul_f32 %0, %0, %1", "v")
CHAIN_KERNEL(k_fma_f64, double, 1.0, "v_fma_f64 %0, %0, %1, %2", "v")
CHAIN_KERNEL(k_mul_f64, double, 1.0, "v_mul_f64 %0, %0, %1", "v")
CHAIN_KERNEL(k_add_f64, double, 1.0, "v_add_f64 %0, %0, %2", "v")
CHAIN_KERNEL(k_rcp_f64, double, 1.5, "v_rcp_f64 %0, %0", "v")
CHAIN_KERNEL(k_rcp_f32, float, 1.5f, "v_rcp_f32 %0, %0", "v")
CHAIN_KERNEL(k_max_f32, float, 1.0f, "v_max_f32 %0, %0, %1", "v")
CHAIN_KERNEL(k_cndmask, float, 1.0f, "v_cndmask_b32 %0, %0, %1, vcc", "v")

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_pk_fma_f32(float* out, unsigned long long* cyc) {
    f2 a[kChains];
    for (int c = 0; c < kChains; ++c) a[c] = f2{1.0f + threadIdx.x + c, 2.0f + c};
    f2 m = f2{1.0000001f, 0.9999999f}, b = f2{0.5f, 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < kChains; ++c) s += a[c].x + a[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_pk_mul_f32(float* out, unsigned long long* cyc) {
    f2 a[kChains];
    for (int c = 0; c < kChains; ++c) a[c] = f2{1.0f + threadIdx.x + c, 2.0f + c};
    f2 m = f2{1.0000001f, 0.9999999f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < kChains; ++c) s += a[c].x + a[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    void* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8); hipMalloc(&cyc, (size_t)cus * 8 * 8);
    unsigned long long* h = new unsigned long long[cus * 8];
    printf("{\"device_cus\": %d, \"iters\": %d, \"chains\": %d, \"cycles_per_wave_instruction\": {", cus, kIters, kChains);
    bool first = true;
#define RUN(kern, T)                                                                                              \
    for (int w : {1, 8}) {                                                                                        \
        int blocks = cus * w;                                                                                     \
        kern<<<blocks, 256>>>((T*)out, cyc); kern<<<blocks, 256>>>((T*)out, cyc);                                 \
        hipDeviceSynchronize();                                                                                   \
        hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);                                                     \
        double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];                                         \
        /* per SIMD: w waves each issue kIters*kChains instructions in (mean) s/blocks cycles */                  \
        double per = (s / blocks) / ((double)kIters * kChains * w);                                               \
        printf("%s\"%s@%dwaves/SIMD\": %.2f", first ? "" : ", ", #kern, w, per); first = false;                  \
    }
    RUN(k_fma_f32, float) RUN(k_mul_f32, float) RUN(k_pk_fma_f32, float) RUN(k_pk_mul_f32, float) RUN(k_max_f32, float) RUN(k_cndmask, float)
    RUN(k_rcp_f32, float) RUN(k_fma_f64, double) RUN(k_mul_f64, double) RUN(k_add_f64, double) RUN(k_rcp_f64, double)
    printf("}, \"note\": \"SIMD cycles per wave64 instruction; @8waves = saturated issue rate, @1wave = one wave alone with 8 independent chains\"}\n");
    return 0;
}
