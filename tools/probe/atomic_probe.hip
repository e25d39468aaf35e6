// What a work-list dequeue and an in-workgroup hand-over cost on MI355X (GPU box):
//   1. latency of a returning atomicAdd on global memory, one wave, dependent chain: agent scope, workgroup scope, a plain load for comparison
//   2. throughput of per-workgroup counters: 512 workgroups x 4 waves, 20 dequeues each, counters padded to 64 B / packed (4 B apart) / one per XCD (8)
//   3. round trip of an LDS flag between two waves of a workgroup (producer / consumer hand-over)
//   hipcc --offload-arch=gfx950 -O2 -o _build/atomic_probe atomic_probe.hip && _build/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
template <int SCOPE>
__global__ void lat_atomic(unsigned* ctr, unsigned long long* out, int iters) {
    unsigned idx = 0; unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { unsigned v = 0; if (threadIdx.x == 0) v = __hip_atomic_fetch_add(&ctr[idx], 1u, __ATOMIC_RELAXED, SCOPE); idx = (unsigned)__builtin_amdgcn_readfirstlane((int)v) & 0u; }
    unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = (c1 - c0) / iters;
}
__global__ void lat_load(const unsigned* p, unsigned long long* out, int iters) {
    unsigned idx = 0; unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { unsigned v = __builtin_nontemporal_load(&p[idx + threadIdx.x * 0]); idx = (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
    unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = (c1 - c0) / iters;
}
// every wave: `n` dependent dequeues from the counter of its workgroup (stride in dwords) or of its XCD (mode 2)
__global__ void thr_counters(unsigned* ctr, int stride, int mode, int n, unsigned long long* out) {
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); x &= 7u;
    unsigned* c = mode == 2 ? ctr + x * 16 : ctr + (size_t)blockIdx.x * stride;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    for (int i = 0; i < n; ++i) { unsigned v = 0; if ((threadIdx.x & 63) == 0) v = atomicAdd(c, 1u); acc += (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (r1 - r0) | ((unsigned long long)(acc & 1u) << 63);
}
// wave 0 writes a payload + flag, wave 1 waits, answers; `iters` round trips
__global__ void lds_pingpong(unsigned long long* out, int iters, int sleep) {
    __shared__ unsigned flag_a, flag_b, payload[64];
    if (threadIdx.x == 0) { flag_a = 0; flag_b = 0; }
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 1; i <= iters; ++i) {
        if (w == 0) {
            payload[lane] = i;
            __hip_atomic_store(&flag_a, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&flag_b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)i) { if (sleep) __builtin_amdgcn_s_sleep(1); }
        } else if (w == 1) {
            while (__hip_atomic_load(&flag_a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)i) { if (sleep) __builtin_amdgcn_s_sleep(1); }
            if (payload[lane] != (unsigned)i) out[1] = 1; // must see the payload
            __hip_atomic_store(&flag_b, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = (c1 - c0) / iters;
}
int main() {
    unsigned* ctr; hipMalloc(&ctr, 1 << 20); hipMemset(ctr, 0, 1 << 20);
    unsigned long long* out; hipMalloc(&out, 1 << 16); std::vector<unsigned long long> h(8192);
    auto get = [&](int n) { hipDeviceSynchronize(); hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost); };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(lat_atomic<__HIP_MEMORY_SCOPE_AGENT>, dim3(1), dim3(64), 0, 0, ctr, out, 2000); get(1);
        if (rep) printf("{\"latency_cycles\": {\"atomic_agent\": %llu", h[0]);
        hipLaunchKernelGGL(lat_atomic<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(1), dim3(64), 0, 0, ctr, out, 2000); get(1);
        if (rep) printf(", \"atomic_workgroup\": %llu", h[0]);
        hipLaunchKernelGGL(lat_atomic<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(1), dim3(64), 0, 0, ctr, out, 2000); get(1);
        if (rep) printf(", \"atomic_system\": %llu", h[0]);
        hipMemset(ctr, 0, 1 << 20);
        hipLaunchKernelGGL(lat_load, dim3(1), dim3(64), 0, 0, ctr, out, 2000); get(1);
        if (rep) printf(", \"load_nt\": %llu}}\n", h[0]);
    }
    struct { const char* name; int stride, mode; } cases[] = {{"per-workgroup counters, 64 B apart", 16, 0}, {"per-workgroup counters, 128 B apart", 32, 0}, {"per-workgroup counters, packed", 1, 0}, {"one counter per XCD", 0, 2}};
    for (auto& c : cases) for (int n : {4, 20}) {
        hipMemset(ctr, 0, 1 << 20);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(thr_counters, dim3(512), dim3(256), 0, 0, ctr, c.stride, c.mode, n, out);
        get(2048);
        double s = 0, mx = 0; for (int i = 0; i < 2048; ++i) { double t = (double)(h[i] & ~(1ull << 63)) / 100.0; s += t; if (t > mx) mx = t; }
        printf("{\"case\": \"%s\", \"dequeues_per_wave\": %d, \"us_per_dequeue_mean\": %.3f, \"slowest_wave_us\": %.2f}\n", c.name, n, s / 2048 / n, mx);
    }
    for (int sl : {0, 1}) { hipMemset(out, 0, 64); hipLaunchKernelGGL(lds_pingpong, dim3(1), dim3(128), 0, 0, out, 2000, sl); get(2);
        printf("{\"lds_flag_round_trip_cycles\": %llu, \"s_sleep\": %d, \"payload_errors\": %llu}\n", h[0], sl, h[1]); }
    return 0;
}
