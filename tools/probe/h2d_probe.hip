// h2d_probe.hip — how fast do a caller's pageable arrays reach the device?  (bvh_device.hip uploads ~93 MB of mesh arrays.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 35u << 20;
    char* h = (char*)malloc(n); memset(h, 1, n);
    char* d = nullptr; hipMalloc((void**)&d, n);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) { double t = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); printf("pageable hipMemcpy %zu MB: %.2f ms\n", n >> 20, now() - t); }
    { char* d2 = nullptr; double t = now(); hipMalloc((void**)&d2, n); double t1 = now(); hipMemcpy(d2, h, n, hipMemcpyHostToDevice); printf("fresh device buffer: malloc %.2f ms, copy %.2f ms\n", t1 - t, now() - t1); }
    { char* h2 = (char*)malloc(n); memset(h2, 2, n); double t = now(); hipMemcpy(d, h2, n, hipMemcpyHostToDevice); printf("another host buffer, first copy: %.2f ms\n", now() - t); t = now(); hipMemcpy(d, h2, n, hipMemcpyHostToDevice); printf("  second copy: %.2f ms\n", now() - t); }
    { double t = now(); hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t2 = now(); hipHostUnregister(h);
      printf("hipHostRegister (%d) %.2f ms, copy %.2f ms, unregister %.2f ms\n", (int)e, t1 - t, t2 - t1, now() - t2); }
    { char* p = nullptr; double t = now(); hipHostMalloc((void**)&p, n, hipHostMallocDefault); double t1 = now();
      const int T = 8; std::vector<std::thread> th; for (int k = 0; k < T; ++k) th.emplace_back([&, k] { memcpy(p + n * k / T, h + n * k / T, n * (k + 1) / T - n * k / T); }); for (auto& x : th) x.join();
      double t2 = now(); hipMemcpy(d, p, n, hipMemcpyHostToDevice); double t3 = now();
      printf("pinned staging: hipHostMalloc %.2f ms, 8-thread memcpy %.2f ms, DMA %.2f ms\n", t1 - t, t2 - t1, t3 - t2);
      t = now(); for (int c = 0; c < 8; ++c) { size_t lo = n * c / 8, hi = n * (c + 1) / 8; memcpy(p + lo, h + lo, hi - lo); hipMemcpyAsync(d + lo, p + lo, hi - lo, hipMemcpyHostToDevice, 0); } hipDeviceSynchronize();
      printf("  chunked: memcpy + async DMA pipelined %.2f ms\n", now() - t); }
    return 0;
}
