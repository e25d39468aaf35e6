// Probe: are f64 div / sqrt / a*b+c on gfx950 bit-identical to the host (IEEE, unfused)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const double* a, const double* b, const double* c, double* q, double* s, double* m, double* nrm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = a[i] / b[i];
    s[i] = sqrt(fabs(a[i]));
    m[i] = a[i] * b[i] + c[i];
    double x = a[i], y = b[i], z = c[i];
    double l = sqrt(x * x + y * y + z * z);
    nrm[i] = x / l;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> a(n), b(n), c(n), q(n), s(n), m(n), r(n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = (rand() / (double)RAND_MAX - 0.5) * 1000; b[i] = (rand() / (double)RAND_MAX - 0.5) * 3 + 1e-3; c[i] = (rand() / (double)RAND_MAX - 0.5) * 7; }
    double *da, *db, *dc, *dq, *ds, *dm, *dr;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dm, n * 8); hipMalloc(&dr, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dc, dq, ds, dm, dr, n);
    hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    long bq = 0, bs = 0, bm = 0, bmf = 0, br = 0;
    for (int i = 0; i < n; ++i) {
        volatile double hq = a[i] / b[i]; volatile double hs = std::sqrt(std::fabs(a[i]));
        volatile double p = a[i] * b[i]; volatile double hm = p + c[i]; double hf = std::fma(a[i], b[i], c[i]);
        volatile double xx = a[i] * a[i]; volatile double yy = b[i] * b[i]; volatile double zz = c[i] * c[i]; volatile double sum = xx + yy; sum = sum + zz;
        volatile double l = std::sqrt(sum); volatile double hr = a[i] / l;
        bq += memcmp((void*)&hq, &q[i], 8) != 0; bs += memcmp((void*)&hs, &s[i], 8) != 0; bm += memcmp((void*)&hm, &m[i], 8) != 0; bmf += memcmp(&hf, &m[i], 8) != 0; br += memcmp((void*)&hr, &r[i], 8) != 0;
    }
    printf("mismatches of %d: div %ld sqrt %ld mul-add(unfused) %ld mul-add(vs fma) %ld normalize %ld\n", n, bq, bs, bm, bmf, br);
    return 0;
}
