// Probe: is cast_triangle's toi on gfx950 bit-identical to the host's for given (triangle, ray) cases?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/nrays_abi.h"
#include "../../nrays_amd/csrc/device_types.h"
#include "../../nrays_amd/csrc/trace_device.h"
using namespace nrays;
__global__ void k(const double* in, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = in + 16 * i;
    d3 a = D3(r[0], r[1], r[2]), b = D3(r[3], r[4], r[5]), c = D3(r[6], r[7], r[8]), o = D3(r[9], r[10], r[11]), d = D3(r[12], r[13], r[14]);
    double toi = -1.0;
    bool hit = cast_triangle(a, b, c, o, d, toi, nullptr, nullptr);
    out[2 * i] = hit ? toi : -1.0;
    // the intermediate quantities
    d3 ab = b - a, ac = c - a; d3 nn = cross(ab, ac); out[2 * i + 1] = dot(nn, d);
}
// host copy of the same function (gcc/clang host pass, -ffp-contract=off)
static bool host_cast(const double* r, double& toi, double& dn_out) {
    auto sub = [](const double* p, const double* q, double* o) { for (int k = 0; k < 3; ++k) o[k] = p[k] - q[k]; };
    double ab[3], ac[3], ap[3], n[3], e[3];
    sub(r + 3, r, ab); sub(r + 6, r, ac);
    n[0] = ab[1] * ac[2] - ab[2] * ac[1]; n[1] = ab[2] * ac[0] - ab[0] * ac[2]; n[2] = ab[0] * ac[1] - ab[1] * ac[0];
    const double* d = r + 12; const double* o = r + 9;
    double dn = n[0] * d[0] + n[1] * d[1] + n[2] * d[2]; dn_out = dn;
    if (dn == 0.0) return false;
    sub(o, r, ap);
    double t = ap[0] * n[0] + ap[1] * n[1] + ap[2] * n[2];
    if ((t < 0.0 && dn < 0.0) || (t > 0.0 && dn > 0.0)) return false;
    double dabs = dn < 0 ? -dn : dn;
    e[0] = -(d[1] * ap[2] - d[2] * ap[1]); e[1] = -(d[2] * ap[0] - d[0] * ap[2]); e[2] = -(d[0] * ap[1] - d[1] * ap[0]);
    double v, w;
    if (t < 0.0) { v = -(ac[0] * e[0] + ac[1] * e[1] + ac[2] * e[2]); if (v < 0 || v > dabs) return false; w = ab[0] * e[0] + ab[1] * e[1] + ab[2] * e[2]; if (w < 0 || v + w > dabs) return false; toi = -t * (1.0 / dabs); }
    else { v = ac[0] * e[0] + ac[1] * e[1] + ac[2] * e[2]; if (v < 0 || v > dabs) return false; w = -(ab[0] * e[0] + ab[1] * e[1] + ab[2] * e[2]); if (w < 0 || v + w > dabs) return false; toi = t * (1.0 / dabs); }
    return true;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argc > 1 ? argv[1] : "tools/probe/tri_cases.bin", "rb");
    if (!f) { printf("no case file\n"); return 1; }
    unsigned n = 0; fread(&n, 4, 1, f);
    std::vector<double> in(16 * n), out(2 * n);
    fread(in.data(), 8, 16 * n, f); fclose(f);
    double *di, *dout; hipMalloc(&di, in.size() * 8); hipMalloc(&dout, out.size() * 8);
    hipMemcpy(di, in.data(), in.size() * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout, (int)n);
    hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
    for (unsigned i = 0; i < n; ++i) {
        double ht = -1.0, hdn = 0.0; bool hh = host_cast(&in[16 * i], ht, hdn);
        unsigned long long a, b, c, d; memcpy(&a, &out[2 * i], 8); memcpy(&b, &ht, 8); memcpy(&c, &out[2 * i + 1], 8); memcpy(&d, &hdn, 8);
        printf("tri %6.0f  gpu toi %.17g (%016llx)  host toi %.17g (%016llx) %s | dn gpu %016llx host %016llx %s\n", in[16 * i + 15], out[2 * i], a, hh ? ht : -1.0, b, a == b ? "same" : "DIFFERENT", c, d, c == d ? "same" : "DIFFERENT");
    }
    return 0;
}
