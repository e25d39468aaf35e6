// Exhaustive check of the division k_primary's PLAIN raygen uses for pixel / resolution (trace_device.h: generate_primary):
//   y = RN(1 / b) (host), q0 = RN(a * y), r = fma(-q0, b, a) (exact), q = fma(r, y, q0)   ==   RN(a / b)
// (Markstein's correction step; a = pixel index, b = width or height: small integers).  Every b in [1, 16384], every a in [0, b + 64).
//   gcc -O2 -mfma -o /tmp/div_markstein tools/probe/div_markstein.c -lm && /tmp/div_markstein
#include <math.h>
#include <stdio.h>
int main(void) {
    unsigned long long bad = 0, n = 0;
    for (int b = 1; b <= 16384; ++b) {
        const double bd = (double)b, y = 1.0 / bd;
        for (int a = 0; a < b + 64; ++a) {
            const double ad = (double)a;
            volatile double want = ad / bd;
            const double q0 = ad * y;
            const double r = fma(-q0, bd, ad);
            const double q = fma(r, y, q0);
            if (q != want) { if (bad < 10) printf("mismatch a=%d b=%d: %.17g vs %.17g\n", a, b, q, want); ++bad; }
            ++n;
        }
    }
    printf("%llu quotients checked, %llu mismatches\n", n, bad);
    return bad != 0;
}
