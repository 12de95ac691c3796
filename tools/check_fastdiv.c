/* Exhaustive check of the division shortcut used for the coordinate normalisation n = 2q/(S-1) - 1.
 *
 * The kernels replace the IEEE division a/d (d = S-1, a small positive integer) by
 *     r = RN(1/d) (host),  q0 = RN(a*r),  e = RN(a - q0*d) (one fma, exact),  q = RN(q0 + e*r) (one fma)
 * which is correctly rounded for most but not all divisors (Brisebarre, Muller, Raina: "Accelerating correctly
 * rounded floating-point division when the divisor is known in advance").  This program compares it with the
 * hardware division for EVERY fp32 significand a in [1,2) and every divisor 1..DMAX; both sequences are invariant
 * under scaling a by a power of two as long as nothing under- or overflows, so this covers all normal a whose
 * quotient is normal.  Quotients below 2^-25 in magnitude need no check: the normalisation subtracts 1 right after
 * and RN(x - 1) = -1 for every such x.
 *
 *   gcc -O2 -mfma -ffp-contract=off -fopenmp tools/check_fastdiv.c -o /tmp/check_fastdiv -lm && /tmp/check_fastdiv 8191
 * prints the divisors for which the shortcut is NOT exact (none up to 8191 -- see DESIGN.md).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    const int dmax = argc > 1 ? atoi(argv[1]) : 8191;
    int nbad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : nbad)
    for (int di = 1; di <= dmax; ++di) {
        const float d = (float)di;
        const float r = 1.0f / d;
        long bad = 0;
        for (uint32_t m = 0; m < (1u << 23); ++m) {
            const uint32_t bits = 0x3f800000u | m;
            float a;
            memcpy(&a, &bits, 4);
            const float ref = a / d;
            const float q0 = a * r;
            const float e = fmaf(-q0, d, a);
            const float q = fmaf(e, r, q0);
            bad += (q != ref);
        }
        if (bad) {
            ++nbad;
#pragma omp critical
            printf("divisor %d: %ld significands differ\n", di, bad);
        }
    }
    printf("checked divisors 1..%d: %d with mismatches\n", dmax, nbad);
    return nbad != 0;
}
