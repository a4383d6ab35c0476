/* Exhaustive check behind the 3D rollout's division-free observation scalars (snac_amd/csrc/snac_hip.hip, Roll3D):
 * for r = RN(1/d), q = RN(n*r), the value RN(q + RN(n - q*d) * r) (two fused multiply-adds) is bit-identical to the IEEE
 * quotient n/d for every integer 0 <= n <= 32767, 1 <= d <= 32767 -- the whole domain of count_brick/total_brick and
 * count_step/total_step (Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py:73-75).  Prints the number of mismatches of
 * the corrected and of the uncorrected product; exit status 1 if the corrected form ever differs. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

int main(void) {
    long bad = 0, bad1 = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad, bad1)
    for (int d = 1; d <= 32767; ++d) {
        const double dd = (double)d, r = 1.0 / dd;
        for (int n = 0; n <= 32767; ++n) {
            const double nn = (double)n, q = nn * r, rem = fma(-q, dd, nn), q2 = fma(rem, r, q), t = nn / dd;
            if (bits(q) != bits(t)) bad1++;
            if (bits(q2) != bits(t)) bad++;
        }
    }
    printf("uncorrected %ld corrected %ld pairs %ld\n", bad1, bad, 32767L * 32768L);
    return bad != 0;
}
