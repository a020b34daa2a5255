/* trig_sweep.c — how far is the oracle's sine / cosine from THIS host's libm sinf / cosf?
 *
 * The reference calls std::sin / std::cos on float (Sophus SO3::expAndTheta, thirdparty/sophus/so3.hpp:538-558; SE3::exp,
 * se3.hpp:723-744): libm's sinf / cosf, whose last bit depends on the libm build.  The oracle (oracle/uwt_oracle.c: sin32 /
 * cos32) and the HIP library (csrc/uwt_math.h) compute (float)sin((double)x): the correctly rounded value in all but
 * double-rounding corner cases.  This sweep evaluates both on EVERY float of [0, limit] (sin is odd, cos even: the negative half
 * is the mirror image) and reports where they differ — exhaustive, so "0 differences" is a statement, not a sample.
 * Also: sqrtf against (float)sqrt((double)x) over every positive float (Sophus takes sqrt(theta_sq), so3.hpp:540).
 *   gcc -O2 -fopenmp -ffp-contract=off tools/trig/trig_sweep.c -o trig_sweep -lm && ./trig_sweep [limit = 0.5]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <gnu/libc-version.h>

static float f_of(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t b_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

int main(int argc, char** argv) {
  const float limit = argc > 1 ? (float)atof(argv[1]) : 0.5f;
  const uint32_t top = b_of(limit);
  long long ds = 0, dc = 0, dq = 0, ds2 = 0, dc2 = 0;
  uint32_t first_s = 0, first_c = 0;
  int have_s = 0, have_c = 0;
#pragma omp parallel for reduction(+ : ds, dc, ds2, dc2) schedule(static, 1 << 20)
  for (long long i = 0; i <= (long long)top; i++) {
    const float x = f_of((uint32_t)i);
    const float s_ref = sinf(x), s_orc = (float)sin((double)x);
    const float c_ref = cosf(x), c_orc = (float)cos((double)x);
    if (b_of(s_ref) != b_of(s_orc)) {
      ds++;
      if (llabs((long long)b_of(s_ref) - (long long)b_of(s_orc)) > 1) ds2++;
#pragma omp critical
      if (!have_s || (uint32_t)i < first_s) { first_s = (uint32_t)i; have_s = 1; }
    }
    if (b_of(c_ref) != b_of(c_orc)) {
      dc++;
      if (llabs((long long)b_of(c_ref) - (long long)b_of(c_orc)) > 1) dc2++;
#pragma omp critical
      if (!have_c || (uint32_t)i < first_c) { first_c = (uint32_t)i; have_c = 1; }
    }
  }
#pragma omp parallel for reduction(+ : dq) schedule(static, 1 << 20)
  for (long long i = 0; i < 0x7f800000LL; i++) {
    const float x = f_of((uint32_t)i);
    if (b_of(sqrtf(x)) != b_of((float)sqrt((double)x))) dq++;
  }
  printf("glibc %s, every float of [0, %g] (%u values; the negative half mirrors it)\n", gnu_get_libc_version(), (double)limit, top + 1u);
  printf("sinf(x) != (float)sin((double)x): %lld values (%lld of them by more than one ulp)", ds, ds2);
  if (have_s) printf("; the smallest: x = %a: sinf %a, oracle %a", (double)f_of(first_s), (double)sinf(f_of(first_s)), (double)(float)sin((double)f_of(first_s)));
  printf("\ncosf(x) != (float)cos((double)x): %lld values (%lld of them by more than one ulp)", dc, dc2);
  if (have_c) printf("; the smallest: x = %a: cosf %a, oracle %a", (double)f_of(first_c), (double)cosf(f_of(first_c)), (double)(float)cos((double)f_of(first_c)));
  printf("\nsqrtf(x) != (float)sqrt((double)x) over every positive finite float: %lld values\n", dq);
  return 0;
}
