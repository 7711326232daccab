// w / n (n = 1..128) by y = RN(1/n): q0 = w*y; r = fma(-n, q0, w); q = fma(r, y, q0)  vs  IEEE division
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ULL;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(void) {
  long bad = 0, tot = 0;
  for (int n = 1; n <= 128; ++n) {
    volatile double dn = (double)n;
    volatile double y = 1.0 / dn;
    for (long i = 0; i < 4000000; ++i) {
      uint64_t u = rnd();
      double w;
      if (i & 1) {   // random mantissa, exponent in [-60, 60], random sign
        uint64_t bits = (u & 0x800FFFFFFFFFFFFFULL) | ((uint64_t)(1023 - 60 + (int)((u >> 52) % 121)) << 52);
        memcpy(&w, &bits, 8);
      } else {       // sums of float-like values (what value_sum holds)
        float a = (float)((int64_t)(u & 0xFFFFFF) - 0x800000) * (1.0f / 65536.0f);
        float b = (float)((int64_t)((u >> 24) & 0xFFFFFF) - 0x800000) * (1.0f / 4194304.0f);
        w = (double)a + 0.997 * (double)b + 0.994009 * (double)a * 0.37;
      }
      volatile double ref = w / dn;
      double q0 = w * y;
      double r = fma(-dn, q0, w);
      double q = fma(r, y, q0);
      if (q != ref) { if (bad < 5) printf("n=%d w=%a got %a ref %a\n", n, w, q, ref); ++bad; }
      ++tot;
    }
  }
  printf("%ld mismatches in %ld quotients\n", bad, tot);
  return bad != 0;
}
