// Is  q = a*y; r = fma(-b, q, a); res = fma(r, y, q)  with y = the Newton-refined reciprocal of the compiler's own f64
// division expansion bit-identical to a / b when no operand scaling is involved?  (b loop-invariant in the descent:
// MinMaxStats.normalize divides by the same span at every level.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__device__ __forceinline__ double refined_rcp(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}
__global__ void k(const double *a, const double *b, unsigned long long *bad, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double y = refined_rcp(b[i]);
  const double q = a[i] * y;
  const double r = __builtin_fma(-b[i], q, a[i]);
  const double res = __builtin_fma(r, y, q);
  const double ref = a[i] / b[i];
  if (__double_as_longlong(res) != __double_as_longlong(ref)) atomicAdd(bad, 1ull);
}
int main() {
  const int n = 1 << 24;
  double *ha = (double *)malloc(n * 8), *hb = (double *)malloc(n * 8);
  double *a, *b; unsigned long long *bad, hbad;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&bad, 8);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int mode = 0; mode < 5; ++mode) {
    for (int i = 0; i < n; ++i) {
      const double u = (rnd() >> 11) * (1.0 / 9007199254740992.0), v = (rnd() >> 11) * (1.0 / 9007199254740992.0);
      double bb, aa;
      if (mode == 0) { bb = ldexp(0.5 + 0.5 * v, (int)(rnd() % 120) - 60); aa = u * bb; }            // a in [0, b]
      else if (mode == 1) { bb = 1e-3 + 100.0 * v; aa = (i & 3) == 0 ? 0.0 : ((i & 3) == 1 ? bb : u * bb); }
      else if (mode == 2) { bb = ldexp(1.0 + v, (int)(rnd() % 120) - 60); aa = ldexp(u, -(int)(rnd() % 200)) * bb; }   // tiny quotients
      else if (mode == 3) { bb = (float)(0.01 + 50 * v); aa = (double)(float)(u * bb); }               // f32-born values
      else { bb = ldexp(1.0 + v, (int)(rnd() % 120) - 60); aa = -u * bb * 3.0; }                        // negative / beyond b
      ha[i] = aa; hb[i] = bb;
    }
    hipMemcpy(a, ha, n * 8, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 8, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, bad, n);
    hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %llu of %d quotients differ from a / b\n", mode, hbad, n);
  }
  return 0;
}
