// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 under the conditions of the fused kernel's fc1 step
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MFMAV(c, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const f32x4 *w, float *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 a[4];
  for (int i = 0; i < 4; ++i) a[i] = w[i * 64 + lane];
  float x = out[lane];
  const char *wb = (const char *)(w + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 4096);
  const unsigned lo = lane * 16;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define LD(buf, it_)                                                                                   \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) buf[p] =                                                 \
      *(const __attribute__((address_space(1))) f32x4 *)((const __attribute__((address_space(1))) char *)wb + \
                                                           (size_t)((((it_) & 31) * 4 + p) * 1024) + lo);
  f32x4 b[4], c[4], d[4];
  if (MODE >= 2) { LD(b, 0) LD(c, 1) LD(d, 2) }
  for (int it = 0; it < iters; it += 4) {
    if (MODE >= 2) {     // ring of 4 buffers, loads 3 steps ahead, 4 loads per 16 MFMAs
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], a[t >> 2][t & 3], x);
      LD(a, it + 4)
      if (MODE == 3) x += 1.0f;
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], b[t >> 2][t & 3], x);
      LD(b, it + 5)
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], c[t >> 2][t & 3], x);
      LD(c, it + 6)
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], d[t >> 2][t & 3], x);
      LD(d, it + 7)
    } else if (MODE == 1) {
      for (int u = 0; u < 4; ++u) { _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMAV(acc[t], a[t >> 2][t & 3], x); }
    } else {
      for (int u = 0; u < 4; ++u) { _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], a[t >> 2][t & 3], x); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x + blockIdx.x * 256] = r;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  f32x4 *w; float *out; unsigned long long *cyc;
  hipMalloc(&w, 1 << 24); hipMemset(w, 0, 1 << 24);
  hipMalloc(&out, 256 * 256 * 4); hipMemset(out, 0, 256 * 256 * 4);
  hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
      hipDeviceSynchronize();
    }
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("mode %d: %.1f cycles per MFMA (avg over 1024 waves)\n", mode, s / 1024 / iters / 16);
  }
  return 0;
}
