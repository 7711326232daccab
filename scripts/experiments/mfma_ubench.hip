// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 under the conditions of the fused kernel's fc1 step
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MFMAV(c, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const f32x4 *w, float *out, unsigned long long *cyc, int iters) {
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
  float junk[8];
  for (int i = 0; i < 8; ++i) junk[i] = x + i;
  if (MODE >= 2 && MODE <= 4) { LD(b, 0) LD(c, 1) LD(d, 2) }
  if (MODE == 4) {
    // loads from inline asm, one after every 4th MFMA; manual counted vmcnt
    const __attribute__((address_space(1))) char *gb = (const __attribute__((address_space(1))) char *)wb;
#define ALD(dst, it_, p)                                                                               \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lo + (unsigned)((((it_) & 31) * 4 + (p)) * 1024)), "s"(gb) : "memory");
#define STEP4(cur, nxt, it_)                                                                           \
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                    \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                      \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) MFMA(acc[4 * g + t], cur[g][t], x);                  \
    ALD(nxt[g], it_, g)                                                                                \
  }
    for (int it = 0; it < iters; it += 4) {
      STEP4(a, a, it + 4) STEP4(b, b, it + 5) STEP4(c, c, it + 6) STEP4(d, d, it + 7)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else
  for (int it = 0; it < iters; it += 4) {
    if (MODE >= 2 && MODE <= 3) {     // ring of 4 buffers, loads 3 steps ahead, 4 loads per 16 MFMAs
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], a[t >> 2][t & 3], x);
      LD(a, it + 4)
      if (MODE == 3) x += 1.0f;
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], b[t >> 2][t & 3], x);
      LD(b, it + 5)
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], c[t >> 2][t & 3], x);
      LD(c, it + 6)
      _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], d[t >> 2][t & 3], x);
      LD(d, it + 7)
    } else if (MODE == 4) { }
    else if (MODE >= 5) {   // resident operands + (MODE-4) pinned VALU ops behind every MFMA: does VALU hide in the MFMA shadow?
      for (int u = 0; u < 4; ++u) {
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {
          MFMA(acc[t], a[t >> 2][t & 3], x);
          _Pragma("unroll") for (int v = 0; v < MODE - 4; ++v) asm volatile("v_max_f32 %0, %0, %0" : "+v"(junk[(t + v) & 7]));
        }
      }
    }
    else if (MODE == 1) {
      for (int u = 0; u < 4; ++u) { _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMAV(acc[t], a[t >> 2][t & 3], x); }
    } else {
      for (int u = 0; u < 4; ++u) { _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], a[t >> 2][t & 3], x); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (MODE >= 5) for (int i = 0; i < 8; ++i) r += junk[i];
  out[threadIdx.x + blockIdx.x * 512] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  f32x4 *w; float *out; unsigned long long *cyc;
  hipMalloc(&w, 1 << 24); hipMemset(w, 0, 1 << 24);
  hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
  hipMalloc(&cyc, 2048 * 8);
  const int iters = 2000;
  for (int nthreads = 256; nthreads <= 256; nthreads += 256)
  for (int mode = 0; mode < 9; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 7) hipLaunchKernelGGL(k<8>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      if (mode == 8) hipLaunchKernelGGL(k<12>, dim3(256), dim3(nthreads), 0, 0, w, out, cyc, iters);
      hipDeviceSynchronize();
    }
    unsigned long long h[2048];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const int nw = nthreads / 64;
    double s = 0;      // slowest wave of every block, averaged over blocks
    for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < nw; ++w) m = h[b * 8 + w] > m ? h[b * 8 + w] : m; s += (double)m * 4; }
    printf("mode %d, %d threads: %.1f cycles per own MFMA per wave -> %.1f cycles per MFMA per SIMD\n", mode, nthreads, s / 1024 / iters / 16, s / 1024 / iters / 16 / (nthreads / 256));
  }
  return 0;
}
