// micro-benchmark: does a latency-bound VALU/LDS/f64 wave (the tree phase) run in the shadow of a dense MFMA
// wave on the same SIMD?  512 threads = 2 waves per SIMD: waves 0-3 stream weights and issue MFMAs (the fc1
// step of the fused kernel), waves 4-7 chase pointers through LDS with an f64 division per hop.
//   mode 0: MFMA waves only     mode 1: tree waves only     mode 2: both
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

__global__ __launch_bounds__(512, 1) void k(const f32x4 *w, float *out, unsigned long long *cyc, int iters, int hops,
                                            int mode) {
  __shared__ int s_next[4096];
  __shared__ double s_val[4096];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 4096; i += 512) { s_next[i] = (i * 1237 + 11) & 4095; s_val[i] = 1.0 + i * 1e-3; }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0;
  if (wv < 4) {
    if (mode != 1) {
      f32x4 acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
      const char *wb = (const char *)(w + (size_t)__builtin_amdgcn_readfirstlane(wv) * 4096);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)wb, 0, 32 * 4096, 0x00020000);
      const int lo = lane * 16;
      float x = out[lane];
#define LD(buf, it_)                                                                                               \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) buf[p] =                                                            \
      __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, ((((it_) & 31) * 4 + p) * 1024), 0));
      f32x4 a[4], b[4], c[4], d[4];
      LD(a, 0) LD(b, 1) LD(c, 2) LD(d, 3)
      for (int it = 0; it < iters; it += 4) {
        _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], a[t >> 2][t & 3], x);
        LD(a, it + 4)
        _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], b[t >> 2][t & 3], x);
        LD(b, it + 5)
        _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], c[t >> 2][t & 3], x);
        LD(c, it + 6)
        _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], d[t >> 2][t & 3], x);
        LD(d, it + 7)
      }
      for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (mode != 0) {
      int p = (tid * 7) & 4095;
      double v = 1.0, span = 3.0 + lane;
      for (int h = 0; h < hops; ++h) {
        const double q = s_val[p];
        p = s_next[p];
        v = (v + q) / span;                   // f64 division on the chain (MinMaxStats.normalize)
        v += __shfl_xor(v, 1, 16);            // a cross-lane step
      }
      r = (float)v + p;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[tid + blockIdx.x * 512] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

int main() {
  f32x4 *w; float *out; unsigned long long *cyc;
  hipMalloc(&w, 1 << 24); hipMemset(w, 0, 1 << 24);
  hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
  hipMalloc(&cyc, 2048 * 8);
  const int iters = 2000, hops = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, w, out, cyc, iters, hops, mode);
      hipDeviceSynchronize();
    }
    unsigned long long h[2048];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double sm = 0, st = 0;
    for (int b = 0; b < 256; ++b) {
      unsigned long long m = 0, t = 0;
      for (int w = 0; w < 4; ++w) m = h[b * 8 + w] > m ? h[b * 8 + w] : m;
      for (int w = 4; w < 8; ++w) t = h[b * 8 + w] > t ? h[b * 8 + w] : t;
      sm += (double)m; st += (double)t;
    }
    // s_memtime ticks at 100 MHz; report in ticks per unit and let the reader compare modes
    printf("mode %d: MFMA waves %.4f ticks per MFMA, tree waves %.4f ticks per hop\n", mode, sm / 256 / iters / 16,
           st / 256 / hops);
  }
  return 0;
}
