// which instruction classes of a second wave run in the shadow of a dense MFMA wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b))

template <int TM>
__global__ __launch_bounds__(512, 1) void k(const f32x4 *w, float *out, unsigned long long *cyc, int iters, int hops, int mode) {
  __shared__ f32x4 s_v[1024];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 1024; i += 512) s_v[i] = f32x4{1.f + i, 2.f, 3.f, 4.f};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0;
  if (wv < 4) {
    if (mode != 1) {
      f32x4 acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
      f32x4 a[4];
      for (int i = 0; i < 4; ++i) a[i] = w[i * 64 + lane];
      float x = out[lane];
      for (int it = 0; it < iters; ++it) {
        _Pragma("unroll") for (int t = 0; t < 16; ++t) MFMA(acc[t], a[t >> 2][t & 3], x);
      }
      for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (mode != 0) {
      float v = out[lane] + 1.5f;
      double d = v;
      f32x2 p2 = {v, v};
      f32x4 q = {v, v, v, v};
      if (TM >= 13) asm volatile("s_setprio 3");
      for (int h = 0; h < hops; ++h) {
        if (TM == 1) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v)); }
        if (TM == 2) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v)); }
        if (TM == 3) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p2)); }
        if (TM == 4) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d)); }
        if (TM == 5) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v)); }
        if (TM == 6) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((lane * 16) & 0x3ff0)); }
        if (TM == 7) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d)); }
        if (TM == 8) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v)); }
        if (TM == 9) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d)); }
        if (TM == 10) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d)); }
        if (TM == 12 || TM == 14) { asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3])); }
        if (TM == 13) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v)); }
        if (TM == 15) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d)); }
        if (TM == 11) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(v)); }
      }
      r = v + (float)d + p2[0] + q[0];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[tid + blockIdx.x * 512] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int TM> void run(const char *name, const f32x4 *w, float *out, unsigned long long *cyc) {
  const int iters = 2000, hops = 4000;
  double res[3][2];
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k<TM>, dim3(256), dim3(512), 0, 0, w, out, cyc, iters, hops, mode);
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
    res[mode][0] = sm / 256 / iters / 16; res[mode][1] = st / 256 / hops / 8;
  }
  printf("%-14s MFMA alone %.1f, with T %.1f cycles/MFMA;   T op alone %.1f, with M %.1f cycles/op\n", name, res[0][0], res[2][0], res[1][1], res[2][1]);
}

int main() {
  f32x4 *w; float *out; unsigned long long *cyc;
  hipMalloc(&w, 1 << 24); hipMemset(w, 0, 1 << 24);
  hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
  hipMalloc(&cyc, 2048 * 8);
  run<1>("v_exp_f32", w, out, cyc); run<2>("v_add_f32_dpp", w, out, cyc); run<3>("v_pk_add_f32", w, out, cyc);
  run<4>("v_fma_f64", w, out, cyc); run<5>("v_add_f32", w, out, cyc); run<6>("ds_read_b128", w, out, cyc);
  run<7>("v_rcp_f64", w, out, cyc); run<8>("v_sqrt_f32", w, out, cyc); run<9>("v_mul_f64", w, out, cyc);
  run<12>("4-way indep add", w, out, cyc); run<13>("add, setprio 3", w, out, cyc); run<14>("indep, setprio 3", w, out, cyc); run<15>("fma64, setprio3", w, out, cyc);
  return 0;
}
