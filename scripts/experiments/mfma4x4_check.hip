// layout and issue rate of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void layout(float *out) {
  const int l = threadIdx.x;
  const float a = 1.0f + l;            // A: lane l
  const float b = 100.0f * (1 + l);    // B: lane l
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ __launch_bounds__(256, 1) void rate(float *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = out[lane], b = out[64 + lane];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x + blockIdx.x * 256] = r;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMemset(out, 0, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  float h[256]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  // expectation: D[block = l/4][row r][col = l%4] = A[lane 4*block + r] * B[lane l]
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const float want = (1.0f + (4 * (l / 4) + r)) * (100.0f * (1 + l));
    if (h[l * 4 + r] != want) ++bad;
  }
  printf("layout D[lane l][vgpr r] = A[lane 4*(l/4)+r] * B[lane l]: %d mismatches (lane 5: %.0f %.0f %.0f %.0f)\n", bad, h[20], h[21], h[22], h[23]);
  const int iters = 4000;
  hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long hc[1024]; hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += (double)hc[i];
  printf("v_mfma_f32_4x4x1_16b_f32: %.2f cycles per instruction (8 independent accumulators)\n", s / 1024 / iters / 8);
  return 0;
}
