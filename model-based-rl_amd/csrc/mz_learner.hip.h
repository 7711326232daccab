// mz_learner.hip.h -- the elementwise ends of the learner step (reference learners.py:164-230) as single launches.  The
// step is launch-bound (a few hundred kernels of a few microseconds of work each): what PyTorch spells as ~45 tiny
// kernels for the targets and ~11 per categorical loss is one kernel each here.  The GEMMs stay rocBLAS / PyTorch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Config.scalar_transform (config.py:51-54): h(x) = sign(x) (sqrt(|x| + 1) - 1) + 0.001 x, float32 like torch's
__device__ __forceinline__ float mzl_scalar_transform(float x) {
  const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  return sg * (sqrtf(fabsf(x) + 1.f) - 1.f) + 0.001f * x;
}

// Config.scalar_to_support (config.py:56-68): two-hot projection of x, clamped to [lo, lo + S - 1], onto S integer bins
__device__ __forceinline__ void mzl_two_hot(float x, int lo, int S, float *out) {
  x = fminf(fmaxf(x, (float)lo), (float)(lo + S - 1));
  const float low = floorf(x), high = ceilf(x);
  const float p_high = x - low;
  for (int s = 0; s < S; ++s) out[s] = 0.f;
  out[(int)high - lo] = p_high;             // (scatter order of the reference: high first, then low -- x integral: low wins, 1)
  out[(int)low - lo] = 1.f - p_high;
}

// Learner targets of one batch (learners.py:176-189).  Threads [0, bs * K1): one per (sample, unroll position) --
//   sup_val [K1][bs][Sv], sup_rew [K1][bs][Sr] = two-hot(h(target)) (h skipped with no_target_transform), position-major.
// Threads [bs * K1, bs * K1 + 32 bs): 32 lanes per sample --
//   new_errors [bs] = inverse_transform(softmax expectation of value0 [bs][Sv]) - target_value[:, 0]
//   (Config.inverse_transform, config.py:27-33, float32).  t_val, t_rew [bs][K1] as sample_batch returns them.
__global__ void k_learner_targets(const float *t_val, const float *t_rew, const float *value0, int bs, int K1, int Sv,
                                  int vmin, int Sr, int rmin, int no_target_transform, float *sup_val, float *sup_rew,
                                  float *new_errors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < bs * K1) {
    const int b = i / K1, p = i % K1;
    float v = t_val[i], r = t_rew[i];
    if (!no_target_transform) { v = mzl_scalar_transform(v); r = mzl_scalar_transform(r); }
    mzl_two_hot(v, vmin, Sv, sup_val + ((size_t)p * bs + b) * Sv);
    mzl_two_hot(r, rmin, Sr, sup_rew + ((size_t)p * bs + b) * Sr);
    return;
  }
  const int j = i - ((bs * K1 + 31) & ~31);      // (the 32-lane groups start on a 32-lane boundary of the wavefronts)
  if (j < 0) return;
  const int b = j >> 5, lane = j & 31;
  if (b >= bs) return;
  const float *lg = value0 + (size_t)b * Sv;
  float mx = -__builtin_inff();
  for (int s = lane; s < Sv; s += 32) mx = fmaxf(mx, lg[s]);
  for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
  float sum = 0.f, ex = 0.f;
  for (int s = lane; s < Sv; s += 32) { const float e = expf(lg[s] - mx); sum += e; ex += e * (float)(vmin + s); }
  for (int o = 16; o >= 1; o >>= 1) { sum += __shfl_xor(sum, o, 32); ex += __shfl_xor(ex, o, 32); }
  if (lane == 0) {
    float x = ex / sum;
    if (!no_target_transform) {
      const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
      const float t = (sqrtf(1.f + 4.f * 0.001f * (fabsf(x) + 1.f + 0.001f)) - 1.f) / (2.f * 0.001f);
      x = sg * (t * t - 1.f);
    }
    new_errors[b] = x - t_val[(size_t)b * K1];
  }
}

// Soft cross-entropy of P (<= 8) positions, summed over the positions (learners.py:191-203 with utils.py:53-60:
// (-target * log_softmax(logits)).sum(1), added up over the unroll positions): logits [P][bs][S], target element (p, b, s) at
// target[p * tp_stride + b * tb_stride + s] -> loss [bs].  Eight adjacent lanes per sample, one per position; an
// 8-lane shuffle reduction adds the positions up.
__global__ void k_soft_ce_fwd(const float *logits, const float *target, int P, int bs, int S, int64_t tp_stride,
                              int64_t tb_stride, float *loss) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = i >> 3, p = i & 7;
  float l = 0.f;
  if (b < bs && p < P) {
    const float *x = logits + ((size_t)p * bs + b) * S;
    const float *t = target + p * tp_stride + b * tb_stride;
    float mx = -__builtin_inff();
    for (int s = 0; s < S; ++s) mx = fmaxf(mx, x[s]);
    float sum = 0.f;
    for (int s = 0; s < S; ++s) sum += expf(x[s] - mx);
    const float lse = mx + logf(sum);
    for (int s = 0; s < S; ++s) l += -t[s] * (x[s] - lse);
  }
  l += __shfl_xor(l, 1, 8); l += __shfl_xor(l, 2, 8); l += __shfl_xor(l, 4, 8);
  if (b < bs && p == 0) loss[b] = l;
}

// d loss[b] / d logits[p][b][s] = softmax(x)[s] * sum_s' t[s'] - t[s], times the upstream gradient of loss[b];
// 8 lanes per (position, sample) row, lane q takes the bins q, q + 8, ...
__global__ void k_soft_ce_bwd(const float *logits, const float *target, const float *gloss, int P, int bs, int S,
                              int64_t tp_stride, int64_t tb_stride, float *glogits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = i >> 3, q = i & 7;
  const bool live = row < P * bs;
  const int p = live ? row / bs : 0, b = live ? row % bs : 0;
  const float *x = logits + (size_t)(live ? row : 0) * S;
  const float *t = target + p * tp_stride + b * tb_stride;
  float mx = -__builtin_inff(), tsum = 0.f;
  for (int s = q; s < S; s += 8) { mx = fmaxf(mx, x[s]); tsum += t[s]; }
  for (int o = 4; o >= 1; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 8)); tsum += __shfl_xor(tsum, o, 8); }
  float sum = 0.f;
  for (int s = q; s < S; s += 8) sum += expf(x[s] - mx);
  for (int o = 4; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 8);
  if (!live) return;
  const float g = gloss[b];
  for (int s = q; s < S; s += 8) glogits[(size_t)row * S + s] = g * ((expf(x[s] - mx) / sum) * tsum - t[s]);
}
