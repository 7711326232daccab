// mz_net.hip.h -- fused FCNetwork inference on the f32 matrix cores (gfx950, v_mfma_f32_16x16x4_f32).
//
// One workgroup = 4 wavefronts (one per SIMD) = 16 rows (trees).  Every layer is evaluated transposed,
// D[n][m] = sum_k W[n][k] * X[k][m], with the WEIGHTS as the MFMA A operand and the 16 rows as the B
// operand (N = 16 columns), so that
//   * the D fragment of an fc1 tile (lane: column m = lane&15, features 16t + 4*(lane>>4) + r) is
//     already the B fragment of the following 512->J layer -- activations of the 512-wide hidden
//     layers never leave the register file (the k order of that second product is the permutation
//     k = 16t + 4*(lane>>4) + r; the weights are pre-packed in the same order);
//   * weights stream L2 -> VGPR once per workgroup in exactly the per-lane order the MFMAs consume
//     (one fully coalesced 1 KiB dwordx4 load = the A operands of four MFMAs), no LDS round trip;
//   * bias is the accumulator's initial value, ReLU is a v_max on the accumulator.
// The 512->J layers split K over the four waves (128 each) and combine through LDS; LayerNorm, the
// softmax-expectation over the support and the inverse value transform run on the combined tile.
// float32 throughout: v_mfma_f32_16x16x4_f32 is an exact f32 fma chain (needed for the 1e-5 bound).
//
// Reference: networks.py:146-174 (representation / prediction / dynamics / attach_action),
// config.py:27-33 (inverse_transform).
#pragma once
#include "mz_common.h"

#define MZ_XT_ROWS 128
struct NetSmem {
  float xT[MZ_XT_ROWS * 16];      // B operand tile, k-major: xT[k][m]
  float red[4 * 8 * 4 * 64];      // split-K partials [wave][tile][r][lane]
  float fin[128 * 16];            // combined outputs [n][m]
};

struct NetSink {
  float *h_base; size_t h_stride;   // hidden out: row m at h_base + m*h_stride (MZ_HS or MZ_H floats)
  int h_pad;                        // write zero padding up to MZ_HS
  float *reward, *value, *logits;   // row m at reward[m], value[m], logits[m*A]
  int rows;                         // valid rows in this tile (<= 16)
};

__device__ __forceinline__ f32x4 mz_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// fc1 stage: acc[t] (+)= W1 tile t of this wave  x  xT, over k-steps [s0, s0+cnt) of `ks` total.
// wp: [4 waves][NT/4][ks][64] f32x4 (element i = tile 4*tg+i), bp: [4][NT][64] f32x4.
template <int NT>
__device__ __forceinline__ void mz_fc1(const f32x4 *__restrict__ wp, const f32x4 *__restrict__ bp,
                                       const float *xT, int ks, int s0, int cnt, bool init, int w, int lane,
                                       f32x4 (&acc)[NT]) {
  constexpr int TG = NT / 4;
  const int g = lane >> 4, m = lane & 15;
  if (init) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = bp[(w * NT + t) * 64 + lane];
  }
  const f32x4 *wb = wp + (size_t)(w * TG) * ks * 64 + lane;
  f32x4 cur[TG], nxt[TG], nx2[TG];
  float xc, xn = 0.f, xn2 = 0.f;
#pragma unroll
  for (int tg = 0; tg < TG; ++tg) cur[tg] = wb[(size_t)(tg * ks + s0) * 64];
  xc = xT[(0 * 4 + g) * 16 + m];
  if (cnt > 1) {
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) nxt[tg] = wb[(size_t)(tg * ks + s0 + 1) * 64];
    xn = xT[(1 * 4 + g) * 16 + m];
  }
  for (int s = 0; s < cnt; ++s) {
    if (s + 2 < cnt) {
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) nx2[tg] = wb[(size_t)(tg * ks + s0 + s + 2) * 64];
      xn2 = xT[((s + 2) * 4 + g) * 16 + m];
    }
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
      acc[4 * tg + 0] = mz_mfma(cur[tg][0], xc, acc[4 * tg + 0]);
      acc[4 * tg + 1] = mz_mfma(cur[tg][1], xc, acc[4 * tg + 1]);
      acc[4 * tg + 2] = mz_mfma(cur[tg][2], xc, acc[4 * tg + 2]);
      acc[4 * tg + 3] = mz_mfma(cur[tg][3], xc, acc[4 * tg + 3]);
    }
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) { cur[tg] = nxt[tg]; nxt[tg] = nx2[tg]; }
    xc = xn; xn = xn2;
  }
}

template <int NT>
__device__ __forceinline__ void mz_relu(f32x4 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t][0] = fmaxf(acc[t][0], 0.f); acc[t][1] = fmaxf(acc[t][1], 0.f);
    acc[t][2] = fmaxf(acc[t][2], 0.f); acc[t][3] = fmaxf(acc[t][3], 0.f);
  }
}

// fc2 stage (512 -> 16*JT outputs), this wave's 128-wide K slice: the hidden tiles hid[OFF..OFF+7]
// are the B operands as they stand.  wp: [JT][4 waves][8 tiles][64] f32x4 (element r).
template <int JT, int OFF, int NT>
__device__ __forceinline__ void mz_fc2(const f32x4 *__restrict__ wp, const f32x4 (&hid)[NT], int w, int lane,
                                       f32x4 *out) {
  f32x4 cur[JT], nxt[JT];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    out[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    cur[jt] = wp[((jt * 4 + w) * 8 + 0) * 64 + lane];
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t + 1 < 8) {
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) nxt[jt] = wp[((jt * 4 + w) * 8 + t + 1) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) out[jt] = mz_mfma(cur[jt][r], hid[OFF + t][r], out[jt]);
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) cur[jt] = nxt[jt];
  }
}

// combine the four waves' split-K partials (+ bias) into fin[n][m]
template <int JTOT>
__device__ __forceinline__ void mz_combine(NetSmem &sm, const f32x4 *out, const float *bias, int tid) {
  const int w = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int jt = 0; jt < JTOT; ++jt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sm.red[((w * 8 + jt) * 4 + r) * 64 + lane] = out[jt][r];
  }
  __syncthreads();
  for (int e = tid; e < JTOT * 256; e += 256) {
    const int jt = e >> 8, r = (e >> 6) & 3, ln = e & 63;
    const int n = 16 * jt + 4 * (ln >> 4) + r;
    float s = bias[n];
    s += sm.red[((0 * 8 + jt) * 4 + r) * 64 + ln];
    s += sm.red[((1 * 8 + jt) * 4 + r) * 64 + ln];
    s += sm.red[((2 * 8 + jt) * 4 + r) * 64 + ln];
    s += sm.red[((3 * 8 + jt) * 4 + r) * 64 + ln];
    sm.fin[n * 16 + (ln & 15)] = s;
  }
  __syncthreads();
}

// Config.inverse_transform (config.py:27-33) for 16 columns, 4 lanes per column (one wavefront):
// softmax over S bins, expectation over the integer support, then the reference's float32 formula
// in its own operation order (its sqrt(...)-1 cancellation makes the result a staircase; the same
// order keeps the steps where the reference has them).
__device__ __forceinline__ float mz_support_to_scalar(const float *fin, int row0, int S, int smin,
                                                      int no_transform, int lane) {
  const int m = lane >> 2, q = lane & 3;
  float mx = -__builtin_inff();
  for (int i = q; i < S; i += 4) mx = fmaxf(mx, fin[(row0 + i) * 16 + m]);
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  float sum = 0.f;
  for (int i = q; i < S; i += 4) sum += expf(fin[(row0 + i) * 16 + m] - mx);
  sum += __shfl_xor(sum, 1);
  sum += __shfl_xor(sum, 2);
  float v = 0.f;
  for (int i = q; i < S; i += 4) v += (float)(smin + i) * (expf(fin[(row0 + i) * 16 + m] - mx) / sum);
  if (no_transform == 2) v = (q == 0) ? fin[row0 * 16 + m] : 0.f;   // --no_support: the head's single output
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  if (!no_transform) {
    const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    float t = (fabsf(v) + 1.f) + 0.001f;
    t = 1.f + 0.004f * t;
    t = (sqrtf(t) - 1.f) / 0.002f;
    v = sgn * (t * t - 1.f);
  }
  return v;
}

// F.relu(LayerNorm(50)) (networks.py:144-149,163-164) of fin rows [row0, row0+50) -> xT rows 0..49
// (+ zero rows 50,51), one wavefront, 4 lanes per column.
__device__ __forceinline__ void mz_ln_relu(NetSmem &sm, const NetView &n, int row0, int lane) {
  const int m = lane >> 2, q = lane & 3;
  float s = 0.f;
  for (int f = q; f < MZ_H; f += 4) s += sm.fin[(row0 + f) * 16 + m];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  const float mean = s / (float)MZ_H;
  float v = 0.f;
  for (int f = q; f < MZ_H; f += 4) { const float d = sm.fin[(row0 + f) * 16 + m] - mean; v += d * d; }
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  const float rstd = 1.0f / sqrtf(v / (float)MZ_H + 1e-5f);
  for (int f = q; f < MZ_HS; f += 4) {
    float y = 0.f;
    if (f < MZ_H) {
      y = (sm.fin[(row0 + f) * 16 + m] - mean) * rstd * n.lnw[f] + n.lnb[f];
      y = fmaxf(y, 0.f);
    }
    sm.xT[f * 16 + m] = y;
  }
}

// hidden tile in xT (rows 0..49) -> global rows
__device__ __forceinline__ void mz_store_hidden(const NetSmem &sm, const NetSink &o, int tid) {
  const int width = o.h_pad ? MZ_HS : MZ_H;
  for (int idx = tid; idx < 16 * width; idx += 256) {
    const int m = idx / width, k = idx % width;
    if (m < o.rows) o.h_base[(size_t)m * o.h_stride + k] = (k < MZ_H) ? sm.xT[k * 16 + m] : 0.f;
  }
}

// FCNetwork.prediction (networks.py:151-156) on the hidden tile in xT: value + policy logits.
template <int JTP>
__device__ __forceinline__ void mz_net_prediction(NetSmem &sm, const NetView &n, const NetSink &o, int tid) {
  const int w = tid >> 6, lane = tid & 63;
  f32x4 acc[16];
  mz_fc1<16>(n.w3, n.b3, sm.xT, n.ks3, 0, n.ks3, true, w, lane, acc);
  mz_relu<16>(acc);
  f32x4 out[2 + JTP];
  mz_fc2<2, 0, 16>(n.w4, acc, w, lane, out);
  mz_fc2<JTP, 8, 16>(n.w4 + 2 * 4 * 8 * 64, acc, w, lane, out + 2);
  mz_combine<2 + JTP>(sm, out, n.b4, tid);
  if (w == 0) {
    const float v = mz_support_to_scalar(sm.fin, 0, n.Sv, n.vmin, n.no_transform, lane);
    if ((lane & 3) == 0 && (lane >> 2) < o.rows) o.value[lane >> 2] = v;
  } else {
    for (int idx = tid - 64; idx < 16 * n.A; idx += 192) {
      const int m = idx / n.A, a = idx % n.A;
      if (m < o.rows) o.logits[(size_t)m * n.A + a] = sm.fin[(32 + a) * 16 + m];
    }
  }
}

// FCNetwork.dynamics (networks.py:158-165) on the [hidden | one-hot action] tile in xT, then
// prediction: BaseNetwork.recurrent_inference (networks.py:31-34).
template <int JTP>
__device__ __forceinline__ void mz_net_recurrent(NetSmem &sm, const NetView &n, const NetSink &o, int tid) {
  const int w = tid >> 6, lane = tid & 63;
  {
    f32x4 acc[16];
    mz_fc1<16>(n.w1, n.b1, sm.xT, n.ks1, 0, n.ks1, true, w, lane, acc);
    mz_relu<16>(acc);
    f32x4 out[6];
    mz_fc2<2, 0, 16>(n.w2, acc, w, lane, out);                       // reward support logits
    mz_fc2<4, 8, 16>(n.w2 + 2 * 4 * 8 * 64, acc, w, lane, out + 2);  // next hidden state (pre-LN)
    __syncthreads();                                                  // every wave is done reading xT
    mz_combine<6>(sm, out, n.b2, tid);
  }
  if (w == 0) {
    mz_ln_relu(sm, n, 32, lane);
  } else if (w == 1) {
    const float r = mz_support_to_scalar(sm.fin, 0, n.Sr, n.rmin, n.no_transform, lane);
    if ((lane & 3) == 0 && (lane >> 2) < o.rows) o.reward[lane >> 2] = r;
  }
  __syncthreads();
  mz_store_hidden(sm, o, tid);
  mz_net_prediction<JTP>(sm, n, o, tid);
}

// ------------------------------------------------------------------ kernels
// recurrent inference for the leaves the last descent selected: gathers parent hidden state +
// action per tree, writes hidden slot `out_slot` and the per-tree outputs in TreeView.
template <int JTP>
__global__ __launch_bounds__(256, 1) void k_net_recurrent_tree(NetView n, TreeView t, int out_slot) {
  __shared__ NetSmem sm;
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * MZ_ROWS;
  const size_t per_tree = (size_t)(t.sims + 1) * MZ_HS;
  for (int idx = tid; idx < 16 * MZ_H; idx += 256) {
    const int m = idx / MZ_H, k = idx % MZ_H;
    sm.xT[k * 16 + m] = t.hpool[(size_t)(b0 + m) * per_tree + (size_t)t.slot[b0 + m] * MZ_HS + k];
  }
  const int extra = n.ks1 * 4 - MZ_H;
  for (int idx = tid; idx < 16 * extra; idx += 256) {
    const int m = idx & 15, kk = idx >> 4;
    sm.xT[(MZ_H + kk) * 16 + m] = (kk == t.act[b0 + m]) ? 1.f : 0.f;
  }
  __syncthreads();
  NetSink o;
  o.h_base = t.hpool + (size_t)b0 * per_tree + (size_t)out_slot * MZ_HS;
  o.h_stride = per_tree; o.h_pad = 1;
  o.reward = t.reward + b0; o.value = t.value + b0; o.logits = t.logits + (size_t)b0 * t.A;
  o.rows = 16;
  mz_net_recurrent<JTP>(sm, n, o, tid);
}

// recurrent inference on caller-provided rows (mz_recurrent_inference)
template <int JTP>
__global__ __launch_bounds__(256, 1) void k_net_recurrent_rows(NetView n, const float *hin, const int32_t *act,
                                                                int nrows, float *hout, float *reward,
                                                                float *value, float *logits) {
  __shared__ NetSmem sm;
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * MZ_ROWS;
  const int rows = (nrows - b0) < 16 ? (nrows - b0) : 16;
  for (int idx = tid; idx < 16 * MZ_H; idx += 256) {
    const int m = idx / MZ_H, k = idx % MZ_H;
    sm.xT[k * 16 + m] = (m < rows) ? hin[(size_t)(b0 + m) * MZ_H + k] : 0.f;
  }
  const int extra = n.ks1 * 4 - MZ_H;
  for (int idx = tid; idx < 16 * extra; idx += 256) {
    const int m = idx & 15, kk = idx >> 4;
    sm.xT[(MZ_H + kk) * 16 + m] = (m < rows && kk == act[b0 + m]) ? 1.f : 0.f;
  }
  __syncthreads();
  NetSink o;
  o.h_base = hout + (size_t)b0 * MZ_H; o.h_stride = MZ_H; o.h_pad = 0;
  o.reward = reward + b0; o.value = value + b0; o.logits = logits + (size_t)b0 * n.A;
  o.rows = rows;
  mz_net_recurrent<JTP>(sm, n, o, tid);
}

// packed[i] = idx[i] >= 0 ? flat[idx[i]] : 0   (weights -> MFMA operand order)
// idx: -1 = padding (0.0), else bits 0-28 the source element, bits 29-30 its scale class (scale[cls], powers of two:
// exact; mz_engine.hip k_relu_scale); idx2: -1 or a second element added to the first (one float32 addition)
static __global__ void k_pack_weights(const float *flat, const int32_t *idx, const int32_t *idx2, float *packed, size_t n,
                                      const float *scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t s = idx[i];
  if (s < 0) { packed[i] = 0.f; return; }
  float v = flat[s & 0x1fffffff];
  const int32_t s2 = idx2[i];
  if (s2 >= 0) v += flat[s2];
  packed[i] = v * scale[(s >> 29) & 3];
}
