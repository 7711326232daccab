// mz_fcl.hip.h -- the FCNetwork learner step (reference learners.py:164-230, networks.py:135-180, config.py:27-33,51-68,
// utils.py:53-60) as FIVE launches (six with gradient clipping): the whole update -- K-step unroll forward, the three heads'
// losses, backward, clipping, AdamW -- without a GEMM library or an autograd tape.  The step is ~1.4 GFLOP (batch 256,
// K = 5): as PyTorch operators it is ~220 kernels of ~4 us of launch floor each; here its shape follows the data
// dependences instead:
//
//   k_fcl_chain_fwd4  one workgroup per 4 samples: h_0 = representation(obs), h_p = dynamics(h_{p-1}, a_{p-1}) -- the only
//                     sequential part -- on v_mfma_f32_4x4x1_16b_f32 (64 output rows x 4 samples per instruction), the
//                     transition's weights resident in registers across positions, activations in LDS
//   k_fcl_heads       one workgroup per (16 samples = the 16 columns of v_mfma_f32_16x16x4_f32, unroll position, head):
//                     value / policy / reward head forward, two-hot targets, soft cross-entropy, and the head's backward
//                     down to d loss / d hidden state
//   k_fcl_chain_bwd4  one workgroup per 4 samples: the chain backwards (gradient hooks 0.5, LayerNorm, ReLU)
//   k_fcl_dw          every weight gradient dW = sum_rows delta (x) input as 16 x 64 MFMA strips over the activation /
//                     delta tapes the three kernels above left in HBM (four waves per strip and unroll position, summed
//                     in a fixed order: deterministic, no atomics)
//   k_fcl_grad        (only with clip_grad) adds the per-position strips up, squares for the global norm
//   k_fcl_adam        (adds the strips up,) clip_grad_norm_, Adam / AdamW (torch's fused-kernel arithmetic), the new weights
//                     into the flat vector AND into the packed copies the next step's MFMAs read; loss sums
//
// Layouts.  Tapes in HBM: [position][row chunk of 16][feature][16 rows]: a slice kernel's workgroup owns one contiguous
// block (a quarter of it: the chain kernels) per position, and k_fcl_dw reads 16 features x 16 rows as ONE contiguous KiB
// per wave and load.  Heads kernel: LDS activations in "k-step layout" -- element (feature f, sample n) at
// ((f >> 2) * 16 + n) * 4 + (f & 3): the B operand of k-step s is one conflict-free ds_read_b32 (s * 64 + 4 n + k), a D
// fragment one ds_write_b128; packed weights P(W; M, K): [ceil(M / 64)][k-steps][64 lanes] f32x4, component i of lane
// (m16, g4) of k-step s of group tg = W[64 tg + 16 i + m16][4 s + g4]: one 16-byte load feeds four MFMAs.  Chain kernels:
// LDS activations sample-major, packed "quad" copies Q(W; M, K): [M / 64][K][64 lanes] floats (see there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FCL_NW 8                    // waves per workgroup of the chain and heads kernels
#define FCL_THREADS (FCL_NW * 64)
#define FCL_MAXP 8                  // unroll positions K + 1
#define FCL_LN_EPS 1e-5f

struct FclPack {          // float offsets into the packed buffer
  size_t F1, F2, B2, B1;
  int ks1, ks2;           // heads: k-steps (even, zero-padded) of fc1 forward / fc2 backward; chain (quad copies): K of fc1 / fc2^T
  int mout, nt;           // fc2 output features, 16-row tiles of them
};

struct FclView {
  int bs, K, O, A, KD, Sv, Sr, vmin, rmin, ntt, R, XR, xks, xq;
  const float *P;           // flat parameters (engine.WEIGHT_ORDER)
  const float *pk;          // packed copies
  size_t rep_b1, rep_b2, tr_b1, tr_b2, ln_w, ln_b, hb1[3], hb2[3];      // bias offsets in P (heads: value, policy, reward)
  FclPack rep, tr, head[3];
  const float *obs; const void *act; int act_i32; const float *t_rew, *t_val, *t_pol; const void *w; int w_f64;
  float *xin, *a1c, *xhat, *rstd, *h, *d2c, *d1c;      // chain tapes, [K + 1][...][R]
  float *a1h, *d2h, *d1h, *dH, *lossb;                 // head tapes, [3][K + 1][...][R]
  float *lnpart;                                        // [bs / 4][128] LayerNorm weight / bias gradient partials (per chain workgroup)
  float *new_errors;
  unsigned long long *prof;      // development: s_memtime stamps of k_fcl_heads' phases (workgroup 0 of every head at position 1), else null
};

// workgroup barrier that waits for this wave's LDS traffic only: the tapes are write-only inside a kernel, so a barrier need
// not wait for the global stores in flight (__syncthreads() would: a microsecond per barrier behind every tape write)
__device__ __forceinline__ void fcl_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// tape element (feature f of F, sample n of the 16 of workgroup / row chunk c): [chunk][feature][16]
__device__ __forceinline__ size_t fcl_tp(int F, int c, int f, int n) { return ((size_t)c * F + f) * 16 + n; }

__device__ __forceinline__ int fcl_at(int f, int n) { return ((f >> 2) * 16 + n) * 4 + (f & 3); }

__device__ __forceinline__ f32x4 fcl_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// k-steps allocated per 64-row group of a wide pack: whole groups of 8 (fcl_wide requests GS k-steps at a time, GS | 8)
#define FCL_KSA(ks) (((ks) + 7) & ~7)

// out[512 x 16] = W[512 x 4 ks] . X[4 ks x 16]: wave w owns the 64 output features [64 w, 64 w + 64) (tiles i = 0..3: rows
// 64 w + 16 i + 4 g4 + r, column m16).  pk: P(W; 512, 4 ks), FCL_KSA(ks) k-steps per group.  X in k-step layout.  The
// weights stream from L2 GS k-steps per request (GS divides 8), one request ahead of the MFMAs.
template <int GS>
__device__ __forceinline__ void fcl_wide(const f32x4 *__restrict__ pk, int ks, const float *X, int w, int lane, f32x4 acc[4]) {
  const int ng = (ks + GS - 1) / GS;
  const f32x4 *p = pk + (size_t)w * FCL_KSA(ks) * 64 + lane;
  const float *x = X + 4 * (lane & 15) + (lane >> 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 c[GS];
#pragma unroll
  for (int j = 0; j < GS; ++j) c[j] = p[j * 64];
  for (int g = 0; g < ng; ++g) {
    f32x4 n[GS];
    if (g + 1 < ng) {
#pragma unroll
      for (int j = 0; j < GS; ++j) n[j] = p[((g + 1) * GS + j) * 64];
    }
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (GS * g + j < ks) {
        const float xs = x[(GS * g + j) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fcl_mfma(c[j][i], xs, acc[i]);
      }
    }
    if (g + 1 < ng) {
#pragma unroll
      for (int j = 0; j < GS; ++j) c[j] = n[j];
    }
  }
}

// partial of out[16 NT x 16] = W[16 NT x 512] . A1[512 x 16] over this wave's 16 k-steps (split-K over the 8 waves);
// pk: P(W; <= 64, 512), 128 k-steps.  The partials go to red[w][4][64] (f32x4); fcl_reduce adds them up.
__device__ __forceinline__ void fcl_load_narrow(f32x4 (&W)[16], const f32x4 *__restrict__ pk, int w, int lane) {
  const f32x4 *p = pk + (size_t)(16 * w) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 16; ++s) W[s] = p[s * 64];
}

template <int NT>
__device__ __forceinline__ void fcl_narrow_res(const f32x4 (&W)[16], const float *A1, f32x4 *red, int w, int lane) {
  const float *x = A1 + (16 * w) * 64 + 4 * (lane & 15) + (lane >> 4);
  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float xs = x[s * 64];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = fcl_mfma(W[s][i], xs, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) red[(w * 4 + i) * 64 + lane] = acc[i];
}

template <int NT>
__device__ __forceinline__ void fcl_narrow(const f32x4 *__restrict__ pk, const float *A1, f32x4 *red, int w, int lane) {
  f32x4 c[16];
  fcl_load_narrow(c, pk, w, lane);
  fcl_narrow_res<NT>(c, A1, red, w, lane);
}

__device__ __forceinline__ void fcl_narrow_nt(int nt, const f32x4 *pk, const float *A1, f32x4 *red, int w, int lane) {
  if (nt == 1) fcl_narrow<1>(pk, A1, red, w, lane);
  else if (nt == 2) fcl_narrow<2>(pk, A1, red, w, lane);
  else fcl_narrow<4>(pk, A1, red, w, lane);
}

// Y (k-step layout, 64 features) = sum of the 8 partials + bias (rows >= mreal: 0).  Call between two barriers.
__device__ __forceinline__ void fcl_reduce(const f32x4 *red, int nt, const float *bias, int mreal, float *Y, int tid) {
  if (tid < 256) {
    const int i = tid >> 6, l = tid & 63;
    f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < nt) {
#pragma unroll
      for (int w = 0; w < FCL_NW; ++w) y += red[(w * 4 + i) * 64 + l];
      const int f0 = 16 * i + 4 * (l >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = (f0 + r < mreal) ? y[r] + (bias ? bias[f0 + r] : 0.f) : 0.f;
    }
    *(f32x4 *)(Y + ((4 * i + (l >> 4)) * 16 + (l & 15)) * 4) = y;
  }
}

// LDS of the heads kernel (floats): X [1024] | A1 [8192] | red [8192] | Y [1024] | S [1024] | PV [576] (PV: the small
// parameter vectors, read once per launch): two workgroups per CU
#define FCL_LDS_HEADS (1024 + 8192 + 8192 + 1024 + 1024 + 576)

// Config.scalar_transform + scalar_to_support (config.py:51-68): bin s of the two-hot target of scalar x
struct FclTwoHot { int lo_i, hi_i; float p_hi; };
__device__ __forceinline__ FclTwoHot fcl_two_hot(float x, int lo, int S, int ntt) {
  if (!ntt) x = mzl_scalar_transform(x);
  x = fminf(fmaxf(x, (float)lo), (float)(lo + S - 1));
  const float low = floorf(x), high = ceilf(x);
  FclTwoHot t;
  t.p_hi = x - low; t.lo_i = (int)low - lo; t.hi_i = (int)high - lo;
  return t;
}
__device__ __forceinline__ float fcl_two_hot_at(const FclTwoHot &t, int s) {
  float v = (s == t.hi_i) ? t.p_hi : 0.f;
  if (s == t.lo_i) v = 1.f - t.p_hi;          // (the reference scatters high first, then low: an integral x ends as 1)
  return v;
}

// Tape addressing of a D fragment (rows 64 w + 16 i + 4 g4 + r, column = sample m16): a wave-uniform row base (scalar
// registers) + ONE per-lane offset, so that no per-(i, r) 64-bit address lives in vector registers across the positions
__device__ __forceinline__ int fcl_lane_off(int lane) { return 64 * (lane >> 4) + (lane & 15); }

// fc1 epilogue: bias (LDS), ReLU, the activations to LDS (k-step layout) and to the tape
__device__ __forceinline__ void fcl_fc1_out(const f32x4 acc[4], const float *b1, float *A1, float *a1t, int loff, int w, int lane) {
  const int g4 = lane >> 4, m16 = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f0 = 64 * w + 16 * i + 4 * g4;
    const f32x4 b = *(const f32x4 *)(b1 + f0);
    f32x4 a;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[r] = fmaxf(acc[i][r] + b[r], 0.f);
      (a1t + (64 * w + 16 * i + r) * 16)[loff] = a[r];
    }
    *(f32x4 *)(A1 + ((f0 >> 2) * 16 + m16) * 4) = a;
  }
}

// d a1 through the ReLU (mask: the activation a > 0), to LDS and to the delta tape
__device__ __forceinline__ void fcl_mask_out(const f32x4 acc[4], const f32x4 msk[4], float *A1, float *d1t, int loff, int w, int lane) {
  const int g4 = lane >> 4, m16 = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f0 = 64 * w + 16 * i + 4 * g4;
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[r] = msk[i][r] > 0.f ? acc[i][r] : 0.f;
      (d1t + (64 * w + 16 * i + r) * 16)[loff] = d[r];
    }
    *(f32x4 *)(A1 + ((f0 >> 2) * 16 + m16) * 4) = d;
  }
}

// ------------------------------------------------------------------------------------------------ chain, forward
// Weights of one layer in registers, requested a phase ahead of their use (scripts/fcl_heads_phases.py: streamed inside the
// phase, the four layers' first requests were four exposed L2 round trips of ~1.5 k cycles in a 17 us workgroup).  Up to 16
// k-steps; the guards are wave-uniform.
__device__ __forceinline__ void fcl_req_wide(f32x4 (&W)[16], const f32x4 *__restrict__ pk, int ks, int w, int lane) {
  const f32x4 *p = pk + (size_t)w * FCL_KSA(ks) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 16; ++s)
    if (s < ks) W[s] = p[s * 64];
}
__device__ __forceinline__ void fcl_wide_regs(const f32x4 (&W)[16], int ks, const float *X, int lane, f32x4 acc[4]) {
  const float *x = X + 4 * (lane & 15) + (lane >> 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s < ks) {
      const float xs = x[s * 64];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fcl_mfma(W[s][i], xs, acc[i]);
    }
  }
}

// sum / max over the 32 lanes of a sample: four DPP exchanges inside the 16-lane rows and one cross-row shuffle (a
// ds_bpermute per step, as __shfl_xor compiles to, is a dependent LDS round trip each)
template <int OFF> __device__ __forceinline__ float fcl_xchg(float x) { return __int_as_float(mz_xchg_i<OFF>(__float_as_int(x))); }
__device__ __forceinline__ float fcl_sum32(float x) {
  x += fcl_xchg<1>(x); x += fcl_xchg<2>(x); x += fcl_xchg<4>(x); x += fcl_xchg<8>(x);
  return x + __shfl_xor(x, 16, 32);
}
// sum over the 16 lanes of a wave that share lane & 3 (the chain kernels' sample j): lane bits 2..5 -- two row rotations
// (DPP row_ror:4, row_ror:8: the four lanes of a 16-lane row with the same lane & 3), then the rows by shuffle
__device__ __forceinline__ float fcl_sum_bits2to5(float x) {
  x += __int_as_float(mz_dpp_i<0x124>(__float_as_int(x)));
  x += __int_as_float(mz_dpp_i<0x128>(__float_as_int(x)));
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}
__device__ __forceinline__ float fcl_max32(float x) {
  x = fmaxf(x, fcl_xchg<1>(x)); x = fmaxf(x, fcl_xchg<2>(x)); x = fmaxf(x, fcl_xchg<4>(x)); x = fmaxf(x, fcl_xchg<8>(x));
  return fmaxf(x, __shfl_xor(x, 16, 32));
}

// ------------------------------------------------------------------------------------------------ heads
// grid (bs / 16, K + 1, 3): head 0 value (input h_p), 1 policy (h_p), 2 reward (x_p = [h_{p-1} | one-hot], p >= 1)
__global__ __launch_bounds__(FCL_THREADS, 4) void k_fcl_heads(FclView v) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  float *X = fcl_smem, *A1 = X + 1024, *redf = A1 + 8192, *Y = redf + 8192, *S = Y + 1024, *PV = S + 1024;
  f32x4 *red = (f32x4 *)redf;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g4 = lane >> 4, m16 = lane & 15;
  const int row0 = blockIdx.x * 16, p = blockIdx.y, hd = blockIdx.z, R = v.R, K1 = v.K + 1, loff = fcl_lane_off(lane), cb = blockIdx.x;
  if (hd == 2 && p == 0) return;
  int stamp_i = 0;
#define FCL_STAMP() if (v.prof && blockIdx.x == 0 && p == 1 && tid == 0) v.prof[hd * 16 + stamp_i++] = __builtin_amdgcn_s_memtime();
  FCL_STAMP()
  const size_t T64 = (size_t)64 * R, T512 = (size_t)512 * R, TX = (size_t)v.XR * R;
  const FclPack &pk = v.head[hd];
  const size_t hp = (size_t)hd * K1 + p;
  // the layers' weights pass through ONE set of 16 registers per lane, each layer's requested as soon as the previous
  // layer's products have been issued (the compiler barriers keep the requests from being hoisted above those products,
  // which would need a second register set: 128 registers is what two workgroups per CU leave); fc1's now
  f32x4 WA[16];
  fcl_req_wide(WA, (const f32x4 *)(v.pk + pk.F1), pk.ks1, w, lane);
  const float *src = hd == 2 ? v.xin + (size_t)p * TX : v.h + (size_t)p * T64;
  for (int idx = tid; idx < 64 * 16; idx += FCL_THREADS) {
    const int f = idx >> 4, n = idx & 15;
    X[fcl_at(f, n)] = src[fcl_tp(hd == 2 ? v.XR : 64, cb, f, n)];
  }
  PV[tid] = v.P[v.hb1[hd] + tid];
  if (tid < 64) PV[512 + tid] = tid < pk.mout ? v.P[v.hb2[hd] + tid] : 0.f;
  // this sample's targets and weight: needed after the two forward layers, requested now
  const int n_s = tid >> 5, q_s = tid & 31, row_s = row0 + n_s, M = pk.mout;
  const bool in0 = q_s < M, in1 = q_s + 32 < M;
  float t0 = 0.f, t1 = 0.f, ts = 0.f;
  if (hd == 1) {
    const float *tp = v.t_pol + ((size_t)row_s * K1 + p) * v.A;
    t0 = in0 ? tp[q_s] : 0.f; t1 = in1 ? tp[q_s + 32] : 0.f;
  } else {
    ts = (hd == 0 ? v.t_val : v.t_rew)[(size_t)row_s * K1 + p];
  }
  const double wb = v.w_f64 ? ((const double *)v.w)[row_s] : (double)((const float *)v.w)[row_s];
  const float tv0 = v.t_val[(size_t)row_s * K1];
  fcl_bar();
  FCL_STAMP()      // 1: inputs in LDS
  f32x4 acc[4];
  fcl_wide_regs(WA, pk.ks1, X, lane, acc);
  asm volatile("" ::: "memory");
  fcl_load_narrow(WA, (const f32x4 *)(v.pk + pk.F2), w, lane);          // fc2's, under the epilogue
  FCL_STAMP()      // 2: fc1 products
  fcl_fc1_out(acc, PV, A1, v.a1h + hp * T512 + fcl_tp(512, cb, 0, 0), loff, w, lane);
  fcl_bar();
  FCL_STAMP()      // 3: fc1 epilogue + barrier
  if (pk.nt == 1) fcl_narrow_res<1>(WA, A1, red, w, lane);
  else if (pk.nt == 2) fcl_narrow_res<2>(WA, A1, red, w, lane);
  else fcl_narrow_res<4>(WA, A1, red, w, lane);
  asm volatile("" ::: "memory");
  fcl_req_wide(WA, (const f32x4 *)(v.pk + pk.B2), pk.ks2, w, lane);     // fc2-transposed's, under the reduce and the loss
  fcl_bar();
  FCL_STAMP()      // 4: fc2 partials + barrier
  fcl_reduce(red, pk.nt, PV + 512, pk.mout, Y, tid);
  fcl_bar();
  FCL_STAMP()      // 5: reduce + barrier
  // soft cross-entropy against the categorical target (utils.py:53-60; learners.py:186-203) and its gradient, 32 lanes per
  // sample, bins q and q + 32; the gradient of the weighted mean and the 1 / K hook (learners.py:205-212) ride in g
  {
    const int n = n_s, q = q_s, row = row_s;
    const float x0 = in0 ? Y[fcl_at(q, n)] : -__builtin_inff(), x1 = in1 ? Y[fcl_at(q + 32, n)] : -__builtin_inff();
    if (hd != 1) {
      const FclTwoHot th = fcl_two_hot(ts, hd == 0 ? v.vmin : v.rmin, M, v.ntt);
      t0 = in0 ? fcl_two_hot_at(th, q) : 0.f; t1 = in1 ? fcl_two_hot_at(th, q + 32) : 0.f;
    }
    const float mx = fcl_max32(fmaxf(x0, x1));
    const float e0 = in0 ? expf(x0 - mx) : 0.f, e1 = in1 ? expf(x1 - mx) : 0.f;
    const float sum = fcl_sum32(e0 + e1), tsum = fcl_sum32(t0 + t1);
    const float ex = fcl_sum32(e0 * (float)(v.vmin + q) + e1 * (float)(v.vmin + q + 32));
    const float lse = mx + logf(sum);
    const float l = fcl_sum32((in0 ? -t0 * (x0 - lse) : 0.f) + (in1 ? -t1 * (x1 - lse) : 0.f));
    const float g = (float)(((1.0 / (double)v.K) / (double)v.bs) * wb);
    S[fcl_at(q, n)] = in0 ? g * ((e0 / sum) * tsum - t0) : 0.f;
    S[fcl_at(q + 32, n)] = in1 ? g * ((e1 / sum) * tsum - t1) : 0.f;
    if (q == 0) {
      v.lossb[hp * R + row] = l;
      if (hd == 0 && p == 0) {          // the priority refresh (learners.py:181-182; Config.inverse_transform, config.py:27-33)
        float xs = ex / sum;
        if (!v.ntt) {
          const float sg = (xs > 0.f) ? 1.f : ((xs < 0.f) ? -1.f : 0.f);
          const float tt = (sqrtf(1.f + 4.f * 0.001f * (fabsf(xs) + 1.f + 0.001f)) - 1.f) / (2.f * 0.001f);
          xs = sg * (tt * tt - 1.f);
        }
        v.new_errors[row] = xs - tv0;
      }
    }
  }
  fcl_bar();
  FCL_STAMP()      // 6: loss + barrier
  for (int idx = tid; idx < 64 * 16; idx += FCL_THREADS) {
    const int f = idx >> 4, n = idx & 15;
    v.d2h[hp * T64 + fcl_tp(64, cb, f, n)] = S[fcl_at(f, n)];
  }
  // backward: d a1 = W2^T d logits, through the ReLU; then d x = W1^T d a1 (its first 50 features: d hidden state)
  fcl_wide_regs(WA, pk.ks2, S, lane, acc);
  asm volatile("" ::: "memory");
  fcl_load_narrow(WA, (const f32x4 *)(v.pk + pk.B1), w, lane);          // fc1-transposed's, under the mask
  FCL_STAMP()      // 7: d2 tape + fc2-transposed products
  {
    f32x4 msk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) msk[i] = *(const f32x4 *)(A1 + (((64 * w + 16 * i + 4 * g4) >> 2) * 16 + m16) * 4);
    fcl_mask_out(acc, msk, A1, v.d1h + hp * T512 + fcl_tp(512, cb, 0, 0), loff, w, lane);
  }
  fcl_bar();
  FCL_STAMP()      // 8: mask + barrier
  fcl_narrow_res<4>(WA, A1, red, w, lane);
  fcl_bar();
  FCL_STAMP()      // 9: fc1-transposed partials + barrier
  fcl_reduce(red, 4, nullptr, MZ_H, Y, tid);
  fcl_bar();
  FCL_STAMP()      // 10: reduce + barrier
  for (int idx = tid; idx < 64 * 16; idx += FCL_THREADS) {
    const int f = idx >> 4, n = idx & 15;
    v.dH[hp * T64 + fcl_tp(64, cb, f, n)] = Y[fcl_at(f, n)];
  }
  FCL_STAMP()        // 11: d hidden stored
#undef FCL_STAMP
}

// ------------------------------------------------------------------------------------------------ chain (4 samples per workgroup)
// The chain is the step's only sequential part.  With 16 samples per workgroup (the 16 x 16 x 4 MFMA's columns, as the
// heads kernel does) it had batch / 16 workgroups: 16 of 256 CUs at batch 256, 42 + 37 us.  v_mfma_f32_4x4x1_16b_f32 (sixteen 4 x 4 outer products per instruction: 64 output rows x 4 columns x 1 k) has the
// same arithmetic rate per instruction byte as the 16 x 16 x 4 shape but needs only FOUR samples to fill its columns:
// batch / 4 workgroups, a quarter of the matrix time each.  Operands: A = lane l's weight of output row l (of the wave's
// 64) for ONE k -- packed "quad" copies Q(W; M, K): [M / 64][K][64 lanes] floats, resident in registers across positions
// like the 16-column version's; B = X[k][sample lane & 3] (LDS, [feature][4]); D register i of lane (b, j) = row 4 b + i,
// sample j.  Tapes as before: a workgroup owns 4 of the 16 rows of its row chunk.
__device__ __forceinline__ f32x4 fcl_mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// out[64 rows of wave w][4] = sum_k W[k] x X[sample][k]: xrow = this lane's sample row (LDS, sample-major, 16-byte aligned,
// at least K rounded up to 4 floats): the whole row is requested in 16-byte reads before the first MFMA (left to
// itself the compiler read one k at a time and waited out an LDS round trip in front of every pair of MFMAs); four
// accumulators in rotation (a 2-pass MFMA's result is not forwarded to an immediately following dependent one)
template <int K>
__device__ __forceinline__ f32x4 fcl_quad_res(const float (&W)[K], const float *xrow) {
  constexpr int NV = (K + 3) / 4;
  f32x4 xv[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) xv[q] = *(const f32x4 *)(xrow + 4 * q);
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k & 3] = fcl_mfma4(W[k], xv[k >> 2][k & 3], acc[k & 3]);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <int K>
__device__ __forceinline__ void fcl_quad_load(float (&W)[K], const float *__restrict__ pk, int lane) {
#pragma unroll
  for (int k = 0; k < K; ++k) W[k] = pk[k * 64 + lane];
}

// the compiler's wait-count bookkeeping treats a register whose load was requested before a loop as pending inside the
// loop and waits at its first use in EVERY iteration -- and a wait for a load also waits for every older store.  Touching
// the registers once in front of the loop settles them there.
template <int K>
__device__ __forceinline__ void fcl_quad_settle(float (&W)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) asm volatile("" : "+v"(W[k]));
}

// the same product with the weights streamed (the representation's fc1: K = observation features, used once)
__device__ __forceinline__ f32x4 fcl_quad_stream(const float *__restrict__ pk, int K, const float *xrow, int lane) {
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = pk[(k + j) * 64 + lane];
    const f32x4 x0 = *(const f32x4 *)(xrow + k), x1 = *(const f32x4 *)(xrow + k + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fcl_mfma4(wv[j], x0[j], acc[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fcl_mfma4(wv[4 + j], x1[j], acc[j]);
  }
  for (; k < K; ++k) acc[k & 3] = fcl_mfma4(pk[k * 64 + lane], xrow[k], acc[k & 3]);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// LDS of the 4-sample chain kernels (floats), activations SAMPLE-major: X [4][xq + 4] | A1 [4][516] | red [2048] | misc [16] | PV [1408] | S [256]
#define FCL_LDS4_FLOATS(xq) (4 * ((xq) + 4) + 4 * 516 + 2048 + 16 + 1408 + 256)
#define FCL_LDA 516

// D fragment (rows 64 w + 4 b + i, sample j) -> tape [chunk][feature][16]: wave-uniform base + one lane offset + i * 16
__device__ __forceinline__ int fcl_lane_off4(int lane, int n0) { return (lane >> 2) * 64 + n0 + (lane & 3); }

// KP: the transition's fc1 input features (50 + actions), padded: 56 (up to 6 actions) or 64
template <int KP>
__global__ __launch_bounds__(FCL_THREADS) void k_fcl_chain_fwd4(FclView v) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  const int LDX = v.xq + 4;
  float *X = fcl_smem, *A1 = X + 4 * LDX, *red = A1 + 4 * FCL_LDA, *misc = red + 2048, *PV = misc + 16;
  float *b1r = PV, *b1t = PV + 512, *b2r = PV + 1024, *b2t = PV + 1088, *lnw = PV + 1152, *lnb = PV + 1216;
  int *acts = (int *)(PV + 1280);          // [4 samples][8]
  float *S = PV + 1408;                    // [4 samples][64] x-hat of the position just finished (for the tapes)
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int cb = blockIdx.x >> 2, n0 = 4 * (blockIdx.x & 3), row0 = blockIdx.x * 4, R = v.R, loff = fcl_lane_off4(lane, n0);
  const size_t T64 = (size_t)64 * R, T512 = (size_t)512 * R, TX = (size_t)v.XR * R;
#define FCL_KSTAMP(k) if (v.prof && blockIdx.x == 0 && tid == 0) v.prof[54 + (k)] = __builtin_amdgcn_s_memtime();
  FCL_KSTAMP(0)      // kernel start
  float WT1[KP], WT2[64];
  fcl_quad_load<KP>(WT1, v.pk + v.tr.F1 + (size_t)w * KP * 64, lane);
  fcl_quad_load<64>(WT2, v.pk + v.tr.F2 + (size_t)w * 64 * 64, lane);
  b1r[tid] = v.P[v.rep_b1 + tid]; b1t[tid] = v.P[v.tr_b1 + tid];
  if (tid < 64) {
    b2r[tid] = tid < MZ_H ? v.P[v.rep_b2 + tid] : 0.f; b2t[tid] = tid < MZ_H ? v.P[v.tr_b2 + tid] : 0.f;
    lnw[tid] = tid < MZ_H ? v.P[v.ln_w + tid] : 0.f; lnb[tid] = tid < MZ_H ? v.P[v.ln_b + tid] : 0.f;
  }
  if (tid < 32) {
    const size_t ai = (size_t)(row0 + (tid >> 3)) * v.K + (tid & 7);
    acts[tid] = (tid & 7) < v.K ? (v.act_i32 ? ((const int32_t *)v.act)[ai] : (int)((const int64_t *)v.act)[ai]) : -1;
  }
  for (int idx = tid; idx < v.xq * 4; idx += FCL_THREADS) {
    const int f = idx >> 2, n = idx & 3;
    const float val = f < v.O ? v.obs[(size_t)(row0 + n) * v.O + f] : 0.f;
    X[n * LDX + f] = val;
    if (f < v.XR) v.xin[fcl_tp(v.XR, cb, f, n0 + n)] = val;
  }
  fcl_bar();
  FCL_KSTAMP(1)      // requests out, observations in LDS
  // (development: stamps of position 2's phases in workgroup 0, mz_fcl_heads_profile, slots 48..)
#define FCL_CSTAMP(k) if (v.prof && blockIdx.x == 0 && p == 2 && tid == 0) v.prof[48 + (k)] = __builtin_amdgcn_s_memtime();
  auto rest = [&](int p, f32x4 acc, const float (&W2)[64], const float *b1, const float *b2) __attribute__((always_inline)) {
    FCL_CSTAMP(1)      // fc1 products done
    {   // fc1 epilogue: bias, ReLU, to LDS (one 16-byte write: the lane's four features of its sample) and to the tape
      const int f0 = 64 * w + 4 * (lane >> 2);
      const f32x4 b = *(const f32x4 *)(b1 + f0);
      float *a1t = v.a1c + (size_t)p * T512 + fcl_tp(512, cb, 64 * w, 0);
      f32x4 a;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = fmaxf(acc[i] + b[i], 0.f);
        (a1t + i * 16)[loff] = a[i];
      }
      *(f32x4 *)(A1 + (lane & 3) * FCL_LDA + f0) = a;
    }
    fcl_bar();
    FCL_CSTAMP(2)      // epilogue + barrier
    *(f32x4 *)(red + (w * 64 + lane) * 4) = fcl_quad_res<64>(W2, A1 + (lane & 3) * FCL_LDA + 64 * w);
    fcl_bar();
    FCL_CSTAMP(3)      // fc2 partials + barrier
    if (w == 0) {
      // wave 0 alone: lane (b, j) adds up rows 4 b + i of the 8 partials, then LayerNorm + ReLU (networks.py:147,165) of
      // sample j across the 16 lanes that share it (shuffles over the lane bits 2..5), the tapes and the next input
      // straight from registers
      const int j = lane & 3, f0 = 4 * (lane >> 2);
      f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ww = 0; ww < FCL_NW; ++ww) y += *(const f32x4 *)(red + (ww * 64 + lane) * 4);
      float yv[4], s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { yv[i] = (f0 + i < MZ_H) ? y[i] + b2[f0 + i] : 0.f; s += yv[i]; }
      s = fcl_sum_bits2to5(s);
      const float mean = s / (float)MZ_H;
      float d[4], var = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { d[i] = (f0 + i < MZ_H) ? yv[i] - mean : 0.f; var += d[i] * d[i]; }
      var = fcl_sum_bits2to5(var);
      const float rstd = 1.0f / sqrtf(var / (float)MZ_H + FCL_LN_EPS);
      const int act_p = acts[j * 8 + (p < 7 ? p : 7)];
      f32x4 xv4, xh4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = f0 + i;
        xh4[i] = d[i] * rstd;
        const float hv = f < MZ_H ? fmaxf(xh4[i] * lnw[f] + lnb[f], 0.f) : 0.f;
        // the next input: [h | one-hot(action) | 0]  (networks.py:167-174); past the last transition: no action
        xv4[i] = f < MZ_H ? hv : ((p < v.K && f - MZ_H == act_p) ? 1.f : 0.f);
      }
      *(f32x4 *)(X + j * LDX + f0) = xv4;
      *(f32x4 *)(S + j * 64 + f0) = xh4;
      if (lane < 4) misc[lane] = rstd;
    }
    FCL_CSTAMP(4)      // wave 0: reduce + LayerNorm
    fcl_bar();
    FCL_CSTAMP(5)      // barrier
    // the tapes of this position, by the upper half of the workgroup (wave 0 above is the serial part: seven waves wait for
    // it); X, S and misc are next written three barriers from here
    if (tid >= 256) {
      const int t = tid - 256, f = t >> 2, n = t & 3;
      const float xv = X[n * LDX + f];
      const size_t tp = fcl_tp(64, cb, f, n0 + n);
      v.xhat[(size_t)p * T64 + tp] = S[n * 64 + f];
      v.h[(size_t)p * T64 + tp] = f < MZ_H ? xv : 0.f;
      if (p < v.K) v.xin[(size_t)(p + 1) * TX + fcl_tp(v.XR, cb, f, n0 + n)] = xv;
      if (t < 4) v.rstd[(size_t)p * R + row0 + t] = misc[t];
    }
  };
  {   // position 0: the representation, its weights streamed
    float WR2[64];
    fcl_quad_load<64>(WR2, v.pk + v.rep.F2 + (size_t)w * 64 * 64, lane);
    const f32x4 acc = fcl_quad_stream(v.pk + v.rep.F1 + (size_t)w * v.O * 64, v.O, X + (lane & 3) * LDX, lane);
    rest(0, acc, WR2, b1r, b2r);
  }
  FCL_KSTAMP(2)      // position 0 done
  fcl_quad_settle(WT1);
  fcl_quad_settle(WT2);
  FCL_KSTAMP(3)      // the transition's weights have arrived
  for (int p = 1; p <= v.K; ++p) {
    FCL_CSTAMP(0)
    rest(p, fcl_quad_res<KP>(WT1, X + (lane & 3) * LDX), WT2, b1t, b2t);
  }
#undef FCL_CSTAMP
  FCL_KSTAMP(4)        // positions 1..K done
#undef FCL_KSTAMP
}

__global__ __launch_bounds__(FCL_THREADS) void k_fcl_chain_bwd4(FclView v) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  const int LDX = v.xq + 4;
  float *X = fcl_smem, *A1 = X + 4 * LDX, *red = A1 + 4 * FCL_LDA, *PV = red + 2048 + 16;
  float *D2 = X;                                     // d (pre-LayerNorm output), sample-major like X (rows of >= 64 floats)
  float *lnw = PV;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int cb = blockIdx.x >> 2, n0 = 4 * (blockIdx.x & 3), row0 = blockIdx.x * 4, R = v.R, K1 = v.K + 1, loff = fcl_lane_off4(lane, n0);
  const size_t T64 = (size_t)64 * R, T512 = (size_t)512 * R;
  float WB2[MZ_H], WB1[64];
  fcl_quad_load<MZ_H>(WB2, v.pk + v.tr.B2 + (size_t)w * MZ_H * 64, lane);
  fcl_quad_load<64>(WB1, v.pk + v.tr.B1 + (size_t)w * 64 * 64, lane);
  float WR2[MZ_H];      // the representation's, for position 0 (requested now)
  fcl_quad_load<MZ_H>(WR2, v.pk + v.rep.B2 + (size_t)w * MZ_H * 64, lane);
  if (tid < 64) lnw[tid] = tid < MZ_H ? v.P[v.ln_w + tid] : 0.f;
  // wave 0 carries the per-sample work in registers: lane (b, j) = features 4 b + i of sample j
  const int j = lane & 3, f0 = 4 * (lane >> 2);
  float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dch = (f32x4){0.f, 0.f, 0.f, 0.f};         // d chain: gradient from the transition of position p + 1 into h_p
  float tv[4][5], trs = 0.f;
  auto request = [&](int p) __attribute__((always_inline)) {
    if (w == 0) {
      const size_t o = fcl_tp(64, cb, f0, n0 + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tv[i][0] = v.dH[((size_t)0 * K1 + p) * T64 + o + i * 16];
        tv[i][1] = v.dH[((size_t)1 * K1 + p) * T64 + o + i * 16];
        tv[i][2] = p < v.K ? v.dH[((size_t)2 * K1 + p + 1) * T64 + o + i * 16] : 0.f;
        tv[i][3] = v.h[(size_t)p * T64 + o + i * 16];
        tv[i][4] = v.xhat[(size_t)p * T64 + o + i * 16];
      }
      trs = v.rstd[(size_t)p * R + row0 + j];
    }
  };
  for (int idx = tid; idx < 4 * LDX; idx += FCL_THREADS) X[idx] = 0.f;
  request(v.K);
  fcl_bar();
  auto body = [&](int p, const float (&W2)[MZ_H]) __attribute__((always_inline)) {
    if (w == 0) {
      // gradient arriving at h_p (value and policy heads of position p, reward head and transition of position p + 1;
      // hook 0.5, learners.py:200), then ReLU and LayerNorm backwards over the sample's 16 lanes
      const float rstd = trs;
      float gy[4], xh[4], dx[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float g = tv[i][0] + tv[i][1];
        if (p < v.K) g = g + tv[i][2] + dch[i];
        if (p >= 1) g *= 0.5f;
        const bool real = f0 + i < MZ_H;
        gy[i] = (real && tv[i][3] > 0.f) ? g : 0.f;
        xh[i] = real ? tv[i][4] : 0.f;
        dgam[i] += gy[i] * xh[i]; dbet[i] += gy[i];
        dx[i] = real ? gy[i] * lnw[f0 + i] : 0.f;
        s1 += dx[i]; s2 += dx[i] * xh[i];
      }
      s1 = fcl_sum_bits2to5(s1); s2 = fcl_sum_bits2to5(s2);
      const float inv = 1.f / (float)MZ_H;
      f32x4 dy;
#pragma unroll
      for (int i = 0; i < 4; ++i) dy[i] = (f0 + i < MZ_H) ? rstd * (dx[i] - s1 * inv - xh[i] * (s2 * inv)) : 0.f;
      *(f32x4 *)(D2 + j * LDX + f0) = dy;
    }
    if (p > 0) request(p - 1);
    const float *a1t = v.a1c + (size_t)p * T512 + fcl_tp(512, cb, 64 * w, 0);
    f32x4 msk;
#pragma unroll
    for (int i = 0; i < 4; ++i) msk[i] = (a1t + i * 16)[loff];
    fcl_bar();
    if (tid >= 256) {      // the delta tape of this position, by the upper half of the workgroup (D2 is next written three barriers on)
      const int t = tid - 256;
      v.d2c[(size_t)p * T64 + fcl_tp(64, cb, t >> 2, n0 + (t & 3))] = D2[(t & 3) * LDX + (t >> 2)];
    }
    const f32x4 acc = fcl_quad_res<MZ_H>(W2, D2 + (lane & 3) * LDX);
    {
      const int g0 = 64 * w + 4 * (lane >> 2);
      float *d1t = v.d1c + (size_t)p * T512 + fcl_tp(512, cb, 64 * w, 0);
      f32x4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i] = msk[i] > 0.f ? acc[i] : 0.f;
        (d1t + i * 16)[loff] = d[i];
      }
      *(f32x4 *)(A1 + (lane & 3) * FCL_LDA + g0) = d;
    }
    fcl_bar();
    if (p >= 1) {
      *(f32x4 *)(red + (w * 64 + lane) * 4) = fcl_quad_res<64>(WB1, A1 + (lane & 3) * FCL_LDA + 64 * w);
      fcl_bar();
      if (w == 0) {
        f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < FCL_NW; ++ww) y += *(const f32x4 *)(red + (ww * 64 + lane) * 4);
        dch = y;            // (rows >= 50 come out of zero weights)
      }
      // (red is next written after two more barriers)
    }
  };
  fcl_quad_settle(WB2);
  fcl_quad_settle(WB1);
  for (int p = v.K; p >= 1; --p) body(p, WB2);
  body(0, WR2);      // position 0: the representation's fc2
  // LayerNorm weight / bias gradients of this workgroup's 4 samples over all positions
  if (w == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = dgam[i], b = dbet[i];
      a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
      b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
      if (j == 0 && f0 + i < 64) {
        v.lnpart[(size_t)blockIdx.x * 128 + f0 + i] = a;
        v.lnpart[(size_t)blockIdx.x * 128 + 64 + f0 + i] = b;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradients
// One workgroup (four waves) per job: G[16 x 64] strip of dW = D . X^T over the R rows of one unroll position's tapes
// (D: deltas, Mp features per row chunk; X: layer inputs, Np features), out rows 16 tm .., out columns 64 ng ..; the bias
// gradient = row sums of D.
struct FclJob {
  size_t d_off, x_off;      // float offsets of the two tapes (feature 0, row 0)
  size_t w_off, b_off;      // flat offsets of W [M][N] and of its bias (b_off used by the ng == 0 strip)
  int M, N, Mp, Np, tm, ng, slab;      // Mp, Np: feature counts of the two tapes (their row-chunk strides / 16)
};

__global__ __launch_bounds__(256) void k_fcl_dw(const FclJob *jobs, int njobs, const float *tapes, float *part, size_t nflat, int R,
                                                float *steps, int nsteps) {
  // (the step counters -- torch keeps one per parameter -- advance here, a launch ahead of the optimiser kernel that reads them)
  if (blockIdx.x == 0 && (int)threadIdx.x < nsteps) steps[threadIdx.x] += 1.f;
  // four waves per strip, each over a quarter of the rows (the kernel is bound by how many load streams are in flight, its
  // MFMAs are ~3 us: 17.1 us with two waves per strip, 20.4 with one); the partials meet in LDS and are added in a fixed order
  __shared__ __attribute__((aligned(16))) float sh[3][17][64];
  const int qt = threadIdx.x >> 6;          // this wave's quarter of the rows
  const int wj = blockIdx.x;
  const bool live = wj < njobs;
  const FclJob j = jobs[live ? wj : 0];
  const int lane = threadIdx.x & 63, g4 = lane >> 4, m16 = lane & 15;
  // tapes: [row chunk][feature][16 rows] -- 16 features x 16 rows of chunk c are one contiguous KiB
  const float *Dp = tapes + j.d_off + (size_t)(16 * j.tm + m16) * 16 + 4 * g4;
  const float *Xp = tapes + j.x_off + (size_t)(64 * j.ng + m16) * 16 + 4 * g4;
  const size_t dstr = (size_t)j.Mp * 16, xstr = (size_t)j.Np * 16;      // floats per row chunk
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  // (k index g4 of k-step jj of chunk c = row 16 c + 4 g4 + jj, in both operands: the sum over rows is order-free.)
  // The loads of a wave's chunks (up to 8: 40 x 16 bytes per lane) are requested before the first MFMA of the group
  const int nch = R >> 4, per = (nch + 3) >> 2, ch0 = qt * per, ch1 = ch0 + per < nch ? ch0 + per : nch;
  for (int c0 = ch0; live && c0 < ch1; c0 += 8) {
    f32x4 a[8], b[8][4];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int c = c0 + cc;
      if (c < ch1) {
        a[cc] = *(const f32x4 *)(Dp + c * dstr);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[cc][i] = *(const f32x4 *)(Xp + c * xstr + 256 * i);
      }
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      if (c0 + cc < ch1) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = fcl_mfma(a[cc][jj], b[cc][i][jj], acc[i]);
        }
        bsum += (a[cc][0] + a[cc][1]) + (a[cc][2] + a[cc][3]);
      }
    }
  }
  if (qt > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sh[qt - 1][4 * i + r][lane] = acc[i][r];
    sh[qt - 1][16][lane] = bsum;
  }
  __syncthreads();
  if (qt > 0 || !live) return;
#pragma unroll
  for (int qq = 0; qq < 3; ++qq) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] += sh[qq][4 * i + r][lane];
    bsum += sh[qq][16][lane];
  }
  float *out = part + (size_t)j.slab * nflat;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = 64 * j.ng + 16 * i + m16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * j.tm + 4 * g4 + r;
      if (m < j.M && n < j.N) out[j.w_off + (size_t)m * j.N + n] = acc[i][r];
    }
  }
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (j.ng == 0 && g4 == 0 && 16 * j.tm + m16 < j.M) out[j.b_off + 16 * j.tm + m16] = bsum;
}

// ------------------------------------------------------------------------------------------------ gradient, optimiser
struct FclOpt {
  double beta1, beta2, eps, wd;
  float clip;
  int adamw, no_update;
};

// grad[i] = sum over the unroll positions' strips (LayerNorm parameters: over the workgroups' partials); per-block sum
// of squares for clip_grad_norm_ (launched only when clipping is on: without it k_fcl_adam adds the strips up itself)
// one parameter's gradient: the unroll positions' strips in order (LayerNorm parameters: the chain workgroups' partials,
// eight independent chains so that the loads of a round are in flight together; the order of the sum is fixed)
__device__ __forceinline__ float fcl_grad_of(size_t i, const float *part, int nslab, const float *lnpart, int nwg, size_t ln_w, size_t nflat) {
  float g = 0.f;
  if (i >= ln_w && i < ln_w + 2 * MZ_H) {
    const int k = (int)(i - ln_w), col = k < MZ_H ? k : 64 + (k - MZ_H);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int wg = 0;
    for (; wg + 8 <= nwg; wg += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += lnpart[(size_t)(wg + j) * 128 + col];
    }
    for (; wg < nwg; ++wg) a[wg & 7] += lnpart[(size_t)wg * 128 + col];
    g = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  } else {
    for (int q = 0; q < nslab; ++q) g += part[(size_t)q * nflat + i];
  }
  return g;
}

__global__ __launch_bounds__(256) void k_fcl_grad(const float *part, int nslab, const float *lnpart, int nwg, size_t ln_w,
                                                  size_t nflat, float *grad, float *bsq) {
  __shared__ float sh[256];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  if (i < nflat) {
    g = fcl_grad_of(i, part, nslab, lnpart, nwg, ln_w, nflat);
    grad[i] = g;
  }
  sh[threadIdx.x] = g * g;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsq[blockIdx.x] = sh[0];
}

// clip_grad_norm_ (learners.py:217-218), Adam / AdamW with torch's fused-kernel arithmetic (utils.py:73-83: eps 1.5e-4),
// new weights -> flat vector + the packed copies; one more block (the last) adds the three weighted loss means up
// (learners.py:205-207,228-230).  steps[0] has been advanced by k_fcl_dw.
__global__ __launch_bounds__(256) void k_fcl_adam(float *P, float *pk, const int32_t *posA, const int32_t *posB, float *grad,
                                                  const float *part, int nslab, const float *lnpart, int nwg, size_t ln_w,
                                                  const float *bsq, int nblk, float *m, float *vv, const float *steps,
                                                  const float *lr_p, FclOpt o, size_t nflat, const float *lossb, const void *w,
                                                  int w_f64, int bs, int K1, double *loss_acc) {
  __shared__ float sh[256];
  __shared__ float shc[4];
  if ((int)blockIdx.x == nblk) {
    // the three heads together: every thread's loads are independent (one round trip), one reduction of three doubles
    // (head by head this block was the kernel's critical path: three dependent rounds of loads + tree reductions)
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < bs; b += 256) {
      float l[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int hd = 0; hd < 3; ++hd)
        for (int p = (hd == 2 ? 1 : 0); p < K1; ++p) l[hd] += lossb[((size_t)hd * K1 + p) * bs + b];
      const double wb = w_f64 ? ((const double *)w)[b] : (double)((const float *)w)[b];
#pragma unroll
      for (int hd = 0; hd < 3; ++hd) acc[hd] += wb * (double)l[hd];
    }
    __shared__ double shd3[3][256];
#pragma unroll
    for (int hd = 0; hd < 3; ++hd) shd3[hd][threadIdx.x] = acc[hd];
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) {
      if ((int)threadIdx.x < k) {
#pragma unroll
        for (int hd = 0; hd < 3; ++hd) shd3[hd][threadIdx.x] += shd3[hd][threadIdx.x + k];
      }
      __syncthreads();
    }
    // _loss_dev order: reward, value, policy
    if (threadIdx.x < 3) loss_acc[threadIdx.x == 2 ? 0 : (threadIdx.x == 0 ? 1 : 2)] += shd3[threadIdx.x][0] / (double)bs;
    return;
  }
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (part && i < nflat) grad[i] = fcl_grad_of(i, part, nslab, lnpart, nwg, ln_w, nflat);      // (no clipping: no k_fcl_grad launch)
  if (o.no_update) return;
  float s = 0.f;
  if (o.clip > 0.f)
    for (int b = threadIdx.x; b < nblk; b += 256) s += bsq[b];
  sh[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    const double step = (double)steps[0];
    shc[0] = (float)(1.0 - pow(o.beta1, step));
    shc[1] = (float)(1.0 - pow(o.beta2, step));
  }
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  const float norm = sqrtf(sh[0]);
  float coef = 1.f;
  if (o.clip > 0.f) coef = fminf(o.clip / (norm + 1e-6f), 1.f);
  if (i < nflat) {
    // (the hyper-parameters are doubles in torch's fused kernel and the moments' updates are evaluated in double there:
    // 1 - 0.999 as a float is 4.7e-5 off)
    const double lr = (double)*lr_p;
    float g = grad[i] * coef, p = P[i];
    if (o.wd != 0.0) {
      if (o.adamw) p = (float)((double)p - lr * o.wd * (double)p);
      else g = (float)((double)g + (double)p * o.wd);
    }
    float ea = m[i], es = vv[i];
    ea = (float)((double)ea + (1.0 - o.beta1) * ((double)g - (double)ea));            // torch lerp, weight < 0.5
    es = (float)(o.beta2 * (double)es + (1.0 - o.beta2) * (double)g * (double)g);
    const float bc1 = shc[0], bc2 = shc[1];
    const float step_size = (float)(lr / (double)bc1), bc2s = sqrtf(bc2);
    const float denom = (float)((double)(sqrtf(es) / bc2s) + o.eps);
    p -= step_size * ea / denom;
    m[i] = ea; vv[i] = es; P[i] = p;
    if (posA[i] >= 0) pk[posA[i]] = p;
    if (posB[i] >= 0) pk[posB[i]] = p;
  }
}

// packed copies from the flat vector (after a load_state_dict / any write that did not go through k_fcl_adam)
__global__ __launch_bounds__(256) void k_fcl_repack(const float *P, float *pk, const int32_t *posA, const int32_t *posB, size_t nflat) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nflat) return;
  const float p = P[i];
  if (posA[i] >= 0) pk[posA[i]] = p;
  if (posB[i] >= 0) pk[posB[i]] = p;
}
