// mz_fcl.hip.h -- the FCNetwork learner step (reference learners.py:164-230, networks.py:135-180, config.py:27-33,51-68,
// utils.py:53-60): the whole update -- K-step unroll forward, the three heads' losses, backward, clipping, AdamW -- without a
// GEMM library or an autograd tape.  The step is ~1.4 GFLOP (batch 256, K = 5): as PyTorch operators it is ~220 kernels of
// ~4 us of launch floor each; here its shape follows the data dependences.  TWO launches at the reference's batch 256:
//
//   k_fcl_fb       chain workgroups (4 samples each: h_0 = representation(obs), h_p = dynamics(h_{p-1}, a_{p-1}) -- the only
//                  sequential part -- on v_mfma_f32_4x4x1_16b_f32: 64 output rows x 4 samples per instruction, the transition's
//                  weights resident in registers across positions, activations in LDS), every heads unit (16 samples = the
//                  16 columns of v_mfma_f32_16x16x4_f32, one unroll position, one head: value / policy / reward head forward,
//                  two-hot targets, soft cross-entropy, the head's backward down to d loss / d hidden state), the chain BACKWARDS
//                  in the same workgroups (gradient hooks 0.5, LayerNorm, ReLU) and the heads' weight-gradient jobs -- handing
//                  over inside the launch through write-through tapes and counters (see "in-launch hand-off"): a unit starts when
//                  its sample group's chain workgroups have announced its position; a chain workgroup that has finished its
//                  forward pass runs one of position K's units itself (or takes a queued one), then walks backwards as the units
//                  of each position announce their d loss / d hidden; a head's jobs start when all its units have finished
//   k_fcl_dwa      the chain layers' weight-gradient jobs + the LayerNorm parameters + the loss sums (+ the native loop's
//                  completion word)
//
// (MZ_FCL_FUSE_FB=0: three launches -- k_fcl_fwd: chain + units; k_fcl_bwd_dw: backward chain + the heads' jobs; k_fcl_dwa -- the
// structure of batch 512, where the chain takes half the chip.  MZ_FCL_FUSE_FWD=0: four.  The same arithmetic in the same order: tests
// hold the three to equal bits.)
// A weight-gradient job = one tile of a layer's dW = sum_rows delta (x) input over EVERY unroll position the layer is applied at
// (MFMA strips over the activation / delta tapes in HBM, summed in a fixed order: deterministic, no atomics); with one row slab
// the tile is the gradient, and Adam / AdamW on exactly those weights (torch's fused-kernel arithmetic; the new weights into the
// flat vector AND into the packed copies the next step's MFMAs read) follows in the same workgroup.
// Larger batches: k_fcl_chain_fwd4 / k_fcl_heads as launches of their own from batch 1024 (the chain fills the chip), row slabs +
// 64-row tiles (k_fcl_dwt) + the optimiser kernel (k_fcl_adam; with k_fcl_grad for clip_grad_norm_) from batch 1024, two / four
// groups of four samples per chain workgroup from batch 2048 / 4096.
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
  float *lnpart;                                        // [bs / 4][128] LayerNorm weight / bias gradient partials (per group of four samples)
  float *new_errors;
  unsigned *flags;               // fused forward launch (k_fcl_fwd): [bs / 16][K + 1] arrival counters -- chain workgroups whose tapes of (sample group, position) are
                                 // written through; 4 = all of the group's; zeroed by k_fcl_bwd_dw for the next step
  int nflags;                    // (k_fcl_fb: [2][bs / 16][K + 1] -- behind the arrival counters, the heads' counters of finished units per (sample group, position
                                 // whose hidden state the units' d loss / d hidden belongs to): 3 = value, policy and the next position's reward unit; 2 at position K)
  unsigned *err;                 // [host, device-mapped] set to 1 where a unit's wait for its counter ran into its bound (never on a healthy box)
  float *steps; int nsteps;      // the optimiser's per-parameter step counters (torch keeps one per parameter), advanced by k_fcl_heads; nsteps = 0: not this step
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

// ------------------------------------------------------------------------------------------------ in-launch hand-off (k_fcl_fwd, k_fcl_fb)
// Workgroups of ONE launch hand data to each other (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the table of forms
// measured without an acquire): the producer's stores of the handed-off bytes are write-through (sc1: relaxed agent-scope atomic
// stores), every storing wave waits for them (s_waitcnt vmcnt(0)), a workgroup barrier, then ONE lane adds to the counter (agent
// scope); the consumer polls the counter with sc1 loads in one lane (bounded), a workgroup barrier (or: the polling wave itself is
// the only reader), then EVERY load of the handed-off bytes is an sc1 load (4 bytes: relaxed agent-scope atomic loads; 16 bytes:
// buffer loads with the sc1 bit).  The hand-offs, counters in FclView::flags (zeroed by the step's last launch):
//   chain forward  -> heads units     hidden states / transition inputs of (sample group, position); counter [group][position] = 4
//   heads units    -> chain backward  d loss / d hidden of (group, position); counter [G K1 + group][position] = 3 (2 at position K)
//   heads units    -> the head's jobs the unit's tapes (fc1 activations, both deltas); one counter per head = its number of units
//   chain forward  -> the heads' jobs (hidden states / transition inputs: covered by the units' counters -- a unit has waited for them)
//   chain workgroup 0 -> the jobs     the optimiser's step counter, stored before its first signal
// Every wait depends only on workgroups dispatched EARLIER in the grid (chain, units, jobs, in this order), and the chain's forward
// passes wait for nothing: the launch ends on any device with CUs left beside the chain workgroups (the handle asks for twice as many CUs).
__device__ __forceinline__ void fcl_store_wt(float *p, float x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float fcl_load_wt(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fcl_signal(unsigned *flag) { __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: until *flag >= want, or ~40 ms of the constant 100 MHz clock have passed (then *err = 1 and the caller goes on: the
// step's numbers are wrong, the launch ends, the host sees the word)
// (SLEEP: s_sleep units of 64 cycles between two polls -- 8 for a unit or a chain workgroup (few pollers per counter, the wait is on the
// step's critical path), 64 for a weight-gradient job (168 of them wait ~10 us for three counters: polled every 0.25 us they slowed the
// units' own counter traffic -- the backward chain's positions arrived 1.5 - 3.6 us later))
template <int SLEEP = 8>
__device__ __forceinline__ void fcl_wait_flag(const unsigned *flag, unsigned want, unsigned *err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(SLEEP);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 4000000ull) {
      if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

// Tape addressing of a D fragment (rows 64 w + 16 i + 4 g4 + r, column = sample m16): a wave-uniform row base (scalar
// registers) + ONE per-lane offset, so that no per-(i, r) 64-bit address lives in vector registers across the positions
__device__ __forceinline__ int fcl_lane_off(int lane) { return 64 * (lane >> 4) + (lane & 15); }

// fc1 epilogue: bias (LDS), ReLU, the activations to LDS (k-step layout) and to the tape
// (WT: tape stores written through -- a unit whose tapes are read by weight-gradient jobs of the same launch)
template <bool WT = false>
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
      if constexpr (WT) fcl_store_wt((a1t + (64 * w + 16 * i + r) * 16) + loff, a[r]);
      else (a1t + (64 * w + 16 * i + r) * 16)[loff] = a[r];
    }
    *(f32x4 *)(A1 + ((f0 >> 2) * 16 + m16) * 4) = a;
  }
}

// d a1 through the ReLU (mask: the activation a > 0), to LDS and to the delta tape
template <bool WT = false>
__device__ __forceinline__ void fcl_mask_out(const f32x4 acc[4], const f32x4 msk[4], float *A1, float *d1t, int loff, int w, int lane) {
  const int g4 = lane >> 4, m16 = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f0 = 64 * w + 16 * i + 4 * g4;
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[r] = msk[i][r] > 0.f ? acc[i][r] : 0.f;
      if constexpr (WT) fcl_store_wt((d1t + (64 * w + 16 * i + r) * 16) + loff, d[r]);
      else (d1t + (64 * w + 16 * i + r) * 16)[loff] = d[r];
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
// sum over the 64 lanes of a wave: the four DPP rows' sums (four exchanges each), read out of a lane of each row and added in a fixed
// order -- no LDS round trip; every lane ends with the same bits
__device__ __forceinline__ float fcl_sum16(float x);
__device__ __forceinline__ float fcl_sum64(float x) {
  x = fcl_sum16(x);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
  return (r0 + r1) + (r2 + r3);
}
// sum over the 16 lanes of a DPP row (no LDS round trip); every lane of the row ends with the same bits
__device__ __forceinline__ float fcl_sum16(float x) {
  x += fcl_xchg<1>(x); x += fcl_xchg<2>(x); x += fcl_xchg<4>(x); x += fcl_xchg<8>(x);
  return x;
}
__device__ __forceinline__ float fcl_max32(float x) {
  x = fmaxf(x, fcl_xchg<1>(x)); x = fmaxf(x, fcl_xchg<2>(x)); x = fmaxf(x, fcl_xchg<4>(x)); x = fmaxf(x, fcl_xchg<8>(x));
  return fmaxf(x, __shfl_xor(x, 16, 32));
}

// ------------------------------------------------------------------------------------------------ heads
// grid (bs / 16, K + 1, 3): head 0 value (input h_p), 1 policy (h_p), 2 reward (x_p = [h_{p-1} | one-hot], p >= 1)
// WAIT: a unit of the fused forward launch -- its inputs come from chain workgroups of the same launch: everything else it needs
// (fc1's weights, biases, targets) is requested FIRST and arrives while one lane polls the group's counter
// SIGD: a unit of k_fcl_fb -- its d loss / d hidden is handed to the backward chain of the SAME launch, its tapes to the head's weight-
// gradient jobs of the same launch: all stored write-through, announced on the counter of the (sample group, position) whose hidden state
// the gradient belongs to and on the head's counter of finished units
struct FclNothing { __device__ __forceinline__ void operator()() const {} };
// requested: called once the unit's own first loads (biases, targets, fc1's weights, its inputs) are on their way -- k_fcl_fb's chain workgroups ask for
// their backward pass's weights there (loads return in order: asked for in front of the unit, they held its first phase back 3 us)
template <bool WAIT, bool SIGD = false, class F = FclNothing>
__device__ __forceinline__ void fcl_heads_body(const FclView &v, const int cb, const int p, const int hd, float *fcl_smem, F requested = F()) {
  float *X = fcl_smem, *A1 = X + 1024, *redf = A1 + 8192, *Y = redf + 8192, *S = Y + 1024, *PV = S + 1024;
  f32x4 *red = (f32x4 *)redf;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g4 = lane >> 4, m16 = lane & 15;
  const int row0 = cb * 16, R = v.R, K1 = v.K + 1, loff = fcl_lane_off(lane);
  // (the step counters advance here, two launches ahead of the optimiser code that reads them)
  if constexpr (!SIGD) { if (cb == 0 && p == 0 && hd == 0 && tid < v.nsteps) v.steps[tid] += 1.f; }      // (k_fcl_fb: chain workgroup 0 does)
  if (hd == 2 && p == 0) return;
  int stamp_i = 0;
#define FCL_STAMP() if (v.prof && cb == 0 && p == 1 && tid == 0) v.prof[hd * 16 + stamp_i++] = __builtin_amdgcn_s_memtime();
  FCL_STAMP()
  const size_t T64 = (size_t)64 * R, T512 = (size_t)512 * R, TX = (size_t)v.XR * R;
  const FclPack &pk = v.head[hd];
  const size_t hp = (size_t)hd * K1 + p;
  // the layers' weights pass through ONE set of 16 registers per lane, each layer's requested as soon as the previous
  // layer's products have been issued (the compiler barriers keep the requests from being hoisted above those products,
  // which would need a second register set: 128 registers is what two workgroups per CU leave); fc1's now
  // (loads return in order: the inputs the first barrier waits for are requested BEFORE fc1's weights -- r05 requested the
  // weights first and every workgroup's first barrier waited for all 128 KB of them)
  f32x4 WA[16];
  const float *src = hd == 2 ? v.xin + (size_t)p * TX : v.h + (size_t)p * T64;
  float xs_in[2] = {0.f, 0.f};
  if constexpr (!WAIT) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * FCL_THREADS;
      xs_in[k] = src[fcl_tp(hd == 2 ? v.XR : 64, cb, idx >> 4, idx & 15)];
    }
  }
  const float pv1 = v.P[v.hb1[hd] + tid];
  const float pv2 = (tid < 64 && tid < pk.mout) ? v.P[v.hb2[hd] + tid] : 0.f;
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
  asm volatile("" ::: "memory");
  fcl_req_wide(WA, (const f32x4 *)(v.pk + pk.F1), pk.ks1, w, lane);
  asm volatile("" ::: "memory");
  if constexpr (!WAIT) requested();
  if constexpr (WAIT) {
    // hidden states of position p (value, policy) / the transition input of position p = what position p - 1 left (reward)
    // (development: the launch's timeline on the constant 100 MHz clock, slots 59..63: chain workgroup 0 start / end, the LAST
    // position's value unit of group 0 at its start / past its wait / at its end)
    if (v.prof && cb == 0 && p == v.K && hd == 0 && tid == 0) v.prof[61] = __builtin_amdgcn_s_memrealtime();
    if (v.prof && cb == 0 && p == v.K - 1 && hd == 0 && tid == 0) v.prof[45] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) fcl_wait_flag(v.flags + (size_t)cb * K1 + (hd == 2 ? p - 1 : p), 4u, v.err);
    if (v.prof && cb == 0 && p == v.K && hd == 0 && tid == 0) v.prof[62] = __builtin_amdgcn_s_memrealtime();
    if (v.prof && cb == 0 && p == v.K - 1 && hd == 0 && tid == 0) v.prof[46] = __builtin_amdgcn_s_memrealtime();
    fcl_bar();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * FCL_THREADS;
      xs_in[k] = fcl_load_wt(src + fcl_tp(hd == 2 ? v.XR : 64, cb, idx >> 4, idx & 15));
    }
    asm volatile("" ::: "memory");
    requested();      // (behind the inputs: 230 KB of backward weights in front of them kept the first barrier 1.6 us longer)
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * FCL_THREADS;
    X[fcl_at(idx >> 4, idx & 15)] = xs_in[k];
  }
  PV[tid] = pv1;
  if (tid < 64) PV[512 + tid] = pv2;
  fcl_bar();
  FCL_STAMP()      // 1: inputs in LDS
  f32x4 acc[4];
  fcl_wide_regs(WA, pk.ks1, X, lane, acc);
  asm volatile("" ::: "memory");
  fcl_load_narrow(WA, (const f32x4 *)(v.pk + pk.F2), w, lane);          // fc2's, under the epilogue
  FCL_STAMP()      // 2: fc1 products
  fcl_fc1_out<SIGD>(acc, PV, A1, v.a1h + hp * T512 + fcl_tp(512, cb, 0, 0), loff, w, lane);
  // (no barrier: the 16 k-steps of fc2's split-K slice of wave w are the 64 features of A1 wave w has just written itself)
  FCL_STAMP()      // 3: fc1 epilogue
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
    if constexpr (SIGD) fcl_store_wt(v.d2h + hp * T64 + fcl_tp(64, cb, f, n), S[fcl_at(f, n)]);
    else v.d2h[hp * T64 + fcl_tp(64, cb, f, n)] = S[fcl_at(f, n)];
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
    fcl_mask_out<SIGD>(acc, msk, A1, v.d1h + hp * T512 + fcl_tp(512, cb, 0, 0), loff, w, lane);
  }
  // (no barrier: wave w's slice of the fc1-transposed product again)
  FCL_STAMP()      // 8: mask
  fcl_narrow_res<4>(WA, A1, red, w, lane);
  fcl_bar();
  FCL_STAMP()      // 9: fc1-transposed partials + barrier
  fcl_reduce(red, 4, nullptr, MZ_H, Y, tid);
  fcl_bar();
  FCL_STAMP()      // 10: reduce + barrier
  for (int idx = tid; idx < 64 * 16; idx += FCL_THREADS) {
    const int f = idx >> 4, n = idx & 15;
    if constexpr (SIGD) fcl_store_wt(v.dH + hp * T64 + fcl_tp(64, cb, f, n), Y[fcl_at(f, n)]);
    else v.dH[hp * T64 + fcl_tp(64, cb, f, n)] = Y[fcl_at(f, n)];
  }
  FCL_STAMP()        // 11: d hidden stored
#undef FCL_STAMP
  if constexpr (SIGD) {
    // every wave has stored (2 x 512 threads cover the 1024 elements): each waits for its stores, the workgroup's barrier, one lane announces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fcl_bar();
    if (tid == 0) {
      fcl_signal(v.flags + (size_t)(v.bs >> 4) * K1 + (size_t)cb * K1 + (hd == 2 ? p - 1 : p));
      fcl_signal(v.flags + (size_t)2 * (v.bs >> 4) * K1 + 32 * (1 + hd));      // finished units of this head: its weight-gradient jobs wait for all of them
    }
  }
  if constexpr (WAIT) {
    if (v.prof && cb == 0 && p == v.K && hd == 0 && tid == 0) v.prof[63] = __builtin_amdgcn_s_memrealtime();
    if (v.prof && cb == 0 && p == v.K - 1 && hd == 0 && tid == 0) v.prof[47] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ __launch_bounds__(FCL_THREADS, 4) void k_fcl_heads(FclView v) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  fcl_heads_body<false>(v, blockIdx.x, blockIdx.y, blockIdx.z, fcl_smem);
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

// LDS of the chain kernels (floats), G groups of 4 samples, activations SAMPLE-major: X [4 G][xq + 4] | A1 [4 G][516] | red [G][2048] | misc [16] | PV [1408] | S [4 G][64]
#define FCL_LDS4_FLOATS(xq, G) (4 * (G) * ((xq) + 4) + 4 * (G) * 516 + (G) * 2048 + 16 + 1408 + 256 * (G))
#define FCL_LDA 516

// Split-K partials of a 64-row x 4-sample product in LDS: a wave's 64 lanes (b, j) = (lane >> 2, lane & 3) each leave one f32x4 (rows
// 4 b + i of sample j); the reducing wave reads them as lane (j, b) = (lane >> 4, lane & 15).  Slot of (j, b) among the wave's 64
// 16-byte slots: 16 j + ((b + 4 j) & 15) -- any 16 consecutive lanes of EITHER side touch 16 different slots modulo 16 (256 bytes = all
// banks once); the plain slot 4 b + j served the reader's 16 lanes out of 4 of them
__device__ __forceinline__ int fcl_red_slot(int j, int b) { return 16 * j + ((b + 4 * j) & 15); }

// D fragment (rows 64 w + 4 b + i, sample j) -> tape [chunk][feature][16]: wave-uniform base + one lane offset + i * 16
__device__ __forceinline__ int fcl_lane_off4(int lane, int n0) { return (lane >> 2) * 64 + n0 + (lane & 3); }

// KP: the transition's fc1 input features (50 + actions), padded: 56 (up to 6 actions) or 64
// SIG: a chain workgroup of the fused forward launch -- the tapes the heads read (hidden states, transition inputs) are stored
// write-through and, one phase later, announced on the group's counter (see "in-launch hand-off").
// G: groups of four samples per workgroup (1, 2 or 4).  A position's MFMAs are ~1.9 k of its ~5 k cycles -- the rest is the
// serial chain epilogue / barrier / LayerNorm on one wave -- so where the batch has more than one workgroup per CU (batch 2048 and
// up) a workgroup takes G groups through every phase together: the resident weights serve G x 4 samples, the G LayerNorms run on G
// different waves side by side, and the phase latencies are paid once per G groups.
template <int KP, bool SIG, int G>
__device__ __forceinline__ void fcl_chain_fwd4_body(const FclView &v, const int blk, float *fcl_smem) {
  static_assert(!SIG || G == 1, "the in-launch hand-off counts four chain workgroups per sample group");
  constexpr int NS = 4 * G;                    // samples of this workgroup
  const int LDX = v.xq + 4;
  float *X = fcl_smem, *A1 = X + NS * LDX, *red = A1 + NS * FCL_LDA, *misc = red + G * 2048, *PV = misc + 16;
  float *b1r = PV, *b1t = PV + 512, *b2r = PV + 1024, *b2t = PV + 1088, *lnw = PV + 1152, *lnb = PV + 1216;
  int *acts = (int *)(PV + 1280);          // [NS samples][8]
  float *S = PV + 1408;                    // [NS samples][64] x-hat of the position just finished (for the tapes)
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int row0 = blk * NS, cb = row0 >> 4, n0 = row0 & 15, R = v.R;
  const size_t T64 = (size_t)64 * R, T512 = (size_t)512 * R, TX = (size_t)v.XR * R;
#define FCL_KSTAMP(k) if (v.prof && blk == 0 && tid == 0) v.prof[54 + (k)] = __builtin_amdgcn_s_memtime();
  FCL_KSTAMP(0)      // kernel start
  if constexpr (SIG) { if (v.prof && blk == 0 && tid == 0) v.prof[59] = __builtin_amdgcn_s_memrealtime(); }
  // Loads return in order: what position 0 needs -- the small parameter vectors, the observations, the representation's
  // fc2 weights -- is requested FIRST; the transition's resident weights (245 KB per workgroup) go out behind position 0's
  // fc1 products and arrive under the rest of position 0 (r05 requested them first: every workgroup's first barrier waited
  // 3.8 us for them)
  float WT1[KP], WT2[64], WR2[64], WR1[16];
  {
    const float pb1r = v.P[v.rep_b1 + tid], pb1t = v.P[v.tr_b1 + tid];
    float pb2r = 0.f, pb2t = 0.f, plnw = 0.f, plnb = 0.f;
    if (tid < MZ_H) { pb2r = v.P[v.rep_b2 + tid]; pb2t = v.P[v.tr_b2 + tid]; plnw = v.P[v.ln_w + tid]; plnb = v.P[v.ln_b + tid]; }
    int pact = -1;
    if (tid < 8 * NS && (tid & 7) < v.K) {
      const size_t ai = (size_t)(row0 + (tid >> 3)) * v.K + (tid & 7);
      pact = v.act_i32 ? ((const int32_t *)v.act)[ai] : (int)((const int64_t *)v.act)[ai];
    }
    const int f_0 = tid / NS, n_0 = tid % NS;
    const float ob0 = (tid < v.xq * NS && f_0 < v.O) ? v.obs[(size_t)(row0 + n_0) * v.O + f_0] : 0.f;
    // up to 16 observation features (LunarLander 8, TicTacToe 9): the representation's fc1 weights ride with these small reads, so that
    // position 0's fc1 products start behind ONE round trip (streamed, their loads queue behind WR2's 128 KB)
    if (v.O <= 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) WR1[k] = k < v.O ? (v.pk + v.rep.F1 + (size_t)w * v.O * 64)[k * 64 + lane] : 0.f;
    }
    asm volatile("" ::: "memory");
    fcl_quad_load<64>(WR2, v.pk + v.rep.F2 + (size_t)w * 64 * 64, lane);
    asm volatile("" ::: "memory");
    b1r[tid] = pb1r; b1t[tid] = pb1t;
    if (tid < 64) { b2r[tid] = pb2r; b2t[tid] = pb2t; lnw[tid] = plnw; lnb[tid] = plnb; }
    if (tid < 8 * NS) acts[tid] = pact;
    if (tid < v.xq * NS) {
      X[n_0 * LDX + f_0] = ob0;
      if (f_0 < v.XR) v.xin[fcl_tp(v.XR, cb, f_0, n0 + n_0)] = ob0;
    }
    for (int idx = tid + FCL_THREADS; idx < v.xq * NS; idx += FCL_THREADS) {      // (more observation features / samples than threads)
      const int f = idx / NS, n = idx % NS;
      const float val = f < v.O ? v.obs[(size_t)(row0 + n) * v.O + f] : 0.f;
      X[n * LDX + f] = val;
      if (f < v.XR) v.xin[fcl_tp(v.XR, cb, f, n0 + n)] = val;
    }
  }
  fcl_bar();
  FCL_KSTAMP(1)      // observations in LDS
  // (development: stamps of position 2's phases in workgroup 0, mz_fcl_heads_profile, slots 48..)
#define FCL_CSTAMP(k) if (v.prof && blk == 0 && p == 2 && tid == 0) v.prof[48 + (k)] = __builtin_amdgcn_s_memtime();
  auto rest = [&](int p, const f32x4 (&acc)[G], const float (&W2)[64], const float *b1, const float *b2) __attribute__((always_inline)) {
    FCL_CSTAMP(1)      // fc1 products done
#pragma unroll
    for (int g = 0; g < G; ++g) {   // fc1 epilogue: bias, ReLU, to LDS (one 16-byte write: the lane's four features of its sample) and to the tape
      const int f0 = 64 * w + 4 * (lane >> 2), loff = fcl_lane_off4(lane, n0 + 4 * g);
      const f32x4 b = *(const f32x4 *)(b1 + f0);
      float *a1t = v.a1c + (size_t)p * T512 + fcl_tp(512, cb, 64 * w, 0);
      f32x4 a;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = fmaxf(acc[g][i] + b[i], 0.f);
        (a1t + i * 16)[loff] = a[i];
      }
      *(f32x4 *)(A1 + (4 * g + (lane & 3)) * FCL_LDA + f0) = a;
    }
    // NO barrier here: fc2's split-K slice of wave w is features [64 w, 64 w + 64) of A1 -- exactly what wave w itself has just written
    // (a wave's LDS instructions execute in order); the waves drift apart and one's epilogue runs under its SIMD neighbour's MFMAs
    FCL_CSTAMP(2)      // epilogue
    // what the LayerNorm wave needs besides the partials, requested in front of the barrier it waits at
    f32x4 lb2 = (f32x4){0.f, 0.f, 0.f, 0.f}, llw = lb2, llb = lb2;
    int act_p = -1;
    if constexpr (G == 1) {
      if (w < 4) {        // (one sample per wave, lane = feature: see below)
        lb2[0] = b2[lane]; llw[0] = lnw[lane]; llb[0] = lnb[lane];
        act_p = acts[w * 8 + (p < 7 ? p : 7)];
      }
    } else if (w < G) {
      const int f0 = 4 * (lane & 15);
      lb2 = *(const f32x4 *)(b2 + f0); llw = *(const f32x4 *)(lnw + f0); llb = *(const f32x4 *)(lnb + f0);
      act_p = acts[(4 * w + (lane >> 4)) * 8 + (p < 7 ? p : 7)];
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
      *(f32x4 *)(red + g * 2048 + w * 256 + fcl_red_slot(lane & 3, lane >> 2) * 4) = fcl_quad_res<64>(W2, A1 + (4 * g + (lane & 3)) * FCL_LDA + 64 * w);
    if constexpr (SIG) {
      // the tapes of position p - 1 (stored write-through behind its last barrier, by the upper waves, two phases of MFMAs ago):
      // every storing wave waits for its stores HERE; one lane announces them behind the barrier that follows
      if (p > 0 && tid >= 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    fcl_bar();
    if constexpr (SIG) {
      if (p > 0 && tid == 0) fcl_signal(v.flags + (size_t)cb * (v.K + 1) + (p - 1));
    }
    FCL_CSTAMP(3)      // fc2 partials + barrier
    if constexpr (G == 1) {
      // ONE group of four samples: waves 0 .. 3 take one sample each, lane = feature -- a quarter of the per-lane work of the one-wave form
      // below, and the four samples' divisions and square roots side by side on four SIMDs (the one wave was 1.0 k of a position's 4.7 k
      // cycles with the other seven waiting); the reductions run over the wave's 64 lanes (fcl_sum64)
      if (w < 4) {
        float y = 0.f;
#pragma unroll
        for (int ww = 0; ww < FCL_NW; ++ww) y += red[ww * 256 + fcl_red_slot(w, lane >> 2) * 4 + (lane & 3)];      // (row `lane` of sample w: written by lane 4 b + w, register lane & 3)
        const bool real = lane < MZ_H;
        const float yv = real ? y + lb2[0] : 0.f;
        const float mean = fcl_sum64(yv) / (float)MZ_H;
        const float d = real ? yv - mean : 0.f;
        const float rstd = 1.0f / sqrtf(fcl_sum64(d * d) / (float)MZ_H + FCL_LN_EPS);
        const float xh = d * rstd;
        const float hv = real ? fmaxf(xh * llw[0] + llb[0], 0.f) : 0.f;
        // the next input: [h | one-hot(action) | 0]  (networks.py:167-174); past the last transition: no action
        X[w * LDX + lane] = real ? hv : ((p < v.K && lane - MZ_H == act_p) ? 1.f : 0.f);
        S[w * 64 + lane] = xh;
        if (lane == 0) misc[w] = rstd;
      }
    } else if (w < G) {
      // wave g alone for group g: lane (j, b) = (lane >> 4, lane & 15) adds up rows 4 b + i of sample j of the 8 partials (written by
      // lane 4 b + j of each wave into slot fcl_red_slot(j, b): both sides touch 16 different 16-byte slots per 16 lanes), then
      // LayerNorm + ReLU (networks.py:147,165) of sample j across ITS DPP ROW -- the two reductions are four DPP exchanges each, no
      // LDS round trip (r05: lane (b, j), two ds_bpermute per reduction: this serial part was 2.0 k of a position's 5.3 k cycles)
      // -- the tapes and the next input straight from registers
      const int j = lane >> 4, f0 = 4 * (lane & 15), sj = 4 * w + j;
      f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ww = 0; ww < FCL_NW; ++ww) y += *(const f32x4 *)(red + w * 2048 + ww * 256 + fcl_red_slot(j, lane & 15) * 4);
      float yv[4], s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { yv[i] = (f0 + i < MZ_H) ? y[i] + lb2[i] : 0.f; s += yv[i]; }
      s = fcl_sum16(s);
      const float mean = s / (float)MZ_H;
      float d[4], var = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { d[i] = (f0 + i < MZ_H) ? yv[i] - mean : 0.f; var += d[i] * d[i]; }
      var = fcl_sum16(var);
      const float rstd = 1.0f / sqrtf(var / (float)MZ_H + FCL_LN_EPS);
      f32x4 xv4, xh4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = f0 + i;
        xh4[i] = d[i] * rstd;
        const float hv = f < MZ_H ? fmaxf(xh4[i] * llw[i] + llb[i], 0.f) : 0.f;
        // the next input: [h | one-hot(action) | 0]  (networks.py:167-174); past the last transition: no action
        xv4[i] = f < MZ_H ? hv : ((p < v.K && f - MZ_H == act_p) ? 1.f : 0.f);
      }
      *(f32x4 *)(X + sj * LDX + f0) = xv4;
      *(f32x4 *)(S + sj * 64 + f0) = xh4;
      if ((lane & 15) == 0) misc[sj] = rstd;
    }
    FCL_CSTAMP(4)      // waves 0 .. G - 1: reduce + LayerNorm
    fcl_bar();
    FCL_CSTAMP(5)      // barrier
    // the tapes of this position, by the upper half of the workgroup (the LayerNorm waves above are the serial part: the others
    // wait for them); X, S and misc are next written two barriers from here
    if (tid >= 256) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int t = tid - 256, f = t >> 2, n = 4 * g + (t & 3);
        const float xv = X[n * LDX + f];
        const size_t tp = fcl_tp(64, cb, f, n0 + n);
        v.xhat[(size_t)p * T64 + tp] = S[n * 64 + f];
        if constexpr (SIG) {
          fcl_store_wt(v.h + (size_t)p * T64 + tp, f < MZ_H ? xv : 0.f);
          if (p < v.K) fcl_store_wt(v.xin + (size_t)(p + 1) * TX + fcl_tp(v.XR, cb, f, n0 + n), xv);
        } else {
          v.h[(size_t)p * T64 + tp] = f < MZ_H ? xv : 0.f;
          if (p < v.K) v.xin[(size_t)(p + 1) * TX + fcl_tp(v.XR, cb, f, n0 + n)] = xv;
        }
      }
      if (tid - 256 < NS) v.rstd[(size_t)p * R + row0 + tid - 256] = misc[tid - 256];
    }
  };
  {   // position 0: the representation, its fc1 weights streamed
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
      acc[g] = v.O <= 16 ? fcl_quad_res<16>(WR1, X + (4 * g + (lane & 3)) * LDX)
                         : fcl_quad_stream(v.pk + v.rep.F1 + (size_t)w * v.O * 64, v.O, X + (4 * g + (lane & 3)) * LDX, lane);
    asm volatile("" ::: "memory");
    // (the wait counter holds 63 loads: with the transition's 120 requests queued BEHIND WR2's, the first use of WR2 would wait
    // until all but 63 of them had returned too -- WR2, requested at the kernel's start, is settled BEFORE they go out)
    fcl_quad_settle(WR2);
    fcl_quad_load<KP>(WT1, v.pk + v.tr.F1 + (size_t)w * KP * 64, lane);
    fcl_quad_load<64>(WT2, v.pk + v.tr.F2 + (size_t)w * 64 * 64, lane);
    asm volatile("" ::: "memory");
    rest(0, acc, WR2, b1r, b2r);
  }
  FCL_KSTAMP(2)      // position 0 done
  fcl_quad_settle(WT1);
  fcl_quad_settle(WT2);
  FCL_KSTAMP(3)      // the transition's weights have arrived
  for (int p = 1; p <= v.K; ++p) {
    FCL_CSTAMP(0)
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = fcl_quad_res<KP>(WT1, X + (4 * g + (lane & 3)) * LDX);
    rest(p, acc, WT2, b1t, b2t);
  }
#undef FCL_CSTAMP
  FCL_KSTAMP(4)        // positions 1..K done
#undef FCL_KSTAMP
  if constexpr (SIG) {          // the last position's tapes
    if (tid >= 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fcl_bar();
    if (tid == 0) fcl_signal(v.flags + (size_t)cb * (v.K + 1) + v.K);
    if (v.prof && blk == 0 && tid == 0) v.prof[60] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int KP, int G>
__global__ __launch_bounds__(FCL_THREADS) void k_fcl_chain_fwd4(FclView v) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  fcl_chain_fwd4_body<KP, false, G>(v, blockIdx.x, fcl_smem);
}

// The forward pass of batches up to 256 as ONE launch: the chain's workgroups (blocks [0, nchain): dispatched first, always
// resident) and, behind them, every heads unit ordered by position (value / policy of position 0, then value / policy / reward of
// position 1, ...).  A unit starts as soon as its sample group's four chain workgroups have announced its position -- the heads of
// positions 0 .. K - 1 run on the 192 CUs the chain leaves idle WHILE the chain computes the later positions, and a waiting unit has
// its weights, biases and targets in flight.  As two launches the heads started when the whole chain had ended, 272 units on 256 CUs
// (a second round on 16 of them): 23.3 + 1.7 + 24.7 us; a unit never waits for anything but chain workgroups, which are resident
// from the start of the launch: no deadlock however few units fit beside them.
template <int KP>
__global__ __launch_bounds__(FCL_THREADS) void k_fcl_fwd(FclView v, int nchain) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  if ((int)blockIdx.x < nchain) {
    fcl_chain_fwd4_body<KP, true, 1>(v, blockIdx.x, fcl_smem);
    return;
  }
  const int G = v.bs >> 4;
  int u = (int)blockIdx.x - nchain, p, hd, cb;
  if (u < 2 * G) { p = 0; hd = u / G; cb = u - hd * G; }
  else {
    u -= 2 * G;
    p = 1 + u / (3 * G);
    const int r = u - (p - 1) * 3 * G;
    hd = r / G; cb = r - hd * G;
  }
  fcl_heads_body<true>(v, cb, p, hd, fcl_smem);
}

// WAITD: the chain workgroups of k_fcl_fb -- the backward pass follows the forward pass in the SAME workgroup, and the heads' d loss / d hidden
// of position p comes from units of the same launch: wave 0 reads the (sample group, position) counter (requested a phase early, so a
// counter that is already full costs no round trip), then loads what the units stored write-through with sc1 loads; the workgroup's
// own tapes of the forward pass (same CU, plain stores drained before the pass ended) are read past the L1 too
// PRE: WB2 and WB1 (the transition's transposed weights) were requested by the caller -- k_fcl_fb asks for them before the workgroup runs its
// heads unit, so that they are in registers when the backward pass begins
template <int G, bool WAITD, bool PRE>
__device__ __forceinline__ void fcl_chain_bwd4_core(const FclView &v, const int blk, float *fcl_smem, float (&WB2)[MZ_H], float (&WB1)[64]) {
  static_assert(!WAITD || G == 1, "the in-launch hand-off counts units per group of 16 samples");
  constexpr int NS = 4 * G;                          // samples of this workgroup: G groups of four (see the forward chain)
  const int LDX = v.xq + 4;
  float *X = fcl_smem, *A1 = X + NS * LDX, *red = A1 + NS * FCL_LDA, *PV = red + G * 2048 + 16;
  float *D2 = X;                                     // d (pre-LayerNorm output), sample-major like X (rows of >= 64 floats)
  float *lnw = PV;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int row0 = blk * NS, cb = row0 >> 4, n0 = row0 & 15, R = v.R, K1 = v.K + 1;
  const size_t T64 = (size_t)64 * R, T512 = (size_t)512 * R;
  float WR2[MZ_H];      // the representation's, for position 0
  const float plnw = tid < MZ_H ? v.P[v.ln_w + tid] : 0.f;
  // wave g < G carries group g's per-sample work in registers: lane (j, b) = (lane >> 4, lane & 15) = features 4 b + i of sample j
  // -- a sample's 16 lanes are one DPP row, its two LayerNorm reductions four DPP exchanges each (as in the forward chain)
  const int j = lane >> 4, f0 = 4 * (lane & 15), sj = 4 * w + j;
  float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dch = (f32x4){0.f, 0.f, 0.f, 0.f};         // d chain: gradient from the transition of position p + 1 into h_p
  float tv[4][5], trs = 0.f;
  const unsigned *dfl = v.flags + (size_t)(v.bs >> 4) * K1 + (size_t)cb * K1;      // (WAITD) finished units per position of this sample group
  auto ld = [&](const float *q) __attribute__((always_inline)) -> float {
    if constexpr (WAITD) return fcl_load_wt(q);
    else return *q;
  };
  auto request = [&](int p) __attribute__((always_inline)) {
    if (w < G) {
      const size_t o = fcl_tp(64, cb, f0, n0 + sj);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tv[i][0] = ld(v.dH + ((size_t)0 * K1 + p) * T64 + o + i * 16);
        tv[i][1] = ld(v.dH + ((size_t)1 * K1 + p) * T64 + o + i * 16);
        tv[i][2] = p < v.K ? ld(v.dH + ((size_t)2 * K1 + p + 1) * T64 + o + i * 16) : 0.f;
        tv[i][3] = ld(v.h + (size_t)p * T64 + o + i * 16);
        tv[i][4] = ld(v.xhat + (size_t)p * T64 + o + i * 16);
      }
      trs = ld(v.rstd + (size_t)p * R + row0 + sj);
    }
  };
  // (WAITD) the units of position p have all stored: fl = the counter as read a phase ago; short of its target, one lane polls (bounded)
  auto arrived = [&](int p, unsigned fl) __attribute__((always_inline)) {
    if constexpr (WAITD) {
      const unsigned want = p < v.K ? 3u : 2u;
      if (w < G && fl < want) {
        if (lane == 0) fcl_wait_flag(dfl + p, want, v.err);
        if (v.prof && blk == 0 && lane == 0) v.prof[14] += 1ull;
      }
      if (v.prof && blk == 0 && tid == 0 && p < 6) v.prof[p == 0 ? 15 : (p < 5 ? 27 + p : 44)] = __builtin_amdgcn_s_memrealtime();
      asm volatile("" ::: "memory");      // (the loads of the handed-off bytes stay behind the counter's read)
    }
  };
  for (int idx = tid; idx < NS * LDX; idx += FCL_THREADS) X[idx] = 0.f;
  if constexpr (!WAITD) request(v.K);
  // (loads return in order: the small reads above first, then the weights in the order of their first use)
  asm volatile("" ::: "memory");
  if constexpr (!PRE) fcl_quad_load<MZ_H>(WB2, v.pk + v.tr.B2 + (size_t)w * MZ_H * 64, lane);
  asm volatile("" ::: "memory");
  if constexpr (WAITD) {
    // every weight of the pass is requested before wave 0 waits for the last position's heads units
    if constexpr (!PRE) fcl_quad_load<64>(WB1, v.pk + v.tr.B1 + (size_t)w * 64 * 64, lane);
    asm volatile("" ::: "memory");
    arrived(v.K, 0u);
    if (v.prof && blk == 0 && tid == 0) v.prof[12] = __builtin_amdgcn_s_memrealtime();
    request(v.K);
  }
  if (tid < 64) lnw[tid] = plnw;
  fcl_bar();
  f32x4 mskn[G];
  auto mask_request = [&](int p) __attribute__((always_inline)) {
    const float *a1t = v.a1c + (size_t)p * T512 + fcl_tp(512, cb, 64 * w, 0);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) mskn[g][i] = ld((a1t + i * 16) + fcl_lane_off4(lane, n0 + 4 * g));
  };
  mask_request(v.K);
#define FCL_BSTAMP(k) if (v.prof && blk == 0 && p == 2 && tid == 0) v.prof[67 + (k)] = __builtin_amdgcn_s_memtime();
  auto body = [&](int p, const float (&W2)[MZ_H]) __attribute__((always_inline)) {
    FCL_BSTAMP(0)      // (development: the phases of backward position 2 in workgroup 0, slots 67..)
    unsigned fl = 0u;
    if constexpr (WAITD) {
      if (w < G && p > 0) fl = __hip_atomic_load(dfl + (p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (w < G) {
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
      s1 = fcl_sum16(s1); s2 = fcl_sum16(s2);
      const float inv = 1.f / (float)MZ_H;
      f32x4 dy;
#pragma unroll
      for (int i = 0; i < 4; ++i) dy[i] = (f0 + i < MZ_H) ? rstd * (dx[i] - s1 * inv - xh[i] * (s2 * inv)) : 0.f;
      *(f32x4 *)(D2 + sj * LDX + f0) = dy;
    }
    FCL_BSTAMP(1)      // wave 0: LayerNorm backwards
    if (p > 0) { arrived(p - 1, fl); request(p - 1); }
    FCL_BSTAMP(2)      // next position's requests out
    // (this position's fc1 activations -- the ReLU mask -- were requested one position ago: asked for here, in front of the barrier, they
    // were a round trip to L2 that the 400 cycles of fc2-transposed MFMAs did not cover)
    f32x4 msk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) msk[g] = mskn[g];
    fcl_bar();
    FCL_BSTAMP(3)      // barrier
    if (tid >= 256) {      // the delta tape of this position, by the upper half of the workgroup (D2 is next written behind the next barrier)
      const int t = tid - 256;
#pragma unroll
      for (int g = 0; g < G; ++g)
        v.d2c[(size_t)p * T64 + fcl_tp(64, cb, t >> 2, n0 + 4 * g + (t & 3))] = D2[(4 * g + (t & 3)) * LDX + (t >> 2)];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f32x4 acc = fcl_quad_res<MZ_H>(W2, D2 + (4 * g + (lane & 3)) * LDX);
      const int g0 = 64 * w + 4 * (lane >> 2), loff = fcl_lane_off4(lane, n0 + 4 * g);
      float *d1t = v.d1c + (size_t)p * T512 + fcl_tp(512, cb, 64 * w, 0);
      f32x4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i] = msk[g][i] > 0.f ? acc[i] : 0.f;
        (d1t + i * 16)[loff] = d[i];
      }
      *(f32x4 *)(A1 + (4 * g + (lane & 3)) * FCL_LDA + g0) = d;
    }
    FCL_BSTAMP(4)      // fc2-transposed products, mask, tapes
    if (p > 0) mask_request(p - 1);
    // (no barrier: wave w's split-K slice of the fc1-transposed product is features [64 w, 64 w + 64) of A1, its own writes)
    if (p >= 1) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        *(f32x4 *)(red + g * 2048 + w * 256 + fcl_red_slot(lane & 3, lane >> 2) * 4) = fcl_quad_res<64>(WB1, A1 + (4 * g + (lane & 3)) * FCL_LDA + 64 * w);
      FCL_BSTAMP(5)      // fc1-transposed partials
      fcl_bar();
      FCL_BSTAMP(6)      // barrier
      if (w < G) {
        f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < FCL_NW; ++ww) y += *(const f32x4 *)(red + w * 2048 + ww * 256 + fcl_red_slot(j, lane & 15) * 4);      // (written by lane 4 b + j)
        dch = y;            // (rows >= 50 come out of zero weights)
      }
      FCL_BSTAMP(7)      // wave 0: reduce
      // (red is next written after one more barrier; D2 after this one)
    }
  };
  // (the last position peeled off the loop: it waits for WB2 alone -- the wait counter holds 63 loads, so WB1's 64 (and the
  // representation's 50) are requested only once WB2 has been settled -- and they arrive under its first phases)
  fcl_quad_settle(WB2);
  if constexpr (!WAITD) fcl_quad_load<64>(WB1, v.pk + v.tr.B1 + (size_t)w * 64 * 64, lane);
  if constexpr (G == 1) fcl_quad_load<MZ_H>(WR2, v.pk + v.rep.B2 + (size_t)w * MZ_H * 64, lane);      // (G > 1: no registers to hold them through the loop)
  asm volatile("" ::: "memory");
  body(v.K, WB2);
  fcl_quad_settle(WB1);
  for (int p = v.K - 1; p >= 1; --p) body(p, WB2);
  if constexpr (G > 1) fcl_quad_load<MZ_H>(WR2, v.pk + v.rep.B2 + (size_t)w * MZ_H * 64, lane);
  body(0, WR2);      // position 0: the representation's fc2
  // LayerNorm weight / bias gradients of every group's 4 samples over all positions: one row of lnpart per GROUP (batch / 4 rows)
  if (w < G) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = dgam[i], b = dbet[i];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (j == 0 && f0 + i < 64) {
        v.lnpart[((size_t)blk * G + w) * 128 + f0 + i] = a;
        v.lnpart[((size_t)blk * G + w) * 128 + 64 + f0 + i] = b;
      }
    }
  }
  if constexpr (WAITD) {
    if (v.prof && blk == 0 && tid == 0) v.prof[13] = __builtin_amdgcn_s_memrealtime();
    if (v.prof && tid == 0) atomicMax(v.prof + 66, (unsigned long long)__builtin_amdgcn_s_memrealtime());      // (the last chain workgroup's end)
  }
}

template <int G, bool WAITD = false>
__device__ __forceinline__ void fcl_chain_bwd4_body(const FclView &v, const int blk, float *fcl_smem) {
  float WB2[MZ_H], WB1[64];
  fcl_chain_bwd4_core<G, WAITD, false>(v, blk, fcl_smem, WB2, WB1);
}

template <int G>
__global__ __launch_bounds__(FCL_THREADS) void k_fcl_chain_bwd4(FclView v) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  fcl_chain_bwd4_body<G>(v, blockIdx.x, fcl_smem);
}

// Forward chain, every heads unit AND the backward chain in ONE launch (batch <= 256: 64 chain workgroups on 256 CUs).  A chain
// workgroup runs the forward pass, then -- same workgroup, same CU, its backward weights requested while it waits -- the backward
// pass, position K first, as soon as the (at most three) heads units that produce d loss / d h_p of its sample group have announced it.
// The launch boundary between the two passes (all 272 units finished, the chip drained, a cold start) is gone, and only position K's
// units are on the critical path: the units of positions K - 1 .. 0 finish while the backward chain works its way down.
// Progress: a unit waits for chain workgroups' FORWARD passes only, and those never wait; a backward pass waits for units, every one of
// which gets a CU (the chain holds 64 of them) -- so the launch ends wherever all chain workgroups are resident, which the handle checks
// (2 x chain workgroups <= CUs); every wait is bounded (-> the pinned error word).
// (the kernel: behind the weight-gradient jobs it also carries)

// ------------------------------------------------------------------------------------------------ weight gradients (+ optimiser)
// One workgroup (NW waves) per job: the 16 x 64 strip G of dW = D . X^T (D: deltas, Mp features per row chunk; X: layer
// inputs, Np features) over EVERY unroll position the layer is applied at -- the positions' tapes lie a constant stride
// apart -- and the rows of one slab of the batch; out rows 16 tm .., out columns 64 ng ..; the bias gradient = row sums of
// D.  With ONE slab (batch <= 512) the strip IS the gradient of those weights, and the optimiser's update of exactly those
// weights follows in the same workgroup: no gradient round trip through HBM, no optimiser launch (r05: one strip per
// position into K + 1 slabs that k_fcl_adam added up: 16.7 + 9.4 us of the 92 us step).
struct FclJob {
  size_t d_off, x_off;      // float offsets of the two tapes at the layer's FIRST position (feature 0, row 0)
  size_t d_ps, x_ps;        // their strides from one position to the next
  size_t w_off, b_off;      // flat offsets of W [M][N] and of its bias (b_off used by the ng == 0 strip)
  int M, N, Mp, Np, tm, ng, npos, na, hd, pad_;        // Mp, Np: feature counts of the two tapes (their row-chunk strides / 16); tile: 16 na rows from row 16 na tm, columns from 16 ni ng (ni: the kernel's)
};

struct FclOpt {
  double beta1, beta2, eps, wd;
  float clip;
  int adamw, no_update;
};

struct FclDw {              // what a weight-gradient workgroup does with its strip
  const float *tapes;
  unsigned tape_bytes;      // (jobs of k_fcl_fb read the tapes through a buffer descriptor: sc1 loads)
  int R, S;                 // batch rows; row slabs (1: a strip is a gradient)
  float *part;              // fuse == 0: strips to part[slab][nflat]
  size_t nflat;
  float *grad;              // fuse == 1: the gradient (kept for mz_fcl_read_grad) ...
  int fuse;                 // ... and Adam / AdamW on the strip's weights, here
  float *P, *pk;
  const int32_t *posA, *posB;
  float *m, *vv;
  const float *steps, *lr_p;
  FclOpt o;
};

// Adam / AdamW on one weight, torch's fused-kernel arithmetic (utils.py:73-83: eps 1.5e-4; the hyper-parameters are doubles
// in torch's kernel and the moments' updates are evaluated in double there: 1 - 0.999 as a float is 4.7e-5 off); the new
// weight goes into the flat vector AND into the packed copies the next step's MFMAs read
__device__ __forceinline__ void fcl_adam_elem(const FclDw &a, size_t i, float g, float p, float ea, float es, int pa, int pb, float bc1,
                                              float bc2, double lr) {
  if (a.o.wd != 0.0) {
    if (a.o.adamw) p = (float)((double)p - lr * a.o.wd * (double)p);
    else g = (float)((double)g + (double)p * a.o.wd);
  }
  ea = (float)((double)ea + (1.0 - a.o.beta1) * ((double)g - (double)ea));            // torch lerp, weight < 0.5
  es = (float)(a.o.beta2 * (double)es + (1.0 - a.o.beta2) * (double)g * (double)g);
  const float step_size = (float)(lr / (double)bc1), bc2s = sqrtf(bc2);
  const float denom = (float)((double)(sqrtf(es) / bc2s) + a.o.eps);
  p -= step_size * ea / denom;
  a.m[i] = ea; a.vv[i] = es; a.P[i] = p;
  if (pa >= 0) a.pk[pa] = p;
  if (pb >= 0) a.pk[pb] = p;
}

// the two bias corrections of this step (steps[0] has been advanced by k_fcl_heads), by one lane, into LDS
__device__ __forceinline__ void fcl_bias_corr(const FclDw &a, float *dst) {
  const double step = (double)a.steps[0];
  dst[0] = (float)(1.0 - pow(a.o.beta1, step));
  dst[1] = (float)(1.0 - pow(a.o.beta2, step));
}

// A job's tile: NA x 16 rows of dW (NA delta fragments per row chunk) by NI x 16 columns (NI input fragments).  One wave's unit of
// work = one row chunk of one position: NA + NI loads of 16 bytes per lane feed 4 NA NI MFMAs.  Small batches want MANY small
// jobs (a job lives on one CU: its MFMAs bound its duration) -- NA = 1, NI = 2 or 4; large batches want few loads per MFMA (the
// kernel is bound by the tapes' way through L2: 16 x 64 strips re-read the input tape once per 16 rows of dW) -- NA up to 4.
#define FCL_DW_Q(NA, NI) ((NA) * (4 * (NI) + 1))                                   // floats per lane a wave leaves in LDS
#define FCL_DW_LDS(NW, NA, NI) (((NW) * FCL_DW_Q(NA, NI) * 64 + 4) * 4)            // bytes: the waves' partial tiles + the two bias corrections

// WT: a job of k_fcl_fb -- its tapes were written (through) by heads units and chain workgroups of the SAME launch: the job has its
// optimiser operands requested, then one lane waits for the head's counter of finished units (`wait_flag` >= `wait_for`), a barrier, and
// every tape load is an sc1 load (buffer loads: 16 bytes per lane as before)
// ONEBUF: ONE register set of NF units per wave -- where NF x NW covers the job's units (the chain layers' jobs of the step's last launch:
// 80 units, NF = 10) every tape load of the job is requested before its first MFMA: one round trip to HBM (the chain's deltas were written
// by other XCDs in the launch before) instead of three
template <int NW, int NA, int NI, int NF, bool FUSABLE, bool WT = false, bool ONEBUF = false>
__device__ __forceinline__ void fcl_dw_job(const FclJob *jp, int slab, const FclDw &a, float *sh, const unsigned *wait_flag = nullptr,
                                           unsigned wait_for = 0u, unsigned *err = nullptr) {
  constexpr int Q = FCL_DW_Q(NA, NI);
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g4 = lane >> 4, m16 = lane & 15;
  const FclJob j = *jp;
  if constexpr (!WT) { if (a.fuse && tid == NW * 64 - 1) fcl_bias_corr(a, sh + NW * Q * 64); }     // (a lane of the last wave: the first waves carry the remainder units)
  const int nch = a.R >> 4, per = (nch + a.S - 1) / a.S, c_lo = slab * per, c_hi = c_lo + per < nch ? c_lo + per : nch;
  const int nchs = c_hi > c_lo ? c_hi - c_lo : 0, U = j.npos * nchs;
  // tapes: [row chunk][feature][16 rows] -- 16 features x 16 rows of chunk c are one contiguous KiB
  const size_t Db = j.d_off + (size_t)c_lo * j.Mp * 16 + (size_t)(16 * NA * j.tm + m16) * 16 + 4 * g4;      // (float offsets into the tapes)
  const size_t Xb = j.x_off + (size_t)c_lo * j.Np * 16 + (size_t)(16 * NI * j.ng + m16) * 16 + 4 * g4;
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void *)a.tapes, 0, WT ? (int)a.tape_bytes : 0, 0x00020000);
  auto ld16 = [&](size_t off) __attribute__((always_inline)) -> f32x4 {
    if constexpr (WT) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, (int)(off * 4), 0, 16));      // (aux 16: sc1)
    else return *(const f32x4 *)(a.tapes + off);
  };
  const size_t dstr = (size_t)j.Mp * 16, xstr = (size_t)j.Np * 16;      // floats per row chunk
  f32x4 acc[NA][NI];
  float bsum[NA];
#pragma unroll
  for (int f = 0; f < NA; ++f) {
    bsum[f] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[f][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // unit u = (position u / nchs, chunk u % nchs); wave w takes u = w, w + NW, ...: a fixed assignment and a fixed order, so the
  // sum is deterministic (k index g4 of k-step jj of chunk c = row 16 c + 4 g4 + jj, in both operands).  The loads of
  // NF units are requested before the first MFMA of the group
  // the fused optimiser's operands (weight, moments, pack positions of this thread's elements of the tile) do not depend on the
  // gradient: requested NOW, they arrive under the tapes' loads and the MFMAs (requested after the reduction, they were one more
  // exposed round trip at the end of the step's last launch)
  constexpr int T_ = NW * 64, EW_ = NA * NI * 256, EB_ = NA * 16, EPT_ = (EW_ + EB_ + T_ - 1) / T_;
  // (the learning rate too: in the native loop it lives in the update's pinned staging -- read where it is used, behind the reduction, it
  // was a round trip over PCIe at the end of every job)
  const float lr_pre = (FUSABLE && a.fuse) ? *a.lr_p : 0.f;
  float pre_p[FUSABLE ? EPT_ : 1], pre_m[FUSABLE ? EPT_ : 1], pre_v[FUSABLE ? EPT_ : 1];
  int pre_a[FUSABLE ? EPT_ : 1], pre_b[FUSABLE ? EPT_ : 1];
  static_assert(!FUSABLE || NA == 1, "the fused optimiser works on 16-row tiles");
  if constexpr (FUSABLE) {
    if (a.fuse) {
#pragma unroll
      for (int k = 0; k < EPT_; ++k) {
        const int e = tid + k * T_;
        size_t ix = 0;
        bool ok_ = false;
        if (e < EW_) {
          const int q = e >> 6, l = e & 63, i = q >> 2;
          const int mm = 16 * j.tm + 4 * (l >> 4) + (q & 3), nn = 16 * (NI * j.ng + i) + (l & 15);
          ix = j.w_off + (size_t)mm * j.N + nn;
          ok_ = mm < j.M && nn < j.N;
        } else if (e < EW_ + EB_) {
          const int m = e - EW_;
          ix = j.b_off + 16 * j.tm + m;
          ok_ = j.ng == 0 && 16 * j.tm + m < j.M;
        }
        pre_p[k] = 0.f; pre_m[k] = 0.f; pre_v[k] = 0.f; pre_a[k] = -1; pre_b[k] = -1;
        if (ok_) { pre_p[k] = a.P[ix]; pre_m[k] = a.m[ix]; pre_v[k] = a.vv[ix]; pre_a[k] = a.posA[ix]; pre_b[k] = a.posB[ix]; }
      }
    }
  }
  // Two register sets in turn: the loads of the NEXT NF units are in flight while the MFMAs of this set run (one set: a wave
  // waited out a round trip to L2 / HBM in front of every NF units of MFMAs, and two waves per SIMD hide little of it)
  int p = 0, c = w;
  while (nchs > 0 && c >= nchs) { c -= nchs; ++p; }
  auto load = [&](f32x4 (&av)[NF][NA], f32x4 (&bv)[NF][NI], int u0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      if (u0 + k * NW < U) {
        const size_t Dp = Db + (size_t)p * j.d_ps + (size_t)c * dstr;
        const size_t Xp = Xb + (size_t)p * j.x_ps + (size_t)c * xstr;
#pragma unroll
        for (int f = 0; f < NA; ++f) av[k][f] = ld16(Dp + 256 * f);
#pragma unroll
        for (int i = 0; i < NI; ++i) bv[k][i] = ld16(Xp + 256 * i);
        c += NW;
        while (c >= nchs) { c -= nchs; ++p; }
      }
    }
  };
  auto comp = [&](const f32x4 (&av)[NF][NA], const f32x4 (&bv)[NF][NI], int u0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      if (u0 + k * NW < U) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
          for (int f = 0; f < NA; ++f)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[f][i] = fcl_mfma(av[k][f][jj], bv[k][i][jj], acc[f][i]);
        }
#pragma unroll
        for (int f = 0; f < NA; ++f) bsum[f] += (av[k][f][0] + av[k][f][1]) + (av[k][f][2] + av[k][f][3]);
      }
    }
  };
  if constexpr (WT) {
    // (the optimiser's operands above are in flight; the step counter was written through by chain workgroup 0 before its first signal)
    if (tid == 0) fcl_wait_flag<64>(wait_flag, wait_for, err);
    fcl_bar();
    if (a.fuse && tid == NW * 64 - 1) {
      const double step = (double)fcl_load_wt(a.steps);
      sh[NW * Q * 64] = (float)(1.0 - pow(a.o.beta1, step));
      sh[NW * Q * 64 + 1] = (float)(1.0 - pow(a.o.beta2, step));
    }
  }
  if constexpr (ONEBUF) {
    f32x4 avA[NF][NA], bvA[NF][NI];
    for (int u0 = w; u0 < U; u0 += NF * NW) {
      load(avA, bvA, u0);
      comp(avA, bvA, u0);
    }
  } else if (w < U) {
    f32x4 avA[NF][NA], bvA[NF][NI], avB[NF][NA], bvB[NF][NI];
    int u0 = w;
    load(avA, bvA, u0);
    for (;;) {
      const int u1 = u0 + NF * NW;
      if (u1 < U) load(avB, bvB, u1);
      comp(avA, bvA, u0);
      if (u1 >= U) break;
      const int u2 = u1 + NF * NW;
      if (u2 < U) load(avA, bvA, u2);
      comp(avB, bvB, u1);
      if (u2 >= U) break;
      u0 = u2;
    }
  }
  // a wave's partial tile in LDS: row q = (f * NI + i) * 4 + r of 64 lanes; the NA bias rows behind them
#pragma unroll
  for (int f = 0; f < NA; ++f) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sh[(w * Q + (f * NI + i) * 4 + r) * 64 + lane] = acc[f][i][r];
    sh[(w * Q + NA * NI * 4 + f) * 64 + lane] = bsum[f];
  }
  __syncthreads();
  // the tile's weights (+ biases): element e = (q, lane l) -> row 16 (NA tm + f) + 4 (l >> 4) + r, column 16 (NI ng + i) + (l & 15);
  // the waves' partials in wave order
  constexpr int T = NW * 64, EW = NA * NI * 256, EB = NA * 16, EPT = (EW + EB + T - 1) / T;
  auto element = [&](int e, float &g, size_t &idx) -> bool {
    g = 0.f;
    if (e < EW) {
      const int q = e >> 6, l = e & 63, fi = q >> 2, f = fi / NI, i = fi - f * NI;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) g += sh[(ww * Q + q) * 64 + l];
      const int mm = 16 * (NA * j.tm + f) + 4 * (l >> 4) + (q & 3), nn = 16 * (NI * j.ng + i) + (l & 15);
      idx = j.w_off + (size_t)mm * j.N + nn;
      return mm < j.M && nn < j.N;
    }
    if (e < EW + EB) {
      const int t = e - EW, f = t >> 4, m = t & 15;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) g += sh[(ww * Q + NA * NI * 4 + f) * 64 + m + 16 * gg];
      idx = j.b_off + 16 * (NA * j.tm + f) + m;
      return j.ng == 0 && 16 * (NA * j.tm + f) + m < j.M;
    }
    idx = 0;
    return false;
  };
  if (!FUSABLE || !a.fuse) {        // (a slab's tile: to part[slab]; no arrays of elements kept in registers)
    float *out = a.part + (size_t)slab * a.nflat;
    for (int e = tid; e < EW + EB; e += T) {
      float g;
      size_t idx;
      if (element(e, g, idx)) out[idx] = g;
    }
    return;
  }
  if constexpr (FUSABLE) {
    // fused optimiser: a thread's elements together, so that their loads (weight, moments, pack positions) are one round trip
    static_assert(EPT == EPT_, "the prefetched operands are this thread's elements");
    const float bc1 = sh[NW * Q * 64], bc2 = sh[NW * Q * 64 + 1];
    const double lr = (double)lr_pre;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      float g;
      size_t idx;
      if (element(tid + k * T, g, idx)) {
        a.grad[idx] = g;
        fcl_adam_elem(a, idx, g, pre_p[k], pre_m[k], pre_v[k], pre_a[k], pre_b[k], bc1, bc2, lr);
      }
    }
  }
}

// LayerNorm weight / bias gradient k (0 .. 2 MZ_H - 1): the chain workgroups' partials, eight independent chains so that the
// loads of a round are in flight together; the order of the sum is fixed
__device__ __forceinline__ float fcl_ln_grad(int k, const float *lnpart, int nwg) {
  const int col = k < MZ_H ? k : 64 + (k - MZ_H);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int wg = 0;
  for (; wg + 8 <= nwg; wg += 8) {
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) a[jx] += lnpart[(size_t)(wg + jx) * 128 + col];
  }
  for (; wg < nwg; ++wg) a[wg & 7] += lnpart[(size_t)wg * 128 + col];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// the three weighted loss means (learners.py:205-207,228-230) added to loss_acc (_loss_dev order: reward, value, policy), by
// one workgroup of 256+ threads; shd3: [3][256] doubles of LDS
__device__ __forceinline__ void fcl_loss_block(const float *lossb, const void *w, int w_f64, int bs, int K1, double *loss_acc, double *shd3) {
  // the three heads together: every thread's loads are independent (one round trip), one reduction of three doubles
  double acc[3] = {0.0, 0.0, 0.0};
  if (threadIdx.x < 256) {
    // four rows of a thread at a time: their 4 x 17 loads are in flight together (row by row, this block was 38 us of the step at
    // batch 2048: eight dependent round trips); per row the positions in order, per thread the rows in order: the sums' order is fixed
    for (int b0 = threadIdx.x; b0 < bs; b0 += 4 * 256) {
      float lv[4][3][FCL_MAXP];
      double wb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = b0 + r * 256;
        wb[r] = 0.0;
        if (b < bs) {
#pragma unroll
          for (int hd = 0; hd < 3; ++hd)
#pragma unroll
            for (int p = 0; p < FCL_MAXP; ++p) lv[r][hd][p] = (p < K1 && p >= (hd == 2 ? 1 : 0)) ? lossb[((size_t)hd * K1 + p) * bs + b] : 0.f;
          wb[r] = w_f64 ? ((const double *)w)[b] : (double)((const float *)w)[b];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (b0 + r * 256 < bs) {
#pragma unroll
          for (int hd = 0; hd < 3; ++hd) {
            float l = 0.f;
#pragma unroll
            for (int p = 0; p < FCL_MAXP; ++p)
              if (p < K1 && p >= (hd == 2 ? 1 : 0)) l += lv[r][hd][p];
            acc[hd] += wb[r] * (double)l;
          }
        }
      }
    }
#pragma unroll
    for (int hd = 0; hd < 3; ++hd) shd3[hd * 256 + threadIdx.x] = acc[hd];
  }
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) {
#pragma unroll
      for (int hd = 0; hd < 3; ++hd) shd3[hd * 256 + threadIdx.x] += shd3[hd * 256 + threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) loss_acc[threadIdx.x == 2 ? 0 : (threadIdx.x == 0 ? 1 : 2)] += shd3[threadIdx.x * 256] / (double)bs;
}

// (the heads' weight-gradient jobs ride in the same launch, behind the units: a job waits for the counter of its head's finished units and
// reads the tapes they wrote through -- it runs on the CUs the units have left, beside the backward chain, as it did in k_fcl_bwd_dw)
// (a chain workgroup that has finished its forward pass does not idle the ~13 us until its last position's units have finished: three of a
// sample group's four chain workgroups run THOSE units themselves -- value, policy, reward of position K: no CU waits 17 us for them with
// nothing to do -- and the fourth takes one of the other positions' units, which are handed out through a counter in position order)
template <int KP>
__global__ __launch_bounds__(FCL_THREADS) void k_fcl_fb(FclView v, int nchain, const FclJob *jobs, FclDw a) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  __shared__ int s_unit;
  float WB2[MZ_H], WB1[64];
  const int G = v.bs >> 4, K1 = v.K + 1, units = G * (3 * v.K - 1);      // (units of positions 0 .. K - 1: the queue)
  const bool chain = (int)blockIdx.x < nchain;
  if (!chain && (int)blockIdx.x - nchain >= units) {
    const FclJob *jp = jobs + ((int)blockIdx.x - nchain - units);
    const int hd = jp->hd;
    fcl_dw_job<FCL_NW, 1, 4, 4, true, true>(jp, 0, a, fcl_smem, v.flags + (size_t)2 * G * K1 + 32 * (1 + hd), (unsigned)(G * (hd == 2 ? v.K : K1)), v.err);      // (a 128-byte line per head's counter)
    if (v.prof && threadIdx.x == 0) atomicMax(v.prof + 64, (unsigned long long)__builtin_amdgcn_s_memrealtime());      // (development: the last job's end)
    return;
  }
  if (chain) {
    // (the optimiser's step counters advance here, written through by a storing wave of chain workgroup 0 -- drained before its first signal)
    if (blockIdx.x == 0 && (int)threadIdx.x >= 256 && (int)threadIdx.x < 256 + v.nsteps)
      fcl_store_wt(v.steps + (threadIdx.x - 256), v.steps[threadIdx.x - 256] + 1.f);
    fcl_chain_fwd4_body<KP, true, 1>(v, blockIdx.x, fcl_smem);
  }
  // the backward pass's resident weights: requested behind the unit's own first loads, in registers through the unit (the heads code
  // leaves them room)
  auto preload = [&]() __attribute__((always_inline)) {
    if (chain) {
      const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
      fcl_quad_load<MZ_H>(WB2, v.pk + v.tr.B2 + (size_t)w * MZ_H * 64, lane);
      fcl_quad_load<64>(WB1, v.pk + v.tr.B1 + (size_t)w * 64 * 64, lane);
    }
  };
  const bool own = chain && (blockIdx.x & 3) < 3;          // chain workgroup j < 3 of its sample group: head j of position K
  if (!own && threadIdx.x == 0) s_unit = (int)__hip_atomic_fetch_add(v.flags + (size_t)2 * G * K1 + 32 * 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int u = own ? 0 : s_unit;
  if (u < units) {
    int p = v.K, hd = (int)(blockIdx.x & 3), cb = (int)(blockIdx.x >> 2);
    if (!own) {        // position by position: 2 G units at position 0 (no reward head), 3 G at the others; value, policy, then reward
      p = 0;
      if (u >= 2 * G) { u -= 2 * G; p = 1 + u / (3 * G); u -= (p - 1) * 3 * G; }
      hd = u / G; cb = u - hd * G;
    }
    fcl_heads_body<true, true>(v, cb, p, hd, fcl_smem, preload);
  } else preload();
  if (v.prof && !chain && threadIdx.x == 0) atomicMax(v.prof + 65, (unsigned long long)__builtin_amdgcn_s_memrealtime());      // (the last unit's end)
  if (chain) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's own tapes (a1c, xhat, rstd: plain stores) are read back below
    fcl_bar();
    fcl_chain_bwd4_core<1, true, true>(v, blockIdx.x, fcl_smem, WB2, WB1);
  }
}

// the fused step's LAST launch (batch <= 512): the chain layers' weight-gradient jobs as 16 x 32 tiles, one slab, Adam in the
// workgroup; two more workgroups: the LayerNorm parameters (gradient from the chain workgroups' partials, Adam) and the loss sums
// (behind k_fcl_fb this launch carries EVERY job: the first nheads of the list are the heads' layers as 16 x 64 strips; it then also
// zeroes the hand-off counters for the next step -- the launch that used them has ended)
// (NFJ, MINW: tape loads in flight per wave and the register budget -- <4, 2>: 217 registers, one workgroup per CU (the chain layers' 224 jobs
// fit the chip); <2, 4>: 128 registers, two per CU, for the launch that carries all 392 jobs)
// (done_ctr / done_flag / seq: the native loop's completion word -- the workgroup that ends LAST (a device counter every workgroup adds to
// when it is done; that workgroup zeroes it again) stores `seq` into the update's pinned host word, system scope: every read of the
// update's staging slot and every store of its results lies before it.  The loop polls that word instead of recording an event behind
// every update: an event record between two updates cost 3.4 us of the GPU's timeline)
template <int NFJ, int MINW>
__global__ __launch_bounds__(FCL_THREADS, MINW) void k_fcl_dwa(const FclJob *jobs, int njobs, FclDw a, int tail, const float *lnpart, int nwg,
                                                                  size_t ln_w, const float *lossb, const void *w, int w_f64, int bs, int K1,
                                                                  double *loss_acc, int nheads, unsigned *flags, int nflags,
                                                                  unsigned *done_ctr, unsigned *done_flag, unsigned seq) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  const int b = blockIdx.x;
  if (b == 0 && flags)
    for (int i = threadIdx.x; i < nflags; i += FCL_THREADS) flags[i] = 0u;
  if (b < nheads) {
    fcl_dw_job<FCL_NW, 1, 4, 4, true>(jobs + b, 0, a, fcl_smem);
  } else if (b < njobs) {
    fcl_dw_job<FCL_NW, 1, 2, NFJ, true, false, (NFJ > 4)>(jobs + b, 0, a, fcl_smem);
  } else if (tail && b == njobs) {
    const int k = threadIdx.x;
    if (k == 2 * MZ_H) fcl_bias_corr(a, fcl_smem);
    float g = 0.f, pv = 0.f, ea = 0.f, es = 0.f;
    const float lr_pre = a.fuse ? *a.lr_p : 0.f;
    int pa = -1, pb = -1;
    const size_t i = ln_w + (size_t)(k < 2 * MZ_H ? k : 0);
    if (k < 2 * MZ_H) {
      pv = a.P[i]; ea = a.m[i]; es = a.vv[i]; pa = a.posA[i]; pb = a.posB[i];
      g = fcl_ln_grad(k, lnpart, nwg);
      a.grad[i] = g;
    }
    __syncthreads();
    if (k < 2 * MZ_H && a.fuse) fcl_adam_elem(a, i, g, pv, ea, es, pa, pb, fcl_smem[0], fcl_smem[1], (double)lr_pre);
  } else if (tail) {
    fcl_loss_block(lossb, w, w_f64, bs, K1, loss_acc, (double *)fcl_smem);
  }
  if (done_flag) {
    __syncthreads();      // (every thread's loads of the staging slot have returned, its stores have been issued)
    // (no fence: the word says that the staging slot has been READ -- the barrier above has every load back -- and that the launch
    // BEFORE this one, which stored the new errors into pinned memory, has ended; agent-scope release / acquire fences here cost 9 us)
    if (threadIdx.x == 0) {
      if (__hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// larger batches: every (job, row slab) is a workgroup of four waves; a job's tile is 16 NA rows x 64 columns, NA = the job's
// `na` (4 where the layer has >= 50 output rows: the input tape is read ONCE per 64 rows of dW instead of once per 16); the slabs'
// tiles go to part[slab] and are added up by k_fcl_adam
__global__ __launch_bounds__(256, 2) void k_fcl_dwt(const FclJob *jobs, int njobs, FclDw a, const float *lnpart, int nwg, float *lngrad) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  const int b = blockIdx.x;
  if (b == njobs * a.S) {
    // one more workgroup: the LayerNorm parameters' gradients from the chain workgroups' partials [nwg][128] -> lngrad [128].
    // Thread (half, col) adds the partials of every second workgroup, 16 loads in flight; the halves meet in LDS.  (Summed by
    // the optimiser kernel's one thread per parameter, batch 2048's 512 partials were 64 dependent round trips: 23 us)
    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
    float acc = 0.f;
    int wg = half;
    for (; wg + 2 * 15 < nwg; wg += 2 * 16) {
      float t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = lnpart[(size_t)(wg + 2 * k) * 128 + col];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += t[k];
    }
    for (; wg < nwg; wg += 2) acc += lnpart[(size_t)wg * 128 + col];
    fcl_smem[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 128) lngrad[threadIdx.x] = fcl_smem[threadIdx.x] + fcl_smem[128 + threadIdx.x];
    return;
  }
  if (b > njobs * a.S) return;
  const FclJob *jp = jobs + b / a.S;
  const int slab = b % a.S, na = jp->na;
  if (na == 4) fcl_dw_job<4, 4, 4, 2, false>(jp, slab, a, fcl_smem);
  else if (na == 2) fcl_dw_job<4, 2, 4, 3, false>(jp, slab, a, fcl_smem);
  else fcl_dw_job<4, 1, 4, 4, false>(jp, slab, a, fcl_smem);
}

// The backward chain (batch / 4 workgroups: 64 of 256 CUs at batch 256) and the heads' weight-gradient jobs -- they read only
// what k_fcl_heads wrote -- in ONE launch: the jobs (16 x 64 strips) run on the CUs the chain leaves idle (r05 tried the same
// overlap with a side stream and two events: the cross-stream waits cost what it saved)
__global__ __launch_bounds__(FCL_THREADS) void k_fcl_bwd_dw(FclView v, int nchain, const FclJob *jobs, FclDw a) {
  extern __shared__ __attribute__((aligned(16))) float fcl_smem[];
  if (blockIdx.x == 0 && v.flags)          // (the fused forward launch's arrival counters, for the next step)
    for (int i = threadIdx.x; i < v.nflags; i += FCL_THREADS) v.flags[i] = 0u;
  if ((int)blockIdx.x < nchain) fcl_chain_bwd4_body<1>(v, blockIdx.x, fcl_smem);
  else fcl_dw_job<FCL_NW, 1, 4, 4, true>(jobs + ((int)blockIdx.x - nchain), 0, a, fcl_smem);
}

// ------------------------------------------------------------------------------------------------ gradient, optimiser (unfused paths)
// one parameter's gradient: the slabs' strips in order (LayerNorm parameters: the chain workgroups' partials)
// (nwg < 0: lnpart is the reduced vector [128] k_fcl_dwt's last workgroup left -- columns as in the partials)
__device__ __forceinline__ float fcl_grad_of(size_t i, const float *part, int nslab, const float *lnpart, int nwg, size_t ln_w, size_t nflat) {
  if (i >= ln_w && i < ln_w + 2 * MZ_H) {
    const int k = (int)(i - ln_w);
    return nwg < 0 ? lnpart[k < MZ_H ? k : 64 + (k - MZ_H)] : fcl_ln_grad(k, lnpart, nwg);
  }
  float g = 0.f;
  int q = 0;
  for (; q + 8 <= nslab; q += 8) {          // eight slabs' loads in flight together, added in slab order
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = part[(size_t)(q + k) * nflat + i];
#pragma unroll
    for (int k = 0; k < 8; ++k) g += t[k];
  }
  for (; q < nslab; ++q) g += part[(size_t)q * nflat + i];
  return g;
}

// grad[i] = sum over the slabs' strips; per-block sum of squares for clip_grad_norm_ (launched only when clipping is on:
// without it k_fcl_adam adds the strips up itself)
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
// (learners.py:205-207,228-230).  steps[0] has been advanced by k_fcl_heads.
__global__ __launch_bounds__(256) void k_fcl_adam(float *P, float *pk, const int32_t *posA, const int32_t *posB, float *grad,
                                                  const float *part, int nslab, const float *lnpart, int nwg, size_t ln_w,
                                                  const float *bsq, int nblk, float *m, float *vv, const float *steps,
                                                  const float *lr_p, FclOpt o, size_t nflat, const float *lossb, const void *w,
                                                  int w_f64, int bs, int K1, double *loss_acc) {
  __shared__ float sh[256];
  __shared__ float shc[4];
  if ((int)blockIdx.x == nblk) {
    __shared__ double shd3[3 * 256];
    fcl_loss_block(lossb, w, w_f64, bs, K1, loss_acc, shd3);
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
