// mz_fused.hip.h -- the persistent fused search kernel: MCTS.run (reference mcts.py:78-102) for 16 trees
// per workgroup, ALL simulations in one launch.
//
//   workgroup = 4 wavefronts (one per SIMD) = 16 trees = the 16 columns of v_mfma_f32_16x16x4_f32.
//   per simulation:  [tree lanes] gather parent hidden + action  ->  [4 waves] dynamics + prediction on
//   the matrix cores  ->  [tree lanes] expand + backup + next descent.   Trees never leave their
//   workgroup, so there is no grid-wide synchronisation and no host round trip between simulations.
//
// Weight streaming.  Each wave consumes its share of the network (~180 KiB per simulation) as one cyclic
// stream of STEPS; a step = four 1-KiB pieces = the A operands of 16 MFMAs, laid out in consumption order
// (bias = the weight column of a constant-1 input, so there are no bias pieces).  The first steps stay resident
// in AGPRs for the whole launch; the others go L2 -> VGPR with buffer_load_dwordx4 into a ring of NB register
// buffers, NB-1 steps (>= 2000 cycles) ahead of use.  The whole per-simulation schedule is unrolled so that every
// buffer index is a compile-time constant; the stream is padded to a multiple of NB steps per simulation so the
// ring position is the same at the top of every simulation and the prefetch runs across simulation boundaries,
// barriers and the tree phase.  Every instruction in the MFMA stream costs issue time (the f32 "matrix core" IS
// the vector ALU), so a streamed step carries exactly five besides its MFMAs: four loads and one counted wait.
// Output rows that would fill a 16-row tile with padding (rows 48..49 of the hidden state, the policy head of a
// 4-action game) run on v_mfma_f32_4x4x1_16b_f32 instead (mz_mfma4_*).
// (Measured dead ends, kept out: LDS-DMA (global_load_lds) rings filled by the MFMA waves themselves -- each 1-KiB
// piece costs the issuing wave ~150 cycles of issue time; a wave-specialised 8-wave variant; see DESIGN.md.)
#pragma once
#include "mz_common.h"
#include "mz_net.hip.h"
#include "mz_tree.hip.h"
#include "mz_selfplay.hip.h"
#include <utility>

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
__device__ __forceinline__ void mz_static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void mz_static_for(F &&f) {
  mz_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#ifndef MZ_NB
#define MZ_NB 4      // register ring depth in steps (prefetch distance NB-1 steps = ~1600 cycles; A/B on one box:
#endif               // 4 buffers + 13 resident steps 406 us, 5 + 12: 410, 3 + 14: 407)
// s_waitcnt vmcnt(N), nothing else waited for (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
#define MZ_VMCNT(N) (((N) & 15) | (((N) >> 4) << 14) | 0x0F70)
// steps of the stream held in registers (AGPRs) across all simulations: less L2 traffic and no load issue in those
// steps (a streamed step runs at ~37 cycles/MFMA, a resident one at 32).  As many as the register file takes
// without scratch: the wide-action instantiations (two tree passes, two policy tiles) have fewer to spare.
#ifndef MZ_RS_MAIN
#define MZ_RS_MAIN 13
#endif
__host__ __device__ constexpr int mz_fused_rs(int ks1, int jtp) { return jtp > 1 ? 9 : (ks1 > 16 ? 11 : (ks1 > 14 ? 12 : MZ_RS_MAIN)); }
#define MZ_XE 36     // row stride of the x-tile extensions (k >= 50): dynamics [one-hot(action) | 0 ...], prediction [1 | 0 ...]

// per-simulation schedule (in steps of 16 MFMAs per wave)
template <int KS1, int JTP>
struct FusedSched {
  static constexpr int FC1 = KS1;               // dynamics fc1: K = 50 + A (the bias rides in the one-hot columns), 4 per step
  static constexpr int FC2 = 12;                // reward (2 tiles) + next hidden (4 tiles): 48 pieces
  static constexpr int P1 = (MZ_H + 1 + 3) / 4; // prediction fc1: K = 51 -> 13 steps
  static constexpr int P2 = 2 * (2 + JTP);      // value (2 tiles) + policy (JTP tiles)
  static constexpr int REAL = FC1 + FC2 + P1 + P2;
  static constexpr int RS = mz_fused_rs(KS1, JTP);            // the first RS steps stay resident in registers for the whole launch
  static constexpr int NRING = (REAL - RS + MZ_NB - 1) / MZ_NB * MZ_NB;   // streamed steps, padded to the ring depth
  static constexpr int NSTEPS = RS + NRING;     // schedule length incl. padding steps (prefetch only)
};

// MFMA from inline asm with the accumulator tied in place ("+a").  Left to itself hipcc, in this kernel,
// stages every accumulator through one scratch tile (4 v_accvgpr_mov per MFMA, all MFMAs serialised on
// it).  Its hazard recogniser does not look inside asm, so mz_mfma_fence() supplies the MFMA-result ->
// VALU wait states once per stage.
__device__ __forceinline__ void mz_mfma_a(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first product of an accumulation chain: SrcC = inline constant 0, so the tile is never zero-filled (the zero
// fill of AGPR tiles shows up as v_accvgpr_mov instructions from a zero tile, inside the MFMA stream)
__device__ __forceinline__ void mz_mfma_a0(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(b));
}
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per instruction (block = lane / 4: row r of the
// result comes from the A operand of lane 4 * block + r, column lane % 4 from the B operand of the lane itself;
// scripts/experiments/mfma4x4_check.hip).  With the B operand = a hidden tile as it stands (lane: tree = lane & 15, feature
// 16 t + 4 (lane >> 4) + r) one instruction multiplies FOUR output rows by the 16 trees over the four features the
// wave's four lane rows hold, at 11.5 cycles instead of 32: output layers with 4 (or 2) useful rows -- the policy head
// of a 4-action game, rows 48..49 of the 50-wide hidden state -- stop paying for a 16-row tile.  The result of lane
// (g, m) is the partial sum over its own lane row's features: the consumer adds up 4 waves x 4 lane rows.
__device__ __forceinline__ void mz_mfma4_a(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mz_mfma4_a0(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(b));
}
// fc1 accumulators live in arch VGPRs: ReLU is then one v_max in place and the tile is the B operand of the
// following layer as it stands (no v_accvgpr_read copies, no second register set).  The first k-step uses the
// inline constant 0 as SrcC, so the tiles are never zero-filled.
__device__ __forceinline__ void mz_mfma_v(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mz_mfma_va(f32x4 &c, float a, float b) {     // A operand from an AGPR
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mz_mfma_v0a(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mz_mfma_v0(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// MFMA results -> VALU: wait states after the LAST MFMA of a stage (the earlier ones are long done): 17, the 8-pass
// MFMA needs 11
__device__ __forceinline__ void mz_mfma_fence16v(f32x4 (&acc)[16]) {
  asm volatile("s_nop 15\n\ts_nop 0"
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),
                 "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
}
// VALU (ReLU) results -> MFMA SrcB: two wait states, once for the whole tile set
__device__ __forceinline__ void mz_valu_fence16v(f32x4 (&acc)[16]) {
  asm volatile("s_nop 1"
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),
                 "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
}
template <int N>
__device__ __forceinline__ void mz_mfma_fence(f32x4 (&acc)[N]) {
  static_assert(N >= 1 && N <= 6, "out tiles");
  // one wait for the LAST MFMA's result (24 wait states), all tiles tied to it
  if constexpr (N == 6) asm volatile("s_nop 15\n\ts_nop 0" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]) :: "memory");
  else if constexpr (N == 5) asm volatile("s_nop 15\n\ts_nop 0" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]) :: "memory");
  else if constexpr (N == 4) asm volatile("s_nop 15\n\ts_nop 0" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) :: "memory");
  else if constexpr (N == 3) asm volatile("s_nop 15\n\ts_nop 0" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]) :: "memory");
  else if constexpr (N == 2) asm volatile("s_nop 15\n\ts_nop 0" : "+a"(acc[0]), "+a"(acc[1]) :: "memory");
  else asm volatile("s_nop 15\n\ts_nop 0" : "+a"(acc[0]) :: "memory");
}

// ReLU as ONE instruction: fmaxf() (and the fmed3 builtin) on a value that comes out of inline asm is preceded by
// a canonicalising v_max x,x,x (maxnum semantics for signalling NaNs).  A float is negative exactly when its bit
// pattern is a negative int32, so max_i32(bits, 0) is relu(x) (-0 -> +0) with nothing to canonicalise.
__device__ __forceinline__ float mz_relu1(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// ReLU as a clamp, two elements per instruction.  gfx950 has no packed f32 maximum, but every VOP3P instruction has the
// clamp modifier: v_pk_add_f32 x, x, 0 clamp = min(max(x, 0), 1) on both halves of a register pair.  That is relu(x)
// wherever x <= 1 -- and the activations can be made to satisfy that without changing a single result bit: the host
// packs the fc1 layers of the weight stream (bias column included) times 2^-k and the layers that consume their
// activations times 2^k (k_relu_scale / k_pack_weights, mz_engine.hip).  Scaling by a power of two commutes with every
// floating-point product and sum (barring underflow below 2^(k-126) / overflow above 2^(127-k), both checked against on
// the host side of the choice of k), so the scaled fc1 accumulators are exactly 2^-k times the unscaled ones and the out
// layers' products exactly the unscaled products.  k is chosen per weight set so that 2^k exceeds an upper bound of
// every fc1 output: |W| . (bound of the LayerNorm'ed hidden state: |x_i - mean| / std <= sqrt(49), times |gamma_i|, plus
// |beta_i|) + the largest one-hot column + |bias|.  When no such k exists (non-finite or absurdly large weights)
// mz_set_weights routes the engine to the stand-alone kernels instead of this one.  64 VALU instructions per
// simulation fewer than v_max on every element: -0.8 % (A/B 412.1 -> 409.3 us per move).
typedef float f32x2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mz_relu_clamp(f32x4 &a) {
  f32x2r lo = __builtin_shufflevector(a, a, 0, 1), hi = __builtin_shufflevector(a, a, 2, 3);
  asm volatile("v_pk_add_f32 %0, %0, 0 clamp\n\tv_pk_add_f32 %1, %1, 0 clamp" : "+v"(lo), "+v"(hi));
  a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

__device__ __forceinline__ void mz_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float mz_xval(const float *xR, const float *xE, int m, int k) {
  const float *p = (k < MZ_H) ? (xR + m * MZ_HS + k) : (xE + m * MZ_XE + (k - MZ_H));
  return *p;
}
// the same read issued from inline asm, so that it STAYS where it is written (one fc1 step ahead of its use,
// in front of that step's 16 MFMAs) instead of being sunk by the scheduler to just before its consumer; the
// value is usable after mz_lds_wait(x).
__device__ __forceinline__ void mz_xval_async(float &x, const float *xR, const float *xE, int m, int k) {
  const float *p = (k < MZ_H) ? (xR + m * MZ_HS + k) : (xE + m * MZ_XE + (k - MZ_H));
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
  asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr));
}
__device__ __forceinline__ void mz_lds_wait(float &x) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x)); }
// the B operands of two consecutive k-steps inside the hidden columns, xR[m][4 st + g] and xR[m][4 st + 4 + g], with
// one ds_read2_b32 (xbase = LDS address of xR[m][g]; ST even): one read and one wait per two fc1 steps, one address
// register for all of them
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int ST>
__device__ __forceinline__ void mz_xpair_async(f32x2 &x, unsigned xbase) {
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(x) : "v"(xbase), "n"(4 * ST), "n"(4 * ST + 4));
}
__device__ __forceinline__ void mz_lds_wait2(f32x2 &x) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x)); }

// combine the four waves' split-K partial tiles (+ bias) into fin[n][m].  Partials are exchanged as one
// 16-byte vector per lane and tile (ds_write_b128 / ds_read_b128): 6 + 4 LDS instructions per lane instead of
// 24 + 30 dword ones.
template <int JTOT>
__device__ __forceinline__ void scombine(float *red, float *fin, const f32x4 (&out)[JTOT], const float *bias,
                                         int tid) {
  const int w = tid >> 6, lane = tid & 63;
  f32x4 *red4 = (f32x4 *)red;
#pragma unroll
  for (int jt = 0; jt < JTOT; ++jt) red4[(w * 6 + jt) * 64 + lane] = out[jt];
  mz_bar();
  for (int it = tid; it < JTOT * 64; it += 256) {
    const int jt = it >> 6, ln = it & 63;
    const int n0 = 16 * jt + 4 * (ln >> 4);
    f32x4 sum = *(const f32x4 *)(bias + n0);
    sum += red4[(0 * 6 + jt) * 64 + ln];
    sum += red4[(1 * 6 + jt) * 64 + ln];
    sum += red4[(2 * 6 + jt) * 64 + ln];
    sum += red4[(3 * 6 + jt) * 64 + ln];
    float *dst = fin + n0 * 16 + (ln & 15);
    dst[0] = sum[0]; dst[16] = sum[1]; dst[32] = sum[2]; dst[48] = sum[3];
  }
  mz_bar();
}

// 8-lane (half-row) all-reduce by DPP: quad_perm xor 1, xor 2, then row_half_mirror
template <int CTRL>
__device__ __forceinline__ float mz_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float mz_sum8(float x) {
  x += mz_dpp<0xB1>(x); x += mz_dpp<0x4E>(x); x += mz_dpp<0x141>(x);
  return x;
}
// (one v_max_f32_dpp per step from asm: through fmaxf the compiler emits v_mov_dpp + a canonicalising v_max + the
// v_max -- the operands here are sums of MFMA results, never signalling NaNs.  s_nop 1: VALU write -> DPP read)
__device__ __forceinline__ float mz_max8(float x) {
  asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "+v"(x));
  return x;
}

// 16-lane (full row) all-reduce by DPP: + row_mirror
__device__ __forceinline__ float mz_sum16(float x) {
  x += mz_dpp<0xB1>(x); x += mz_dpp<0x4E>(x); x += mz_dpp<0x141>(x); x += mz_dpp<0x140>(x);
  return x;
}
__device__ __forceinline__ float mz_max16(float x) {
  x = fmaxf(x, mz_dpp<0xB1>(x)); x = fmaxf(x, mz_dpp<0x4E>(x)); x = fmaxf(x, mz_dpp<0x141>(x));
  x = fmaxf(x, mz_dpp<0x140>(x));
  return x;
}

// relu(LayerNorm) of fin rows [row0,row0+50), column m -> xR[m][0..51]; 16 lanes per column
__device__ __forceinline__ void sln_relu16(const float *fin, float *xR, const float *lnw, const float *lnb, int row0,
                                           int m, int q) {
  float x[4], s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = q + 16 * i;
    x[i] = f < MZ_H ? fin[(row0 + f) * 16 + m] : 0.f;
    s += x[i];
  }
  s = mz_sum16(s);
  const float mean = s / (float)MZ_H;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float d = (q + 16 * i < MZ_H) ? x[i] - mean : 0.f;
    v += d * d;
  }
  v = mz_sum16(v);
  const float rstd = 1.0f / sqrtf(v / (float)MZ_H + 1e-5f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = q + 16 * i;
    if (f < MZ_HS) {
      float y = 0.f;
      if (f < MZ_H) y = fmaxf((x[i] - mean) * rstd * lnw[f] + lnb[f], 0.f);
      xR[m * MZ_HS + f] = y;
    }
  }
}

// Config.inverse_transform (config.py:27-33), column m, 16 lanes per column, S <= 32 bins.
// no_transform: 0 = transform, 1 = --no_target_transform, 2 = --no_support (S == 1; networks.py:153,161 return the
// head's scalar untouched)
__device__ __forceinline__ float mz_support_to_scalar16(const float *fin, int row0, int S, int smin, int no_transform,
                                                        int m, int q) {
  float x[2], mx = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int bin = q + 16 * i;
    x[i] = bin < S ? fin[(row0 + bin) * 16 + m] : -__builtin_inff();
    mx = fmaxf(mx, x[i]);
  }
  mx = mz_max16(mx);
  float e[2], sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    e[i] = (q + 16 * i < S) ? expf(x[i] - mx) : 0.f;
    sum += e[i];
  }
  sum = mz_sum16(sum);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) v += (float)(smin + q + 16 * i) * (e[i] / sum);
  if (no_transform == 2) v = (q == 0) ? x[0] : 0.f;   // --no_support: the head's single output as it is
  v = mz_sum16(v);
  if (!no_transform) {
    const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    float t = (fabsf(v) + 1.f) + 0.001f;
    t = 1.f + 0.004f * t;
    t = (sqrtf(t) - 1.f) / 0.002f;
    v = sgn * (t * t - 1.f);
  }
  return v;
}

// ---- epilogues straight from the split-K partials (no combined tile, one barrier less per epilogue).
// Every wave leaves its JTOT partial tiles in LDS as one 16-byte vector per lane and tile: the vector of lane
// (g, m) = rows 16jt+4g..+3 of column m sits at red4[(w*6+jt)*64 + 16g + (m ^ ((g + 4(jt&1)) & 7))]; a consumer
// lane then adds up bias + the four waves' partials of exactly the outputs it needs, in wave order.  The XOR
// spreads the eight row quads a column's consumer lanes fetch in one instruction over the eight 16-byte bank
// groups (unswizzled they all fall into bank group m mod 8: an 8-way conflict on every read).
// SMALL: bit mask of the tiles that hold 4-row results of the small MFMA (one vector per lane, kept in lane order)
template <int JTOT, int SMALL = 0>
__device__ __forceinline__ void mz_partials_out(float *red, const f32x4 (&out)[JTOT], int tid) {
  const int w = tid >> 6, lane = tid & 63;
  f32x4 *red4 = (f32x4 *)red;
#pragma unroll
  for (int jt = 0; jt < JTOT; ++jt) {
    if ((SMALL >> jt) & 1) red4[(w * 6 + jt) * 64 + lane] = out[jt];
    else red4[(w * 6 + jt) * 64 + (lane ^ (((lane >> 4) + 4 * (jt & 1)) & 7))] = out[jt];
  }
  mz_bar();
}
// The epilogue lanes fetch their inputs as 16-byte vectors (four consecutive output rows = one lane's slice of a
// partial tile) with ds_read_b128 issued from inline asm, all of them back to back, and wait ONCE.  Left to the
// compiler (VGPR budget exhausted in this kernel) every read was followed by its own s_waitcnt lgkmcnt(0): ~25
// exposed LDS round trips per epilogue.
// (One address register per base + immediate offsets: a separate "v" address per asm read would be loop
// invariant, get hoisted out of the simulation loop and spill.)
template <int OFF>
__device__ __forceinline__ void mz_lds128(f32x4 &dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ unsigned mz_lds_addr(const void *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}
struct MzQuad { f32x4 b, p0, p1, p2, p3; };      // bias + the four waves' partials of rows n0..n0+3, column m
// LDS byte address of wave 0's partial vector of rows n0..n0+3 (n0 % 4 == 0), column m
__device__ __forceinline__ unsigned mz_quad_addr(const float *red, int n0, int m) {
  const int jt = n0 >> 4, g = (n0 >> 2) & 3;
  return mz_lds_addr((const f32x4 *)red + (jt * 64 + g * 16 + (m ^ ((g + 4 * (jt & 1)) & 7))));
}
template <int BOFF>
__device__ __forceinline__ void mz_quad_issue(MzQuad &o, unsigned paddr, unsigned baddr) {
  mz_lds128<BOFF>(o.b, baddr);
  mz_lds128<0>(o.p0, paddr); mz_lds128<6 * 64 * 16>(o.p1, paddr); mz_lds128<12 * 64 * 16>(o.p2, paddr);
  mz_lds128<18 * 64 * 16>(o.p3, paddr);
}
__device__ __forceinline__ f32x4 mz_quad_sum(const MzQuad &o) {      // same order as the tile-wide combine
  f32x4 s = o.b;
  s += o.p0; s += o.p1; s += o.p2; s += o.p3;
  return s;
}
#define MZ_Q(o) "+v"(o.b), "+v"(o.p0), "+v"(o.p1), "+v"(o.p2), "+v"(o.p3)

// relu(LayerNorm) of output rows [row0,row0+50), column m -> row-major tile xR[m][0..51]; 8 lanes per column,
// lane q owns the feature quads q and q+8.  Rows up to row0+63 exist and are exact zeros beyond 50 (zero
// weights, bias and LayerNorm affine), so nothing is conditional but the variance term and the last store.
// SMALLT: rows 48..51 come out of the small MFMA (as described below); false: they are an ordinary 16-row tile.
// SINK: gets the lane's two output quads as well (features 4q.. and 32 + 4q..), e.g. to keep a second copy of the tile.
struct MzNoLnSink { __device__ __forceinline__ void operator()(int, int, const f32x4 &, const f32x4 &) const {} };
template <bool SMALLT = true, class SINK = MzNoLnSink>
__device__ __forceinline__ void sln_relu8p(const float *red, const float *bias, float *xR, const float *lnw,
                                           const float *lnb, int row0, int m, int q, const SINK &sink = SINK()) {
  MzQuad A, B;
  f32x4 wA, wB, bA, bB;
  const unsigned ba = mz_lds_addr(bias + row0 + 4 * q), wa = mz_lds_addr(lnw + 4 * q);
  mz_quad_issue<0>(A, mz_quad_addr(red, row0 + 4 * q, m), ba);
  // second quad, features 32 + 4q..: lanes q < 4 own rows of the 16-row tile row0/16 + 2 as before; features 48..51
  // come out of the small MFMA (tile row0/16 + 3's slot: one vector per wave and lane row g) -- lanes q = 4..7 add up
  // the four lane rows of wave q - 4 each, a quad reduction joins the waves, lane q = 4 keeps the sum (features >= 52
  // do not exist: exact zeros).  One address + one stride per lane serve both cases.
  const int jt5 = (row0 >> 4) + 3;
  const unsigned pb = (!SMALLT || q < 4) ? mz_quad_addr(red, row0 + 4 * q + 32, m)
                                         : mz_lds_addr((const f32x4 *)red + (((q - 4) * 6 + jt5) * 64 + m));
  const unsigned st = (!SMALLT || q < 4) ? 6u * 1024u : 256u;
  mz_lds128<128>(B.b, ba);
  mz_lds128<0>(B.p0, pb); mz_lds128<0>(B.p1, pb + st); mz_lds128<0>(B.p2, pb + 2 * st); mz_lds128<0>(B.p3, pb + 3 * st);
  asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(A), MZ_Q(B));
  const f32x4 xa = mz_quad_sum(A);
  f32x4 xb;
  {
    f32x4 ps = B.p0;
    ps += B.p1; ps += B.p2; ps += B.p3;
    f32x4 tot;
#pragma unroll
    for (int r = 0; r < 4; ++r) { float v = ps[r]; v += mz_dpp<0xB1>(v); v += mz_dpp<0x4E>(v); tot[r] = v; }   // over the quad q = 4..7
    const f32x4 big = B.b + ps, small = B.b + tot;
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[r] = (!SMALLT || q < 4) ? big[r] : ((q == 4) ? small[r] : 0.f);
  }
  // the affine parameters (lnb = lnw + 64 floats) arrive under the two reductions
  mz_lds128<0>(wA, wa); mz_lds128<128>(wB, wa);
  mz_lds128<256>(bA, wa); mz_lds128<384>(bB, wa);
  float s = ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xb[0] + xb[1]) + (xb[2] + xb[3]));
  s = mz_sum8(s);
  const float mean = s * (1.0f / (float)MZ_H);
  float v = 0.f;
  f32x4 da, db;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    da[r] = xa[r] - mean;                                      // features 4q+r < 32: always valid
    db[r] = (4 * q + 32 + r < MZ_H) ? xb[r] - mean : 0.f;
    v += da[r] * da[r];
    v += db[r] * db[r];
  }
  v = mz_sum8(v);
  // v_rsq_f32 (1 ulp) instead of a correctly rounded sqrt followed by a correctly rounded division (two roundings,
  // ~25 instructions on the epilogue's critical chain): either is within an ulp of the true value
  const float rstd = __builtin_amdgcn_rsqf(v * (1.0f / (float)MZ_H) + 1e-5f);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wA), "+v"(wB), "+v"(bA), "+v"(bB));
  f32x4 ya, yb;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ya[r] = fmaxf(da[r] * rstd * wA[r] + bA[r], 0.f);
    yb[r] = fmaxf(db[r] * rstd * wB[r] + bB[r], 0.f);          // f >= 50: affine is 0 -> 0
  }
  *(f32x4 *)(xR + m * MZ_HS + 4 * q) = ya;
  if (4 * q + 32 < MZ_HS) *(f32x4 *)(xR + m * MZ_HS + 4 * q + 32) = yb;
  sink(m, q, ya, yb);
}

// policy logits of a 4-action game out of the small MFMA (tile 2's slot of the prediction partials): lane tl of the
// tree's 16 lanes fetches the vector of (wave tl >> 2, lane row tl & 3), a 16-lane reduction adds them up: every
// lane ends up with all four logits (+ bias)
__device__ __forceinline__ f32x4 mz_logits4(const float *red, const float *bias, int m, int tl) {
  f32x4 pv, bv;
  mz_lds128<0>(pv, mz_lds_addr((const f32x4 *)red + (((tl >> 2) * 6 + 2) * 64 + 16 * (tl & 3) + m)));
  mz_lds128<0>(bv, mz_lds_addr(bias));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv), "+v"(bv));
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = bv[r] + mz_sum16(pv[r]);
  return o;
}

// Config.inverse_transform (config.py:27-33) of the 32 bins in raw (lane q of the 8 lanes of a column holds bins
// 4q..4q+3).  Bins beyond the support size arrive as MZ_PAD_BIN (their bias in LDS, weights zero): exp(d) of those is
// 2^(about -1.4e9) = 0 exactly, so they drop out of the maximum, the sum and the expectation without a mask
// (exact as long as the real logits stay within +-1e8).
// no_transform: 0 = transform, 1 = --no_target_transform, 2 = --no_support (one output; networks.py:153,161 return it as it is)
#define MZ_PAD_BIN (-1.0e9f)
__device__ __forceinline__ float mz_support_to_scalar_q(const f32x4 &raw, int smin, int no_transform, int q) {
  float mx = fmaxf(fmaxf(raw[0], raw[1]), fmaxf(raw[2], raw[3]));
  mx = mz_max8(mx);
  float e[4], sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // expf(d), d = x - max <= 0, written out: the device library's own arithmetic (hi/lo split of d * log2(e), exp2 of
    // the fraction, ldexp) without its overflow / underflow clamps -- d <= 0 cannot overflow, results below
    // 2^-149 may come out as that instead of 0
    const float d = raw[r] - mx;
    const float th = d * 0x1.715476p+0f;
    const float tl = __builtin_fmaf(d, 0x1.4ae0bep-26f, __builtin_fmaf(d, 0x1.715476p+0f, -th));
    const float ri = __builtin_rintf(th);
    const float y = __builtin_amdgcn_exp2f((th - ri) + tl);
    e[r] = __builtin_ldexpf(y, (int)ri);
    sum += e[r];
  }
  sum = mz_sum8(sum);
  const float rs = __builtin_amdgcn_rcpf(sum);      // (1 ulp; the probabilities were e * (1 / sum) already, not e / sum)
  // --no_support (one output, 31 padding bins): p is 1 for that output and 0 elsewhere, so with the output itself
  // in the place of bin 0's support value the expectation below returns it exactly (this select sits beside the
  // softmax chain, not in it)
  // (as arithmetic with a 0 / 1 scalar rather than a select: a lane mask held across the simulation loop costs two
  // scalar registers this kernel does not have; the host passes smin = 0 with --no_support)
  const float k2 = (no_transform == 2) ? 1.f : 0.f;
  const float c0 = __builtin_fmaf(k2, raw[0], (float)(smin + 4 * q));
  float v = c0 * (e[0] * rs);
#pragma unroll
  for (int r = 1; r < 4; ++r) v += (float)(smin + 4 * q + r) * (e[r] * rs);
  v = mz_sum8(v);
  // (computed unconditionally and selected: a branch here would end the basic block and with it the scheduler's
  // freedom to interleave this chain with the caller's other work)
  const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
  float t = (fabsf(v) + 1.f) + 0.001f;
  t = 1.f + 0.004f * t;
  t = (sqrtf(t) - 1.f) / 0.002f;
  const float vt = sgn * (t * t - 1.f);
  return no_transform ? v : vt;
}

#define MZ_FUSED_MAXPL 64   // search-path slots per tree kept in LDS: num_simulations + 2 <= 64
#define MZ_FUSED_LDS_BASE (16 * MZ_HS + 4 * 6 * 256 + 16 + 16 + 16 * 32 + 96 + 64 + 64 + 64 + 2 * 16 * MZ_XE + 16 * MZ_FUSED_MAXPL + 2 * MZ_FUSED_MAXPL)
// + the tree step's own staging [16][96] doubles, except beside large trees (LT = 2), where it shares the partials' space
// + [16][MZ_ENVW] words of per-environment state a whole-moves launch keeps across its moves (mz_root_body, envs)
#define MZ_ENVW 16
__host__ __device__ constexpr int mz_fused_lds_floats(int lt) { return MZ_FUSED_LDS_BASE + 16 * 96 * 2 + 16 * MZ_ENVW; }

// PROF: diagnostic build only (mz_search_phase_profile): per-wave cycle totals of each phase of the loop.
#define MZ_NPHASE 14
#define STAMP(ph)                                                              \
  if (PROF) {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
    __builtin_amdgcn_sched_barrier(0);                                         \
    pacc[ph] += now_ - tlast;                                                  \
    tlast = now_;                                                              \
  }

// dynamic LDS of the fused kernel: pb_c table, then the 16 trees' node arrays (lt = 1: everything + Q cache, rows of the
// table 64 apart; lt = 2: N, E, P, to_play per node, W, R, X per expansion slot, rows sims + 2 apart; lt = 0: table only)
__host__ __device__ inline size_t mz_fused_dyn_lds(int sims, int NN, int lt) {
  size_t b = (size_t)(sims + 2) * (lt == 2 ? sims + 2 : 64) * 8;
  if (lt == 1) b += (size_t)16 * NN * (8 + 8 + 8 + 4 + 2 + 2 + 1) + 64;
  if (lt == 2) b += (size_t)16 * NN * (8 + 2 + 2 + 1) + (size_t)16 * (sims + 2) * (8 + 8 + 4) + 64;
  return b;
}

// What Actor.play_game does after MCTS.run (actors.py:147-158) for one tree of the self-play loop, by the tree's
// own lanes at the end of the search launch while its nodes are still in LDS: Config.select_action,
// Game.store_search_statistics, root error, Game.apply on the synthetic env and the experience record
// (the stand-alone k_env_step_record does the same from the global pool).  Lane a stages child a's visit count
// in LDS, lane 0 then runs the reference's sequential arithmetic (mz_sample_index) on the staged vector.
// GAME: the environment is the device TicTacToe (mz_ttt_apply) instead of the synthetic one; the uniform may be the host's.
template <int TL, int LT, bool GAME = false>
__device__ __forceinline__ void mz_finalize_record(const TreeView &t, const TreeMem<LT> &tm, const SelfplayState &sp,
                                                   int b, int lane, uint32_t legal, uint64_t seed, double *stage,
                                                   int O, int *envs = nullptr) {
  // envs (LDS, optional): this environment's 8 words of state kept by the whole-moves launch (mz_root_body): read here
  // instead of five dependent global loads, updated for the next move of the launch
  const int A = t.A;
  const bool ok = lane < A && ((legal >> lane) & 1u);
  const int c = ok ? (int)tm.N[1 + lane] : 0;
  int sumv = c;
#pragma unroll
  for (int off = TL / 2; off >= 1; off >>= 1) sumv += __shfl_xor(sumv, off, TL);
  double *d = stage;
  int *acts = (int *)(stage + 32);
  const int pos = __popc(legal & ((1u << lane) - 1u));
  if (ok) { d[pos] = (double)c; acts[pos] = lane; }
  const unsigned long long move = envs ? *(const unsigned long long *)envs : sp.movecnt[b];
  float *rec = sp.ring + ((size_t)(move % (unsigned long long)sp.ring_moves) * t.B + b) * sp.rec_floats;
  const int OS = sp.obs_slots;       // float slots of the observation in the record (O, or ceil(O / 4) for packed bytes)
  if (envs && O <= MZ_ENVW - 8 && sp.obs_u8 != 2) { if (lane < O) rec[lane] = ((const float *)envs)[8 + lane]; }
  else for (int k = lane; k < OS; k += TL) rec[k] = mz_rec_obs_slot(sp, sp.obs + (size_t)b * O, O, k);
  if (lane < A) rec[OS + lane] = (float)(ok ? (double)c / (double)sumv : 0.0);
  if (lane == 0) {
    const int n = __popc(legal);
    const int N0 = (int)tm.N[0];
    const double rv = N0 == 0 ? 0.0 : tm.W[0] / (double)N0;
    const double err = rv - (double)(envs ? __builtin_bit_cast(float, envs[6]) : t.root_value[b]);
    const uint32_t env = (uint32_t)(sp.env_offset + b);
    const mz_u4 r = mz_philox(seed, env, (uint32_t)move, (uint32_t)(move >> 32), MZ_RNG_ACTION << 24);
    double u = mz_u01(r.x, r.y);
    if constexpr (GAME) { if (sp.draw_uniform) u = sp.draw_uniform[b]; }
    const int idx = mz_sample_index(d, n, envs ? *(const double *)(envs + 4) : sp.temp[b], u);
    const int action = acts[idx];
    if constexpr (GAME) {
      mz_ttt_apply(sp, b, action, rv, err, rec, A);
      sp.movecnt[b] = move + 1ull;
      return;
    }
    const int tt = envs ? envs[2] : sp.t[b], ep = envs ? envs[3] : sp.episode[b];
    const int done = (tt + 1 >= sp.episode_len) ? 1 : 0;
    mz_rec_put_double(rec + OS + A + 0, rv);
    mz_rec_put_double(rec + OS + A + 2, err);
    rec[OS + A + 4] = mz_synth_reward(seed, env, (uint32_t)ep, (uint32_t)tt);
    int32_t *ri = (int32_t *)(rec + OS + A + 5);
    ri[0] = action; ri[1] = done; ri[2] = tt; ri[3] = (int32_t)env; ri[4] = ep;
    if (done) {      // the next game starts: its temperature is evaluated now (actors.py:128-129)
      const double tn = *sp.temp_next;
      sp.t[b] = 0; sp.episode[b] = ep + 1; sp.temp[b] = tn;
      if (envs) { envs[2] = 0; envs[3] = ep + 1; *(double *)(envs + 4) = tn; }
    } else {
      sp.t[b] = tt + 1;
      if (envs) envs[2] = tt + 1;
    }
    sp.movecnt[b] = move + 1ull;
    if (envs) *(unsigned long long *)envs = move + 1ull;
  }
}

// (mz_root.hip.h) the root of a move for the 16 rows of a workgroup
template <int JTP, int G, bool SELFPLAY, bool GAME, class STAMPF>
__device__ __forceinline__ void mz_root_body(const NetView &n, const TreeView &t, const float *obs_in,
                                             const f32x4 *istream, int nst0, const SelfplayState &sp, uint64_t seed,
                                             double alpha, double frac, float *smem, int tid, double *root_stage,
                                             STAMPF stampf, int *envs = nullptr);
// HEAD instantiations: what the root needs, and how many moves the launch plays
struct MzRootArgs {
  const f32x4 *istream;      // the root network's weight stream (k_root's)
  int nst0;                  // steps of its run-time first stage
  int nmoves;                // moves per launch
  double alpha, frac;        // root_dirichlet_alpha, root_exploration_fraction
};

// HEAD (self-play loop only): ONE launch plays ra.nmoves whole moves of its 16 environments -- observation, initial
// inference, root expansion with Dirichlet noise and first descent (mz_root_body, on the LDS the trees are about to
// occupy), all simulations, then the end of the move -- so a move costs no kernel launch, no grid-wide drain between
// root and search, and the resident weight steps are loaded once per launch instead of once per move.  The
// workgroups drift apart freely: nothing is exchanged between them.
// GAME (HEAD, two players): the moves are those of the device TicTacToe environment (mz_root_body<.., GAME>, mz_ttt_apply).
template <int KS1, int JTP, int G, int LT, bool PROF, bool SP, bool HEAD = false, bool GAME = false>
__global__ __launch_bounds__(256, 1) void k_search_fused(NetView n, TreeView t, const f32x4 *wstream, int nsims,
                                                          int slot0, unsigned long long *prof, SelfplayState sp,
                                                          int record, uint64_t seed, MzRootArgs ra) {
  static_assert(!HEAD || (LT != 0 && !PROF), "HEAD: trees in LDS, no phase stamps");
  static_assert(!GAME || (HEAD && !SP), "GAME: whole moves of a two-player game environment");
  using SC = FusedSched<KS1, JTP>;
  constexpr int NB = MZ_NB, NSTEPS = SC::NSTEPS, RS = SC::RS, NRING = SC::NRING;
  static_assert(RS <= SC::FC1, "resident steps must be fc1 steps of the dynamics stage");
  constexpr int E_FC1 = SC::FC1, E_FC2 = E_FC1 + SC::FC2, E_P1 = E_FC2 + SC::P1, E_P2 = E_P1 + SC::P2;
  constexpr int NJ2 = 2 + JTP;
  constexpr bool P4 = (G == 4);          // at most 4 actions: the policy head runs on the small MFMA

  __shared__ __attribute__((aligned(16))) float smem[mz_fused_lds_floats(LT)];
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  double *s_pbc = (double *)dyn_lds;       // pb_c(Np, Nc) table (host-computed, exact), rows PBS entries apart
  const int PBS = (LT == 2) ? t.sims + 2 : 64;
  // (LT = 1, 2) the workgroup's 16 trees live in LDS for the whole launch (LT = 2: the descent's fields only)
  double *l_P = s_pbc + (t.sims + 2) * PBS;
  // LT = 1: every field per node; LT = 2: W, R and the X cache per expansion slot (mz_tree.hip.h, TreeMem)
  const int NV = (LT == 2) ? t.sims + 2 : t.NN;      // entries per tree of the value arrays
  double *l_Q = l_P + 16 * t.NN;                     // X cache
  double *l_W = l_Q + 16 * NV;
  float *l_R = (float *)(l_W + 16 * NV);
  int16_t *l_N = (int16_t *)(l_R + 16 * NV);
  int16_t *l_E = l_N + 16 * t.NN;
  int8_t *l_TP = (int8_t *)(l_E + 16 * t.NN);
  float *xR = smem;                       // [16][MZ_HS] x tile, row-major
  float *red = xR + 16 * MZ_HS;           // split-K partials [4][6][4][64]
  float *s_val = red + 4 * 6 * 256;
  float *s_rew = s_val + 16;
  float *s_lg = s_rew + 16;               // [16][32]
  float *s_b2 = s_lg + 16 * 32;
  float *s_b4 = s_b2 + 96;
  float *s_lnw = s_b4 + 64;
  float *s_lnb = s_lnw + 64;
  float *xEd = s_lnb + 64;                // [16][MZ_XE] dynamics extension: one-hot(action) (the bias rides in its weights)
  float *xEp = xEd + 16 * MZ_XE;          // [16][MZ_XE] prediction extension: 1 (bias column), then 0
  int *s_path = (int *)(xEp + 16 * MZ_XE); // [16][MZ_FUSED_MAXPL] pending search path of every tree
  double *s_rcp = (double *)(s_path + 16 * MZ_FUSED_MAXPL);      // [MZ_FUSED_MAXPL] 1 / n (mz_tree_backup_select_f)
  // [16][96] staging of the tree step: its own (the tree step of one wave overlaps other waves' epilogue), except
  // beside large trees, where it shares the partials' space and the tree step starts behind a barrier
  double *s_stage = (double *)(s_rcp + MZ_FUSED_MAXPL);
  int *s_env = (int *)(s_stage + 16 * 96);      // [16][MZ_ENVW], HEAD launches only

  const int tid0 = threadIdx.x;
  const int b0 = blockIdx.x * MZ_ROWS;
  const bool full = b0 + MZ_ROWS <= t.B;       // (wave-uniform) all 16 trees of this workgroup exist
  const size_t per_tree = (size_t)(t.sims + 1) * MZ_HS;

  // (the padding bins of the two 32-row support tiles get MZ_PAD_BIN as their bias: their softmax terms come out as
  // exact zeros in mz_support_to_scalar_q with no per-bin masking there)
  if (tid0 < 96) s_b2[tid0] = (tid0 < 32 && tid0 >= n.Sr) ? MZ_PAD_BIN : n.b2[tid0];
  if (tid0 < 32 + 16 * JTP) s_b4[tid0] = (tid0 < 32 && tid0 >= n.Sv) ? MZ_PAD_BIN : n.b4[tid0];
  if (tid0 < 64) { s_lnw[tid0] = n.lnw[tid0]; s_lnb[tid0] = n.lnb[tid0]; }
  if (tid0 < MZ_FUSED_MAXPL) s_rcp[tid0] = 1.0 / (double)(tid0 > 0 ? tid0 : 1);
  for (int i = tid0; i < 16 * MZ_XE; i += 256) xEp[i] = (i % MZ_XE == 0) ? 1.f : 0.f;
  for (int i = tid0; i < (t.sims + 2) * (t.sims + 2); i += 256) s_pbc[(i / (t.sims + 2)) * PBS + i % (t.sims + 2)] = t.pbctab[i];

  // tree-lane mapping: TL lanes per tree (16, or 32 when A > 16), 256/TL trees per pass
  constexpr int TL = (G <= 16) ? 16 : 32;
  constexpr int NPASS = 16 * TL / 256;

  unsigned long long pacc[MZ_NPHASE];
  unsigned long long tlast = 0;
  if (PROF) {
    for (int i = 0; i < MZ_NPHASE; ++i) pacc[i] = 0;
  }

  const int nmoves = HEAD ? ra.nmoves : 1;
  constexpr bool ENVS = HEAD && !GAME;      // per-environment scalars of the synthetic env kept in LDS across the moves
  if constexpr (ENVS) {
    if (tid0 < 16 && b0 + tid0 < t.B) {
      const int b = b0 + tid0;
      *(unsigned long long *)(s_env + tid0 * MZ_ENVW) = sp.movecnt[b];
      s_env[tid0 * MZ_ENVW + 2] = sp.t[b]; s_env[tid0 * MZ_ENVW + 3] = sp.episode[b];
      *(double *)(s_env + tid0 * MZ_ENVW + 4) = sp.temp[b];
    }
  }
  // HEAD launches: cycles per phase of a move (s_memtime), accumulated over the launch when `prof` is given
  // (mz_selfplay_phase_profile): 0 root's tree part, 1 ring + barrier, 2 simulations, 3 end of move, 4 root first
  // stage, 5 representation + LayerNorm, 6 prediction, 7 resident steps + tree set-up
  unsigned long long hs_t = 0, hs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define HSTAMP(k) if (HEAD && prof) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); hs_acc[k] += now_ - hs_t; hs_t = now_; }
  if (HEAD && prof) hs_t = __builtin_amdgcn_s_memtime();
  for (int mv = 0; mv < nmoves; ++mv) {
  if constexpr (HEAD) {
    // the root of this move (k_root's body) on the trees' LDS: the previous move's tail is done with it, this move's
    // trees are built from what the root leaves in the pool (same workgroup, same L1: a barrier publishes it)
    __syncthreads();
    // (thread index laundered here and below: nothing derived from it is loop-invariant to the compiler, so the root's
    // per-lane addresses are not kept in registers through the simulations, nor the search's through the root)
    int tid_r = threadIdx.x;
    asm volatile("" : "+v"(tid_r));
    // (nst0 is never negative.  The branch is there for the compiler: with the call unconditional it treats the
    // root's address arithmetic as invariants of the move loop and carries them through the simulations --
    // 413.5 vs 410.5 us per move, A/B on one box)
    if (ra.nst0 >= 0)
    mz_root_body<JTP, G, true, GAME>(n, t, nullptr, ra.istream, ra.nst0, sp, seed, ra.alpha, ra.frac,
                               (float *)(dyn_lds + (((t.sims + 2) * PBS * 8 + 15) & ~15)), tid_r, s_stage,
                               [&](int k) __attribute__((always_inline)) { HSTAMP(4 + k) }, ENVS ? s_env : nullptr);
    __syncthreads();
    HSTAMP(0)
  }
  int tid = threadIdx.x;
  if constexpr (HEAD) asm volatile("" : "+v"(tid));
  const int w = tid >> 6, lane = tid & 63;
  const int g4 = lane >> 4, m16 = lane & 15;
  const int tl = tid % TL;
  int my_slot[NPASS], my_act[NPASS];
  TreeRegs tr[NPASS];
  TreeMem<LT> tm[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int mt = tid / TL + i * (256 / TL);
    const int b = b0 + mt;
    if constexpr (LT == 1) {
      const int o = mt * t.NN;
      tm[i].N = l_N + o; tm[i].W = l_W + o; tm[i].P = l_P + o; tm[i].R = l_R + o; tm[i].E = l_E + o; tm[i].TP = l_TP + o;
      tm[i].X = l_Q + o;
    } else if constexpr (LT == 2) {
      const int o = mt * t.NN, ov = mt * NV;
      tm[i].N = l_N + o; tm[i].E = l_E + o; tm[i].P = l_P + o; tm[i].TP = l_TP + o;
      tm[i].X = l_Q + ov; tm[i].W = l_W + ov; tm[i].R = l_R + ov;
    } else {
      const size_t o = mz_slab(t, b < t.B ? b : 0);
      tm[i].N = t.N + o; tm[i].W = t.W + o; tm[i].P = t.P + o; tm[i].R = t.R + o; tm[i].E = t.E + o; tm[i].TP = t.TP + o;
    }
  }

  // hidden state of the pending descent's parent, 16 bytes per lane (lanes 13.. of a tree duplicate column 12);
  // refreshed by every descent (speculative gather, mz_tree_step_fused)
  f32x4 hv[NPASS];
  unsigned hoff[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int b = b0 + tid / TL + i * (256 / TL);
    hoff[i] = (unsigned)(((size_t)(b < t.B ? b : 0) * per_tree) * 4) + (unsigned)((tl < MZ_HS / 4 ? tl : MZ_HS / 4 - 1) * 16);
  }

  // this wave's stream: [NSTEPS][4 pieces][64 lanes] f32x4.  Wave-uniform base in SGPRs + per-lane byte
  // offset in one VGPR: every piece is then "s_base + const, v_off" (saddr form) and no per-piece 64-bit
  // VGPR address exists that the compiler could hoist out of the simulation loop and spill.
  // stream of one wave: [RS resident steps][NRING streamed steps], 4 pieces of 64 lanes x f32x4 each
  // Buffer addressing: the wave's stream is one buffer resource (4 SGPRs), the lane offset one VGPR, the piece a
  // scalar offset + immediate -- no VALU address arithmetic in the MFMA stream (with flat/global addressing the
  // compiler kept base + lane offset as a 64-bit VGPR pair and spent two VALU adds per step on it; every
  // non-MFMA instruction in a dense MFMA stream costs issue time).
  const char *wbase = (const char *)(wstream + (size_t)__builtin_amdgcn_readfirstlane(w) * NSTEPS * 256);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, NSTEPS * 4096, 0x00020000);
  const int lane_off = lane * 16;
#define MZ_BLOAD(byteoff) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off, (byteoff), 0))
  // the resident steps (per move in a HEAD launch: the root needs the register file for its own ring, and 208 KB from
  // L2 under the tree set-up below are cheaper than spilling around it)
  f32x4 Rw[RS][4];
#pragma unroll
  for (int s = 0; s < RS; ++s) {
#pragma unroll
    for (int p = 0; p < 4; ++p) Rw[s][p] = MZ_BLOAD((s * 4 + p) * 1024);
  }
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int mt = tid / TL + i * (256 / TL);
    const int b = b0 + mt;
    my_slot[i] = 0; my_act[i] = 0;
    tr[i].len = 1; tr[i].tp = 1; tr[i].root_tp = 1; tr[i].legal = 0; tr[i].mn = 0.0; tr[i].mx = 0.0; tr[i].root_n = 0;
    if constexpr (HEAD) {
      // the root this launch has just made (mz_tree_root, single player): a fresh Node(0) with its A children, the
      // first descent already taken -- everything but the priors and that descent's action is known without
      // reading the pool back (mz_tree_root left those two in the staging area)
      if (b < t.B) {
        const double *st = s_stage + mt * 96;
        const int best = (int)st[32];
        my_act[i] = best;
        tr[i].len = 2;
        tr[i].legal = (t.A >= 32) ? 0xFFFFFFFFu : ((1u << t.A) - 1u);
        if constexpr (GAME) {         // the root's mover and legal moves (mz_root_body left them in the staging area)
          tr[i].root_tp = (int)st[33];
          tr[i].tp = -tr[i].root_tp;                 // to_play at the leaf of the first descent (mcts.py:90-92)
          tr[i].legal = (uint32_t)st[34];
        }
        tr[i].mn = t.has_min ? t.min_bound : __builtin_inf();
        tr[i].mx = t.has_max ? t.max_bound : -__builtin_inf();
        if (tl == 0) { s_path[mt * MZ_FUSED_MAXPL] = 0; s_path[mt * MZ_FUSED_MAXPL + 1] = 1 + best; }
        const int have = 1 + t.A;
        if constexpr (LT == 1) {
          for (int k = tl; k < t.NN; k += TL) {
            tm[i].N[k] = 0; tm[i].W[k] = 0.0; tm[i].R[k] = 0.f; tm[i].E[k] = (k == 0) ? 0 : -1; tm[i].TP[k] = 1;
          }
          for (int k = tl; k < have; k += TL) tm[i].X[k] = 0.0;
        } else {
          for (int k = tl; k < have; k += TL) { tm[i].N[k] = 0; tm[i].E[k] = (k == 0) ? 0 : -1; tm[i].TP[k] = (GAME && k == 0) ? (int8_t)tr[i].root_tp : (int8_t)1; }
          if (tl == 0) { tm[i].W[0] = 0.0; tm[i].R[0] = 0.f; tm[i].X[0] = 0.0; }      // the root's expansion slot
        }
        for (int k = tl; k < have; k += TL) tm[i].P[k] = (k == 0) ? 0.0 : st[k - 1];
      }
    } else
    if (b < t.B) {
      my_slot[i] = t.slot[b];
      my_act[i] = t.act[b];
      tr[i].len = t.plen[b];
      tr[i].tp = t.leaf_tp[b];
      tr[i].root_tp = t.TP[mz_slab(t, b)];
      tr[i].root_n = t.N[mz_slab(t, b)];
      tr[i].legal = t.legal[b];
      tr[i].mn = t.mn[b];
      tr[i].mx = t.mx[b];
      for (int k = tl; k < tr[i].len; k += TL) s_path[mt * MZ_FUSED_MAXPL + k] = t.path[(size_t)b * t.PL + k];
      if constexpr (LT == 1) {       // bring the existing part of the tree (root, expanded slabs) into LDS
        const size_t o = mz_slab(t, b);
        const int have = 1 + (slot0 + 1) * t.A;
        // every node this launch can create starts as a fresh Node (mcts.py:30-37): visit_count 0, value_sum 0, reward 0,
        // no children, to_play 1 -- written once here, so that an expansion only has to set the priors
        for (int k = have + tl; k < t.NN; k += TL) {
          tm[i].N[k] = 0; tm[i].W[k] = 0.0; tm[i].R[k] = 0.f; tm[i].E[k] = -1; tm[i].TP[k] = 1;
        }
        for (int k = tl; k < have; k += TL) {
          tm[i].N[k] = (int16_t)t.N[o + k]; tm[i].W[k] = t.W[o + k]; tm[i].P[k] = t.P[o + k]; tm[i].R[k] = t.R[o + k];
          const double qk = t.N[o + k] > 0 ? t.W[o + k] / (double)t.N[o + k] : 0.0;      // a continued search: what the backup
          const double rk = (double)t.R[o + k];                                           // would have cached
          tm[i].X[k] = t.two_players ? rk - t.discount * qk : rk + t.discount * qk;
          tm[i].E[k] = (int16_t)t.E[o + k]; tm[i].TP[k] = t.TP[o + k];
        }
      } else if constexpr (LT == 2) {
        const size_t o = mz_slab(t, b);
        const int have = 1 + (slot0 + 1) * t.A;
        for (int k = tl; k < have; k += TL) {
          const int nk = t.N[o + k], ek = t.E[o + k];
          tm[i].N[k] = (int16_t)nk; tm[i].P[k] = t.P[o + k]; tm[i].E[k] = (int16_t)ek; tm[i].TP[k] = t.TP[o + k];
          if (ek >= 0) {       // an expanded node: its value fields live in its expansion slot
            const double qk = nk > 0 ? t.W[o + k] / (double)nk : 0.0;      // a continued search: what the backup would have cached
            const double rk = (double)t.R[o + k];
            tm[i].W[ek] = t.W[o + k]; tm[i].R[ek] = t.R[o + k];
            tm[i].X[ek] = t.two_players ? rk - t.discount * qk : rk + t.discount * qk;
          }
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < NPASS; ++i)
    hv[i] = *(const f32x4 *)((const char *)t.hpool + hoff[i] + (size_t)my_slot[i] * (MZ_HS * 4));

  // the streamed part follows the resident one: ring step r at byte RS*4096 + r*4096
  // (piece offset as the instruction's immediate, step offset as the scalar offset: one s_mov per step instead of four)
#define MZ_WLOAD(step, piece) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off + (piece) * 1024, sbase + (step) * 4096, 0))
  // scalar offset of ring step 0.  Laundered through asm at the top of every simulation: as a known constant the
  // compiler materialises one SGPR per step of the unrolled schedule (33 of them), spills other scalars to make room
  // and pays v_readlane reloads inside the MFMA stream; as an opaque base a step costs one s_add
  HSTAMP(7)
  int sbase = RS * 4096;
  f32x4 Bf[NB][4];
#pragma unroll
  for (int s = 0; s < NB - 1; ++s) {
#pragma unroll
    for (int p = 0; p < 4; ++p) Bf[s][p] = MZ_WLOAD(s, p);
  }

  if (PROF) tlast = __builtin_amdgcn_s_memtime();
  __syncthreads();
  HSTAMP(1)

  for (int sim = 0; sim < nsims; ++sim) {
    asm volatile("" : "+s"(sbase));
    int lane_e = lane;                  // epilogue lane index, laundered: the LDS addresses derived from it are
    asm volatile("" : "+v"(lane_e));    // recomputed every simulation instead of living in registers all along
    // ---- gather: x tile = [hidden of search_path[-2] | one-hot(action)]  (mcts.py:94-96; networks.py:167-174)
    // (the parent's hidden state was requested during the descent that chose it -- hv[], see mz_tree_step_fused)
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int mt = tid / TL + i * (256 / TL);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[i]));
      if (tl < MZ_HS / 4) *(f32x4 *)(xR + mt * MZ_HS + 4 * tl) = hv[i];
      // (only the columns the KS1 k-steps of the dynamics fc1 reach: k < 4 KS1)
      for (int c = tl; c < 4 * KS1 - MZ_H; c += TL) xEd[mt * MZ_XE + c] = (c == my_act[i] || (G == 4 && c == n.A)) ? 1.f : 0.f;
      // (column A, where it exists, meets zero weights: the bias rides in the one-hot columns, fill_fc1_foldbias.  The
      // 4-action instantiations still write it as 1: without the second compare their simulation loop came out 13
      // instructions longer and 0.8 % slower on the LunarLander shapes, with it the 8-lane instantiation is 1 % slower
      // on the Pong-ram shapes -- register allocation, not arithmetic)
    }
    STAMP(0)
    mz_bar();
    STAMP(1)

    // ---- network: dynamics + prediction (networks.py:31-34), NSTEPS steps fully unrolled
    f32x4 acc[16];       // fc1 tiles (arch VGPRs); after the in-place ReLU they are the hidden activations
    f32x4 out2[6];
    f32x4 out4[NJ2];
    float xq = 0.f;       // B operand of the NEXT fc1 step (read one step ahead; the steps beyond the hidden columns)
    f32x2 xpair[2];       // B operands of fc1 steps 0..11, two per read
    const unsigned xbase = mz_lds_addr(xR + m16 * MZ_HS + g4);
    mz_static_for<NSTEPS>([&](auto S_) __attribute__((always_inline)) {
      constexpr int s = decltype(S_)::value;
      // streamed steps: prefetch ring step r + NB - 1 (cyclic: the tail of a simulation prefetches the head of
      // the next); resident steps issue no loads
      // fc2 steps: the compiler's hazard recogniser wants one instruction between two groups of four MFMAs that
      // accumulate into the same four tiles and fills the slot with s_nop 0 (three per step) -- so those steps issue
      // their four prefetch loads one per slot instead of up front (SPREAD), and wait for their own pieces first
      constexpr bool SPREAD = (s >= E_FC1 && s < E_FC2) || (s >= E_P1 && s < E_P2);
      constexpr int pf_r = s >= RS ? s - RS : 0;
      constexpr int pf_ps = (pf_r + NB - 1) % NRING, pf_pb = (pf_r + NB - 1) % NB;
      if constexpr (s >= RS) {
        if constexpr (SPREAD) {
          __builtin_amdgcn_s_waitcnt(MZ_VMCNT(4 * (NB - 2)));     // (12 for NB = 5) this step's loads are not out yet
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) Bf[pf_pb][p] = MZ_WLOAD(pf_ps, p);
          // this step's own pieces were requested NB - 1 steps ago: exactly 4 (NB - 1) younger requests are allowed to
          // be outstanding.  ONE explicit wait per step -- left alone the compiler emits a counted wait in front of each
          // of the step's four pieces, and every instruction in the MFMA stream costs issue time.
          __builtin_amdgcn_s_waitcnt(MZ_VMCNT(4 * (NB - 1)));     // (16 for NB = 5)
        }
      }
      constexpr int cb = (s >= RS ? s - RS : 0) % NB;
      if constexpr (s < E_FC1 || (s >= E_FC2 && s < E_P1)) {
        // fc1 step: 16 tiles of this wave x one k-step; x from the tile (+ extension for k >= 50)
        constexpr bool dyn = s < E_FC1;
        constexpr int st = dyn ? s : s - E_FC2;
        constexpr int NST = dyn ? SC::FC1 : SC::P1;
        float x;
        if constexpr (st < 12) {        // k-steps 0..11 lie inside the 50 hidden columns: fetched in pairs
          f32x2 &cur = xpair[(st / 2) % 2];
          if constexpr (st % 2 == 0) {
            if constexpr (st == 0) mz_xpair_async<0>(cur, xbase);
            mz_lds_wait2(cur);
            x = cur[0];
            if constexpr (st + 2 < 12) mz_xpair_async<st + 2>(xpair[(st / 2 + 1) % 2], xbase);
          } else {
            x = cur[1];
            if constexpr (st == 11 && NST > 12) mz_xval_async(xq, xR, dyn ? xEd : xEp, m16, 4 * 12 + g4);
          }
        } else {                        // the last k-steps reach into the extension (one-hot / bias columns)
          mz_lds_wait(xq);
          x = xq;
          if constexpr (st + 1 < NST) mz_xval_async(xq, xR, dyn ? xEd : xEp, m16, 4 * (st + 1) + g4);
        }
#pragma unroll
        for (int tg = 0; tg < 4; ++tg) {
          if constexpr (s < RS) {
            if constexpr (st == 0) {
              mz_mfma_v0a(acc[4 * tg + 0], Rw[s][tg][0], x);
              mz_mfma_v0a(acc[4 * tg + 1], Rw[s][tg][1], x);
              mz_mfma_v0a(acc[4 * tg + 2], Rw[s][tg][2], x);
              mz_mfma_v0a(acc[4 * tg + 3], Rw[s][tg][3], x);
            } else {
              mz_mfma_va(acc[4 * tg + 0], Rw[s][tg][0], x);
              mz_mfma_va(acc[4 * tg + 1], Rw[s][tg][1], x);
              mz_mfma_va(acc[4 * tg + 2], Rw[s][tg][2], x);
              mz_mfma_va(acc[4 * tg + 3], Rw[s][tg][3], x);
            }
          } else if constexpr (st == 0) {
            mz_mfma_v0(acc[4 * tg + 0], Bf[cb][tg][0], x);
            mz_mfma_v0(acc[4 * tg + 1], Bf[cb][tg][1], x);
            mz_mfma_v0(acc[4 * tg + 2], Bf[cb][tg][2], x);
            mz_mfma_v0(acc[4 * tg + 3], Bf[cb][tg][3], x);
          } else {
            mz_mfma_v(acc[4 * tg + 0], Bf[cb][tg][0], x);
            mz_mfma_v(acc[4 * tg + 1], Bf[cb][tg][1], x);
            mz_mfma_v(acc[4 * tg + 2], Bf[cb][tg][2], x);
            mz_mfma_v(acc[4 * tg + 3], Bf[cb][tg][3], x);
          }
        }
        if constexpr (s == E_FC1 - 1 || s == E_P1 - 1) {
          // (ascending tile order = the order of the stage's last 16 MFMAs: 30 instructions lie between the last MFMA and
          // the read of its tile, more than the 17 wait states its result needs -- no fence in front)
#ifndef MZ_RELU_VMAX
#pragma unroll
          for (int tt = 0; tt < 16; ++tt) mz_relu_clamp(acc[tt]);
#else      // (development switch: v_max on an unscaled stream -- scripts/experiments/relu_clamp_check.py compares the two builds)
          mz_mfma_fence16v(acc);
#pragma unroll
          for (int tt = 0; tt < 16; ++tt) {
            acc[tt][0] = mz_relu1(acc[tt][0]); acc[tt][1] = mz_relu1(acc[tt][1]);
            acc[tt][2] = mz_relu1(acc[tt][2]); acc[tt][3] = mz_relu1(acc[tt][3]);
          }
#endif
          // VALU write -> MFMA SrcB read needs wait states the compiler cannot know about (the MFMAs are asm)
          mz_valu_fence16v(acc);
          if constexpr (s == E_FC1 - 1) { STAMP(2) } else { STAMP(6) }
        }
      } else if constexpr (s < E_FC2) {
        // dynamics fc2: pieces in (t, jt) order, 6 tiles per hidden tile: jt 0,1 reward (hid[t]), 2..5 next hidden (hid[8+t])
        constexpr int step = s - E_FC1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int q = 4 * step + q4, tt = q / 6, jt = q % 6;
            if (jt == 5) {         // rows 48..51 of the next hidden state: small MFMA
              if (q < 6 && r == 0) mz_mfma4_a0(out2[jt], Bf[cb][q4][r], acc[8 + tt][r]);
              else mz_mfma4_a(out2[jt], Bf[cb][q4][r], acc[8 + tt][r]);
            } else if (q < 6 && r == 0) mz_mfma_a0(out2[jt], Bf[cb][q4][r], acc[jt < 2 ? tt : 8 + tt][r]);
            else mz_mfma_a(out2[jt], Bf[cb][q4][r], acc[jt < 2 ? tt : 8 + tt][r]);
          }
          Bf[pf_pb][r] = MZ_WLOAD(pf_ps, r);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (s == E_FC2 - 1) {
          mz_mfma_fence<6>(out2);
          STAMP(3)
          mz_partials_out<6, 1 << 5>(red, out2, tid);
          STAMP(4)
          // waves 0,1: LayerNorm+ReLU of 8 trees each -> xR; waves 2,3: reward scalar of 8 trees each -> s_rew
          // (8 lanes per tree; the two chains run side by side on different SIMDs)
          {
            const int col = 8 * (w & 1) + (lane_e >> 3), q = lane_e & 7;
            if (w < 2) {
              sln_relu8p(red, s_b2, xR, s_lnw, s_lnb, 32, col, q);
            } else {
              MzQuad Q;
              mz_quad_issue<0>(Q, mz_quad_addr(red, 4 * q, col), mz_lds_addr(s_b2 + 4 * q));
              asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(Q));
              const float r = mz_support_to_scalar_q(mz_quad_sum(Q), n.rmin, n.no_transform, q);
              if (q == 0) s_rew[col] = r;
            }
          }
          mz_bar();
          STAMP(5)
          if (tid < 16 * (MZ_HS / 4)) {      // next hidden state -> pool slot of this expansion
            const int m = tid / (MZ_HS / 4), c = tid % (MZ_HS / 4);
            f32x4 *dst = (f32x4 *)(t.hpool + (size_t)(b0 + m) * per_tree + (size_t)(slot0 + sim + 1) * MZ_HS);
            dst[c] = *(const f32x4 *)(xR + m * MZ_HS + 4 * c);
          }
        }
      } else if constexpr (s >= E_P1 && s < E_P2) {
        // prediction fc2: NJ2 tiles per hidden tile: jt 0,1 value (hid[t]), 2.. policy (hid[8+t])
        constexpr int step = s - E_P1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int q = 4 * step + q4, tt = q / NJ2, jt = q % NJ2;
            if (P4 && jt == 2) {   // the four policy logits: small MFMA
              // two small MFMAs on the same accumulator back to back (the step's last piece at r, its first at r + 1):
              // a 2-pass MFMA's result is not forwarded to an immediately following SrcC read, and the compiler does
              // not see inside asm -- without these wait states the second product went missing
              if (q4 == 0 && r > 0 && (4 * step + 3) % NJ2 == 2) asm volatile("s_nop 7" : "+a"(out4[jt]));
              if (q < NJ2 && r == 0) mz_mfma4_a0(out4[jt], Bf[cb][q4][r], acc[8 + tt][r]);
              else mz_mfma4_a(out4[jt], Bf[cb][q4][r], acc[8 + tt][r]);
            } else if (q < NJ2 && r == 0) mz_mfma_a0(out4[jt], Bf[cb][q4][r], acc[jt < 2 ? tt : 8 + tt][r]);
            else mz_mfma_a(out4[jt], Bf[cb][q4][r], acc[jt < 2 ? tt : 8 + tt][r]);
          }
          Bf[pf_pb][r] = MZ_WLOAD(pf_ps, r);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (s == E_P2 - 1) {
          mz_mfma_fence<NJ2>(out4);
          STAMP(7)
          // every wave's hidden-state stores must have landed before the tree lanes may gather them: they are older
          // than the last 16 weight loads, so vmcnt(16) covers them without draining the prefetch ring; the barrier
          // inside mz_partials_out then publishes them
          __builtin_amdgcn_s_waitcnt(MZ_VMCNT(4 * (NB - 1)));
          mz_partials_out<NJ2, P4 ? 1 << 2 : 0>(red, out4, tid);
          STAMP(8)
          if constexpr (TL != 16) {
              // every wave: value scalar + policy logits of 4 trees -> LDS; both halves of a tree's 16 lanes compute the
              // value (8 lanes x 4 bins), lane q < ceil(A/4) forwards 4 logits
            const int col = 4 * w + (lane_e >> 4), q = lane_e & 15, q8 = q & 7;
            MzQuad V, L;
            const unsigned ba = mz_lds_addr(s_b4 + 4 * q8);
            mz_quad_issue<0>(V, mz_quad_addr(red, 4 * q8, col), ba);
            asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(V));
            const f32x4 vs = mz_quad_sum(V);
            if constexpr (!P4) mz_quad_issue<128>(L, mz_quad_addr(red, 32 + 4 * q8, col), ba);      // arrives under the value chain
            const float v = mz_support_to_scalar_q(vs, n.vmin, n.no_transform, q8);
            if (q == 0) s_val[col] = v;
            if constexpr (!P4) asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(L));
            if constexpr (P4) {
              const f32x4 lg4 = mz_logits4(red, s_b4 + 32, col, q);
              if (q == 0) *(f32x4 *)(s_lg + col * 32) = lg4;
            } else {
              if (q < 8 && 4 * q < n.A) *(f32x4 *)(s_lg + col * 32 + 4 * q) = mz_quad_sum(L);
            }
          }
        }
      }
      // s >= E_P2: padding steps (prefetch only)
    });
    if constexpr (TL == 16) {
      // ---- value scalar, logits, then the tree step (expand + backup, mcts.py:97-99; next descent, mcts.py:83-92) of
      // the 4 trees this wave's lanes own, with no barrier in between: nothing here is exchanged across waves (the
      // wave that owns a tree reads its value and logits straight from the partials), and written as ONE straight
      // line so that the two long dependent chains -- f32 softmax + inverse transform of the value, f64 exp +
      // normalisation of the priors -- interleave in the instruction stream instead of running back to back.
      const int mt = tid / TL;
      const int q8 = tl & 7;
      MzQuad V, L;
      const unsigned ba = mz_lds_addr(s_b4 + 4 * q8);
      mz_quad_issue<0>(V, mz_quad_addr(red, 4 * q8, mt), ba);
      if constexpr (P4) {
        asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(V));
      } else {
        mz_quad_issue<128>(L, mz_quad_addr(red, 32 + 4 * q8, mt), ba);
        asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(V), MZ_Q(L));
      }
      float lgl;          // this lane's logit (lane a = action a)
      if constexpr (P4) {
        const f32x4 lg4 = mz_logits4(red, s_b4 + 32, mt, tl);
        lgl = (tl & 2) ? ((tl & 1) ? lg4[3] : lg4[2]) : ((tl & 1) ? lg4[1] : lg4[0]);
      } else {
        if (tl < 8 && 4 * tl < n.A) *(f32x4 *)(s_lg + mt * 32 + 4 * tl) = mz_quad_sum(L);
        lgl = s_lg[mt * 32 + (tl < n.A ? tl : 0)];
      }
      const float v = mz_support_to_scalar_q(mz_quad_sum(V), n.vmin, n.no_transform, q8);
      const double pe = exp((double)lgl);     // mcts.py:52 (unconditional: no branch)
      double pr = (tl < n.A) ? pe : 0.0;
      asm volatile("" : "+v"(pr));       // pinned here: left alone, the compiler sinks the exp chain into the branch below
      STAMP(9)
      auto stampf = [&](int k) __attribute__((always_inline)) { STAMP(10 + k) };
      if (full || b0 + mt < t.B) {
        float vv = v, rew = s_rew[mt];
        // test instrumentation (mz_sim_io, inject mode; a zero in production, and absent from the whole-moves launch):
        // the caller's value / reward / logits -- the goldens' recorded network outputs -- replace this simulation's, so
        // that they reach THIS tree code.  Loads issued and awaited inside asm (mz_sim_io_load): the production path
        // behind the never-taken branch gets no s_waitcnt for them.
        if constexpr (!HEAD) {
          if (MZ_SIM_IO_ON && __builtin_expect(t.sim_io_keep < 0, 0)) {
            const float *io = mz_sim_io_row(t, b0 + mt, 0ull, slot0 + sim + 1);
            float lgi = 0.f;
            mz_sim_io_load(vv, io); mz_sim_io_load(rew, io + 1); mz_sim_io_load(lgi, io + 2 + (tl < n.A ? tl : 0));
            pr = (tl < n.A) ? exp((double)lgi) : 0.0;
          }
        }
        mz_tree_expand_f<TL, G, LT, SP>(t, tm[0], tl, slot0 + sim + 1, rew, pr, s_path + mt * MZ_FUSED_MAXPL,
                                        s_stage + mt * 96, tr[0]);
        stampf(0);
        mz_tree_backup_select_f<TL, G, LT, SP>(t, tm[0], tl, vv, rew, s_path + mt * MZ_FUSED_MAXPL,
                                               s_stage + mt * 96, s_pbc, s_rcp, tr[0], sim + 1 < nsims, my_slot[0],
                                               my_act[0], MzHiddenPrefetch{t.hpool, hoff[0], hv[0]}, stampf);
        // test instrumentation (mz_sim_io, log mode; a zero in production): what the tree step above consumed -- value,
        // reward, the A logits -- stored for the parity tests' replay through the CPU checker's tree.  BEHIND the tree step and
        // recomputed from the LDS the values came from (split-K partials, s_rew, s_lg: untouched until the next
        // simulation's first barrier) by the very same device functions: nothing of the production path lives longer or
        // is scheduled differently for this block, which costs it one scalar compare + branch per simulation.
        if (MZ_SIM_IO_ON && __builtin_expect(t.sim_io_keep > 0, 0)) {
          unsigned long long mvx = 0;
          if constexpr (ENVS) mvx = *(const unsigned long long *)(s_env + mt * MZ_ENVW);
          else if (record) mvx = sp.movecnt[b0 + mt];
          float *io = mz_sim_io_row(t, b0 + mt, mvx, slot0 + sim + 1);
          MzQuad V2;
          mz_quad_issue<0>(V2, mz_quad_addr(red, 4 * q8, mt), mz_lds_addr(s_b4 + 4 * q8));
          asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(V2));
          const float v2 = mz_support_to_scalar_q(mz_quad_sum(V2), n.vmin, n.no_transform, q8);
          float lgx;
          if constexpr (P4) {
            const f32x4 lg4 = mz_logits4(red, s_b4 + 32, mt, tl);
            lgx = (tl & 2) ? ((tl & 1) ? lg4[3] : lg4[2]) : ((tl & 1) ? lg4[1] : lg4[0]);
          } else {
            lgx = s_lg[mt * 32 + (tl < n.A ? tl : 0)];
          }
          if (tl < n.A) io[2 + tl] = lgx;
          if (tl == 0) { io[0] = v2; io[1] = s_rew[mt]; }
        }
      }
    } else {
      mz_bar();
      STAMP(9)
      // ---- tree: expand + backup (mcts.py:97-99), then the next descent (mcts.py:83-92)
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int mt = tid / TL + i * (256 / TL);
        const int b = b0 + mt;
        if (b < t.B) {
          auto stampf = [&](int k) __attribute__((always_inline)) { STAMP(10 + k) };
          float vv = s_val[mt], rew = s_rew[mt];
          // test instrumentation (mz_sim_io), as in the 16-lane path above; lane a owns logit a of s_lg
          if constexpr (!HEAD) {
            if (MZ_SIM_IO_ON && __builtin_expect(t.sim_io_keep < 0, 0)) {
              const float *io = mz_sim_io_row(t, b, 0ull, slot0 + sim + 1);
              float lgi = 0.f;
              mz_sim_io_load(vv, io); mz_sim_io_load(rew, io + 1); mz_sim_io_load(lgi, io + 2 + (tl < n.A ? tl : 0));
              if (tl < n.A) s_lg[mt * 32 + tl] = lgi;
            }
          }
          if (MZ_SIM_IO_ON && __builtin_expect(t.sim_io_keep > 0, 0)) {
            unsigned long long mvx = 0;
            if constexpr (ENVS) mvx = *(const unsigned long long *)(s_env + mt * MZ_ENVW);
            else if (record) mvx = sp.movecnt[b];
            float *io = mz_sim_io_row(t, b, mvx, slot0 + sim + 1);
            if (tl < n.A) io[2 + tl] = s_lg[mt * 32 + tl];
            if (tl == 0) { io[0] = vv; io[1] = rew; }
          }
          mz_tree_step_fused<TL, G, LT, SP>(t, tm[i], tl, slot0 + sim + 1, vv, rew, s_lg + mt * 32,
                                            s_path + mt * MZ_FUSED_MAXPL, s_stage + mt * 96, s_pbc, s_rcp, tr[i],
                                            sim + 1 < nsims, my_slot[i], my_act[i], t.hpool, hoff[i], hv[i], stampf);
        }
      }
    }
    STAMP(13)
  }
  HSTAMP(2)
  if (record) {     // self-play loop: action, visit distribution, env step and experience record of this move
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int mt = tid / TL + i * (256 / TL);
      if (b0 + mt < t.B) mz_finalize_record<TL, LT, GAME>(t, tm[i], sp, b0 + mt, tl, tr[i].legal, seed, s_stage + mt * 96, n.O,
                                                           ENVS ? s_env + mt * MZ_ENVW : nullptr);
    }
  }
  // per-tree scalars back to the pool (what export / a later mz_select continue from)
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int b = b0 + tid / TL + i * (256 / TL);
    if (b < t.B && tl == 0) { t.mn[b] = tr[i].mn; t.mx[b] = tr[i].mx; t.nexp[b] = slot0 + nsims + 1; }
    // the trees themselves: only where somebody can read them -- a search call (export, continued search, finalize
    // kernel) or a self-play loop asked to keep them; the self-play loop itself rebuilds every tree at the next root
    // and has finalized this one from LDS above (15 MB of dead stores per launch at 4096 x 125 nodes)
    if constexpr (LT != 0) {
      if (b < t.B && (!record || sp.export_trees)) {
        const size_t o = mz_slab(t, b);
        const int have = 1 + (slot0 + nsims + 1) * t.A;
        for (int k = tl; k < have; k += TL) {
          t.N[o + k] = tm[i].N[k]; t.P[o + k] = tm[i].P[k]; t.E[o + k] = tm[i].E[k];
          if constexpr (LT == 1) { t.W[o + k] = tm[i].W[k]; t.R[o + k] = tm[i].R[k]; t.TP[o + k] = tm[i].TP[k]; }
          if constexpr (LT == 2) {
            const int ek = tm[i].E[k];
            t.W[o + k] = ek >= 0 ? tm[i].W[ek] : 0.0; t.R[o + k] = ek >= 0 ? tm[i].R[ek] : 0.f; t.TP[o + k] = tm[i].TP[k];
          }
        }
      }
    }
  }
  HSTAMP(3)
  }      // (moves of a HEAD launch)
  if (HEAD && prof && (tid0 & 63) == 0)
    for (int i = 0; i < 8; ++i) prof[((size_t)blockIdx.x * 4 + (tid0 >> 6)) * 8 + i] = hs_acc[i];
  if (PROF && (tid0 & 63) == 0)
    for (int i = 0; i < MZ_NPHASE; ++i) prof[((size_t)blockIdx.x * 4 + (tid0 >> 6)) * MZ_NPHASE + i] = pacc[i];
#undef MZ_BLOAD
#undef MZ_WLOAD
#undef HSTAMP
}
