// mz_fused_h2.hip.h -- k_search_h2: the persistent fused search kernel (mz_fused.hip.h) with the FCNetwork GEMMs on the
// real matrix pipe: every float32 operand is split into two float16 parts, x = xh + xl (xh = f16(x), xl = f16(x - xh):
// 22 significant bits), and every product block is three v_mfma_f32_16x16x32_f16 instructions with float32 accumulation
//     W X  ~=  Wh Xh + Wh Xl + Wl Xh          (the dropped Wl Xl term is <= 2^-22 |W X|)
// at 16 cycles per 16x16x32 block instead of the eight 32-cycle v_mfma_f32_16x16x4_f32 of the exact-f32 kernel:
// 4.8 k instead of 21.7 k matrix-pipe cycles per simulation, and the f16 MFMA does not occupy the vector ALU the way the
// f32 one does.  Measured deviation from a float64 evaluation of the same network: 1-2x that of the exact-f32 path
// (hidden 3.8e-6 vs 1.9e-6, logits < 1e-6; scripts/split_f16_error.py), i.e. inside the 1e-5 bound of north_star.
// NOT bit-identical to the f32 kernel, so it is opt-in (mz_config.split_f16 / MZ_SPLIT_F16=1) and reported with its own
// dtype string; the exact-f32 kernel stays the default and the headline.
//
// Everything outside the four GEMM stages -- gather, LayerNorm / support epilogues, expand / backup / descent, the end of
// the move -- is the code of mz_fused.hip.h.  Differences:
//   * weights: per wave one cyclic stream of GROUPS (8 pieces of 1 KiB = the A operands, high and low halves, of four
//     16x32 blocks = 12 MFMAs); the first RSG groups stay in AGPRs for the whole launch, the others go L2 -> AGPR with
//     buffer_load_dwordx4 into a ring of NBG group buffers, issued one per MFMA slot two groups ahead of use;
//   * the B operand of a 16x16x32 MFMA is eight consecutive k per lane: fc1 reads it with one ds_read_b128 from f16 copies
//     (high / low) of the x tile; fc2 builds it from two fc1 tiles of the wave (D fragment = features 16t + 4(lane>>4) + r
//     of column lane & 15 -- the weights are packed in that k order), split into halves with v_cvt_pkrtz_f16_f32;
//   * every accumulator is touched at most once per four consecutive MFMAs (the MFMAs are inline asm: no hazard
//     recogniser), out tiles that would be hit more often are kept as two or four partial accumulators.
// (Tried: refilling every piece in place, right behind the MFMA that reads it last, with the same piece of the group NBG
// ahead -- a deeper prefetch from the same registers.  No gain (the stages are bound by bytes, not by latency), and the
// 4 + 3 configuration computed wrong values; dropped.)
// Supported: action_space <= 13 (dynamics fc1 K = 50 + A + 1 <= 64), 16 lanes per tree, one policy tile.
#pragma once
#include "mz_fused.hip.h"

#ifndef MZ_H2_NBG
#define MZ_H2_NBG 3       // ring depth (groups)
#endif
#ifndef MZ_H2_RSG
#define MZ_H2_RSG 4       // resident groups
#endif
// (A/B on one box, us per launch of the headline workload: RSG / NBG = 4 / 3: 269, 1 / 4: 283, 5 / 2: 276 -- both the
// bytes streamed and the prefetch distance matter.  Resident groups in arch VGPRs (5 / 4 with one of them there) were
// tried: under the register pressure that creates the compiler moves operands around with VALU copies right in front of
// the asm MFMAs, whose wait states it cannot know -- wrong results on a full grid; dropped.)
#define MZ_H2_XS 72       // row stride of the f16 x tiles (halfs): 144 B, rows 36 banks apart
#define MZ_H2_MAXA 13

struct H2Sched {
  static constexpr int D1 = 8, D2 = 6, P1 = 8, P2 = 3;                      // groups per stage
  static constexpr int E_D1 = D1, E_D2 = E_D1 + D2, E_P1 = E_D2 + P1, E_P2 = E_P1 + P2;
  static constexpr int REAL = E_P2;                                          // 25 groups = 200 KiB per wave and simulation
  static constexpr int RSG = MZ_H2_RSG, NBG = MZ_H2_NBG;
  static constexpr int NRING = (REAL - RSG + NBG - 1) / NBG * NBG;
  static constexpr int NGROUPS = RSG + NRING;
};
static_assert(H2Sched::NRING == H2Sched::REAL - H2Sched::RSG, "no padding groups: (REAL - RSG) must be a multiple of NBG");

typedef _Float16 mz_h16;
typedef mz_h16 mz_h16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void h2_mfma(f32x4 &c, const f32x4 &a, const f32x4 &b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
}
__device__ __forceinline__ void h2_mfma0(f32x4 &c, const f32x4 &a, const f32x4 &b) {     // first product: SrcC = 0
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
}
// one 1-KiB piece of the weight stream straight into AGPRs (the compiler does not see a load: waits are explicit)
template <int IMM>
__device__ __forceinline__ void h2_load(f32x4 &dst, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=a"(dst) : "v"(voff), "s"(rs), "s"(soff), "n"(IMM));
}
// all pieces of a group have arrived once at most `N` younger vector-memory operations are outstanding
template <int N>
__device__ __forceinline__ void h2_wait(f32x4 (&g)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+a"(g[0]), "+a"(g[1]), "+a"(g[2]), "+a"(g[3]), "+a"(g[4]), "+a"(g[5]), "+a"(g[6]), "+a"(g[7])
               : "n"(N));
}
// MFMA results -> VALU (wait states after the LAST MFMA of a stage) / VALU results -> MFMA SrcB
__device__ __forceinline__ void h2_fence4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void h2_valu_fence(f32x4 &a, f32x4 &b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }

// two float32 -> the packed float16 high parts and the packed float16 low parts (x = h + l up to 2^-22 |x|)
__device__ __forceinline__ void h2_split2(float x0, float x1, float &hp, float &lp) {
  const mz_h16x2 h = __builtin_bit_cast(mz_h16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const float l0 = x0 - (float)h[0], l1 = x1 - (float)h[1];           // exact: h keeps the leading 11 bits of x
  const mz_h16x2 l = __builtin_bit_cast(mz_h16x2, __builtin_amdgcn_cvt_pkrtz(l0, l1));
  hp = __builtin_bit_cast(float, h);
  lp = __builtin_bit_cast(float, l);
}
// B operand of a K = 32 block out of two fc1 tiles of this wave: element j < 4 = t0[j], j >= 4 = t1[j - 4]
__device__ __forceinline__ void h2_split_pair(const f32x4 &t0, const f32x4 &t1, f32x4 &bh, f32x4 &bl) {
  float h[4], l[4];
  h2_split2(t0[0], t0[1], h[0], l[0]);
  h2_split2(t0[2], t0[3], h[1], l[1]);
  h2_split2(t1[0], t1[1], h[2], l[2]);
  h2_split2(t1[2], t1[3], h[3], l[3]);
  bh = f32x4{h[0], h[1], h[2], h[3]};
  bl = f32x4{l[0], l[1], l[2], l[3]};
}

// LayerNorm output -> the f16 x tile (high / low) of the prediction stage, written by the LayerNorm lanes themselves
// (8 per column: lane q holds features 4q.. and 32 + 4q..): columns 0..49 the hidden state, 50 the bias column (1),
// zero beyond
struct H2LnSink {
  mz_h16 *xH, *xL;
  __device__ __forceinline__ void operator()(int m, int q, const f32x4 &ya, const f32x4 &yb) const {
    float h0, l0, h1, l1;
    h2_split2(ya[0], ya[1], h0, l0);
    h2_split2(ya[2], ya[3], h1, l1);
    float *dh = (float *)(xH + m * MZ_H2_XS + 4 * q), *dl = (float *)(xL + m * MZ_H2_XS + 4 * q);
    dh[0] = h0; dh[1] = h1; dl[0] = l0; dl[1] = l1;
    h2_split2(yb[0], yb[1], h0, l0);
    h2_split2(yb[2], yb[3], h1, l1);
    if (q < 4) { dh[16] = h0; dh[17] = h1; dl[16] = l0; dl[17] = l1; }
    else if (q == 4) {                  // columns 48, 49, then the prediction's extension (the low parts there stay zero)
      dh[16] = h0; dl[16] = l0;
      dh[17] = __builtin_bit_cast(float, mz_h16x2{(mz_h16)1.f, (mz_h16)0.f});
#pragma unroll
      for (int z = 18; z < 24; ++z) dh[z] = 0.f;
    }
  }
};

// prefetch hook of a ring group: slot i requests piece i of the group NBG - 1 ahead (cyclic over the streamed groups)
// into the buffer the previous group has released
struct H2Prefetch {
  f32x4 (&dst)[8]; const __amdgpu_buffer_rsrc_t &rs; int voff, soff;
  template <int I> __device__ __forceinline__ void slot() const {
    h2_load<(I & 3) * 1024>(dst[I], rs, voff, soff + (I >> 2) * 4096);
  }
};

// the three products of four blocks that share nothing but the schedule slot: accumulators c0..c3 (all different),
// A operands = pieces 0..3 (high) and 4..7 (low) of the group, B operands (high, low) per block.
// FIRST: the accumulators start here (SrcC = 0).  Between the MFMAs, one piece of the group NBG - 1 ahead is requested
// per slot (PF) into the ring buffer this group's predecessor has released.
struct H2NoPrefetch { template <int I> __device__ __forceinline__ void slot() const {} };
template <bool FIRST, class PF>
__device__ __forceinline__ void h2_group(f32x4 &c0, f32x4 &c1, f32x4 &c2, f32x4 &c3, const f32x4 (&A)[8],
                                         const f32x4 &b0h, const f32x4 &b0l, const f32x4 &b1h, const f32x4 &b1l,
                                         const f32x4 &b2h, const f32x4 &b2l, const f32x4 &b3h, const f32x4 &b3l,
                                         const PF &pf) {
  if constexpr (FIRST) { h2_mfma0(c0, A[0], b0h); pf.template slot<0>(); h2_mfma0(c1, A[1], b1h); pf.template slot<1>();
                         h2_mfma0(c2, A[2], b2h); pf.template slot<2>(); h2_mfma0(c3, A[3], b3h); pf.template slot<3>(); }
  else { h2_mfma(c0, A[0], b0h); pf.template slot<0>(); h2_mfma(c1, A[1], b1h); pf.template slot<1>();
         h2_mfma(c2, A[2], b2h); pf.template slot<2>(); h2_mfma(c3, A[3], b3h); pf.template slot<3>(); }
  h2_mfma(c0, A[0], b0l); pf.template slot<4>(); h2_mfma(c1, A[1], b1l); pf.template slot<5>();
  h2_mfma(c2, A[2], b2l); pf.template slot<6>(); h2_mfma(c3, A[3], b3l); pf.template slot<7>();
  h2_mfma(c0, A[4], b0h); h2_mfma(c1, A[5], b1h); h2_mfma(c2, A[6], b2h); h2_mfma(c3, A[7], b3h);
}

// static LDS (floats): hidden tile, split-K partials, value / reward / logits, biases, LayerNorm affine, search paths,
// 1 / n table, two f16 x tiles [16][XS] (high, low; shared by the dynamics and the prediction stage), and -- except beside
// large trees (LT = 2), where it shares the partials' space -- the tree step's staging
#define MZ_H2_LDS_BASE (16 * MZ_HS + 4 * 6 * 256 + 16 + 16 + 16 * 32 + 96 + 64 + 64 + 64 + 16 * MZ_FUSED_MAXPL + 2 * MZ_FUSED_MAXPL + 2 * 16 * MZ_H2_XS / 2)
__host__ __device__ constexpr int mz_h2_lds_floats(int lt) { return MZ_H2_LDS_BASE + 16 * 96 * 2; }

// HEAD: whole self-play moves inside the launch, as in k_search_fused (mz_fused.hip.h); the resident groups live in AGPRs
// the compiler never allocates to the root, so here they do stay across the moves of a launch.
template <int G, int LT, bool PROF, bool SP, bool HEAD = false>
__global__ __launch_bounds__(256, 1) void k_search_h2(NetView n, TreeView t, const f32x4 *wstream, int nsims, int slot0,
                                                       unsigned long long *prof, SelfplayState sp, int record,
                                                       uint64_t seed, MzRootArgs ra) {
  static_assert(!HEAD || (LT != 0 && !PROF && SP), "HEAD: single player, trees in LDS, no phase stamps");
  using SC = H2Sched;
  constexpr int NBG = SC::NBG, RSG = SC::RSG, NGROUPS = SC::NGROUPS, NRING = SC::NRING;
  static_assert(G <= 16, "16 lanes per tree");
  constexpr int XS = MZ_H2_XS;

  __shared__ __attribute__((aligned(16))) float smem[mz_h2_lds_floats(LT)];
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  double *s_pbc = (double *)dyn_lds;
  const int PBS = (LT == 2) ? t.sims + 2 : 64;
  double *l_P = s_pbc + (t.sims + 2) * PBS;
  // LT = 1: every field per node; LT = 2: W, R and the X cache per expansion slot (mz_tree.hip.h, TreeMem)
  const int NV = (LT == 2) ? t.sims + 2 : t.NN;      // entries per tree of the value arrays
  double *l_Q = l_P + 16 * t.NN;                     // X cache
  double *l_W = l_Q + 16 * NV;
  float *l_R = (float *)(l_W + 16 * NV);
  int16_t *l_N = (int16_t *)(l_R + 16 * NV);
  int16_t *l_E = l_N + 16 * t.NN;
  int8_t *l_TP = (int8_t *)(l_E + 16 * t.NN);
  float *xR = smem;                       // [16][MZ_HS] float32 hidden tile (LayerNorm output -> hidden-state pool)
  float *red = xR + 16 * MZ_HS;
  float *s_val = red + 4 * 6 * 256;
  float *s_rew = s_val + 16;
  float *s_lg = s_rew + 16;
  float *s_b2 = s_lg + 16 * 32;
  float *s_b4 = s_b2 + 96;
  float *s_lnw = s_b4 + 64;
  float *s_lnb = s_lnw + 64;
  int *s_path = (int *)(s_lnb + 64);
  double *s_rcp = (double *)(s_path + 16 * MZ_FUSED_MAXPL);
  // f16 x tile [16][XS], high and low parts: columns 0..49 hidden state, then for the dynamics one-hot(action) and 1 (bias
  // column), for the prediction 1 (bias column); zero beyond.  One pair of tiles serves both stages: each rewrites its
  // extension columns (the low parts of those are always zero)
  mz_h16 *xH = (mz_h16 *)(s_rcp + MZ_FUSED_MAXPL);
  mz_h16 *xL = xH + 16 * XS;
  double *s_stage = (double *)(xL + 16 * XS);

  const int tid0 = threadIdx.x;
  const int b0 = blockIdx.x * MZ_ROWS;
  const bool full = b0 + MZ_ROWS <= t.B;
  const size_t per_tree = (size_t)(t.sims + 1) * MZ_HS;

  if (tid0 < 96) s_b2[tid0] = (tid0 < 32 && tid0 >= n.Sr) ? MZ_PAD_BIN : n.b2[tid0];
  if (tid0 < 48) s_b4[tid0] = (tid0 < 32 && tid0 >= n.Sv) ? MZ_PAD_BIN : n.b4[tid0];
  if (tid0 < 64) { s_lnw[tid0] = n.lnw[tid0]; s_lnb[tid0] = n.lnb[tid0]; }
  if (tid0 < MZ_FUSED_MAXPL) s_rcp[tid0] = 1.0 / (double)(tid0 > 0 ? tid0 : 1);
  for (int i = tid0; i < 16 * XS; i += 256) { xH[i] = (mz_h16)0.f; xL[i] = (mz_h16)0.f; }
  for (int i = tid0; i < (t.sims + 2) * (t.sims + 2); i += 256) s_pbc[(i / (t.sims + 2)) * PBS + i % (t.sims + 2)] = t.pbctab[i];

  constexpr int TL = 16;
  // this wave's stream: [NGROUPS][8 pieces][64 lanes] f32x4 (8 KiB per group); the resident groups once per launch
  f32x4 Rw[RSG][8];          // resident groups (AGPRs)
#define H2_LOAD_GROUP(dst, byteoff)                                                                   \
  do {                                                                                                \
    h2_load<0>((dst)[0], wrsrc, lane_off, (byteoff)); h2_load<1024>((dst)[1], wrsrc, lane_off, (byteoff));          \
    h2_load<2048>((dst)[2], wrsrc, lane_off, (byteoff)); h2_load<3072>((dst)[3], wrsrc, lane_off, (byteoff));       \
    h2_load<0>((dst)[4], wrsrc, lane_off, (byteoff) + 4096); h2_load<1024>((dst)[5], wrsrc, lane_off, (byteoff) + 4096);   \
    h2_load<2048>((dst)[6], wrsrc, lane_off, (byteoff) + 4096); h2_load<3072>((dst)[7], wrsrc, lane_off, (byteoff) + 4096); \
  } while (0)
  {
    const char *wbase = (const char *)(wstream + (size_t)__builtin_amdgcn_readfirstlane(tid0 >> 6) * NGROUPS * 512);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, NGROUPS * 8192, 0x00020000);
    const int lane_off = (tid0 & 63) * 16;
#pragma unroll
    for (int s = 0; s < RSG; ++s) H2_LOAD_GROUP(Rw[s], s * 8192);
    // the resident groups are never waited for again: have them in, with every register tied to the wait (the compiler
    // does not know these are loads and must not touch the registers before the data has arrived)
#pragma unroll
    for (int s = 0; s < RSG; ++s) h2_wait<0>(Rw[s]);
  }

  unsigned long long pacc[MZ_NPHASE];
  unsigned long long tlast = 0;
  if (PROF) {
    for (int i = 0; i < MZ_NPHASE; ++i) pacc[i] = 0;
  }

  const int nmoves = HEAD ? ra.nmoves : 1;
  for (int mv = 0; mv < nmoves; ++mv) {
  if constexpr (HEAD) {
    // the root of this move on the trees' LDS (see k_search_fused)
    __syncthreads();
    int tid_r = threadIdx.x;
    asm volatile("" : "+v"(tid_r));
    if (ra.nst0 >= 0)
    mz_root_body<1, G, true>(n, t, nullptr, ra.istream, ra.nst0, sp, seed, ra.alpha, ra.frac,
                             (float *)(dyn_lds + (((t.sims + 2) * PBS * 8 + 15) & ~15)), tid_r, s_stage, MzNoStamp());
    __syncthreads();
  }
  int tid = threadIdx.x;
  if constexpr (HEAD) asm volatile("" : "+v"(tid));
  const int w = tid >> 6, lane = tid & 63;
  const int g4 = lane >> 4, m16 = lane & 15;
  const int tl = tid % TL;
  int my_slot = 0, my_act = 0;
  TreeRegs tr;
  TreeMem<LT> tm;
  const int mt = tid / TL;
  {
    const int b = b0 + mt;
    if constexpr (LT == 1) {
      const int o = mt * t.NN;
      tm.N = l_N + o; tm.W = l_W + o; tm.P = l_P + o; tm.R = l_R + o; tm.E = l_E + o; tm.TP = l_TP + o; tm.X = l_Q + o;
    } else if constexpr (LT == 2) {
      const int o = mt * t.NN, ov = mt * NV;
      tm.N = l_N + o; tm.E = l_E + o; tm.P = l_P + o; tm.TP = l_TP + o;
      tm.X = l_Q + ov; tm.W = l_W + ov; tm.R = l_R + ov;
    } else {
      const size_t o = mz_slab(t, b < t.B ? b : 0);
      tm.N = t.N + o; tm.W = t.W + o; tm.P = t.P + o; tm.R = t.R + o; tm.E = t.E + o; tm.TP = t.TP + o;
    }
    tr.len = 1; tr.tp = 1; tr.root_tp = 1; tr.legal = 0; tr.mn = 0.0; tr.mx = 0.0; tr.root_n = 0;
    if constexpr (HEAD) {      // the root this launch has just made: taken from LDS and from what is known (k_search_fused)
      if (b < t.B) {
        const double *st = s_stage + mt * 96;
        const int best = (int)st[32];
        my_act = best;
        tr.len = 2;
        tr.legal = (t.A >= 32) ? 0xFFFFFFFFu : ((1u << t.A) - 1u);
        tr.mn = t.has_min ? t.min_bound : __builtin_inf();
        tr.mx = t.has_max ? t.max_bound : -__builtin_inf();
        if (tl == 0) { s_path[mt * MZ_FUSED_MAXPL] = 0; s_path[mt * MZ_FUSED_MAXPL + 1] = 1 + best; }
        const int have = 1 + t.A;
        if constexpr (LT == 1) {
          for (int k = tl; k < t.NN; k += TL) {
            tm.N[k] = 0; tm.W[k] = 0.0; tm.R[k] = 0.f; tm.E[k] = (k == 0) ? 0 : -1; tm.TP[k] = 1;
          }
          for (int k = tl; k < have; k += TL) tm.X[k] = 0.0;
        } else {
          for (int k = tl; k < have; k += TL) { tm.N[k] = 0; tm.E[k] = (k == 0) ? 0 : -1; tm.TP[k] = 1; }
          if (tl == 0) { tm.W[0] = 0.0; tm.R[0] = 0.f; tm.X[0] = 0.0; }      // the root's expansion slot
        }
        for (int k = tl; k < have; k += TL) tm.P[k] = (k == 0) ? 0.0 : st[k - 1];
      }
    } else
    if (b < t.B) {
      my_slot = t.slot[b];
      my_act = t.act[b];
      tr.len = t.plen[b];
      tr.tp = t.leaf_tp[b];
      tr.root_tp = t.TP[mz_slab(t, b)];
      tr.root_n = t.N[mz_slab(t, b)];
      tr.legal = t.legal[b];
      tr.mn = t.mn[b];
      tr.mx = t.mx[b];
      for (int k = tl; k < tr.len; k += TL) s_path[mt * MZ_FUSED_MAXPL + k] = t.path[(size_t)b * t.PL + k];
      if constexpr (LT == 1) {
        const size_t o = mz_slab(t, b);
        const int have = 1 + (slot0 + 1) * t.A;
        for (int k = have + tl; k < t.NN; k += TL) {
          tm.N[k] = 0; tm.W[k] = 0.0; tm.R[k] = 0.f; tm.E[k] = -1; tm.TP[k] = 1;
        }
        for (int k = tl; k < have; k += TL) {
          tm.N[k] = (int16_t)t.N[o + k]; tm.W[k] = t.W[o + k]; tm.P[k] = t.P[o + k]; tm.R[k] = t.R[o + k];
          const double qk = t.N[o + k] > 0 ? t.W[o + k] / (double)t.N[o + k] : 0.0;
          const double rk = (double)t.R[o + k];
          tm.X[k] = t.two_players ? rk - t.discount * qk : rk + t.discount * qk;
          tm.E[k] = (int16_t)t.E[o + k]; tm.TP[k] = t.TP[o + k];
        }
      } else if constexpr (LT == 2) {
        const size_t o = mz_slab(t, b);
        const int have = 1 + (slot0 + 1) * t.A;
        for (int k = tl; k < have; k += TL) {
          const int nk = t.N[o + k], ek = t.E[o + k];
          tm.N[k] = (int16_t)nk; tm.P[k] = t.P[o + k]; tm.E[k] = (int16_t)ek; tm.TP[k] = t.TP[o + k];
          if (ek >= 0) {       // an expanded node: its value fields live in its expansion slot
            const double qk = nk > 0 ? t.W[o + k] / (double)nk : 0.0;
            const double rk = (double)t.R[o + k];
            tm.W[ek] = t.W[o + k]; tm.R[ek] = t.R[o + k];
            tm.X[ek] = t.two_players ? rk - t.discount * qk : rk + t.discount * qk;
          }
        }
      }
    }
  }

  f32x4 hv;
  unsigned hoff;
  {
    const int b = b0 + mt;
    hoff = (unsigned)(((size_t)(b < t.B ? b : 0) * per_tree) * 4) + (unsigned)((tl < MZ_HS / 4 ? tl : MZ_HS / 4 - 1) * 16);
    hv = *(const f32x4 *)((const char *)t.hpool + hoff + (size_t)my_slot * (MZ_HS * 4));
  }

  const char *wbase = (const char *)(wstream + (size_t)__builtin_amdgcn_readfirstlane(w) * NGROUPS * 512);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, NGROUPS * 8192, 0x00020000);
  const int lane_off = lane * 16;
  f32x4 Bf[NBG][8];          // ring (AGPRs)
  int sbase = RSG * 8192;      // byte offset of ring group 0 (laundered per simulation, see mz_fused.hip.h)
#pragma unroll
  for (int s = 0; s < NBG - 1; ++s) H2_LOAD_GROUP(Bf[s], sbase + s * 8192);

  if (PROF) tlast = __builtin_amdgcn_s_memtime();
  __syncthreads();

  for (int sim = 0; sim < nsims; ++sim) {
    asm volatile("" : "+s"(sbase));
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    // ---- gather: f16 x tile (high / low) = [hidden of search_path[-2] | one-hot(action) | 1]  (mcts.py:94-96)
    {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv));
      float h0, l0, h1, l1;
      h2_split2(hv[0], hv[1], h0, l0);
      h2_split2(hv[2], hv[3], h1, l1);
      float *dh = (float *)(xH + mt * XS + 4 * tl), *dl = (float *)(xL + mt * XS + 4 * tl);
      if (tl < 12) { dh[0] = h0; dh[1] = h1; dl[0] = l0; dl[1] = l1; }
      else if (tl == 12) { dh[0] = h0; dl[0] = l0; }                  // columns 48, 49 (50.. belong to the extension)
      if (tl < 14) xH[mt * XS + MZ_H + tl] = (tl == my_act || tl == n.A) ? (mz_h16)1.f : (mz_h16)0.f;
    }
    STAMP(0)
    mz_bar();
    STAMP(1)

    f32x4 acc[16];
    // B operands of an fc1 stage: eight consecutive k per lane, chunk c = columns 32c + 8g .. + 7 of row m
    f32x4 bh0, bl0, bh1, bl1;
    auto load_x = [&](const mz_h16 *xh, const mz_h16 *xl) __attribute__((always_inline)) {
      bh0 = *(const f32x4 *)(xh + m16 * XS + 8 * g4); bh1 = *(const f32x4 *)(xh + m16 * XS + 32 + 8 * g4);
      bl0 = *(const f32x4 *)(xl + m16 * XS + 8 * g4); bl1 = *(const f32x4 *)(xl + m16 * XS + 32 + 8 * g4);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh0), "+v"(bl0), "+v"(bh1), "+v"(bl1));
    };
    // group GI of the schedule: body(A operands (resident or ring buffer), prefetch hook)
    auto run = [&](auto GI_, auto &&body) __attribute__((always_inline)) {
      constexpr int GI = decltype(GI_)::value;
      if constexpr (GI < RSG) {
        body(Rw[GI], H2NoPrefetch{});
      } else {
        constexpr int r_ = GI - RSG, cb_ = r_ % NBG, pb_ = (r_ + NBG - 1) % NBG, ps_ = (r_ + NBG - 1) % NRING;
        h2_wait<8 * (NBG - 2)>(Bf[cb_]);      // this group's pieces: all but the next group's requests have returned
        body(Bf[cb_], H2Prefetch{Bf[pb_], wrsrc, lane_off, sbase + ps_ * 8192});
      }
    };
#define H2_GI(v) std::integral_constant<int, (v)>{}
    // ---- dynamics fc1: groups (tg, c): tiles 4tg..4tg+3, K chunk c
    load_x(xH, xL);
    mz_static_for<SC::D1>([&](auto G_) __attribute__((always_inline)) {
      constexpr int gi = decltype(G_)::value, tg = gi / 2, c = gi % 2;
      const f32x4 &bh = c ? bh1 : bh0, &bl = c ? bl1 : bl0;
      run(H2_GI(gi), [&](const f32x4 (&A_)[8], const auto &pf_) __attribute__((always_inline)) {
        h2_group<c == 0>(acc[4 * tg], acc[4 * tg + 1], acc[4 * tg + 2], acc[4 * tg + 3], A_, bh, bl, bh, bl, bh, bl, bh, bl, pf_);
      });
    });
    mz_mfma_fence16v(acc);
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      acc[tt][0] = mz_relu1(acc[tt][0]); acc[tt][1] = mz_relu1(acc[tt][1]);
      acc[tt][2] = mz_relu1(acc[tt][2]); acc[tt][3] = mz_relu1(acc[tt][3]);
    }
    STAMP(2)
    // ---- dynamics fc2: out tiles o0, o1 (reward, K pairs from tiles 0..7) and o2..o5 (next hidden, tiles 8..15)
    f32x4 o[4], oa[2], ob[2];
    {
      f32x4 rh, rl, th[2], tll[2];
      mz_static_for<SC::D2>([&](auto G_) __attribute__((always_inline)) {
        constexpr int k = decltype(G_)::value, gi = SC::E_D1 + k;
        constexpr int half = k / 3, sub = k % 3;         // order: (o0..o3; p), (o0..o3; p+1), (o4, o5; p, p+1) for p = 0, 2
        if constexpr (sub < 2) {
          constexpr int p = 2 * half + sub;
          h2_split_pair(acc[2 * p], acc[2 * p + 1], rh, rl);
          h2_split_pair(acc[8 + 2 * p], acc[9 + 2 * p], th[sub], tll[sub]);
          h2_valu_fence(rh, rl); h2_valu_fence(th[sub], tll[sub]);
          run(H2_GI(gi), [&](const f32x4 (&A_)[8], const auto &pf_) __attribute__((always_inline)) {
            h2_group<p == 0>(o[0], o[1], o[2], o[3], A_, rh, rl, rh, rl, th[sub], tll[sub], th[sub], tll[sub], pf_);
          });
        } else {
          run(H2_GI(gi), [&](const f32x4 (&A_)[8], const auto &pf_) __attribute__((always_inline)) {
            h2_group<half == 0>(oa[0], oa[1], ob[0], ob[1], A_, th[0], tll[0], th[0], tll[0], th[1], tll[1], th[1], tll[1], pf_);
          });
        }
      });
    }
    f32x4 out2[6];
    h2_fence4(o[0], o[1], o[2], o[3]);
    h2_fence4(oa[0], oa[1], ob[0], ob[1]);
    out2[0] = o[0]; out2[1] = o[1]; out2[2] = o[2]; out2[3] = o[3]; out2[4] = oa[0] + ob[0]; out2[5] = oa[1] + ob[1];
    STAMP(3)
    mz_partials_out<6, 0>(red, out2, tid);
    STAMP(4)
    {
      const int col = 8 * (w & 1) + (lane_e >> 3), q = lane_e & 7;
      if (w < 2) {
        sln_relu8p<false>(red, s_b2, xR, s_lnw, s_lnb, 32, col, q, H2LnSink{xH, xL});
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
    // ---- prediction fc1
    load_x(xH, xL);
    mz_static_for<SC::P1>([&](auto G_) __attribute__((always_inline)) {
      constexpr int k = decltype(G_)::value, gi = SC::E_D2 + k, tg = k / 2, c = k % 2;
      const f32x4 &bh = c ? bh1 : bh0, &bl = c ? bl1 : bl0;
      run(H2_GI(gi), [&](const f32x4 (&A_)[8], const auto &pf_) __attribute__((always_inline)) {
        h2_group<c == 0>(acc[4 * tg], acc[4 * tg + 1], acc[4 * tg + 2], acc[4 * tg + 3], A_, bh, bl, bh, bl, bh, bl, bh, bl, pf_);
      });
    });
    mz_mfma_fence16v(acc);
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      acc[tt][0] = mz_relu1(acc[tt][0]); acc[tt][1] = mz_relu1(acc[tt][1]);
      acc[tt][2] = mz_relu1(acc[tt][2]); acc[tt][3] = mz_relu1(acc[tt][3]);
    }
    STAMP(6)
    // ---- prediction fc2: value tiles v0, v1 (K pairs from tiles 0..7; partial accumulators a: even pairs, b: odd pairs),
    // policy tile (tiles 8..15; one partial accumulator per pair)
    f32x4 va[2], vb[2], pl[4];
    {
      f32x4 ph[4], pll[4];
      mz_static_for<SC::P2>([&](auto G_) __attribute__((always_inline)) {
        constexpr int k = decltype(G_)::value, gi = SC::E_P1 + k;
        if constexpr (k < 2) {
          f32x4 v0h, v0l, v1h, v1l;
          h2_split_pair(acc[4 * k], acc[4 * k + 1], v0h, v0l);
          h2_split_pair(acc[4 * k + 2], acc[4 * k + 3], v1h, v1l);
          h2_valu_fence(v0h, v0l); h2_valu_fence(v1h, v1l);
          run(H2_GI(gi), [&](const f32x4 (&A_)[8], const auto &pf_) __attribute__((always_inline)) {
            h2_group<k == 0>(va[0], va[1], vb[0], vb[1], A_, v0h, v0l, v0h, v0l, v1h, v1l, v1h, v1l, pf_);
          });
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            h2_split_pair(acc[8 + 2 * p], acc[9 + 2 * p], ph[p], pll[p]);
            h2_valu_fence(ph[p], pll[p]);
          }
          run(H2_GI(gi), [&](const f32x4 (&A_)[8], const auto &pf_) __attribute__((always_inline)) {
            h2_group<true>(pl[0], pl[1], pl[2], pl[3], A_, ph[0], pll[0], ph[1], pll[1], ph[2], pll[2], ph[3], pll[3], pf_);
          });
        }
      });
    }
    f32x4 out4[3];
    h2_fence4(va[0], va[1], vb[0], vb[1]);
    h2_fence4(pl[0], pl[1], pl[2], pl[3]);
    out4[0] = va[0] + vb[0]; out4[1] = va[1] + vb[1]; out4[2] = (pl[0] + pl[1]) + (pl[2] + pl[3]);
    STAMP(7)
    // every wave's hidden-state stores must have landed before the tree lanes may gather them (they are older than the
    // weight requests in flight); the barrier inside mz_partials_out then publishes them
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NBG - 1)) : "memory");
    mz_partials_out<3, 0>(red, out4, tid);
    STAMP(8)
    {
      const int q8 = tl & 7;
      MzQuad V, L;
      const unsigned ba = mz_lds_addr(s_b4 + 4 * q8);
      mz_quad_issue<0>(V, mz_quad_addr(red, 4 * q8, mt), ba);
      mz_quad_issue<128>(L, mz_quad_addr(red, 32 + 4 * q8, mt), ba);
      asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(V), MZ_Q(L));
      if (tl < 8 && 4 * tl < n.A) *(f32x4 *)(s_lg + mt * 32 + 4 * tl) = mz_quad_sum(L);
      const float lgl = s_lg[mt * 32 + (tl < n.A ? tl : 0)];
      const float v = mz_support_to_scalar_q(mz_quad_sum(V), n.vmin, n.no_transform, q8);
      const double pe = exp((double)lgl);
      double pr = (tl < n.A) ? pe : 0.0;
      asm volatile("" : "+v"(pr));
      STAMP(9)
      auto stampf = [&](int k) __attribute__((always_inline)) { STAMP(10 + k) };
      if (full || b0 + mt < t.B) {
        float vv = v, rew = s_rew[mt];
        // test instrumentation (mz_sim_io), zeros in production -- see k_search_fused: inject in front of the tree step
        // (not in the whole-moves launch), log behind it, recomputed from LDS
        if constexpr (!HEAD) {
          if (MZ_SIM_IO_ON && __builtin_expect(t.sim_io_keep < 0, 0)) {
            const float *io = mz_sim_io_row(t, b0 + mt, 0ull, slot0 + sim + 1);
            float lgi = 0.f;
            mz_sim_io_load(vv, io); mz_sim_io_load(rew, io + 1); mz_sim_io_load(lgi, io + 2 + (tl < n.A ? tl : 0));
            pr = (tl < n.A) ? exp((double)lgi) : 0.0;
          }
        }
        mz_tree_expand_f<TL, G, LT, SP>(t, tm, tl, slot0 + sim + 1, rew, pr, s_path + mt * MZ_FUSED_MAXPL,
                                        s_stage + mt * 96, tr);
        stampf(0);
        mz_tree_backup_select_f<TL, G, LT, SP>(t, tm, tl, vv, rew, s_path + mt * MZ_FUSED_MAXPL, s_stage + mt * 96,
                                               s_pbc, s_rcp, tr, sim + 1 < nsims, my_slot, my_act,
                                               MzHiddenPrefetch{t.hpool, hoff, hv}, stampf);
        if (MZ_SIM_IO_ON && __builtin_expect(t.sim_io_keep > 0, 0)) {
          unsigned long long mvx = 0;
          if (record) mvx = sp.movecnt[b0 + mt];
          float *io = mz_sim_io_row(t, b0 + mt, mvx, slot0 + sim + 1);
          MzQuad V2;
          mz_quad_issue<0>(V2, mz_quad_addr(red, 4 * q8, mt), mz_lds_addr(s_b4 + 4 * q8));
          asm volatile("s_waitcnt lgkmcnt(0)" : MZ_Q(V2));
          const float v2 = mz_support_to_scalar_q(mz_quad_sum(V2), n.vmin, n.no_transform, q8);
          if (tl < n.A) io[2 + tl] = s_lg[mt * 32 + tl];
          if (tl == 0) { io[0] = v2; io[1] = s_rew[mt]; }
        }
      }
    }
    STAMP(13)
  }
#undef H2_GI
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's requests in flight target registers of this wave
  if (record) {
    if (b0 + mt < t.B) mz_finalize_record<TL, LT>(t, tm, sp, b0 + mt, tl, tr.legal, seed, s_stage + mt * 96, n.O);
  }
  {
    const int b = b0 + mt;
    if (b < t.B && tl == 0) { t.mn[b] = tr.mn; t.mx[b] = tr.mx; t.nexp[b] = slot0 + nsims + 1; }
    if constexpr (LT != 0) {
      if (b < t.B && (!record || sp.export_trees)) {
        const size_t o = mz_slab(t, b);
        const int have = 1 + (slot0 + nsims + 1) * t.A;
        for (int k = tl; k < have; k += TL) {
          t.N[o + k] = tm.N[k]; t.P[o + k] = tm.P[k]; t.E[o + k] = tm.E[k];
          if constexpr (LT == 1) { t.W[o + k] = tm.W[k]; t.R[o + k] = tm.R[k]; t.TP[o + k] = tm.TP[k]; }
          if constexpr (LT == 2) {
            const int ek = tm.E[k];
            t.W[o + k] = ek >= 0 ? tm.W[ek] : 0.0; t.R[o + k] = ek >= 0 ? tm.R[ek] : 0.f; t.TP[o + k] = tm.TP[k];
          }
        }
      }
    }
  }
  }      // (moves of a HEAD launch)
#undef H2_LOAD_GROUP
  if (PROF && (tid0 & 63) == 0)
    for (int i = 0; i < MZ_NPHASE; ++i) prof[((size_t)blockIdx.x * 4 + (tid0 >> 6)) * MZ_NPHASE + i] = pacc[i];
}
