// mz_fused.hip.h -- the persistent fused search kernel: MCTS.run (reference mcts.py:78-102) for 16 trees
// per workgroup, ALL simulations in one launch.
//
//   workgroup = 4 wavefronts (one per SIMD) = 16 trees = the 16 columns of v_mfma_f32_16x16x4_f32.
//   per simulation:  [tree lanes] gather parent hidden + action  ->  [4 waves] dynamics + prediction on
//   the matrix cores  ->  [tree lanes] expand + backup + next descent.   Trees never leave their
//   workgroup, so there is no grid-wide synchronisation and no host round trip between simulations.
//
// Weight streaming: every wave consumes its share of the network (~210 KiB per simulation) as ONE cyclic
// stream of 1-KiB pieces laid out in exactly the order the MFMAs need them.  Pieces travel L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4) into a per-wave ring and are read back with one ds_read_b128 per
// piece; the ring runs WS_A0 pieces ahead of consumption behind a COUNTED s_waitcnt vmcnt(WS_A0), also
// across the barriers and the tree phases (raw s_barrier, never a vmcnt(0) drain in the loop), so the
// L2 latency is paid once per launch instead of once per piece.  Because the stream is cyclic the
// prefetch for simulation s+1 is in flight while simulation s finishes.
#pragma once
#include "mz_common.h"
#include "mz_net.hip.h"
#include "mz_tree.hip.h"

#define WS_R 28     // ring slots (1 KiB each) per wave
#define WS_A0 20    // pieces in flight ahead of consumption; WS_A0 + 8 <= WS_R

struct WS {
  const f32x4 *src;   // this wave's stream + lane
  f32x4 *ring;        // this wave's ring (wave-uniform LDS base)
  unsigned ring_lds;  // its LDS byte address
  int gpos, islot, cslot, np;
  unsigned long long wait_cycles;
};

// One 1-KiB piece L2 -> LDS (global_load_lds_dwordx4: LDS address = M0 + lane*16, global address per lane).
// Issued from inline asm ON PURPOSE: hipcc models the builtin form as a FLAT access that may touch LDS
// and, while one is pending, turns every counted s_waitcnt lgkmcnt(N) it inserts into lgkmcnt(0) -- which
// puts a full LDS-latency stall in front of the first MFMA of every step.  The RAW ordering DMA -> ds_read
// is done by hand with the counted vmcnt in pipe_fetch.
__device__ __forceinline__ void ws_issue(WS &s) {
  {
    const unsigned lds = __builtin_amdgcn_readfirstlane(s.ring_lds + (unsigned)s.islot * 1024u);
    const f32x4 *g = s.src + (size_t)s.gpos * 64;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory", "m0");
  }
  s.gpos = (s.gpos + 1 == s.np) ? 0 : s.gpos + 1;
  s.islot = (s.islot + 1 == WS_R) ? 0 : s.islot + 1;
}

// MFMA from inline asm with the accumulator tied in place ("+a"): left to itself hipcc, in this kernel,
// stages every accumulator through one scratch tile (4 v_accvgpr_mov per MFMA, all MFMAs serialised on it).
// Nothing else in an asm statement is visible to its hazard recogniser, so mz_mfma_fence() supplies the
// MFMA-result -> VALU wait states once per stage.
__device__ __forceinline__ void mz_mfma_a(f32x4 &c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void mz_mfma_fence(f32x4 (&acc)[N]) {
  if constexpr (N == 16) {
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]),
                   "+a"(acc[7]), "+a"(acc[8]), "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]),
                   "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]));
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i]));
  }
}

// Pipeline invariant at the start of every 4-piece step: the step's own pieces are already in registers
// (`cur`), WS_A0 younger pieces are in flight and the oldest four of those belong to the next step.
// pipe_fetch: wait for those four (counted vmcnt), start reading them (ds_read_b128, consumed one step
// later so their LDS latency hides under this step's 16 MFMAs).  The four new DMA pieces of the step are
// issued one per MFMA group so that they take issue slots the matrix pipe leaves free.
__device__ unsigned long long g_wait_cycles_dummy;
__device__ __forceinline__ void pipe_fetch(WS &s, int lane, f32x4 (&nxt)[4]) {
#ifdef MZ_PROF_WAIT
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WS_A0 - 4) : "memory");
#ifdef MZ_PROF_WAIT
  __builtin_amdgcn_sched_barrier(0);
  s.wait_cycles += __builtin_amdgcn_s_memtime() - t0_;
  __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    nxt[p] = s.ring[s.cslot * 64 + lane];
    s.cslot = (s.cslot + 1 == WS_R) ? 0 : s.cslot + 1;
  }
}

__device__ __forceinline__ void mz_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// B operand of an fc1 k-step: x[m][k] from the row-major tile (k < 50); one-hot(action) (networks.py:167-174)
// and the constant-1 column that carries the bias (k - 50 == ones_k) above that.
#define MZ_XE 36   // row stride of the extension tile: [one-hot(action) | 1 | 0...] for k >= 50
__device__ __forceinline__ float mz_xval(const float *xR, const float *xE, int m, int k) {
  const float *p = (k < MZ_H) ? (xR + m * MZ_HS + k) : (xE + m * MZ_XE + (k - MZ_H));
  return *p;
}

// fc1 (two 512-wide heads, 16 tiles per wave) from the stream: ks steps of 4 pieces; bias = weight column
// of the constant-1 input.
__device__ __forceinline__ void pfc1(WS &s, const float *xR, const float *xE, int ks, int lane,
                                     f32x4 (&cur)[4], f32x4 (&acc)[16]) {
  const int g = lane >> 4, m = lane & 15;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float xc = mz_xval(xR, xE, m, g);
  for (int st = 0; st < ks; ++st) {
    f32x4 nxt[4];
    pipe_fetch(s, lane, nxt);
    const float xn = mz_xval(xR, xE, m, 4 * (st + 1) + g);   // beyond the last step: reads pad, unused
#pragma unroll
    for (int tg = 0; tg < 4; ++tg) {
      ws_issue(s);
      mz_mfma_a(acc[4 * tg + 0], cur[tg][0], xc);
      mz_mfma_a(acc[4 * tg + 1], cur[tg][1], xc);
      mz_mfma_a(acc[4 * tg + 2], cur[tg][2], xc);
      mz_mfma_a(acc[4 * tg + 3], cur[tg][3], xc);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) cur[p] = nxt[p];
    xc = xn;
  }
  mz_mfma_fence<16>(acc);
}

// fc2 from the stream: NJ output tiles per hidden tile t (the first JA take hid[t], the rest hid[8+t]);
// pieces arrive in (t, jt) order, four per step; the r loop is outermost inside a step so that
// consecutive MFMAs hit different accumulators.
template <int NJ, int JA>
__device__ __forceinline__ void pfc2(WS &s, int lane, f32x4 (&cur)[4], const f32x4 (&hid)[16], f32x4 (&out)[NJ]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) out[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int step = 0; step < 2 * NJ; ++step) {
    f32x4 nxt[4];
    pipe_fetch(s, lane, nxt);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ws_issue(s);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int q = 4 * step + q4, t = q / NJ, jt = q % NJ;
        mz_mfma_a(out[jt], cur[q4][r], hid[jt < JA ? t : 8 + t][r]);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) cur[p] = nxt[p];
  }
  mz_mfma_fence<NJ>(out);
}

template <int JTOT>
__device__ __forceinline__ void scombine(float *red, float *fin, const f32x4 (&out)[JTOT], const float *bias,
                                         int tid) {
  const int w = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int jt = 0; jt < JTOT; ++jt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((w * 6 + jt) * 4 + r) * 64 + lane] = out[jt][r];
  }
  mz_bar();
  for (int e = tid; e < JTOT * 256; e += 256) {
    const int jt = e >> 8, r = (e >> 6) & 3, ln = e & 63;
    const int n = 16 * jt + 4 * (ln >> 4) + r;
    float sum = bias[n];
    sum += red[((0 * 6 + jt) * 4 + r) * 64 + ln];
    sum += red[((1 * 6 + jt) * 4 + r) * 64 + ln];
    sum += red[((2 * 6 + jt) * 4 + r) * 64 + ln];
    sum += red[((3 * 6 + jt) * 4 + r) * 64 + ln];
    fin[n * 16 + (ln & 15)] = sum;
  }
  mz_bar();
}

// relu(LayerNorm) of fin rows [row0,row0+50), column m -> row-major tile xR[m][0..51]; 8 lanes per column
__device__ __forceinline__ void sln_relu8(const float *fin, float *xR, const float *lnw, const float *lnb, int row0,
                                          int m, int q) {
  float s = 0.f;
  for (int f = q; f < MZ_H; f += 8) s += fin[(row0 + f) * 16 + m];
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  const float mean = s / (float)MZ_H;
  float v = 0.f;
  for (int f = q; f < MZ_H; f += 8) { const float d = fin[(row0 + f) * 16 + m] - mean; v += d * d; }
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  const float rstd = 1.0f / sqrtf(v / (float)MZ_H + 1e-5f);
  for (int f = q; f < MZ_HS; f += 8) {
    float y = 0.f;
    if (f < MZ_H) {
      y = (fin[(row0 + f) * 16 + m] - mean) * rstd * lnw[f] + lnb[f];
      y = fmaxf(y, 0.f);
    }
    xR[m * MZ_HS + f] = y;
  }
}

// Config.inverse_transform (config.py:27-33), column m, 8 lanes per column, S <= 32 bins
__device__ __forceinline__ float mz_support_to_scalar8(const float *fin, int row0, int S, int smin, int no_transform,
                                                       int m, int q) {
  float x[4], mx = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bin = q + 8 * i;
    x[i] = bin < S ? fin[(row0 + bin) * 16 + m] : -__builtin_inff();
    mx = fmaxf(mx, x[i]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
  float e[4], sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e[i] = (q + 8 * i < S) ? expf(x[i] - mx) : 0.f;
    sum += e[i];
  }
  sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) v += (float)(smin + q + 8 * i) * (e[i] / sum);
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  if (!no_transform) {
    const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    float t = (fabsf(v) + 1.f) + 0.001f;
    t = 1.f + 0.004f * t;
    t = (sqrtf(t) - 1.f) / 0.002f;
    v = sgn * (t * t - 1.f);
  }
  return v;
}

#define MZ_FUSED_LDS_FLOATS (4 * WS_R * 256 + 16 * MZ_HS + 4 * 6 * 256 + 96 * 16 + 16 + 16 + 16 * 32 + 16 + 96 + 64 + 64 + 64 + 2 * 16 * MZ_XE)

// PROF: diagnostic build only (mz_search_phase_profile): per-wave cycle totals of each phase of the loop.
#define MZ_NPHASE 12
#define STAMP(ph)                                                              \
  if (PROF) {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
    __builtin_amdgcn_sched_barrier(0);                                         \
    pacc[ph] += now_ - tlast;                                                  \
    tlast = now_;                                                              \
  }

template <int JTP, int G, bool PROF>
__global__ __launch_bounds__(256, 1) void k_search_fused(NetView n, TreeView t, const f32x4 *wstream, int np,
                                                          int nsims, int slot0, unsigned long long *prof) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4 *ring = (f32x4 *)smem;
  float *xR = smem + 4 * WS_R * 256;
  float *red = xR + 16 * MZ_HS;
  float *fin = red + 4 * 6 * 256;
  float *s_val = fin + 96 * 16;
  float *s_rew = s_val + 16;
  float *s_lg = s_rew + 16;
  int *s_act = (int *)(s_lg + 16 * 32);
  float *s_b2 = (float *)(s_act + 16);
  float *s_b4 = s_b2 + 96;
  float *s_lnw = s_b4 + 64;
  float *s_lnb = s_lnw + 64;
  float *xEd = s_lnb + 64;            // [16][MZ_XE] dynamics input extension: one-hot(action), then 1 (bias column)
  float *xEp = xEd + 16 * MZ_XE;      // [16][MZ_XE] prediction input extension: 1 (bias column), then 0

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * MZ_ROWS;
  const size_t per_tree = (size_t)(t.sims + 1) * MZ_HS;

  // constants -> LDS (ordinary loads only BEFORE the first LDS-DMA is in flight)
  if (tid < 96) s_b2[tid] = n.b2[tid];
  if (tid < 32 + 16 * JTP) s_b4[tid] = n.b4[tid];
  if (tid < 64) { s_lnw[tid] = n.lnw[tid]; s_lnb[tid] = n.lnb[tid]; }
  for (int i = tid; i < 16 * MZ_XE; i += 256) xEp[i] = (i % MZ_XE == 0) ? 1.f : 0.f;

  // tree-lane mapping: group of G lanes per tree, 256/G trees per pass
  const int tl = tid % G;
  int my_slot[(16 * G + 255) / 256], my_act[(16 * G + 255) / 256];
  {
    int i = 0;
    for (int mt = tid / G; mt < 16; mt += 256 / G, ++i) {
      const int b = b0 + mt;
      my_slot[i] = (b < t.B) ? t.slot[b] : 0;
      my_act[i] = (b < t.B) ? t.act[b] : 0;
    }
  }

  WS ws;
  ws.src = wstream + (size_t)w * np * 64 + lane;
  ws.ring = ring + w * WS_R * 64;
  ws.ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(ws.ring);
  ws.np = np; ws.gpos = 0; ws.islot = 0; ws.cslot = 0; ws.wait_cycles = 0;
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll 1
  for (int i = 0; i < WS_A0; ++i) ws_issue(ws);
  f32x4 cur[4];
  pipe_fetch(ws, lane, cur);
#pragma unroll
  for (int i = 0; i < 4; ++i) ws_issue(ws);
  const int ks1f = (MZ_H + n.A + 1 + 3) / 4, ks3f = (MZ_H + 1 + 3) / 4;

  unsigned long long pacc[MZ_NPHASE];
  unsigned long long tlast = 0;
  if (PROF) {
    for (int i = 0; i < MZ_NPHASE; ++i) pacc[i] = 0;
    tlast = __builtin_amdgcn_s_memtime();
  }
  for (int sim = 0; sim < nsims; ++sim) {
    // ---- gather: x tile = [hidden of search_path[-2] | action]  (mcts.py:94-96)
    {
      int i = 0;
      for (int mt = tid / G; mt < 16; mt += 256 / G, ++i) {
        const int b = b0 + mt;
        const f32x4 *src = (const f32x4 *)(t.hpool + (size_t)b * per_tree + (size_t)my_slot[i] * MZ_HS);
        for (int c = tl; c < MZ_HS / 4; c += G) *(f32x4 *)(xR + mt * MZ_HS + 4 * c) = src[c];
        for (int c = tl; c < MZ_XE; c += G) xEd[mt * MZ_XE + c] = (c == my_act[i] || c == n.A) ? 1.f : 0.f;
      }
    }
    STAMP(0)
    mz_bar();
    STAMP(1)

    // ---- network: dynamics + prediction (networks.py:31-34)
    {
      f32x4 acc[16];
      pfc1(ws, xR, xEd, ks1f, lane, cur, acc);
      mz_relu<16>(acc);
      STAMP(2)
      f32x4 out[6];
      pfc2<6, 2>(ws, lane, cur, acc, out);
      STAMP(3)
      scombine<6>(red, fin, out, s_b2, tid);
      STAMP(4)
    }
    {
      const int col = 8 * (w & 1) + (lane >> 3), q = lane & 7;
      if (w < 2) {
        sln_relu8(fin, xR, s_lnw, s_lnb, 32, col, q);
      } else {
        const float r = mz_support_to_scalar8(fin, 0, n.Sr, n.rmin, n.no_transform, col, q);
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
    {
      f32x4 acc[16];
      pfc1(ws, xR, xEp, ks3f, lane, cur, acc);
      mz_relu<16>(acc);
      STAMP(6)
      f32x4 out[2 + JTP];
      pfc2<2 + JTP, 2>(ws, lane, cur, acc, out);
      STAMP(7)
      scombine<2 + JTP>(red, fin, out, s_b4, tid);
      STAMP(8)
    }
    if (w < 2) {
      const int col = 8 * w + (lane >> 3), q = lane & 7;
      const float v = mz_support_to_scalar8(fin, 0, n.Sv, n.vmin, n.no_transform, col, q);
      if (q == 0) s_val[col] = v;
    } else {
      for (int idx = tid - 128; idx < 16 * n.A; idx += 128) {
        const int m = idx / n.A, a = idx % n.A;
        s_lg[m * 32 + a] = fin[(32 + a) * 16 + m];
      }
    }
    // the hidden-state stores are older than the last WS_A0 DMA pieces: complete after this wait
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WS_A0) : "memory");
    mz_bar();
    STAMP(9)

    // ---- tree: expand + backup (mcts.py:97-99), then the next descent (mcts.py:83-92)
    {
      int i = 0;
      for (int mt = tid / G; mt < 16; mt += 256 / G, ++i) {
        const int b = b0 + mt;
        if (b < t.B) {
          mz_tree_expand_backup<G>(t, b, tl, s_val[mt], s_rew[mt], s_lg + mt * 32);
          if (sim + 1 < nsims) {
            __threadfence_block();
            mz_tree_select<G>(t, b, tl, my_slot[i], my_act[i]);
          }
        }
      }
    }
    // No ordinary VMEM load may be pending (from the compiler's point of view either) when the network
    // phase starts: a pending VGPR-destination load makes hipcc put s_waitcnt vmcnt(0) -- a full drain of
    // the DMA ring -- in front of the first overwrite of that VGPR INSIDE the MFMA loops.  A real
    // S_WAITCNT (not inline asm) is what its wait-count pass understands.  vmcnt(0) only: 0x0F70.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    STAMP(10)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (PROF) pacc[11] = ws.wait_cycles;
  if (PROF && lane == 0)
    for (int i = 0; i < MZ_NPHASE; ++i) prof[((size_t)blockIdx.x * 4 + w) * MZ_NPHASE + i] = pacc[i];
}
