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
#define WS_A0 16    // pieces in flight ahead of consumption; WS_A0 + 2*6 <= WS_R

struct WS {
  const f32x4 *src;   // this wave's stream + lane
  f32x4 *ring;        // this wave's ring (wave-uniform LDS base)
  int gpos, islot, cslot, np;
};

__device__ __forceinline__ void ws_issue(WS &s) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s.src + (size_t)s.gpos * 64),
                                   (__attribute__((address_space(3))) void *)(s.ring + s.islot * 64), 16, 0, 0);
  s.gpos = (s.gpos + 1 == s.np) ? 0 : s.gpos + 1;
  s.islot = (s.islot + 1 == WS_R) ? 0 : s.islot + 1;
}

// issue P new pieces, wait until the P oldest outstanding ones have landed, read them
template <int P>
__device__ __forceinline__ void ws_step(WS &s, int lane, f32x4 (&v)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) ws_issue(s);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WS_A0) : "memory");
#pragma unroll
  for (int p = 0; p < P; ++p) {
    v[p] = s.ring[s.cslot * 64 + lane];
    s.cslot = (s.cslot + 1 == WS_R) ? 0 : s.cslot + 1;
  }
}

__device__ __forceinline__ void mz_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fc1 from the stream: 16 bias pieces, then ks x 4 weight pieces.  B operand: x[m][k] from the row-major
// tile xR (k < 50), one-hot(action) for 50 <= k < 50 + A (networks.py:167-174), else 0.
__device__ __forceinline__ void sfc1(WS &s, const float *xR, int act_m, int ks, int lane, f32x4 (&acc)[16]) {
  const int g = lane >> 4, m = lane & 15;
#pragma unroll
  for (int tg = 0; tg < 4; ++tg) {
    f32x4 v[4];
    ws_step<4>(s, lane, v);
    acc[4 * tg + 0] = v[0]; acc[4 * tg + 1] = v[1]; acc[4 * tg + 2] = v[2]; acc[4 * tg + 3] = v[3];
  }
  for (int st = 0; st < ks; ++st) {
    f32x4 v[4];
    ws_step<4>(s, lane, v);
    const int k = 4 * st + g;
    const float x = (k < MZ_H) ? xR[m * MZ_HS + k] : ((k - MZ_H == act_m) ? 1.f : 0.f);
#pragma unroll
    for (int tg = 0; tg < 4; ++tg) {
      acc[4 * tg + 0] = mz_mfma(v[tg][0], x, acc[4 * tg + 0]);
      acc[4 * tg + 1] = mz_mfma(v[tg][1], x, acc[4 * tg + 1]);
      acc[4 * tg + 2] = mz_mfma(v[tg][2], x, acc[4 * tg + 2]);
      acc[4 * tg + 3] = mz_mfma(v[tg][3], x, acc[4 * tg + 3]);
    }
  }
}

// fc2 from the stream: head A (JA tiles, B operands hid[0..7]) and head B (JB tiles, hid[8..15]) together
template <int JA, int JB>
__device__ __forceinline__ void sfc2(WS &s, int lane, const f32x4 (&hid)[16], f32x4 (&out)[JA + JB]) {
#pragma unroll
  for (int j = 0; j < JA + JB; ++j) out[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    f32x4 v[JA + JB];
    ws_step<JA + JB>(s, lane, v);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < JA; ++j) out[j] = mz_mfma(v[j][r], hid[t][r], out[j]);
#pragma unroll
      for (int j = 0; j < JB; ++j) out[JA + j] = mz_mfma(v[JA + j][r], hid[8 + t][r], out[JA + j]);
    }
  }
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

// relu(LayerNorm) of fin rows [row0,row0+50) -> row-major tile xR[m][0..51]
__device__ __forceinline__ void sln_relu(const float *fin, float *xR, const float *lnw, const float *lnb, int row0,
                                         int lane) {
  const int m = lane >> 2, q = lane & 3;
  float s = 0.f;
  for (int f = q; f < MZ_H; f += 4) s += fin[(row0 + f) * 16 + m];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  const float mean = s / (float)MZ_H;
  float v = 0.f;
  for (int f = q; f < MZ_H; f += 4) { const float d = fin[(row0 + f) * 16 + m] - mean; v += d * d; }
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  const float rstd = 1.0f / sqrtf(v / (float)MZ_H + 1e-5f);
  for (int f = q; f < MZ_HS; f += 4) {
    float y = 0.f;
    if (f < MZ_H) {
      y = (fin[(row0 + f) * 16 + m] - mean) * rstd * lnw[f] + lnb[f];
      y = fmaxf(y, 0.f);
    }
    xR[m * MZ_HS + f] = y;
  }
}

#define MZ_FUSED_LDS_FLOATS (4 * WS_R * 256 + 16 * MZ_HS + 4 * 6 * 256 + 96 * 16 + 16 + 16 + 16 * 32 + 16 + 96 + 64 + 64 + 64)

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

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * MZ_ROWS;
  const size_t per_tree = (size_t)(t.sims + 1) * MZ_HS;

  // constants -> LDS (ordinary loads only BEFORE the first LDS-DMA is in flight)
  if (tid < 96) s_b2[tid] = n.b2[tid];
  if (tid < 32 + 16 * JTP) s_b4[tid] = n.b4[tid];
  if (tid < 64) { s_lnw[tid] = n.lnw[tid]; s_lnb[tid] = n.lnb[tid]; }

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
  ws.np = np; ws.gpos = 0; ws.islot = 0; ws.cslot = 0;
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll 1
  for (int i = 0; i < WS_A0; ++i) ws_issue(ws);

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
        if (tl == 0) s_act[mt] = my_act[i];
      }
    }
    STAMP(0)
    mz_bar();
    STAMP(1)

    // ---- network: dynamics + prediction (networks.py:31-34)
    {
      f32x4 acc[16];
      sfc1(ws, xR, s_act[lane & 15], n.ks1, lane, acc);
      mz_relu<16>(acc);
      STAMP(2)
      f32x4 out[6];
      sfc2<2, 4>(ws, lane, acc, out);
      STAMP(3)
      scombine<6>(red, fin, out, s_b2, tid);
      STAMP(4)
    }
    if (w == 0) {
      sln_relu(fin, xR, s_lnw, s_lnb, 32, lane);
    } else if (w == 1) {
      const float r = mz_support_to_scalar(fin, 0, n.Sr, n.rmin, n.no_transform, lane);
      if ((lane & 3) == 0) s_rew[lane >> 2] = r;
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
      sfc1(ws, xR, -1, n.ks3, lane, acc);
      mz_relu<16>(acc);
      STAMP(6)
      f32x4 out[2 + JTP];
      sfc2<2, JTP>(ws, lane, acc, out);
      STAMP(7)
      scombine<2 + JTP>(red, fin, out, s_b4, tid);
      STAMP(8)
    }
    if (w == 0) {
      const float v = mz_support_to_scalar(fin, 0, n.Sv, n.vmin, n.no_transform, lane);
      if ((lane & 3) == 0) s_val[lane >> 2] = v;
    } else {
      for (int idx = tid - 64; idx < 16 * n.A; idx += 192) {
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
  if (PROF && lane == 0)
    for (int i = 0; i < MZ_NPHASE; ++i) prof[((size_t)blockIdx.x * 4 + w) * MZ_NPHASE + i] = pacc[i];
}
