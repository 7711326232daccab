// mz_root.hip.h -- the root of every move in one kernel: BaseNetwork.initial_inference (reference
// networks.py:26-29: representation 146-149 + prediction 151-156, inverse_transform config.py:27-33) for 16
// observations per workgroup on the f32 matrix cores and, in the self-play loop, everything Actor.play_game does
// between env.observe and MCTS.run (actors.py:134-143): the observation of the synthetic env, Node(0) +
// root.expand + add_exploration_noise with the Dirichlet draw (mcts.py:47-61) and the first descent.
//
// Same scheme as the search kernel (mz_fused.hip.h): weights are the MFMA A operand, the 16 rows the B operand,
// bias = weight column of a constant-1 input, every wave walks its own weight stream of 4-KiB steps (16 MFMAs)
// in consumption order through a register ring.  The stream is consumed once per launch, so there is no
// residency and the ring is only 4 deep.  The first stage (O+1 -> 512) has a run-time length (obs_dim is not a
// template parameter): a double-buffered loop over steps of two k-steps x 8 tiles, the observation tile chunked
// through LDS 256 columns at a time; the rest (512 -> 50, LayerNorm, 51 -> 2x512, 512 -> 31 | A) is unrolled.
#pragma once
#include "mz_fused.hip.h"
#include "mz_selfplay.hip.h"

#define MZ_ROOT_NB 6       // register ring depth of the unrolled part (steps): the stream is L2-cold at every launch
#define MZ_ROOT_KCH 256     // observation columns per LDS chunk (32 steps)
#define MZ_ROOT_XS 260      // row stride of the observation tile (== 4 mod 32: rows 4 banks apart)

template <int JTP>
struct RootSched {
  static constexpr int R2 = 8;                   // representation out: 4 tiles x this wave's 8 hidden tiles
  static constexpr int P1 = (MZ_H + 1 + 3) / 4;  // prediction fc1: K = 51 -> 13 steps
  static constexpr int P2 = 2 * (2 + JTP);       // value (2 tiles) + policy (JTP tiles)
  static constexpr int NS = R2 + P1 + P2;
};
// steps of the run-time first stage: two k-steps of K0 = obs_dim + 1 (bias column) per step
__host__ __device__ inline int mz_root_nst0(int O) { return ((O + 1 + 3) / 4 + 1) / 2; }

#define MZ_ROOT_LDS_FLOATS (16 * MZ_ROOT_XS + 16 * MZ_HS + 4 * 6 * 256 + 96 * 16 + 16 + 16 * 32 + 64 + 64 + 64 + 64 + 16 * MZ_XE)

// The whole root of one move for the 16 rows of this workgroup (256 threads), on MZ_ROOT_LDS_FLOATS floats of LDS the
// caller provides: the body of k_root, and the head of every move of the persistent self-play kernel (mz_fused.hip.h,
// HEAD), which runs it on the space its trees are about to occupy.
// stampf(k): phase hook (k = 0 first stage done, 1 representation + LayerNorm done, 2 prediction done); MzNoStamp outside
// the profiled persistent kernel
// GAME (whole-moves launch of a game environment, mz_selfplay_set_env; SELFPLAY only): the observation is turn * board of
// the TicTacToe state (custom_environments/tic_tac_toe.py:24,50), the root is expanded over the legal moves for the player
// to move (actors.py:141-142), the Dirichlet draw covers the legal actions only (mcts.py:58-59) or is the host's
// (parity runs); root_stage additionally receives to_play (slot 33) and the legal mask (slot 34).
template <int JTP, int G, bool SELFPLAY, bool GAME = false, class STAMPF>
__device__ __forceinline__ void mz_root_body(const NetView &n, const TreeView &t, const float *obs_in,
                                             const f32x4 *istream, int nst0, const SelfplayState &sp, uint64_t seed,
                                             double alpha, double frac, float *smem, int tid, double *root_stage,
                                             STAMPF stampf, int *envs) {
  // envs (LDS, optional; the whole-moves launch): per row MZ_ENVW words of environment state the launch keeps across its
  // moves -- [0,1] move counter, [2] step, [3] episode, [4,5] temperature, [6] root value of this move, [8..] the raw
  // observation if it has at most MZ_ENVW - 8 elements -- so that
  // neither the root nor the end of the move waits for dependent global loads of per-environment scalars
  using SC = RootSched<JTP>;
  constexpr int NB = MZ_ROOT_NB, NS = SC::NS, E_R2 = SC::R2, E_P1 = E_R2 + SC::P1, NJ2 = 2 + JTP;
  constexpr int XS = MZ_ROOT_XS;
  float *xO = smem;                       // [16][XS] observation chunk, row-major, column O = 1 (bias)
  float *xR = xO + 16 * XS;               // [16][MZ_HS] hidden tile
  float *red = xR + 16 * MZ_HS;           // split-K partials
  float *fin = red + 4 * 6 * 256;         // combined outputs [96][16]
  float *s_val = fin + 96 * 16;
  float *s_lg = s_val + 16;               // [16][32]
  float *s_b0o = s_lg + 16 * 32;
  float *s_b4 = s_b0o + 64;
  float *s_lnw = s_b4 + 64;
  float *s_lnb = s_lnw + 64;
  float *xEp = s_lnb + 64;                // [16][MZ_XE] prediction extension: 1 (bias column), then 0

  const int w = tid >> 6, lane = tid & 63;
  const int g4 = lane >> 4, m16 = lane & 15;
  const int b0 = blockIdx.x * MZ_ROWS;
  const size_t per_tree = (size_t)(t.sims + 1) * MZ_HS;
  const int O = n.O;

  if (tid < 64) { s_b0o[tid] = n.b0o[tid]; s_lnw[tid] = n.lnw[tid]; s_lnb[tid] = n.lnb[tid]; }
  if (tid < 32 + 16 * JTP) s_b4[tid] = n.b4[tid];
  for (int i = tid; i < 16 * MZ_XE; i += 256) xEp[i] = (i % MZ_XE == 0) ? 1.f : 0.f;

  // this wave's stream: [nst0 + NS steps][4 pieces][64 lanes] f32x4
  const char *wb = (const char *)(istream + (size_t)__builtin_amdgcn_readfirstlane(w) * (size_t)(nst0 + NS) * 256);
  const char *ws = wb + (size_t)nst0 * 4096;
  const unsigned lane_off = (unsigned)lane * 16u;
#define MZ_RLOAD(base, piece)                                                                                  \
  (*(const __attribute__((address_space(1))) f32x4 *)((const __attribute__((address_space(1))) char *)(base) + \
                                                       (size_t)(piece) * 1024 + lane_off))
  f32x4 A0[4], A1[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) A0[p] = MZ_RLOAD(wb, p);
  f32x4 Bf[NB][4];
#pragma unroll
  for (int s = 0; s < NB - 1; ++s) {
#pragma unroll
    for (int p = 0; p < 4; ++p) Bf[s][p] = MZ_RLOAD(ws, s * 4 + p);
  }

  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  mz_valu_fence16v(acc);

  // ---- representation fc1: K0 = O + 1 columns, 512 outputs = 8 tiles per wave; one step = 2 k-steps
  auto rstep = [&](const f32x4 (&Aw)[4], int sl) __attribute__((always_inline)) {
    const float x0 = xO[m16 * XS + 8 * sl + g4], x1 = xO[m16 * XS + 8 * sl + 4 + g4];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      mz_mfma_v(acc[4 * tg + 0], Aw[tg][0], x0); mz_mfma_v(acc[4 * tg + 1], Aw[tg][1], x0);
      mz_mfma_v(acc[4 * tg + 2], Aw[tg][2], x0); mz_mfma_v(acc[4 * tg + 3], Aw[tg][3], x0);
    }
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      mz_mfma_v(acc[4 * tg + 0], Aw[2 + tg][0], x1); mz_mfma_v(acc[4 * tg + 1], Aw[2 + tg][1], x1);
      mz_mfma_v(acc[4 * tg + 2], Aw[2 + tg][2], x1); mz_mfma_v(acc[4 * tg + 3], Aw[2 + tg][3], x1);
    }
  };
  for (int s0 = 0; s0 < nst0; s0 += MZ_ROOT_KCH / 8) {
    const int s1 = (nst0 < s0 + MZ_ROOT_KCH / 8) ? nst0 : s0 + MZ_ROOT_KCH / 8;
    const int kc = 8 * (s1 - s0), c0 = 8 * s0;
    if (s0) __syncthreads();
    for (int idx = tid; idx < 16 * kc; idx += 256) {
      const int m = idx / kc, kk = idx - m * kc, k = c0 + kk, b = b0 + m;
      float v = 0.f;
      if (k < O) {
        if (b < t.B) {
          if constexpr (GAME) {       // tic_tac_toe.py:24,50: observation = turn * board, np.float32 (actors.py:134)
            v = (float)((int)sp.turn[b] * (int)sp.board[(size_t)b * 9 + k]);
            sp.obs[(size_t)b * O + k] = v;
          } else
          if constexpr (SELFPLAY) {   // Game.get_observation(-1) of the synthetic env (game.py:117-121)
            const uint32_t env = (uint32_t)(sp.env_offset + b);
            const uint32_t ep = (uint32_t)(envs ? envs[m * MZ_ENVW + 3] : sp.episode[b]), tt = (uint32_t)(envs ? envs[m * MZ_ENVW + 2] : sp.t[b]);
            v = sp.obs_u8 ? mz_synth_obs_u8(seed, env, ep, tt, (uint32_t)k) : mz_synth_obs_elem(seed, env, ep, tt, (uint32_t)k);
            sp.obs[(size_t)b * O + k] = v;            // History.observations keeps the raw observation (game.py:93-96)
            if (envs && O <= MZ_ENVW - 8) ((float *)envs)[m * MZ_ENVW + 8 + k] = v;      // (small observations: a copy for the record)
            if (sp.obs_min) v = (v - sp.obs_min[k]) / sp.obs_rng[k];      // actors.py:134-137, float32 like numpy's
          } else {
            v = obs_in[(size_t)b * O + k];
          }
        }
      } else if (k == O) {
        v = 1.f;
      }
      xO[m * XS + kk] = v;
    }
    __syncthreads();
    for (int s = s0; s < s1; s += 2) {
      if (s + 1 < nst0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) A1[p] = MZ_RLOAD(wb, (s + 1) * 4 + p);
      }
      rstep(A0, s - s0);
      if (s + 1 < s1) {
        if (s + 2 < nst0) {
#pragma unroll
          for (int p = 0; p < 4; ++p) A0[p] = MZ_RLOAD(wb, (s + 2) * 4 + p);
        }
        rstep(A1, s + 1 - s0);
      }
    }
  }
  mz_mfma_fence16v(acc);
  stampf(0);
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    acc[tt][0] = fmaxf(acc[tt][0], 0.f); acc[tt][1] = fmaxf(acc[tt][1], 0.f);
    acc[tt][2] = fmaxf(acc[tt][2], 0.f); acc[tt][3] = fmaxf(acc[tt][3], 0.f);
  }
  mz_valu_fence16v(acc);

  // ---- representation out + LayerNorm, prediction: NS steps unrolled
  f32x4 out0[4];
  f32x4 out4[NJ2];
  float xq = 0.f;
  mz_static_for<NS>([&](auto S_) __attribute__((always_inline)) {
    constexpr int s = decltype(S_)::value;
    if constexpr (s + NB - 1 < NS) {
      constexpr int pb = (s + NB - 1) % NB;
#pragma unroll
      for (int p = 0; p < 4; ++p) Bf[pb][p] = MZ_RLOAD(ws, (s + NB - 1) * 4 + p);
    }
    constexpr int cb = s % NB;
    if constexpr (s < E_R2) {
      // hidden tile s of this wave x the 4 output tiles (50 features padded to 64)
      if constexpr (s == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) mz_mfma_a(out0[jt], Bf[cb][jt][r], acc[s][r]);
      }
      if constexpr (s == E_R2 - 1) {
        mz_mfma_fence<4>(out0);
        scombine<4>(red, fin, out0, s_b0o, tid);
        {
          const int col = 4 * w + (lane >> 4), q = lane & 15;
          sln_relu16(fin, xR, s_lnw, s_lnb, 0, col, q);
        }
        mz_bar();
        stampf(1);
        if (tid < 16 * (MZ_HS / 4)) {      // hidden state of the root -> pool slot 0
          const int m = tid / (MZ_HS / 4), c = tid % (MZ_HS / 4);
          if (b0 + m < t.B) {
            f32x4 *dst = (f32x4 *)(t.hpool + (size_t)(b0 + m) * per_tree);
            dst[c] = *(const f32x4 *)(xR + m * MZ_HS + 4 * c);
          }
        }
      }
    } else if constexpr (s < E_P1) {
      constexpr int st = s - E_R2;
      if constexpr (st == 0) mz_xval_async(xq, xR, xEp, m16, g4);
      mz_lds_wait(xq);
      const float x = xq;
      if constexpr (st + 1 < SC::P1) mz_xval_async(xq, xR, xEp, m16, 4 * (st + 1) + g4);
#pragma unroll
      for (int tg = 0; tg < 4; ++tg) {
        if constexpr (st == 0) {
          mz_mfma_v0(acc[4 * tg + 0], Bf[cb][tg][0], x); mz_mfma_v0(acc[4 * tg + 1], Bf[cb][tg][1], x);
          mz_mfma_v0(acc[4 * tg + 2], Bf[cb][tg][2], x); mz_mfma_v0(acc[4 * tg + 3], Bf[cb][tg][3], x);
        } else {
          mz_mfma_v(acc[4 * tg + 0], Bf[cb][tg][0], x); mz_mfma_v(acc[4 * tg + 1], Bf[cb][tg][1], x);
          mz_mfma_v(acc[4 * tg + 2], Bf[cb][tg][2], x); mz_mfma_v(acc[4 * tg + 3], Bf[cb][tg][3], x);
        }
      }
      if constexpr (s == E_P1 - 1) {
        mz_mfma_fence16v(acc);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
          acc[tt][0] = fmaxf(acc[tt][0], 0.f); acc[tt][1] = fmaxf(acc[tt][1], 0.f);
          acc[tt][2] = fmaxf(acc[tt][2], 0.f); acc[tt][3] = fmaxf(acc[tt][3], 0.f);
        }
        mz_valu_fence16v(acc);
      }
    } else {
      constexpr int step = s - E_P1;
      if constexpr (step == 0) {
#pragma unroll
        for (int j = 0; j < NJ2; ++j) out4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int q = 4 * step + q4, tt = q / NJ2, jt = q % NJ2;
          mz_mfma_a(out4[jt], Bf[cb][q4][r], acc[jt < 2 ? tt : 8 + tt][r]);
        }
      }
      if constexpr (s == NS - 1) {
        mz_mfma_fence<NJ2>(out4);
        scombine<NJ2>(red, fin, out4, s_b4, tid);
        const int col = 4 * w + (lane >> 4), q = lane & 15;
        const bool live = b0 + col < t.B;
        const float v = mz_support_to_scalar16(fin, 0, n.Sv, n.vmin, n.no_transform, col, q);
        if (q == 0) {
          s_val[col] = v;
          if (live) t.root_value[b0 + col] = v;
          if (envs) envs[col * MZ_ENVW + 6] = __builtin_bit_cast(int, v);
        }
        for (int a = q; a < n.A; a += 16) {
          const float lg = fin[(32 + a) * 16 + col];
          s_lg[col * 32 + a] = lg;
          if (live) t.root_logits[(size_t)(b0 + col) * n.A + a] = lg;
        }
      }
    }
  });

  stampf(2);
  if constexpr (SELFPLAY) {
    // Node(0), root.expand(all actions legal), add_exploration_noise (Dirichlet from the device RNG: lane a
    // draws Gamma(alpha) for action a, the group normalises), then the first descent of MCTS.run
    mz_bar();
    constexpr int TL = (G <= 16) ? 16 : 32;
    constexpr int NPASS = 16 * TL / 256;
    const int tl = tid % TL, A = t.A;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int mt = tid / TL + i * (256 / TL);
      const int b = b0 + mt;
      if (b < t.B && tl < G) {
        const uint64_t move = envs ? (uint64_t)*(const unsigned long long *)(envs + mt * MZ_ENVW) : (uint64_t)sp.movecnt[b];
        double nz;
        if constexpr (GAME) {
          // legal_actions() = the empty cells (tic_tac_toe.py:27-28), game.to_play = env.turn
          const int to_play = (int)sp.turn[b];
          const bool ok = tl < A && sp.board[(size_t)b * 9 + (tl < 9 ? tl : 0)] == 0;
          uint32_t mask = 0;
          for (int a = 0; a < A; ++a) mask |= (__shfl((int)ok, a, G) ? 1u : 0u) << a;
          if (sp.draws_noise) {
            nz = tl < A ? t.noise[(size_t)b * A + tl] : 0.0;      // the host's draw, at the legal positions (mz_selfplay_set_draws)
          } else {
            const double gam = ok ? mz_gamma(alpha, seed, (uint32_t)(sp.env_offset + b), move, (uint32_t)tl) : 0.0;
            double sum = 0.0;
            for (int a = 0; a < A; ++a) sum += __shfl(gam, a, G);
            nz = ok ? (sum > 0.0 ? gam / sum : 1.0 / (double)__popc(mask)) : 0.0;
            if (tl < A) t.noise[(size_t)b * A + tl] = nz;
          }
          __threadfence_block();
          mz_tree_root<G, true>(t, b, tl, to_play, mask, s_lg + mt * 32, t.noise + (size_t)b * A, frac,
                                root_stage ? root_stage + mt * 96 : nullptr);
          if (tl == 0 && root_stage) { root_stage[mt * 96 + 33] = (double)to_play; root_stage[mt * 96 + 34] = (double)mask; }
        } else {
          const double gam = tl < A ? mz_gamma(alpha, seed, (uint32_t)(sp.env_offset + b), move, (uint32_t)tl) : 0.0;
          double sum = 0.0;
          for (int a = 0; a < A; ++a) sum += __shfl(gam, a, G);
          nz = sum > 0.0 ? gam / sum : 1.0 / A;
          if (tl < A) t.noise[(size_t)b * A + tl] = nz;
          __threadfence_block();
          const uint32_t mask = (A >= 32) ? 0xFFFFFFFFu : ((1u << A) - 1u);
          mz_tree_root<G, true>(t, b, tl, 1, mask, s_lg + mt * 32, t.noise + (size_t)b * A, frac,
                                root_stage ? root_stage + mt * 96 : nullptr);
        }
        // test instrumentation (mz_selfplay_noise_log), off in production.  AFTER the root: a second, possibly aliasing
        // store between the one above and mz_tree_root's read of it keeps the compiler from forwarding the value in
        // registers and costs a whole store -> load round trip per move (A/B on one box: 0.3 % of the move)
        if (sp.noise_log && tl < A)
          sp.noise_log[((size_t)((uint32_t)move % (uint32_t)sp.ring_moves) * (uint32_t)t.B + (uint32_t)b) * (uint32_t)A + (uint32_t)tl] = nz;
        // test instrumentation (mz_sim_io, log mode), off in production: the root's value and logits of THIS move in slot 0
        // of the move's row (the search kernel that follows fills slots 1..)
        if (t.sim_io_keep > 0 && tl < A) {
          float *io = mz_sim_io_row(t, b, move, 0);
          io[2 + tl] = s_lg[mt * 32 + tl];
          if (tl == 0) { io[0] = s_val[mt]; io[1] = 0.f; }
        }
      }
    }
  }
#undef MZ_RLOAD
}

template <int JTP, int G, bool SELFPLAY>
__global__ __launch_bounds__(256, 1) void k_root(NetView n, TreeView t, const float *obs_in, const f32x4 *istream,
                                                  int nst0, SelfplayState sp, uint64_t seed, double alpha,
                                                  double frac) {
  __shared__ __attribute__((aligned(16))) float smem[MZ_ROOT_LDS_FLOATS];
  mz_root_body<JTP, G, SELFPLAY, false>(n, t, obs_in, istream, nst0, sp, seed, alpha, frac, smem, (int)threadIdx.x, nullptr, MzNoStamp(), nullptr);
}
