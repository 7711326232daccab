// mz_common.h -- device-side views shared by the tree and network kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MZ_H 50          // FCNetwork hidden_dim (reference networks.py:135)
#define MZ_HS 52         // hidden-state row stride in the pool (16-byte aligned rows, pad = 0)
#define MZ_F 512         // FC head width
#define MZ_ROWS 16       // trees per workgroup = MFMA N dimension of v_mfma_f32_16x16x4_f32

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Node pool in HBM: one slab of NN nodes per tree, ONE 32-byte record per node (array of structs), children of one node
// contiguous.  Node numbering: root 0; child a of the node with expansion index e is 1 + e*A + a.  The fields a descent reads
// of a child -- visit count, expansion index, prior, value sum, reward -- sit in one record, the A children of a node in 32 A
// contiguous bytes: a level of MCTS.select_child (mcts.py:104-113) over 4 children is ONE 128-byte line (until r04 the pool
// was six arrays: a level touched five lines and used 16-32 bytes of each -- 4-7 x the algorithmic bytes by the FETCH_SIZE /
// WRITE_SIZE counters, profiles/r03_tree_traffic.json).  A slab starts MZ_NODE_OFF records into its NS-record stride so that
// the child blocks of a 4-action tree are line-aligned.  Code addresses fields as t.N[slab + node] etc. (MzField).
struct MzNode {
  double W;          // value_sum              (mcts.py:33)
  double P;          // prior                  (mcts.py:36)
  int32_t N;         // visit_count            (mcts.py:32)
  int32_t E;         // expansion index / hidden slot, -1 = leaf (mcts.py:39-40)
  float R;           // reward (float32 network scalar, exact in double) (mcts.py:34)
  int8_t TP;         // to_play                (mcts.py:37)
  int8_t pad_[3];
};
#define MZ_NODE_OFF 3
template <class T, int OFF>
struct MzField {       // one field of the records, indexable like the array it used to be
  MzNode *base;
  __device__ __forceinline__ T &operator[](size_t i) const { return *(T *)((char *)(base + i) + OFF); }
  __host__ __device__ __forceinline__ MzField operator+(size_t o) const { return MzField{base + o}; }
};
struct TreeView {
  MzField<int32_t, 16> N;
  MzField<double, 0> W;
  MzField<double, 8> P;
  MzField<float, 24> R;
  MzField<int32_t, 20> E;
  MzField<int8_t, 28> TP;
  int NS;            // records per tree slab (NN + MZ_NODE_OFF rounded up to a multiple of 4)
  uint32_t *legal;   // [B] bit a = root child a exists (actors.py:141-142)
  double *mn, *mx;   // [B] MinMaxStats                (mcts.py:6-25)
  int32_t *nexp;     // [B] expansions so far
  int32_t *path;     // [B][PL] last search path (node indices)
  int32_t *plen;     // [B]
  int8_t *leaf_tp;   // [B] to_play at the selected leaf
  int32_t *leaf, *slot, *act, *depth;   // [B] outputs of the last descent
  float *hpool;      // [B][sims+1][MZ_HS] hidden states of expanded nodes
  float *value, *reward, *logits;       // network outputs of the current simulation: [B],[B],[B][A]
  float *root_value; // [B] initial_inference value (actors.py:147)
  float *root_logits;// [B][A]
  double *noise;     // [B][A] Dirichlet draw mixed into the root
  const double *sqrttab;  // [sims+2] sqrt(n)                      (host libm, mcts.py:117)
  const double *logtab;   // [sims+2] log((n + base + 1) / base) + init    (host libm, mcts.py:116)
  const double *pbctab;   // [sims+2][sims+2] pb_c(Np, Nc) = logtab[Np] * (sqrttab[Np] / (Nc + 1))  (mcts.py:116-117)
  int B, A, sims, NN, PL;
  int two_players, has_min, has_max;
  double min_bound, max_bound, discount, init_value_score;
  // Test instrumentation of the fused search kernels (mz_sim_io), NULL in production: per move, tree and expansion slot
  // the three things a simulation's tree step consumes -- value, reward (float32 scalars) and the A policy logits --
  // [keep moves][B][sims + 1][2 + A] float32; slot 0 = the root of a self-play move (value, 0, logits), slot 1 + s =
  // simulation s.  sim_io_keep > 0: the kernels store what they computed (log); < 0: they read these values instead of
  // their own (inject), |sim_io_keep| = moves the buffer holds.
  float *sim_io;
  int sim_io_keep;
};

// first record index of tree b's slab (node k of tree b = record mz_slab(t, b) + k)
__device__ __forceinline__ size_t mz_slab(const TreeView &t, int b) { return (size_t)b * (size_t)t.NS + MZ_NODE_OFF; }

// (development switch -DMZ_NO_SIM_IO: the kernels without the instrumentation's branches, for A/B runs of its cost)
#ifdef MZ_NO_SIM_IO
#define MZ_SIM_IO_ON false
#else
#define MZ_SIM_IO_ON true
#endif
// the row of (move, tree b, expansion slot) in the sim_io buffer
__device__ __forceinline__ float *mz_sim_io_row(const TreeView &t, int b, unsigned long long move, int slot) {
  const int keep = t.sim_io_keep < 0 ? -t.sim_io_keep : t.sim_io_keep;
  return t.sim_io + (((size_t)(move % (unsigned long long)keep) * (size_t)t.B + (size_t)b) * (size_t)(t.sims + 1) + (size_t)slot) * (size_t)(2 + t.A);
}

// inject mode's reads, issued and awaited inside one asm statement: the compiler never sees a pending load, so the
// production path behind the (never taken) branch gets no s_waitcnt for it, and "+v" overwrites the value where it lives
__device__ __forceinline__ void mz_sim_io_load(float &dst, const float *src) {
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(dst) : "v"(src) : "memory");
}

// Packed FCNetwork weights in MFMA operand order (see mz_net.hip.h).
struct NetView {
  const f32x4 *w0o;       // representation out   [4 jt][4][8][64] (source of the root kernel's stream)
  const float *b0o;       // [64]
  const f32x4 *w1, *b1;   // dynamics fc1 (reward | transition)  [4][4][ks1][64], [4][16][64]
  const f32x4 *w2;        // reward out (2 jt) | transition out (4 jt): [6][4][8][64]
  const float *b2;        // [32 | 64]
  const f32x4 *w3, *b3;   // prediction fc1 (value | policy)     [4][4][13][64], [4][16][64]
  const f32x4 *w4;        // value out (2 jt) | policy out (JTP jt)
  const float *b4;        // [32 | 16*JTP]
  const float *lnw, *lnb; // [64] LayerNorm affine (pad 0)
  int ks0, ks1, ks3;      // k-steps (K/4) of the three fc1 stages
  int O, A, jtp;
  int Sr, Sv, rmin, vmin; // support sizes / minima (config.py:12-19)
  int no_transform;       // 1 = --no_target_transform, 2 = --no_support (scalar heads, Sr = Sv = 1)
};
