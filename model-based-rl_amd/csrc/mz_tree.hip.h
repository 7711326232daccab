// mz_tree.hip.h -- MCTS tree kernels over the node pool (32-byte records, mz_common.h: MzNode; gfx950).
//
// One group of G lanes (G = power of two >= A, <= 32) owns one tree; a 64-wide wavefront carries 64/G
// trees.  Children of a node are contiguous, so the G lanes of a group read one coalesced segment per
// array per level; the arg-max over children is a log2(G)-step shuffle reduction inside the group.
// All tree arithmetic is IEEE double in the reference's operation order (this TU is compiled with
// -ffp-contract=off); log() and sqrt() of the integer parent visit count come from host-built tables
// (libm, the same values CPython's math.log/math.sqrt return), exp() is the device's.
#pragma once
#include "mz_common.h"
#include "mz_rng.h"

// MinMaxStats.normalize, reference mcts.py:16-21
__device__ __forceinline__ double mz_normalize(double v, double mn, double mx) {
  if (mx > mn) return (v - mn) / (mx - mn);
  if (mx == mn) return 1.0;
  return v;
}

// whole-record access to the node pool: two 16-byte loads / stores per node
typedef int mz_i32x4 __attribute__((ext_vector_type(4)));
union MzNodeBits { MzNode n; mz_i32x4 q[2]; };
__device__ __forceinline__ MzNode mz_node_load(const TreeView &t, size_t i) {
  MzNodeBits u;
  const mz_i32x4 *src = (const mz_i32x4 *)(t.W.base + i);
  u.q[0] = src[0]; u.q[1] = src[1];
  return u.n;
}
// a fresh Node(prior) (mcts.py:30-40): visit_count 0, value_sum 0, reward 0, no children
__device__ __forceinline__ void mz_node_fresh(const TreeView &t, size_t i, double prior, int8_t to_play) {
  MzNodeBits u;
  u.q[0] = mz_i32x4{0, 0, 0, 0}; u.q[1] = mz_i32x4{0, 0, 0, 0};
  u.n.W = 0.0; u.n.P = prior; u.n.N = 0; u.n.E = -1; u.n.R = 0.f; u.n.TP = to_play;
  mz_i32x4 *dst = (mz_i32x4 *)(t.W.base + i);
  dst[0] = u.q[0]; dst[1] = u.q[1];
}

// The descent of MCTS.run (mcts.py:83-94) with MCTS.select_child (104-113) and ucb_score (115-124).
// One dependent memory round trip per level: every child lane fetches its child's whole 32-byte record -- the A children of
// a node are 32 A contiguous bytes -- and the winner's expansion index and visit count (the next level's `e` and parent
// count) ride along in the arg-max instead of being fetched afterwards.
template <int G>
__device__ __forceinline__ void mz_tree_select(const TreeView &t, int b, int lane, int &slot_out, int &act_out) {
  const int A = t.A;
  const size_t o = mz_slab(t, b);
  int32_t *path = t.path + (size_t)b * t.PL;
  const double mn = t.mn[b], mx = t.mx[b];
  const uint32_t legal = t.legal[b];
  int node = 0, parent_e = 0, len = 1, a_sel = -1;
  const MzNode root = mz_node_load(t, o);
  int tp = root.TP;
  if (lane == 0) path[0] = 0;
  int e = root.E, Np = root.N;
  while (e >= 0) {
    const int ch = 1 + e * A + lane;
    const bool valid = lane < A && (node != 0 || ((legal >> lane) & 1u));
    double score = 0.0;
    int best = -1, ce = -1, cn = 0;
    // log((Np + base + 1) / base) + init and sqrt(Np) of THIS level's parent count (host libm tables): requested beside the
    // children's records, not behind them -- pb_c is then the reference's own two operations on them (mcts.py:116-117:
    // pb_c = log(..) + init; pb_c *= sqrt(Np) / (Nc + 1)), IEEE double like the host-built pb_c table of the fused kernels
    const double lgn = t.logtab[Np], sqn = t.sqrttab[Np];
    if (valid) {
      const MzNode c = mz_node_load(t, o + ch);
      const int Nc = c.N;
      const double p = c.P;
      ce = c.E; cn = Nc;
      if (Np == 0) {
        score = p;                                     // fresh root: rank by prior (mcts.py:105-108)
      } else {
        const double pb_c = lgn * (sqn / (double)(Nc + 1));
        const double prior_score = pb_c * p;
        double value_score;
        if (Nc > 0) {
          const double q = c.W / (double)Nc;
          const double v = t.two_players ? -q : q;
          value_score = mz_normalize((double)c.R + t.discount * v, mn, mx);
        } else {
          value_score = t.init_value_score;
        }
        score = prior_score + value_score;
      }
      best = lane;
    }
    // tuple max over (score, action): ties go to the largest action (mcts.py:106-112)
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
      const double os = __shfl_xor(score, off, G);
      const int ob = __shfl_xor(best, off, G), oe = __shfl_xor(ce, off, G), on = __shfl_xor(cn, off, G);
      const bool take = ob >= 0 && (best < 0 || os > score || (os == score && ob > best));
      if (take) { score = os; best = ob; ce = oe; cn = on; }
    }
    a_sel = best;
    parent_e = e;
    node = 1 + e * A + a_sel;
    if (lane == 0) path[len] = node;
    ++len;
    if (t.two_players) tp = -tp;
    e = ce;
    Np = cn;
  }
  slot_out = parent_e;
  act_out = a_sel;
  if (lane == 0) {
    t.plen[b] = len;
    t.leaf_tp[b] = (int8_t)tp;
    t.leaf[b] = node;
    t.slot[b] = slot_out;
    t.act[b] = a_sel;
    t.depth[b] = len - 1;
  }
}

template <int G>
__device__ __forceinline__ void mz_tree_select(const TreeView &t, int b, int lane) {
  int s_, a_;
  mz_tree_select<G>(t, b, lane, s_, a_);
}

// Node.expand for the selected leaf (mcts.py:47-55, all actions: mcts.py:97) followed by
// MCTS.backpropagate (mcts.py:126-143).  value/reward/logits: this tree's network outputs.
// The path's nodes are fetched G at a time, one record per lane (one round trip per G nodes instead of one per node);
// lane 0 then runs the reference's sequential recurrence on the shuffled fields and stores the results.
template <int G>
__device__ __forceinline__ void mz_tree_expand_backup(const TreeView &t, int b, int lane, float value,
                                                      float reward, const float *logits) {
  const int A = t.A;
  const size_t o = mz_slab(t, b);
  const int32_t *path = t.path + (size_t)b * t.PL;
  const int len = t.plen[b];
  const int tp = t.leaf_tp[b];
  const int e = t.nexp[b];
  const int leafnode = path[len - 1];
  // priors: exp(logit) in double, no max-shift, Python sum() order 0 + p0 + p1 + ...
  const double p = (lane < A) ? exp((double)logits[lane]) : 0.0;
  double sum = 0.0;
  for (int a = 0; a < A; ++a) sum = sum + __shfl(p, a, G);
  if (lane < A) mz_node_fresh(t, o + 1 + e * A + lane, p / sum, 1);
  const double g = t.discount;
  double mn = 0.0, mx = 0.0, v = (double)value;
  if (lane == 0) { mn = t.mn[b]; mx = t.mx[b]; t.nexp[b] = e + 1; }
  for (int base = 0; base < len; base += G) {
    const int j = base + lane;
    const int mynode = path[len - 1 - (j < len ? j : len - 1)];
    const MzNode c = mz_node_load(t, o + mynode);
    const int cnt = (len - base) < G ? (len - base) : G;
    for (int jj = 0; jj < cnt; ++jj) {
      const int idx = base + jj;
      const int node = __shfl(mynode, jj, G);
      const double Wn = __shfl(c.W, jj, G);
      const int Nn = __shfl(c.N, jj, G);
      const float Rn = __shfl(c.R, jj, G);
      const int TPn = __shfl((int)c.TP, jj, G);
      if (lane == 0) {
        const int ntp = (idx == 0) ? tp : TPn;
        const double r_node = (idx == 0) ? (double)reward : (double)Rn;
        const double w = Wn + ((ntp == tp) ? v : -v);
        const int n = Nn + 1;
        t.W[o + node] = w;
        t.N[o + node] = n;
        const double r = (t.two_players && ntp == tp) ? -r_node : r_node;
        if (idx < len - 1) {
          const double q = w / (double)n;
          const double new_q = t.two_players ? r_node - g * q : r_node + g * q;
          mn = new_q < mn ? new_q : mn;
          mx = new_q > mx ? new_q : mx;
        }
        v = r + g * v;
      }
    }
  }
  if (lane == 0) {
    t.E[o + leafnode] = e;
    t.TP[o + leafnode] = (int8_t)tp;
    t.R[o + leafnode] = reward;
    t.mn[b] = mn;
    t.mx[b] = mx;
  }
}

// Node(0) + root.expand(legal) + add_exploration_noise + MinMaxStats.reset
// (actors.py:132,141-143; mcts.py:47-61,79).
// SELECT: also perform the first descent of MCTS.run from the registers.  At a fresh root (N = 0) select_child
// ranks the children by (prior, action) (mcts.py:105-108) and the chosen child is a leaf, so the descent needs
// nothing from memory: path = [0, 1 + a*], parent slot 0.
// stage (LDS, optional): the search kernel that continues in the same launch takes the root from there instead of
// reading the pool back -- stage[a] = prior of child a (after the noise), stage[32] = the first descent's action.
template <int G, bool SELECT = false>
__device__ __forceinline__ void mz_tree_root(const TreeView &t, int b, int lane, int to_play, uint32_t legal,
                                             const float *logits, const double *noise, double frac,
                                             double *stage = nullptr) {
  const int A = t.A;
  const size_t o = mz_slab(t, b);
  const bool ok = lane < A && ((legal >> lane) & 1u);
  const double p = ok ? exp((double)logits[lane]) : 0.0;
  double sum = 0.0;
  for (int a = 0; a < A; ++a) sum = sum + __shfl(p, a, G);      // + 0.0 for illegal slots is exact
  if (lane < A) {
    const int ch = 1 + lane;
    double prior = ok ? p / sum : 0.0;
    if (ok && noise) prior = prior * (1 - frac) + noise[lane] * frac;
    mz_node_fresh(t, o + ch, prior, ok ? 1 : 0);
    if (stage) stage[lane] = prior;
  }
  if (lane == 0) {
    t.N[o] = 0; t.W[o] = 0.0; t.R[o] = 0.f; t.E[o] = 0; t.TP[o] = (int8_t)to_play; t.P[o] = 0.0;
    t.nexp[b] = 1;
    t.legal[b] = legal;
    t.mn[b] = t.has_min ? t.min_bound : __builtin_inf();
    t.mx[b] = t.has_max ? t.max_bound : -__builtin_inf();
    t.plen[b] = 0;
  }
  if constexpr (SELECT) {
    double score = 0.0;
    int best = -1;
    if (ok) {
      score = p / sum;
      if (noise) score = score * (1 - frac) + noise[lane] * frac;
      best = lane;
    }
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
      const double os = __shfl_xor(score, off, G);
      const int ob = __shfl_xor(best, off, G);
      const bool take = ob >= 0 && (best < 0 || os > score || (os == score && ob > best));
      if (take) { score = os; best = ob; }
    }
    if (lane == 0) {
      int32_t *path = t.path + (size_t)b * t.PL;
      path[0] = 0; path[1] = 1 + best;
      t.plen[b] = 2;
      t.leaf_tp[b] = (int8_t)(t.two_players ? -to_play : to_play);
      t.leaf[b] = 1 + best;
      t.slot[b] = 0;
      t.act[b] = best;
      t.depth[b] = 1;
      if (stage) stage[32] = (double)best;
    }
  }
}

// ------------------------------------------------------------------ fused-kernel tree step
// Cross-lane moves inside one row of 16 lanes by DPP (one VALU op per 32 bits, no LDS crossbar trip):
// quad_perm xor 1 / xor 2, row_half_mirror (i <-> 7-i), row_mirror (i <-> 15-i).  For a reduction under a
// total order any pairing that eventually joins all lanes gives every lane the same result.
template <int CTRL>
__device__ __forceinline__ int mz_dpp_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double mz_dpp_d(double x) {
  const int lo = mz_dpp_i<CTRL>(__double2loint(x)), hi = mz_dpp_i<CTRL>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
// partner exchange for step `off` (1, 2, 4, 8 inside a row; 16 across the two rows of a 32-lane group)
template <int OFF>
__device__ __forceinline__ int mz_xchg_i(int x) {
  if constexpr (OFF == 1) return mz_dpp_i<0xB1>(x);
  else if constexpr (OFF == 2) return mz_dpp_i<0x4E>(x);
  else if constexpr (OFF == 4) return mz_dpp_i<0x141>(x);
  else if constexpr (OFF == 8) return mz_dpp_i<0x140>(x);
  else return __shfl_xor(x, 16);
}
template <int OFF>
__device__ __forceinline__ double mz_xchg_d(double x) {
  if constexpr (OFF <= 8) return __hiloint2double(mz_xchg_i<OFF>(__double2hiint(x)), mz_xchg_i<OFF>(__double2loint(x)));
  else return __shfl_xor(x, 16);
}


template <bool V> struct MzBool { static constexpr bool value = V; };

// State a tree's lanes keep in registers between the simulations of one fused launch.
struct TreeRegs {
  int len;          // len(search_path) of the pending descent
  int tp;           // to_play at its leaf
  int root_tp;      // root.to_play
  uint32_t legal;   // root child mask
  int root_n;       // root.visit_count (every backup adds one: kept here instead of being fished out of the path)
  double mn, mx;    // MinMaxStats
};

// One tree's node arrays, three placements (LT):
//   0  its slab of the global pool;
//   1  a copy the fused kernel keeps in LDS for the whole launch (visit counts and expansion indices narrowed to
//      16 bit), plus a cache X = reward + discount * (+-Q) of every visited node, the term MinMaxStats normalises: the
//      backup computes it anyway for the MinMaxStats update (same operations, same order as the descent's own
//      expression), so the descent reads one double instead of R and W/N and divides nothing;
//   2  trees too large for (1): still everything in LDS, but compact -- per NODE only what exists for every node (visit
//      count, expansion index, prior, to_play), per EXPANSION SLOT (num_simulations + 2 of them, not 1 + slots * A) what
//      exists only for visited nodes, and a node is visited exactly when it is expanded: value_sum, reward and the X
//      cache.  W[i], R[i], X[i] are therefore indexed by the expansion index E[node] here (root = slot 0); the descent
//      fetches X[E[child]] beside its pb_c lookup (both hang on the first round trip), the backup E[node] first.
//      Pong-ram shapes (307 nodes per tree): 97 KB instead of the 162 KB of placement (1).
template <int LT> struct TreeMem;
template <> struct TreeMem<0> {      // the tree's slab of the global pool: fields of its 32-byte node records (MzField)
  MzField<int32_t, 16> N; MzField<double, 0> W; MzField<double, 8> P; MzField<float, 24> R; MzField<int32_t, 20> E; MzField<int8_t, 28> TP;
};
template <> struct TreeMem<1> { int16_t *N; double *W; double *P; float *R; int16_t *E; int8_t *TP; double *X; };
template <> struct TreeMem<2> { int16_t *N; double *W; double *P; float *R; int16_t *E; int8_t *TP; double *X; };

// expand + backpropagate for the pending leaf, then (do_select) the next descent -- the same arithmetic as
// mz_tree_expand_backup / mz_tree_select, reorganised so that the chain of DEPENDENT memory round trips is
// short: TL (16 or 32) lanes per tree; the search path lives in LDS (s_path); the backup loads every path
// node at once (lane j = j-th node from the leaf) and runs the value recurrence through shuffles; in the
// descent each child lane also loads its own expansion index and visit count, so the winner's are
// forwarded by shuffle and a level costs ONE round trip.
struct MzNoStamp { __device__ __forceinline__ void operator()(int) const {} };

// Hook of the descent loop: enter(e) at the top of every level (e = expansion index of the node being descended
// through, i.e. the leaf's parent if this level turns out to be the last), leave() at its bottom.
struct MzNoLevelHook {
  __device__ __forceinline__ void enter(int) const {}
  __device__ __forceinline__ void leave() const {}
};
// Speculative gather of the parent's hidden state (4-wave fused kernel): the node being descended through is the
// leaf's parent if this turns out to be the last level, so its hidden state (this lane's 16 bytes of it) is requested
// now and the L2 round trip runs under the level's arithmetic; a deeper level simply overwrites the request (loads
// return in order).  Issued from asm: the compiler would put a vmcnt(0) in front of every re-use of the destination
// register.  The consumer waits with s_waitcnt vmcnt(0) (mz_fused.hip.h, gather).
// ("+v": the destination stays allocated to hv between requests -- an earlier request may still be in flight, so
// nothing else, not even this request's address, may be computed in those registers)
struct MzHiddenPrefetch {
  const float *hpool; unsigned hoff; f32x4 &hv;
  __device__ __forceinline__ void enter(int e) const {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(hv) : "v"(hoff + (unsigned)e * (unsigned)(MZ_HS * 4)), "s"(hpool));
  }
  __device__ __forceinline__ void leave() const { asm volatile("" : "+v"(hv)); }   // in flight: keep the destination reserved
};

// ---- Node.expand (mcts.py:47-55) for the pending leaf: priors of the new children, leaf bookkeeping.
// p = math.exp(logit of action `lane`) (0 for lane >= A), computed by the caller (so that it can be scheduled
// beside other work)
// SP: single-player game known at compile time (every to_play is +1: no sign flips, no to_play loads or stores);
// SP = false reads t.two_players at run time (both kinds of game).
template <int TL, int G, int LT, bool SP = false>
__device__ __forceinline__ void mz_tree_expand_f(const TreeView &t, const TreeMem<LT> &tm, int lane, int e_new,
                                                 float reward, double p, const int *s_path,
                                                 double *s_stage, const TreeRegs &tr) {
  const int A = t.A;
  const int leafnode = s_path[tr.len - 1];
  s_stage[lane] = p;                            // every lane then adds them up in Python's sum() order (G >= A terms:
  double sum = 0.0;                             // the lanes beyond A contribute exact zeros at the end of the sum)
#pragma unroll
  for (int a = 0; a < G; ++a) sum = sum + s_stage[a];
  if (lane < A) {
    const int ch = 1 + __mul24(e_new, A) + lane;
    // (LT = 1: the kernel's prologue has written the fresh-Node fields of every node this launch can create)
    if constexpr (LT == 0) { tm.N[ch] = 0; tm.W[ch] = 0.0; tm.R[ch] = 0.f; tm.E[ch] = -1; tm.TP[ch] = 1; }
    if constexpr (LT == 2) { tm.N[ch] = 0; tm.E[ch] = -1; tm.TP[ch] = 1; }
    tm.P[ch] = p / sum;
  }
  if (lane == 0) {
    tm.E[leafnode] = e_new;
    if constexpr (LT == 2) { tm.W[e_new] = 0.0; tm.R[e_new] = reward; }      // (per expansion slot: a fresh Node's value_sum)
    else tm.R[leafnode] = reward;
    if constexpr (!SP) tm.TP[leafnode] = (int8_t)tr.tp;        // (single player: it was created with to_play 1 and stays so)
  }
}

template <int TL, int G, int LT, bool SP, class LEVELF, class STAMPF>
__device__ __forceinline__ void mz_tree_backup_select_f(const TreeView &t, const TreeMem<LT> &tm, int lane,
                                                        float value, float reward, int *s_path, double *s_stage,
                                                        const double *pbctab, const double *rcptab, TreeRegs &tr,
                                                        bool do_select, int &slot_out, int &act_out,
                                                        const LEVELF &levelf, STAMPF stampf);

template <int TL, int G, int LT, bool SP = false, class STAMPF = MzNoStamp>
__device__ __forceinline__ void mz_tree_step_fused(const TreeView &t, const TreeMem<LT> &tm, int lane, int e_new,
                                                   float value, float reward, const float *logits, int *s_path,
                                                   double *s_stage, const double *pbctab, const double *rcptab,
                                                   TreeRegs &tr, bool do_select, int &slot_out, int &act_out,
                                                   const float *hpool, unsigned hoff, f32x4 &hv,
                                                   STAMPF stampf = STAMPF()) {
  mz_tree_expand_f<TL, G, LT, SP>(t, tm, lane, e_new, reward, (lane < t.A) ? exp((double)logits[lane]) : 0.0, s_path, s_stage, tr);
  stampf(0);
  mz_tree_backup_select_f<TL, G, LT, SP>(t, tm, lane, value, reward, s_path, s_stage, pbctab, rcptab, tr, do_select,
                                         slot_out, act_out, MzHiddenPrefetch{hpool, hoff, hv}, stampf);
}

// MCTS.backpropagate for the pending leaf, then (do_select) the next descent.
// rcptab[n] = 1.0 / n (IEEE double, n = 1 .. num_simulations + 1): node.value() = value_sum / visit_count
// (mcts.py:42-45) is taken as q0 = w * y, r = fma(-n, q0, w), q = fma(r, y, q0) with y = RN(1 / n) -- the correction
// step of a correctly rounded division (Markstein): bit-identical to w / n for every finite w
// (scripts/experiments/divcheck.c: 0 differences in 5e8 quotients), three dependent operations instead of the eleven of the
// division's expansion.
template <int TL, int G, int LT, bool SP, class LEVELF, class STAMPF>
__device__ __forceinline__ void mz_tree_backup_select_f(const TreeView &t, const TreeMem<LT> &tm, int lane,
                                                        float value, float reward, int *s_path, double *s_stage,
                                                        const double *pbctab, const double *rcptab, TreeRegs &tr,
                                                        bool do_select, int &slot_out, int &act_out,
                                                        const LEVELF &levelf, STAMPF stampf) {
  const int A = t.A;
  const int len = tr.len, tp = tr.tp;
  const double g = t.discount;
  const bool two = SP ? false : (t.two_players != 0);

  // ---- MCTS.backpropagate (mcts.py:126-143), TL path nodes per round
  double v_cur = (double)value;
  double mn_c = __builtin_inf(), mx_c = -__builtin_inf();
  // MinMaxStats.update over the path: when no path in this wave has more than four nodes below the root, the nodes'
  // values go through LDS (one write, two 16-byte reads, three v_min_f64 + three v_max_f64, quiet NaN = "no node",
  // which minNum / maxNum ignore) instead of a four-step DPP reduction of two doubles over the tree's 16 lanes
  const bool shortp = __builtin_amdgcn_ballot_w64(len > 5) == 0;
  for (int base = 0; base < len; base += TL) {
    // (written for few branches: every lane loads -- lanes beyond the path's end load the root's fields, which exist --
    // and computes; only the stores are predicated.  Divergent ifs around the loads and the division cost more in
    // exec-mask bookkeeping than the work they skipped.)
    const int j = base + lane;
    const bool act = j < len;
    const int node = s_path[act ? len - 1 - j : 0];
    int vi = node;                                  // index of the node's value_sum / reward / X
    if constexpr (LT == 2) vi = tm.E[node];         // (its expansion slot: every path node is expanded, the leaf just now)
    const double Wn = tm.W[vi];
    const int Nn = tm.N[node];
    const double r_node = (j == 0) ? (double)reward : (double)tm.R[vi];
    int ntp = tp;
    if constexpr (!SP) ntp = (j == 0) ? tp : (int)tm.TP[node];
    const double r_signed = (two && ntp == tp) ? -r_node : r_node;
    const int cnt = (len - base) < TL ? (len - base) : TL;
    double my_v = 0.0;
    s_stage[32 + lane] = r_signed;                 // the recurrence value = reward + discount*value runs in
    {                                              // every lane; lane j keeps the value its node receives.
      // The first four staged rewards come in two 16-byte reads and the recurrence runs on registers (most paths are
      // no longer than that); a select keeps the steps beyond the path's end from taking effect.
      typedef double f64x2 __attribute__((ext_vector_type(2)));
      const f64x2 ra = *(const f64x2 *)(s_stage + 32), rb = *(const f64x2 *)(s_stage + 34);
      const double rj[4] = {ra[0], ra[1], rb[0], rb[1]};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const double vn = rj[jj] + g * v_cur;
        my_v = (lane == jj) ? v_cur : my_v;
        v_cur = (jj < cnt) ? vn : v_cur;
      }
    }
    for (int jj = 4; jj < cnt; ++jj) {
      if (lane == jj) my_v = v_cur;
      v_cur = s_stage[32 + jj] + g * v_cur;
    }
    const double w = Wn + ((ntp == tp) ? my_v : -my_v);
    const int n = Nn + 1;
    const double yn = rcptab[n], dn = (double)n;
    const double q0 = w * yn;
    const double q = __builtin_fma(__builtin_fma(-dn, q0, w), yn, q0);     // = w / n, see above
    const double new_q = two ? r_node - g * q : r_node + g * q;     // = reward + discount * (two ? -Q : Q)
    if (act) {
      tm.W[vi] = w;
      tm.N[node] = n;
      if constexpr (LT != 0) tm.X[vi] = new_q;       // what the descent normalises (the root's is never read)
    }
    const bool inner = act & (j < len - 1);          // MinMaxStats.update for every node but the root (mcts.py:136-141)
    mn_c = inner ? new_q : mn_c;
    mx_c = inner ? new_q : mx_c;
    if (shortp) s_stage[64 + lane] = inner ? new_q : __builtin_nan("");
  }
  tr.root_n += 1;
  if (shortp) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const f64x2 qa = *(const f64x2 *)(s_stage + 64), qb = *(const f64x2 *)(s_stage + 66);
    double n01, n23, x01, x23;
    asm("v_min_f64 %0, %1, %2" : "=v"(n01) : "v"(qa[0]), "v"(qa[1]));
    asm("v_min_f64 %0, %1, %2" : "=v"(n23) : "v"(qb[0]), "v"(qb[1]));
    asm("v_max_f64 %0, %1, %2" : "=v"(x01) : "v"(qa[0]), "v"(qa[1]));
    asm("v_max_f64 %0, %1, %2" : "=v"(x23) : "v"(qb[0]), "v"(qb[1]));
    asm("v_min_f64 %0, %1, %2" : "=v"(mn_c) : "v"(n01), "v"(n23));
    asm("v_max_f64 %0, %1, %2" : "=v"(mx_c) : "v"(x01), "v"(x23));
  } else {
    // A path longer than TL nodes took more than one round above, and a later round's nodes REPLACED the earlier rounds' in
    // the lanes' (mn_c, mx_c) -- until r04 those nodes were missing from MinMaxStats.update (found by the injected-output test
    // of a 26-node chain, tests/test_gpu_fused_exact.py; 31-node chains occur in real runs).  Rare, so it is repaired here,
    // outside the loop every simulation runs (anything added INSIDE it cost 0.3-0.4 % of a move in A/B runs, whatever its
    // form): every inner node's value is read back (the X cache holds exactly new_q; pool trees recompute it from what was
    // stored) and accumulated -- min / max are idempotent, the last round's nodes may be seen twice.
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(len > TL) != 0, 0)) {
      if constexpr (LT == 0) __threadfence_block();      // (the loop's stores to the pool, read back by other lanes of the wave)
      for (int base = 0; base < len; base += TL) {
        const int j = base + lane;
        const int node = s_path[j < len ? len - 1 - j : 0];
        double x;
        if constexpr (LT == 0) {
          const double wq = tm.W[node], dq = (double)tm.N[node], yq = rcptab[tm.N[node]];
          const double q0 = wq * yq;
          const double q = __builtin_fma(__builtin_fma(-dq, q0, wq), yq, q0);
          const double rq = (j == 0) ? (double)reward : (double)tm.R[node];
          x = two ? rq - g * q : rq + g * q;
        } else {
          int vi = node;
          if constexpr (LT == 2) vi = tm.E[node];
          x = tm.X[vi];
        }
        if (j < len - 1) {
          mn_c = x < mn_c ? x : mn_c;
          mx_c = x > mx_c ? x : mx_c;
        }
      }
    }
#define MZ_MM_STEP(OFF)                                                         \
  {                                                                             \
    const double a_ = mz_xchg_d<OFF>(mn_c), c_ = mz_xchg_d<OFF>(mx_c);         \
    asm("v_min_f64 %0, %1, %2" : "=v"(mn_c) : "v"(mn_c), "v"(a_));             \
    asm("v_max_f64 %0, %1, %2" : "=v"(mx_c) : "v"(mx_c), "v"(c_));             \
  }
  MZ_MM_STEP(1) MZ_MM_STEP(2) MZ_MM_STEP(4) MZ_MM_STEP(8)
  if constexpr (TL == 32) MZ_MM_STEP(16)
#undef MZ_MM_STEP
  }
  tr.mn = mn_c < tr.mn ? mn_c : tr.mn;         // (a NaN -- no node below the root -- compares false: nothing changes)
  tr.mx = mx_c > tr.mx ? mx_c : tr.mx;
  stampf(1);
  if (!do_select) return;

  // ---- the next descent (mcts.py:83-92, 104-124).  Written branch-free: every lane evaluates the whole
  // score expression on safe operands and the conditions of the reference (fresh root ranks by prior,
  // unvisited child gets init_value_score, MinMaxStats.normalize's three cases, illegal root slots) pick
  // among the results -- a divergent if/else ladder cost more than the arithmetic it skipped.
  if constexpr (LT == 0) __threadfence_block();
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the descent reads LDS only
  const double mn = tr.mn, mx = tr.mx;
  const double span = mx - mn;
  const bool span_pos = mx > mn, span_zero = mx == mn;
  // MinMaxStats.normalize divides by the same span at every level of this descent: the reciprocal (the rcp + two
  // Newton steps of the compiler's own f64 division) is taken once, a level then costs mul + fma + fma -- the tail
  // of that same expansion, bit-identical to (x - mn) / span as long as v_div_scale would not rescale the operands
  // (scripts/experiments/fastdiv_check.hip: 0 differences in 8e7 quotients).  Outside a generous exponent window the level
  // falls back to the division itself.
  double yspan = 0.0;
  bool fast_ok = true;
  if (span_pos) {
    fast_ok = (span > 0x1p-200) & (span < 0x1p200);
    double y0 = __builtin_amdgcn_rcp(span);
    double e0 = __builtin_fma(-span, y0, 1.0);
    y0 = __builtin_fma(y0, e0, y0);
    e0 = __builtin_fma(-span, y0, 1.0);
    yspan = __builtin_fma(y0, e0, y0);
  }
  // The common case of a descent -- every tree of the wave has max > min in its MinMaxStats (true from the second or
  // third backup of a move on), the span inside the window of the reciprocal, init_value_score 0 (the reference's
  // default): a wave-uniform branch inside the level skips the selects among normalize's three cases, the division's
  // fallback branch and the moves of init_value_score into vector registers.  (One loop, not two specialised ones:
  // the hidden-state prefetch of the level hook is an asm-issued load into a fixed register tuple, and a second loop
  // instance made the compiler copy that tuple between two homes while a load was in flight.)
  const bool common = __builtin_amdgcn_ballot_w64(!(span_pos & fast_ok & (t.init_value_score == 0.0))) == 0;
  int node = 0, e = 0, Np = tr.root_n, tpc = tr.root_tp, len2 = 1, a_sel = -1, parent_e = 0;
  if (lane == 0) s_path[0] = 0;
  // the TL lanes of a tree evaluate the children in TL/G redundant copies (child = lane % G), so the
  // arg-max needs only log2(G) exchange steps and every lane ends up with the result
  const int cl = lane % G;
  while (e >= 0) {
    levelf.enter(e);
    const bool valid = (cl < A) & ((node != 0) | (((tr.legal >> cl) & 1u) != 0));
    const int ch = valid ? 1 + __mul24(e, A) + cl : 0;
    const int Nc = tm.N[ch];
    const int Ec = tm.E[ch];
    const double p = tm.P[ch];
    // pb_c table in LDS: rows 64 entries apart where it fits (a shift, no multiply), sims + 2 apart beside large trees
    const double prior_score = pbctab[(LT == 2 ? __mul24(Np, t.sims + 2) : (Np << 6)) + Nc] * p;
    double x;        // reward + discount * (+-Q) of the child; unused (may be stale or garbage) while Nc == 0
    if constexpr (LT == 2) {
      x = tm.X[Ec > 0 ? Ec : 0];                   // cached by the backup, per expansion slot
    } else if constexpr (LT == 1) {
      x = tm.X[ch];                                // cached by the backup
    } else {
      const double q = tm.W[ch] / (double)(Nc > 0 ? Nc : 1);
      x = (double)tm.R[ch] + g * (two ? -q : q);
    }
    const double xm = x - mn;
    const double q0 = xm * yspan;
    const double r0 = __builtin_fma(-span, q0, xm);
    double nrm = __builtin_fma(r0, yspan, q0);
    double ucb;
    if (common) {
      ucb = prior_score + ((Nc > 0) ? nrm : 0.0);
    } else {
      if (!fast_ok) {                            // (never in practice)
        double sp_ = span;
        asm volatile("" : "+v"(sp_));            // not speculatable: left alone, the compiler turns this branch into a
        nrm = xm / sp_;                          // select and evaluates the division at every level after all
      }
      const double visited = span_pos ? nrm : (span_zero ? 1.0 : x);
      ucb = prior_score + ((Nc > 0) ? visited : t.init_value_score);
    }
    // tuple max over (score, action); absent children carry score -inf and action -1 (no finite score loses to them,
    // and among themselves nothing is taken), the winner's expansion index and visit count ride along in one word
    // (a descent of this function always follows a backup: the root and every expanded node on the way down have been
    // visited, so mcts.py:105's fresh-root case, rank by prior, cannot occur here)
    double score = valid ? ucb : -__builtin_inf();
    // one word per child: action + 1 (0 = absent) above expansion index + 1 above visit count -- between two lanes of a
    // group the actions differ, so comparing the words compares the actions, and the winner's E / N ride along
    int key = Nc | ((Ec + 1) << 8) | (valid ? (cl + 1) << 16 : 0);
#ifdef MZ_ARGMAX_TUPLE       // (development switch: the pairwise tuple comparison this replaced, 11 instructions per step)
#define MZ_AM_STEP(OFF)                                                                          \
  {                                                                                              \
    const double os = mz_xchg_d<OFF>(score);                                                     \
    const int ok = mz_xchg_i<OFF>(key);                                                          \
    const bool take = (os > score) | ((os == score) & (ok > key));                               \
    score = take ? os : score;                                                                   \
    key = take ? ok : key;                                                                       \
  }
    if constexpr (G > 1) MZ_AM_STEP(1)
    if constexpr (G > 2) MZ_AM_STEP(2)
    if constexpr (G > 4) MZ_AM_STEP(4)
    if constexpr (G > 8) MZ_AM_STEP(8)
    if constexpr (G > 16) MZ_AM_STEP(16)
#undef MZ_AM_STEP
#else
    // max of the tuples (score, key) in two plain reductions instead of one reduction of pairs: the group's largest score
    // first (v_max_f64 on exchanged halves: 3 instructions per step), then the largest key among the lanes that hold it
    // (the others contribute 0; keys of a group differ in their action bits, so this is the winner's own word).  Equal
    // scores compare equal here exactly where the tuple comparison's == did (+0 and -0 included); a NaN score -- a
    // network that produced one -- is ignored by v_max_f64 where the tuple comparison kept whichever came first.
#ifdef MZ_ARGMAX_F32
    // (r05 experiment) a float32 screen in front of the float64 reductions: rounding to float32 is monotone, so a lane whose
    // ROUNDED score is strictly the group's largest holds the largest float64 score too -- one DPP-fused v_max_f32 per step
    // instead of two moves and a v_max_f64; where two lanes of any group of the wave share the largest rounded score (exact
    // ties included) the wave takes the float64 reductions below.  Exact by construction.
    bool screened = false;
    {
      const float s32 = (float)score;
      float m32 = s32;
#define MZ_AM_MAX32(OFF) m32 = fmaxf(m32, __int_as_float(mz_xchg_i<OFF>(__float_as_int(m32))));
      if constexpr (G > 1) MZ_AM_MAX32(1)
      if constexpr (G > 2) MZ_AM_MAX32(2)
      if constexpr (G > 4) MZ_AM_MAX32(4)
      if constexpr (G > 8) MZ_AM_MAX32(8)
      if constexpr (G > 16) MZ_AM_MAX32(16)
#undef MZ_AM_MAX32
      const bool top = s32 == m32;
      int k2 = top ? key : 0;
#define MZ_AM_KEY32(OFF) { const int ok = mz_xchg_i<OFF>(k2); k2 = ok > k2 ? ok : k2; }
      if constexpr (G > 1) MZ_AM_KEY32(1)
      if constexpr (G > 2) MZ_AM_KEY32(2)
      if constexpr (G > 4) MZ_AM_KEY32(4)
      if constexpr (G > 8) MZ_AM_KEY32(8)
      if constexpr (G > 16) MZ_AM_KEY32(16)
#undef MZ_AM_KEY32
      if (__builtin_amdgcn_ballot_w64(top & (key != k2)) == 0) { key = k2; screened = true; }
    }
    if (!screened)
#endif
    {
      double mxs = score;
#define MZ_AM_MAX(OFF)                                                              \
  {                                                                                 \
    const double os = mz_xchg_d<OFF>(mxs);                                          \
    asm("v_max_f64 %0, %1, %2" : "=v"(mxs) : "v"(mxs), "v"(os));                    \
  }
      if constexpr (G > 1) MZ_AM_MAX(1)
      if constexpr (G > 2) MZ_AM_MAX(2)
      if constexpr (G > 4) MZ_AM_MAX(4)
      if constexpr (G > 8) MZ_AM_MAX(8)
      if constexpr (G > 16) MZ_AM_MAX(16)
#undef MZ_AM_MAX
      key = (score == mxs) ? key : 0;
#define MZ_AM_KEY(OFF)                            \
  {                                               \
    const int ok = mz_xchg_i<OFF>(key);           \
    key = ok > key ? ok : key;                    \
  }
      if constexpr (G > 1) MZ_AM_KEY(1)
      if constexpr (G > 2) MZ_AM_KEY(2)
      if constexpr (G > 4) MZ_AM_KEY(4)
      if constexpr (G > 8) MZ_AM_KEY(8)
      if constexpr (G > 16) MZ_AM_KEY(16)
#undef MZ_AM_KEY
    }
#endif
    a_sel = (key >> 16) - 1;
    parent_e = e;
    node = 1 + __mul24(e, A) + a_sel;
    if (lane == 0) s_path[len2] = node;
    ++len2;
    if (two) tpc = -tpc;
    e = ((key >> 8) & 0xff) - 1;
    Np = key & 0xff;
    levelf.leave();
  }
  tr.len = len2;
  tr.tp = tpc;
  slot_out = parent_e;
  act_out = a_sel;
  stampf(2);
}

// ------------------------------------------------------------------ kernels (one launch per phase)
template <int G>
__global__ void k_tree_select(TreeView t) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gt / G, lane = gt % G;
  if (b >= t.B) return;
  mz_tree_select<G>(t, b, lane);
}

template <int G>
__global__ void k_tree_expand_backup(TreeView t, const float *value, const float *reward, const float *logits) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gt / G, lane = gt % G;
  if (b >= t.B) return;
  mz_tree_expand_backup<G>(t, b, lane, value[b], reward[b], logits + (size_t)b * t.A);
}

// expand+backup of simulation s fused with the descent of simulation s+1 (one launch per simulation)
template <int G>
__global__ void k_tree_step(TreeView t, int do_select) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gt / G, lane = gt % G;
  if (b >= t.B) return;
  mz_tree_expand_backup<G>(t, b, lane, t.value[b], t.reward[b], t.logits + (size_t)b * t.A);
  if (do_select) {
    __threadfence_block();          // lane 0's node updates -> visible to the group's other lanes
    mz_tree_select<G>(t, b, lane);
  }
}

// the same step for an external network (mz_expand_backup_select): expand + backup on the caller's outputs, the next descent,
// and what mz_select hands back -- one launch per simulation instead of three, and the descent finds the lines the backup has
// just written in cache
template <int G>
__global__ void k_tree_step_ext(TreeView t, const float *value, const float *reward, const float *logits, int32_t *leaf,
                                int32_t *slot, int32_t *act, int32_t *depth) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gt / G, lane = gt % G;
  if (b >= t.B) return;
  mz_tree_expand_backup<G>(t, b, lane, value[b], reward[b], logits + (size_t)b * t.A);
  __threadfence_block();          // lane 0's node updates -> visible to the group's other lanes
  mz_tree_select<G>(t, b, lane);
  if (lane == 0) {                // (lane 0 wrote them)
    if (leaf) leaf[b] = t.leaf[b];
    if (slot) slot[b] = t.slot[b];
    if (act) act[b] = t.act[b];
    if (depth) depth[b] = t.depth[b];
  }
}

template <int G>
__global__ void k_tree_root(TreeView t, const int8_t *to_play, const uint8_t *legal, const double *noise,
                            double frac, int then_select) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gt / G, lane = gt % G;
  if (b >= t.B) return;
  uint32_t mask = 0;
  if (legal) {
    for (int a = 0; a < t.A; ++a) mask |= (legal[(size_t)b * t.A + a] ? 1u : 0u) << a;
  } else {
    mask = (t.A >= 32) ? 0xFFFFFFFFu : ((1u << t.A) - 1u);
  }
  mz_tree_root<G>(t, b, lane, to_play ? (int)to_play[b] : 1, mask, t.root_logits + (size_t)b * t.A,
                  noise ? noise + (size_t)b * t.A : nullptr, frac);
  if (then_select) {
    __threadfence_block();
    mz_tree_select<G>(t, b, lane);
  }
}

// root children priors given directly (a caller that already ran Node.expand / add_exploration_noise on
// its own objects, e.g. the batch-1 MCTS.run front-end): Node(0) + children + MinMaxStats.reset
template <int G>
__global__ void k_tree_root_priors(TreeView t, const int8_t *to_play, const uint8_t *legal, const double *priors) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gt / G, lane = gt % G;
  if (b >= t.B) return;
  const int A = t.A;
  const size_t o = mz_slab(t, b);
  uint32_t mask = 0;
  for (int a = 0; a < A; ++a) mask |= ((legal ? legal[(size_t)b * A + a] : 1) ? 1u : 0u) << a;
  if (lane < A) {
    const bool ok = (mask >> lane) & 1u;
    const int ch = 1 + lane;
    mz_node_fresh(t, o + ch, ok ? priors[(size_t)b * A + lane] : 0.0, ok ? 1 : 0);
  }
  if (lane == 0) {
    t.N[o] = 0; t.W[o] = 0.0; t.R[o] = 0.f; t.E[o] = 0; t.TP[o] = to_play ? to_play[b] : (int8_t)1; t.P[o] = 0.0;
    t.nexp[b] = 1;
    t.legal[b] = mask;
    t.mn[b] = t.has_min ? t.min_bound : __builtin_inf();
    t.mx[b] = t.has_max ? t.max_bound : -__builtin_inf();
    t.plen[b] = 0;
  }
  __threadfence_block();
  mz_tree_select<G>(t, b, lane);
}

// numpy's pairwise float64 sum for n < 128 (ndarray.sum of a contiguous vector), config.py:76
__device__ inline double mz_np_sum(const double *a, int n) {
  if (n < 8) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i];
    return s;
  }
  double r[8];
  int i;
  for (i = 0; i < 8; ++i) r[i] = a[i];
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

// Config.select_action (config.py:70-81) on the n visit counts d[0..n) (as doubles, child insertion order) with
// the uniform u that np.random.choice / the tie-break consumes; returns the index into the legal-action list.
// d is overwritten.  d may live in LDS (the fused kernel stages it there) or in private memory.
__device__ __forceinline__ int mz_sample_index(double *d, int n, double T, double u) {
  int idx = 0;
  if (T != 0.0) {
    const double ex = 1 / T;
    if (ex != 1.0) for (int i = 0; i < n; ++i) d[i] = pow(d[i], ex);
    const double s = mz_np_sum(d, n);
    for (int i = 0; i < n; ++i) d[i] = d[i] / s;
    double c = 0.0;
    for (int i = 0; i < n; ++i) { c = c + d[i]; d[i] = c; }
    const double last = d[n - 1];
    for (int i = 0; i < n; ++i) d[i] = d[i] / last;
    while (idx < n && d[idx] <= u) ++idx;            // searchsorted(side='right')
    if (idx >= n) idx = n - 1;
  } else {
    double m = -1.0;
    int nt = 0;
    for (int i = 0; i < n; ++i) m = d[i] > m ? d[i] : m;
    for (int i = 0; i < n; ++i) nt += (d[i] == m);
    int k = (int)(u * nt);
    if (k >= nt) k = nt - 1;
    for (int i = 0; i < n; ++i)
      if (d[i] == m) { if (k == 0) { idx = i; break; } --k; }
  }
  return idx;
}

// Config.select_action (config.py:70-81) + Game.store_search_statistics (game.py:106-115) + root error
// (actors.py:147-148).  One thread per tree (A <= 32 children, once per move).
__device__ __forceinline__ void mz_finalize_tree(const TreeView &t, int b, const double *temperature,
                                                 const double *uniform, uint64_t seed, uint64_t move, int env_offset,
                                                 int32_t *action, double *child_visits, double *root_value,
                                                 double *error, int32_t *visit_counts) {
  const int A = t.A;
  const size_t o = mz_slab(t, b);
  const uint32_t legal = t.legal[b];
  int acts[MZ_MAX_ACTIONS_K];
  double d[MZ_MAX_ACTIONS_K];
  int n = 0;
  long sumv = 0;
  for (int a = 0; a < A; ++a) {
    const bool ok = (legal >> a) & 1u;
    const int c = ok ? t.N[o + 1 + a] : 0;
    if (visit_counts) visit_counts[(size_t)b * A + a] = c;
    if (!ok) continue;
    acts[n] = a; d[n] = (double)c; sumv += c; ++n;
  }
  if (child_visits)
    for (int a = 0; a < A; ++a)
      child_visits[(size_t)b * A + a] = ((legal >> a) & 1u) ? (double)t.N[o + 1 + a] / (double)sumv : 0.0;
  const double rv = t.N[o] == 0 ? 0.0 : t.W[o] / (double)t.N[o];
  if (root_value) root_value[b] = rv;
  if (error) error[b] = rv - (double)t.root_value[b];
  if (!action) return;
  const double T = temperature[b];
  double u;
  if (uniform) {
    u = uniform[b];
  } else {
    mz_u4 r = mz_philox(seed, (uint32_t)(env_offset + b), (uint32_t)move, (uint32_t)(move >> 32), MZ_RNG_ACTION << 24);
    u = mz_u01(r.x, r.y);
  }
  const int idx = mz_sample_index(d, n, T, u);
  action[b] = acts[idx];
}


static __global__ void k_tree_finalize(TreeView t, const double *temperature, const double *uniform, uint64_t seed,
                                uint64_t move_val, const unsigned long long *move_ptr, int env_offset,
                                int32_t *action, double *child_visits, double *root_value, double *error,
                                int32_t *visit_counts) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  const uint64_t move = move_ptr ? (uint64_t)*move_ptr : move_val;
  mz_finalize_tree(t, b, temperature, uniform, seed, move, env_offset, action, child_visits, root_value, error,
                   visit_counts);
}

// hidden_out[b] = pool[b][slot[b]]  (what the reference hands to recurrent_inference, mcts.py:94-96)
static __global__ void k_gather_hidden(TreeView t, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.B * MZ_H) return;
  const int b = i / MZ_H, k = i % MZ_H;
  out[i] = t.hpool[((size_t)b * (t.sims + 1) + t.slot[b]) * MZ_HS + k];
}

// store an externally computed hidden state into the slot the next expansion will own
static __global__ void k_scatter_hidden(TreeView t, const float *in, int root) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.B * MZ_HS) return;
  const int b = i / MZ_HS, k = i % MZ_HS;
  const int slot = root ? 0 : t.nexp[b];
  t.hpool[((size_t)b * (t.sims + 1) + slot) * MZ_HS + k] = k < MZ_H ? in[(size_t)b * MZ_H + k] : 0.f;
}
