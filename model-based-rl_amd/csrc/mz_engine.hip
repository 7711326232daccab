// mz_engine.hip -- host side of libmz_hip.so: the C ABI of include/mz_engine.h over the gfx950
// kernels in mz_tree.hip.h / mz_net.hip.h / mz_selfplay.hip.h.  No torch types, no CPU fallback:
// every entry point launches HIP kernels on the stream it is given.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <deque>
#include <functional>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/mz_engine.h"
#include "../../include/mz_engine_debug.h"      // instrumentation / test hooks (not the boundary)

#define MZ_MAX_ACTIONS_K MZ_MAX_ACTIONS
#include "mz_common.h"
#include "mz_net.hip.h"
#include "mz_rng.h"
#include "mz_tree.hip.h"
#include "mz_selfplay.hip.h"
#include "mz_fused.hip.h"
#include "mz_root.hip.h"
#include "mz_fused_h2.hip.h"
#include "mz_learner.hip.h"
#include "mz_fcl.hip.h"
// the search kernels are compiled in their own translation units, one per shape (mz_inst.hip); here they are launched
#include "mz_kernels.inc"
MZ_ALL_FUSED(extern)
MZ_ALL_H2(extern)
#ifndef MZ_DEV_ONLY
// whole moves of the device TicTacToe environment (two players, 9 actions: the <15, 1, 16> shape, compact LDS trees)
extern template __global__ void k_search_fused<15, 1, 16, 2, false, false, true, true> MZ_KARGS;
#endif

static thread_local std::string g_err;

static int fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return -1;
}

#define HIPCHECK(x)                                                                        \
  do {                                                                                     \
    hipError_t err_ = (x);                                                                 \
    if (err_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(err_), __FILE__, __LINE__); \
  } while (0)

#define MZ_GRAPH_MOVES 16    // most self-play moves captured into one hipGraph (3 kernel nodes per move)

// offsets of the 22 tensors inside the flat weight vector (flat_layout below)
struct FlatLayout {
  size_t rep_w1, rep_b1, rep_w2, rep_b2, val_w1, val_b1, val_w2, val_b2, pol_w1, pol_b1, pol_w2, pol_b2, rew_w1,
      rew_b1, rew_w2, rew_b2, tr_w1, tr_b1, tr_w2, tr_b2, ln_w, ln_b, total;
};

struct mz_engine {
  mz_config cfg;
  int device = 0;                   // HIP device the engine's pools live on (the current device at mz_create)
  int B, Bp, A, O, sims, NN, PL, G, jtp;
  int NS = 0;                       // records per tree slab of the node pool
  MzNode *nodes = nullptr;          // the node pool [Bp][NS]
  TreeView tv;
  NetView nv;
  std::vector<void *> allocs;
  float *flat_dev = nullptr;
  int32_t *pack_idx = nullptr;
  int32_t *pack_idx2 = nullptr;     // a second source element per packed weight (-1: none), added before scaling
  float *packed = nullptr;
  size_t n_flat = 0, n_packed = 0;
  bool weights_set = false;
  int sims_done = 0;
  bool selection_valid = false;
  bool root_ready = false;
  hipStream_t cap_stream = nullptr;
  hipGraphExec_t search_graph = nullptr;
  int search_graph_sims = 0;
  bool use_graph = true;
  SelfplayState sp;
  hipGraphExec_t move_graph[MZ_GRAPH_MOVES + 1] = {};   // [k] = hipGraph of k consecutive self-play moves
  const f32x4 *wstream = nullptr;   // per-wave cyclic weight stream for the fused search kernel
  const f32x4 *istream = nullptr;   // per-wave weight stream of the root kernel (initial inference)
  int nst0 = 0;                     // its run-time first-stage steps (obs_dim + 1 columns, two k-steps per step)
  int ks1sel = 0;                   // fc1 k-steps of the fused kernel instantiation chosen for this A
  bool use_fused = true;
  bool use_lds_trees = true;
  bool use_lds_hybrid = true;
  bool fuse_record = false;         // set by the self-play loop around its search launch: finalize + record in the kernel tail
  unsigned long long *prof_buf = nullptr;   // non-null only inside mz_search_phase_profile
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // non-null only inside mz_search_timed: bracket the search kernel's dispatch
  bool lds_attr_set = false, lds_attr_set_prof = false;   // hipFuncAttributeMaxDynamicSharedMemorySize is per device: set once per engine
  bool lds_attr_set_head = false;
  bool use_persist = true;          // self-play loop: whole moves inside ONE launch of the search kernel (its HEAD instantiation)
  int persist_moves = 0;            // > 0 only around that launch: moves it plays
  float *direct_records = nullptr;  // mz_selfplay_steps_into: device mapping of the caller's pinned buffer, around its launches
  unsigned long long *head_prof = nullptr;   // non-null only inside mz_selfplay_phase_profile
  // record drain on a copy stream (mz_selfplay_drain): event behind the last copy, and how far the compute stream
  // has been ordered behind the copies
  hipEvent_t drain_ev = nullptr;    // the LATEST drain's copy (chains drains issued on different streams)
  hipStream_t drain_stream = nullptr;
  bool drain_pending = false;
  // every drain whose copy may still be reading the ring: (first move not covered = sp.drained after it, its event),
  // oldest first; events come from / go back to drain_ev_pool (mz_selfplay_steps waits for the OLDEST drain that still
  // covers the slots it is about to overwrite, not for the latest one)
  std::deque<std::pair<unsigned long long, hipEvent_t>> drain_q;
  std::vector<hipEvent_t> drain_ev_pool;
  unsigned long long drain_q_start = 0;     // first move covered by drain_q.front()
  double prof_spread[3] = {0, 0, 0};   // mean / min / max over workgroups of the last phase profile's total cycles
  bool split_f16 = false;           // FCNetwork GEMMs as float16 high/low splits (mz_fused_h2.hip.h)
  int32_t *pack_idx_h2 = nullptr;   // gather table of the split-f16 weight stream (bit 30: low part)
  uint16_t *packed_h2 = nullptr;    // [4 waves][NGROUPS][8 pieces][64 lanes][8] float16
  size_t n_packed_h2 = 0;
  float *obs_norm = nullptr;        // [2][O] --norm_obs minimum and range (device)
  double *noise_log = nullptr;      // [ring_moves][B][A] per-move Dirichlet draws (mz_selfplay_noise_log), allocated on first use
  FlatLayout layout;
  float relu_scale_host[4] = {1.f, 1.f, 1.f, 0.f};
  bool scale_host_valid = true;     // relu_scale_host mirrors relu_scale_dev (not after mz_set_weights_async: read back on demand)
  float *wstage[2] = {nullptr, nullptr};      // pinned staging of mz_set_weights_async's host source, used alternately
  hipEvent_t wstage_ev[2] = {nullptr, nullptr};
  int wstage_next = 0;
  bool stream_scaled = false;       // the last weight set admitted a scale: the fused search kernels may run
  float *relu_scale_dev = nullptr;     // {1, 2^-k, 2^k, flag} of the current weight set (k_relu_scale)
  unsigned *relu_bound_dev = nullptr;  // k_relu_bound's two maxima
  bool root_hidden_external = false;  // the root's hidden state came in through mz_root_load: no bound on it is known
  unsigned *absmax_dev = nullptr;   // scratch of the split_f16 weight-range check (mz_set_weights)
  double *draw_uniform = nullptr;   // [B] host-given uniforms of a game environment's parity run (mz_selfplay_set_draws)
  bool draws_noise = false, draws_set = false;
  std::vector<float> obs_norm_host;
};

// Every ABI entry runs on the engine's own device, whatever the calling thread's current device is (an engine may be
// driven from a worker thread: rayshim actors, a replay/drain thread).
// every entry runs on the handle's own device and leaves the calling thread's current device as it found it
struct MzDeviceGuard {
  int prev = -1;
  bool switched = false;
  int enter(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) {
      if (hipSetDevice(device) != hipSuccess) return -1;
      switched = true;
    }
    return 0;
  }
  ~MzDeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};
#define MZ_ENTER(e)                                                                          \
  MzDeviceGuard mz_guard_;                                                                   \
  if (mz_guard_.enter((e)->device)) return fail("hipSetDevice(%d) failed", (e)->device)

__global__ void k_store_double(double *dst, double v) { *dst = v; }

// what mz_select hands back, in one launch: any of the four outputs may be null
__global__ void k_export_selection(TreeView t, int32_t *leaf, int32_t *slot, int32_t *act, int32_t *depth) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  if (leaf) leaf[b] = t.leaf[b];
  if (slot) slot[b] = t.slot[b];
  if (act) act[b] = t.act[b];
  if (depth) depth[b] = t.depth[b];
}

// y = relu(y * scale[c] + shift[c] (+ residual)) in place over an NCHW float32 tensor, HW a multiple of 4: one 16-byte
// access per thread and operand (a float4 never straddles a channel), grid-stride.  HBM-bound: 2 or 3 tensor passes.
template <bool RES>
__global__ __launch_bounds__(256) void k_affine_relu(float4 *__restrict__ y, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, const float4 *__restrict__ res,
                                                      size_t n4, int C, int hw4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i / (size_t)hw4) % (size_t)C);
    const float a = scale[c], b = shift[c];
    float4 v = y[i];
    v.x = v.x * a + b; v.y = v.y * a + b; v.z = v.z * a + b; v.w = v.w * a + b;
    if constexpr (RES) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    y[i] = v;
  }
}

// the fused search kernels (mz_fused.hip.h, mz_fused_h2.hip.h) can run: trees fit their path buffer, and the exact-f32
// kernel's weight stream could be scaled for its clamp ReLU (mz_set_weights; the split-f16 kernel has its own stream)
static inline bool fused_usable(const mz_engine *e) {
  return e->use_fused && e->sims + 2 <= MZ_FUSED_MAXPL && (e->split_f16 || e->stream_scaled || !e->weights_set);
}

template <typename T>
static int dmalloc(mz_engine *e, T **p, size_t n) {
  void *q = nullptr;
  HIPCHECK(hipMalloc(&q, n * sizeof(T) + 64));
  HIPCHECK(hipMemset(q, 0, n * sizeof(T) + 64));
  e->allocs.push_back(q);
  *p = (T *)q;
  return 0;
}

// ---------------------------------------------------------------- weight layout

static FlatLayout flat_layout(int O, int A, int Sv, int Sr) {
  FlatLayout L;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += n; return o; };
  const int K = MZ_H + A;
  L.rep_w1 = take((size_t)MZ_F * O); L.rep_b1 = take(MZ_F); L.rep_w2 = take((size_t)MZ_H * MZ_F); L.rep_b2 = take(MZ_H);
  L.val_w1 = take((size_t)MZ_F * MZ_H); L.val_b1 = take(MZ_F); L.val_w2 = take((size_t)Sv * MZ_F); L.val_b2 = take(Sv);
  L.pol_w1 = take((size_t)MZ_F * MZ_H); L.pol_b1 = take(MZ_F); L.pol_w2 = take((size_t)A * MZ_F); L.pol_b2 = take(A);
  L.rew_w1 = take((size_t)MZ_F * K); L.rew_b1 = take(MZ_F); L.rew_w2 = take((size_t)Sr * MZ_F); L.rew_b2 = take(Sr);
  L.tr_w1 = take((size_t)MZ_F * K); L.tr_b1 = take(MZ_F); L.tr_w2 = take((size_t)MZ_H * MZ_F); L.tr_b2 = take(MZ_H);
  L.ln_w = take(MZ_H); L.ln_b = take(MZ_H);
  L.total = off;
  return L;
}

// fc1 pack: weights [4 waves][NT/4][ks][64][4], bias [4][NT][64][4]; NT = 8*NH tiles per wave, tile t of
// wave w = head t/8, features 128w + 16(t%8) + 0..15.
static void fill_fc1(std::vector<int32_t> &idx, size_t wpos, size_t bpos, int NH, const size_t *woff,
                     const size_t *boff, int K, int ks) {
  const int NT = 8 * NH, TG = NT / 4;
  for (int w = 0; w < 4; ++w)
    for (int tg = 0; tg < TG; ++tg)
      for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int t = 4 * tg + i, head = t / 8, tt = t % 8;
            const int nf = 128 * w + 16 * tt + (lane & 15), k = 4 * s + (lane >> 4);
            idx[wpos + ((((size_t)(w * TG + tg) * ks + s) * 64 + lane) * 4 + i)] =
                k < K ? (int32_t)(woff[head] + (size_t)nf * K + k) : -1;
          }
  for (int w = 0; w < 4; ++w)
    for (int t = 0; t < NT; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int head = t / 8, tt = t % 8;
          const int nf = 128 * w + 16 * tt + 4 * (lane >> 4) + r;
          idx[bpos + (((size_t)(w * NT + t) * 64 + lane) * 4 + r)] = (int32_t)(boff[head] + nf);
        }
}

// fc1 pack for the fused kernel: as fill_fc1 (two heads), but input column K carries the bias
static void fill_fc1_biascol(std::vector<int32_t> &idx, size_t wpos, const size_t *woff, const size_t *boff, int K,
                             int ks) {
  for (int w = 0; w < 4; ++w)
    for (int tg = 0; tg < 4; ++tg)
      for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int t = 4 * tg + i, head = t / 8, tt = t % 8;
            const int nf = 128 * w + 16 * tt + (lane & 15), k = 4 * s + (lane >> 4);
            int32_t v = -1;
            if (k < K) v = (int32_t)(woff[head] + (size_t)nf * K + k);
            else if (k == K) v = (int32_t)(boff[head] + nf);
            idx[wpos + ((((size_t)(w * 4 + tg) * ks + s) * 64 + lane) * 4 + i)] = v;
          }
}

// dynamics fc1 pack for the fused kernel: as fill_fc1_biascol, but without a bias column -- the input's one-hot part holds
// exactly one 1, so the bias rides in the one-hot columns: their packed weight is W[n][50 + a] + b[n] (idx2 = the
// bias element to add, k_pack_weights).  K = 50 + A columns instead of 51 + A: one k-step of four fewer for
// A = 6, 10, 14, ... (Pong-ram: 14 steps instead of 15).
static void fill_fc1_foldbias(std::vector<int32_t> &idx, std::vector<int32_t> &idx2, size_t wpos, const size_t *woff,
                              const size_t *boff, int K, int ks) {
  for (int w = 0; w < 4; ++w)
    for (int tg = 0; tg < 4; ++tg)
      for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int t = 4 * tg + i, head = t / 8, tt = t % 8;
            const int nf = 128 * w + 16 * tt + (lane & 15), k = 4 * s + (lane >> 4);
            const size_t o = wpos + ((((size_t)(w * 4 + tg) * ks + s) * 64 + lane) * 4 + i);
            idx[o] = k < K ? (int32_t)(woff[head] + (size_t)nf * K + k) : -1;
            idx2[o] = (k >= MZ_H && k < K) ? (int32_t)(boff[head] + nf) : -1;
          }
}

// fc2 pack: [JT][4 waves][8 tiles][64][4]: A operand row j = 16jt + (lane&15), k = 128w + 16t + 4(lane>>4) + r
static void fill_fc2(std::vector<int32_t> &idx, size_t wpos, int JT, size_t woff, int J) {
  for (int jt = 0; jt < JT; ++jt)
    for (int w = 0; w < 4; ++w)
      for (int t = 0; t < 8; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + (lane & 15), nf = 128 * w + 16 * t + 4 * (lane >> 4) + r;
            idx[wpos + ((((size_t)(jt * 4 + w) * 8 + t) * 64 + lane) * 4 + r)] =
                j < J ? (int32_t)(woff + (size_t)j * MZ_F + nf) : -1;
          }
}

static void fill_vec(std::vector<int32_t> &idx, size_t pos, int padded, size_t off, int n) {
  for (int i = 0; i < padded; ++i) idx[pos + i] = i < n ? (int32_t)(off + i) : -1;
}

// ---- split-f16 weight stream (mz_fused_h2.hip.h): per wave [NGROUPS][8 pieces][64 lanes][8 halfs]; a group = the A
// operands of four 16 x 32 blocks ("units"): pieces 0..3 their high parts, 4..7 their low parts.  Lane l of a piece holds
// row l & 15, k = 8 (l >> 4) + j of the block.
__global__ void k_pack_weights_h2(const float *flat, const int32_t *idx, _Float16 *packed, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t s = idx[i];
  float v = 0.f;
  if (s >= 0) v = flat[s & 0x3fffffff];
  const _Float16 h = (_Float16)v;                                  // round to nearest
  packed[i] = (s >= 0 && (s & 0x40000000)) ? (_Float16)(v - (float)h) : h;
}

// max |w| over the flat weights, or +inf if any is not finite (bit pattern compare: non-negative floats order like uints)
__global__ void k_absmax(const float *w, size_t n, unsigned *out) {
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned u = __float_as_uint(w[i]) & 0x7fffffffu;
    m = u > m ? u : m;
  }
  atomicMax(out, m);
}

// The power of two the search kernel's weight stream is scaled by (mz_fused.hip.h, "ReLU as a clamp"): 2^k above an upper
// bound of every output of the four 512-wide fc1 layers the search runs (reward, transition: input [hidden | one-hot];
// value, policy: input hidden), given that the hidden state is relu(LayerNorm(.)) -- |(x_i - mean) / std| <= sqrt(49)
// for 50 features, so 0 <= hidden_i <= 7 |gamma_i| + |beta_i| -- and a one-hot row holds a single 1:
//   |out_n| <= sum_i |W[n][i]| hb_i + max_a |W[n][50 + a]| + |b_n|       (every partial sum obeys the same bound).
// k_relu_bound: 64 workgroups, 8 lanes per fc1 row; the largest row bound and the largest |w| of the layers that consume
// the activations (they are multiplied by 2^k) land in mx[0], mx[1] as bit patterns (non-negative floats order like
// unsigned integers, and every NaN pattern lies above +inf: a NaN sticks).
__global__ void k_relu_bound(const float *flat, FlatLayout L, int A, int Sr, int Sv, unsigned *mx) {
  __shared__ float hb[MZ_H];
  const int tid = threadIdx.x;
  if (tid < MZ_H) hb[tid] = 7.01f * fabsf(flat[L.ln_w + tid]) + fabsf(flat[L.ln_b + tid]);
  __syncthreads();
  const size_t w1[4] = {L.rew_w1, L.tr_w1, L.val_w1, L.pol_w1}, b1[4] = {L.rew_b1, L.tr_b1, L.val_b1, L.pol_b1};
  const int row = blockIdx.x * 32 + (tid >> 3), q = tid & 7;       // 2048 rows = 4 heads x 512
  const int h = row >> 9, n = row & 511;
  const int K = h < 2 ? MZ_H + A : MZ_H;
  const float *r = flat + w1[h] + (size_t)n * K;
  float s = 0.f, oh = 0.f;
  for (int i = q; i < MZ_H; i += 8) s += fabsf(r[i]) * hb[i];
  for (int a = MZ_H + q; a < K; a += 8) oh = fmaxf(oh, fabsf(r[a]));
  unsigned nanbit = (s != s || oh != oh) ? 1u : 0u;
  for (int o = 1; o < 8; o <<= 1) {
    s += __shfl_xor(s, o); oh = fmaxf(oh, __shfl_xor(oh, o)); nanbit |= __shfl_xor(nanbit, o);
  }
  s += oh + fabsf(flat[b1[h] + n]);
  unsigned m = __float_as_uint(s) & 0x7fffffffu;
  if (nanbit) m = 0x7fc00000u;
  // largest |w| of the consuming layers, grid-strided
  const size_t w2[4] = {L.rew_w2, L.tr_w2, L.val_w2, L.pol_w2};
  const int J[4] = {Sr, MZ_H, Sv, A};
  unsigned m2 = 0;
  for (int hh = 0; hh < 4; ++hh)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < (size_t)J[hh] * MZ_F; i += (size_t)gridDim.x * blockDim.x) {
      const unsigned u = __float_as_uint(flat[w2[hh] + i]) & 0x7fffffffu;
      m2 = u > m2 ? u : m2;
    }
  __shared__ unsigned red[2];
  if (tid < 2) red[tid] = 0;
  __syncthreads();
  if (q == 0) atomicMax(&red[0], m);
  atomicMax(&red[1], m2);
  __syncthreads();
  if (tid < 2) atomicMax(&mx[tid], red[tid]);
}

// scale = {1, 2^-k, 2^k, 1}; {1, 1, 1, 0} when the bound is not finite, k would exceed 40, or 2^k times the largest
// weight of a consuming layer would leave the float32 range: the host then routes the engine to the stand-alone kernels.
// force < 0: decided here; force = 0 / 1: the HOST's decision (mz_weights_scale_ok on a host copy of the same weights,
// mz_set_weights_async) -- its limits are tighter than the ones below, so a 1 it hands over is one this kernel agrees with;
// should it ever not, scale[3] = NaN poisons the packed stream (k_pack_weights multiplies by it): every value the search
// computes is NaN, loudly, instead of a clamp applied to an unscaled stream.
__global__ void k_relu_scale(const unsigned *mx, float *scale, int force) {
  const float bound = __uint_as_float(mx[0]) * 1.01f;     // (rounding of the sums above and of the kernel's own accumulation)
  const float w2max = __uint_as_float(mx[1]);
  int k = 0;
  if (bound > 1.f) (void)frexpf(bound, &k);               // bound = f 2^k, 0.5 <= f < 1: bound < 2^k
  const bool own = bound == bound && bound < 0x1p40f && w2max == w2max && w2max < ldexpf(1.f, 100 - k);
  const bool ok = own && force != 0;
  if (force > 0 && !own) {
    scale[0] = scale[1] = scale[2] = scale[3] = __uint_as_float(0x7fc00000u);
    return;
  }
#ifdef MZ_RELU_VMAX
  k = 0;
#endif
  scale[0] = 1.f;
  scale[1] = ok ? ldexpf(1.f, -k) : 1.f;
  scale[2] = ok ? ldexpf(1.f, k) : 1.f;
  scale[3] = ok ? 1.f : 0.f;
}

static int build_packing_h2(mz_engine *e, const FlatLayout &L, int Sv, int Sr) {
  using SC = H2Sched;
  const int A = e->A;
  const size_t per_wave = (size_t)SC::NGROUPS * 8 * 512;
  e->n_packed_h2 = 4 * per_wave;
  std::vector<int32_t> idx(e->n_packed_h2, -1);
  for (int w = 0; w < 4; ++w) {
    size_t group = 0;
    // unit: source index of (row 0..15, k 0..31) of one block, or -1
    auto emit = [&](const std::function<int32_t(int, int)> (&unit)[4]) {
      for (int part = 0; part < 2; ++part)
        for (int u = 0; u < 4; ++u) {
          const size_t base = (size_t)w * per_wave + (group * 8 + (size_t)part * 4 + u) * 512;
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int32_t sidx = unit[u](lane & 15, 8 * (lane >> 4) + j);
              idx[base + (size_t)lane * 8 + j] = sidx < 0 ? -1 : (sidx | (part ? 0x40000000 : 0));
            }
        }
      ++group;
    };
    // fc1 block: tile tt of this wave (head tt / 8, feature 128 w + 16 (tt % 8) + row), K chunk c; column K1 = bias
    auto fc1_unit = [&](int tt, int c, const size_t *woff, const size_t *boff, int K1) {
      return [=](int row, int kl) -> int32_t {
        const int head = tt / 8, nf = 128 * w + 16 * (tt % 8) + row, k = 32 * c + kl;
        if (k < K1) return (int32_t)(woff[head] + (size_t)nf * K1 + k);
        if (k == K1) return (int32_t)(boff[head] + nf);
        return -1;
      };
    };
    // fc2 block: output rows 16 ot + row (< J) of the layer at woff, K pair p of this wave's eight tiles of the head:
    // k = 8 g + j  <->  feature 128 w + 16 (2 p + (j >= 4)) + 4 g + (j & 3)   (the D fragments of tiles 2p, 2p+1)
    auto fc2_unit = [&](size_t woff, int ot, int J, int p) {
      return [=](int row, int kl) -> int32_t {
        const int r = 16 * ot + row, g = kl / 8, j = kl % 8;
        if (r >= J) return -1;
        const int feat = 128 * w + 16 * (2 * p + (j >= 4 ? 1 : 0)) + 4 * g + (j & 3);
        return (int32_t)(woff + (size_t)r * MZ_F + feat);
      };
    };
    const size_t d1w[2] = {L.rew_w1, L.tr_w1}, d1b[2] = {L.rew_b1, L.tr_b1};
    const size_t p1w[2] = {L.val_w1, L.pol_w1}, p1b[2] = {L.val_b1, L.pol_b1};
    for (int tg = 0; tg < 4; ++tg)
      for (int c = 0; c < 2; ++c) {
        const std::function<int32_t(int, int)> u[4] = {fc1_unit(4 * tg, c, d1w, d1b, MZ_H + A), fc1_unit(4 * tg + 1, c, d1w, d1b, MZ_H + A),
                                                         fc1_unit(4 * tg + 2, c, d1w, d1b, MZ_H + A), fc1_unit(4 * tg + 3, c, d1w, d1b, MZ_H + A)};
        emit(u);
      }
    for (int half = 0; half < 2; ++half) {
      for (int sub = 0; sub < 2; ++sub) {
        const int p = 2 * half + sub;
        const std::function<int32_t(int, int)> u[4] = {fc2_unit(L.rew_w2, 0, Sr, p), fc2_unit(L.rew_w2, 1, Sr, p),
                                                         fc2_unit(L.tr_w2, 0, MZ_H, p), fc2_unit(L.tr_w2, 1, MZ_H, p)};
        emit(u);
      }
      const std::function<int32_t(int, int)> u[4] = {fc2_unit(L.tr_w2, 2, MZ_H, 2 * half), fc2_unit(L.tr_w2, 3, MZ_H, 2 * half),
                                                       fc2_unit(L.tr_w2, 2, MZ_H, 2 * half + 1), fc2_unit(L.tr_w2, 3, MZ_H, 2 * half + 1)};
      emit(u);
    }
    for (int tg = 0; tg < 4; ++tg)
      for (int c = 0; c < 2; ++c) {
        const std::function<int32_t(int, int)> u[4] = {fc1_unit(4 * tg, c, p1w, p1b, MZ_H), fc1_unit(4 * tg + 1, c, p1w, p1b, MZ_H),
                                                         fc1_unit(4 * tg + 2, c, p1w, p1b, MZ_H), fc1_unit(4 * tg + 3, c, p1w, p1b, MZ_H)};
        emit(u);
      }
    for (int k = 0; k < 2; ++k) {
      const std::function<int32_t(int, int)> u[4] = {fc2_unit(L.val_w2, 0, Sv, 2 * k), fc2_unit(L.val_w2, 1, Sv, 2 * k),
                                                       fc2_unit(L.val_w2, 0, Sv, 2 * k + 1), fc2_unit(L.val_w2, 1, Sv, 2 * k + 1)};
      emit(u);
    }
    {
      const std::function<int32_t(int, int)> u[4] = {fc2_unit(L.pol_w2, 0, A, 0), fc2_unit(L.pol_w2, 0, A, 1),
                                                       fc2_unit(L.pol_w2, 0, A, 2), fc2_unit(L.pol_w2, 0, A, 3)};
      emit(u);
    }
    if ((int)group != SC::REAL) return fail("internal: split-f16 stream has %zu groups, expected %d", group, SC::REAL);
  }
  if (dmalloc(e, &e->pack_idx_h2, e->n_packed_h2)) return -1;
  if (dmalloc(e, &e->packed_h2, e->n_packed_h2)) return -1;
  HIPCHECK(hipMemcpy(e->pack_idx_h2, idx.data(), e->n_packed_h2 * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}

static int build_packing(mz_engine *e) {
  const int O = e->O, A = e->A;
  const int Sv = e->cfg.no_support ? 1 : e->cfg.value_support_max - e->cfg.value_support_min + 1;   // networks.py:135-136
  const int Sr = e->cfg.no_support ? 1 : e->cfg.reward_support_max - e->cfg.reward_support_min + 1;
  const FlatLayout L = flat_layout(O, A, Sv, Sr);
  e->layout = L;
  e->n_flat = L.total;
  const int ks0 = (O + 3) / 4, ks1 = (MZ_H + A + 3) / 4, ks3 = (MZ_H + 3) / 4;
  const int jtp = e->jtp;
  size_t pos = 0;
  auto seg = [&](size_t n) { size_t p = pos; pos += (n + 3) & ~(size_t)3; return p; };
  const size_t p_w0o = seg(4 * 4 * 8 * 256), p_b0o = seg(64);
  const size_t p_w1 = seg((size_t)4 * 4 * ks1 * 256), p_b1 = seg(4 * 16 * 256), p_w2 = seg(6 * 4 * 8 * 256), p_b2 = seg(96);
  const size_t p_w3 = seg((size_t)4 * 4 * ks3 * 256), p_b3 = seg(4 * 16 * 256);
  const size_t p_w4 = seg((size_t)(2 + jtp) * 4 * 8 * 256), p_b4 = seg(32 + 16 * jtp);
  const size_t p_lnw = seg(64), p_lnb = seg(64);
  // fused kernel: fc1 weights with the bias inside the packed matrix -- prediction: one more input column (constant-1
  // input); dynamics: added to the one-hot columns (fill_fc1_foldbias); the k-step count is that of the kernel
  // instantiation chosen for this action count (zero-padded above 50+A)
  const int ks1f = e->ks1sel, ks3f = (MZ_H + 1 + 3) / 4;
  const size_t p_w1f = seg((size_t)4 * 4 * ks1f * 256), p_w3f = seg((size_t)4 * 4 * ks3f * 256);
  const int nj2 = 2 + jtp;
  const int real_steps = ks1f + 12 + ks3f + 2 * nj2;
  const int rs = mz_fused_rs(ks1f, jtp);
  const int nsteps = rs + (real_steps - rs + MZ_NB - 1) / MZ_NB * MZ_NB;   // FusedSched::NSTEPS
  const size_t p_ws = seg((size_t)4 * nsteps * 4 * 256);
  // A operands of the small MFMA (mz_fused.hip.h): rows 48..51 of the next hidden state; the policy head when A <= 4
  const size_t p_h4 = seg((size_t)4 * 8 * 256), p_p4 = seg((size_t)4 * 8 * 256);
  const bool p4 = A <= 4;
  // root kernel stream: [nst0 first-stage steps][8 representation-out][13 prediction fc1][2*nj2 prediction out]
  const int nst0 = mz_root_nst0(O), nroot = nst0 + 8 + ks3f + 2 * nj2;
  const size_t p_is = seg((size_t)4 * nroot * 4 * 256);
  e->n_packed = pos;
  std::vector<int32_t> idx(pos, -1);
  std::vector<int32_t> idx2(pos, -1);      // second source element, added to the first (the folded bias of fill_fc1_foldbias)
  {
    fill_fc2(idx, p_w0o, 4, L.rep_w2, MZ_H);
    fill_vec(idx, p_b0o, 64, L.rep_b2, MZ_H);
  }
  {
    size_t wo[2] = {L.rew_w1, L.tr_w1}, bo[2] = {L.rew_b1, L.tr_b1};
    fill_fc1(idx, p_w1, p_b1, 2, wo, bo, MZ_H + A, ks1);
    fill_fc2(idx, p_w2, 2, L.rew_w2, Sr);
    fill_fc2(idx, p_w2 + (size_t)2 * 4 * 8 * 256, 4, L.tr_w2, MZ_H);
    fill_vec(idx, p_b2, 32, L.rew_b2, Sr);
    fill_vec(idx, p_b2 + 32, 64, L.tr_b2, MZ_H);
  }
  {
    size_t wo[2] = {L.val_w1, L.pol_w1}, bo[2] = {L.val_b1, L.pol_b1};
    fill_fc1(idx, p_w3, p_b3, 2, wo, bo, MZ_H, ks3);
    fill_fc2(idx, p_w4, 2, L.val_w2, Sv);
    fill_fc2(idx, p_w4 + (size_t)2 * 4 * 8 * 256, jtp, L.pol_w2, A);
    fill_vec(idx, p_b4, 32, L.val_b2, Sv);
    fill_vec(idx, p_b4 + 32, 16 * jtp, L.pol_b2, A);
  }
  fill_vec(idx, p_lnw, 64, L.ln_w, MZ_H);
  fill_vec(idx, p_lnb, 64, L.ln_b, MZ_H);
  {
    size_t wo[2] = {L.rew_w1, L.tr_w1}, bo[2] = {L.rew_b1, L.tr_b1};
    fill_fc1_foldbias(idx, idx2, p_w1f, wo, bo, MZ_H + A, ks1f);
    size_t wo3[2] = {L.val_w1, L.pol_w1}, bo3[2] = {L.val_b1, L.pol_b1};
    fill_fc1_biascol(idx, p_w3f, wo3, bo3, MZ_H, ks3f);
  }
  // small-MFMA pieces, [wave][t][lane][r]: lane l carries output row l & 3 for the features 128w + 16t + 4(l >> 4) + r
  for (int w = 0; w < 4; ++w)
    for (int t = 0; t < 8; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int row = lane & 3, nf = 128 * w + 16 * t + 4 * (lane >> 4) + r;
          const size_t o = ((size_t)(w * 8 + t) * 64 + lane) * 4 + r;
          idx[p_h4 + o] = 48 + row < MZ_H ? (int32_t)(L.tr_w2 + (size_t)(48 + row) * MZ_F + nf) : -1;
          idx[p_p4 + o] = row < A ? (int32_t)(L.pol_w2 + (size_t)row * MZ_F + nf) : -1;
        }
  // the fused kernel's weight stream: per wave [nsteps][4 pieces][64 lanes][4], consumption order
  for (int w = 0; w < 4; ++w) {
    size_t piece = 0;
    // cls: the scale class of the piece (bits 29-30 of its gather indices, k_pack_weights): 1 = an fc1 layer of the
    // search (times 2^-k), 2 = a layer that consumes its activations (times 2^k) -- k_relu_scale, mz_fused.hip.h
    auto put = [&](size_t src, int cls) {
      const size_t d0 = p_ws + ((size_t)w * nsteps * 4 + piece) * 256;
      for (int i = 0; i < 256; ++i) {
        idx[d0 + i] = idx[src + i] < 0 ? -1 : (idx[src + i] | (cls << 29));
        idx2[d0 + i] = idx2[src + i];
      }
      ++piece;
    };
    for (int st = 0; st < ks1f; ++st)
      for (int tg = 0; tg < 4; ++tg) put(p_w1f + ((size_t)(w * 4 + tg) * ks1f + st) * 256, 1);
    for (int t = 0; t < 8; ++t)
      for (int jt = 0; jt < 6; ++jt) put(jt == 5 ? p_h4 + (size_t)(w * 8 + t) * 256 : p_w2 + ((size_t)(jt * 4 + w) * 8 + t) * 256, 2);
    for (int st = 0; st < ks3f; ++st)
      for (int tg = 0; tg < 4; ++tg) put(p_w3f + ((size_t)(w * 4 + tg) * ks3f + st) * 256, 1);
    for (int t = 0; t < 8; ++t)
      for (int jt = 0; jt < nj2; ++jt)
        put(p4 && jt == 2 ? p_p4 + (size_t)(w * 8 + t) * 256 : p_w4 + ((size_t)(jt * 4 + w) * 8 + t) * 256, 2);
    if ((int)piece != real_steps * 4) return fail("internal: weight stream has %zu pieces, expected %d", piece, real_steps * 4);
    // the remaining (nsteps - real_steps) * 4 pieces are padding (index -1 -> 0.0), loaded but never used
  }

  // the root kernel's stream (mz_root.hip.h).  First stage, wave w, step st, piece p: k-step 2*st + (p>>1),
  // tiles 4*(p&1)..+3 of the wave's 8 (features 128w + 16*tile + (lane&15)), k = 4*kstep + (lane>>4); column O
  // carries the bias.
  for (int w = 0; w < 4; ++w) {
    const size_t base = p_is + (size_t)w * nroot * 4 * 256;
    for (int st = 0; st < nst0; ++st)
      for (int p = 0; p < 4; ++p)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int tile = 4 * (p & 1) + i, nf = 128 * w + 16 * tile + (lane & 15);
            const int k = 4 * (2 * st + (p >> 1)) + (lane >> 4);
            int32_t v = -1;
            if (k < O) v = (int32_t)(L.rep_w1 + (size_t)nf * O + k);
            else if (k == O) v = (int32_t)(L.rep_b1 + nf);
            idx[base + ((size_t)(st * 4 + p) * 64 + lane) * 4 + i] = v;
          }
    size_t piece = (size_t)nst0 * 4;
    // (the root's prediction stage shares the search's scaled layers: its v_max ReLU is scale-blind; the representation
    // layers take raw observations, for which no bound exists: unscaled)
    auto put = [&](size_t src, int cls) {
      int32_t *dst = &idx[base + piece * 256];
      for (int i = 0; i < 256; ++i) dst[i] = idx[src + i] < 0 ? -1 : (idx[src + i] | (cls << 29));
      ++piece;
    };
    for (int t = 0; t < 8; ++t)
      for (int jt = 0; jt < 4; ++jt) put(p_w0o + ((size_t)(jt * 4 + w) * 8 + t) * 256, 0);
    for (int st = 0; st < ks3f; ++st)
      for (int tg = 0; tg < 4; ++tg) put(p_w3f + ((size_t)(w * 4 + tg) * ks3f + st) * 256, 1);
    for (int t = 0; t < 8; ++t)
      for (int jt = 0; jt < nj2; ++jt) put(p_w4 + ((size_t)(jt * 4 + w) * 8 + t) * 256, 2);
    if ((int)piece != nroot * 4) return fail("internal: root stream has %zu pieces, expected %d", piece, nroot * 4);
  }

  if (L.total >= ((size_t)1 << 29)) return fail("internal: %zu weights do not fit the gather index", L.total);
  if (dmalloc(e, &e->pack_idx, pos)) return -1;
  if (dmalloc(e, &e->pack_idx2, pos)) return -1;
  if (dmalloc(e, &e->packed, pos)) return -1;
  if (dmalloc(e, &e->relu_scale_dev, (size_t)4)) return -1;
  if (dmalloc(e, &e->relu_bound_dev, (size_t)2)) return -1;
  if (dmalloc(e, &e->flat_dev, L.total)) return -1;
  HIPCHECK(hipMemcpy(e->pack_idx, idx.data(), pos * sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(e->pack_idx2, idx2.data(), pos * sizeof(int32_t), hipMemcpyHostToDevice));
  NetView &n = e->nv;
  const float *P = e->packed;
  n.w0o = (const f32x4 *)(P + p_w0o); n.b0o = P + p_b0o;
  n.w1 = (const f32x4 *)(P + p_w1); n.b1 = (const f32x4 *)(P + p_b1); n.w2 = (const f32x4 *)(P + p_w2); n.b2 = P + p_b2;
  n.w3 = (const f32x4 *)(P + p_w3); n.b3 = (const f32x4 *)(P + p_b3); n.w4 = (const f32x4 *)(P + p_w4); n.b4 = P + p_b4;
  n.lnw = P + p_lnw; n.lnb = P + p_lnb;
  e->wstream = (const f32x4 *)(P + p_ws);
  e->istream = (const f32x4 *)(P + p_is);
  e->nst0 = nst0;
  n.ks0 = ks0; n.ks1 = ks1; n.ks3 = ks3; n.O = O; n.A = A; n.jtp = jtp;
  n.Sr = Sr; n.Sv = Sv;
  n.rmin = e->cfg.no_support ? 0 : e->cfg.reward_support_min;   // (0: mz_support_to_scalar_q adds the raw output to it)
  n.vmin = e->cfg.no_support ? 0 : e->cfg.value_support_min;
  n.no_transform = e->cfg.no_support ? 2 : (e->cfg.no_target_transform ? 1 : 0);
  if (e->split_f16 && build_packing_h2(e, L, Sv, Sr)) return -1;
  return 0;
}

#define NET_LAUNCH(kern, grid, s, ...)                                                        \
  do {                                                                                        \
    if (e->jtp == 1) hipLaunchKernelGGL(kern<1>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);    \
    else hipLaunchKernelGGL(kern<2>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);                \
  } while (0)

// (e->ev_start set: the dispatch carries start / stop events -- mz_tree_pair_timed, the clock of bench.py --workload tree)
#define TREE_GO_(kern, G_, s, ...)                                                                                \
  do {                                                                                                            \
    if (e->ev_start) hipExtLaunchKernelGGL(kern<G_>, dim3(blocks_), dim3(threads_), 0, s, e->ev_start, e->ev_stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern<G_>, dim3(blocks_), dim3(threads_), 0, s, __VA_ARGS__);                          \
  } while (0)
#define TREE_LAUNCH(kern, s, ...)                                                             \
  do {                                                                                        \
    const int threads_ = 256;                                                                 \
    const int total_ = e->B * e->G;                                                           \
    const int blocks_ = (total_ + threads_ - 1) / threads_;                                   \
    switch (e->G) {                                                                           \
      case 4: TREE_GO_(kern, 4, s, __VA_ARGS__); break;                                       \
      case 8: TREE_GO_(kern, 8, s, __VA_ARGS__); break;                                       \
      case 16: TREE_GO_(kern, 16, s, __VA_ARGS__); break;                                     \
      default: TREE_GO_(kern, 32, s, __VA_ARGS__); break;                                     \
    }                                                                                         \
  } while (0)


// BaseNetwork.initial_inference for all B rows; selfplay: + synthetic observation, root expansion, Dirichlet
// noise and first descent (mz_root.hip.h)
template <int JTP>
static int launch_root_j(mz_engine *e, const float *obs, bool selfplay, hipStream_t s) {
  const dim3 grid(e->Bp / MZ_ROWS), block(256);
  const double alpha = e->cfg.root_dirichlet_alpha, frac = e->cfg.root_exploration_fraction;
#define MZ_ROOT_GO(G_, SP_)                                                                                     \
  hipLaunchKernelGGL((k_root<JTP, G_, SP_>), grid, block, 0, s, e->nv, e->tv, obs, e->istream, e->nst0, e->sp, \
                     (uint64_t)e->cfg.seed, alpha, frac)
  if (!selfplay) MZ_ROOT_GO(4, false);
  else switch (e->G) {
    case 4: MZ_ROOT_GO(4, true); break;
    case 8: MZ_ROOT_GO(8, true); break;
    case 16: MZ_ROOT_GO(16, true); break;
    default: MZ_ROOT_GO(32, true); break;
  }
#undef MZ_ROOT_GO
  HIPCHECK(hipGetLastError());
  return 0;
}
static int launch_root(mz_engine *e, const float *obs, bool selfplay, hipStream_t s) {
  return e->jtp == 1 ? launch_root_j<1>(e, obs, selfplay, s) : launch_root_j<2>(e, obs, selfplay, s);
}

// dynamic LDS of a HEAD launch: the root's working set (mz_root_body) lies over the trees, behind the pb_c table
static size_t fused_head_dyn_lds(int sims, int NN, int lt) {
  const size_t table = ((size_t)(sims + 2) * (lt == 2 ? sims + 2 : 64) * 8 + 15) & ~(size_t)15;
  const size_t root = table + sizeof(float) * MZ_ROOT_LDS_FLOATS, trees = mz_fused_dyn_lds(sims, NN, lt);
  return root > trees ? root : trees;
}

template <int KS1, int JTP, int G, int LT, bool SP>
static int launch_fused_sp(mz_engine *e, int num_simulations, int sims_done, hipStream_t s) {
  const size_t dyn = mz_fused_dyn_lds(e->sims, e->NN, LT);
  const MzRootArgs ra0 = {nullptr, 0, 1, 0.0, 0.0};
  if constexpr (LT != 0 && SP) {
    if (e->persist_moves > 0) {      // the self-play loop: persist_moves whole moves in this launch
      const size_t dynh = fused_head_dyn_lds(e->sims, e->NN, LT);
      const MzRootArgs ra = {e->istream, e->nst0, e->persist_moves, e->cfg.root_dirichlet_alpha,
                             e->cfg.root_exploration_fraction};
      if (!e->lds_attr_set_head) {
        HIPCHECK(hipFuncSetAttribute((const void *)k_search_fused<KS1, JTP, G, LT, false, SP, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(float) * mz_fused_lds_floats(LT)));
        e->lds_attr_set_head = true;
      }
      if (e->ev_start)
        hipExtLaunchKernelGGL((k_search_fused<KS1, JTP, G, LT, false, SP, true>), dim3(e->Bp / MZ_ROWS), dim3(256), dynh, s,
                              e->ev_start, e->ev_stop, 0, e->nv, e->tv, e->wstream, num_simulations, 0,
                              (unsigned long long *)nullptr, e->sp, 1, (uint64_t)e->cfg.seed, ra);
      else
        hipLaunchKernelGGL((k_search_fused<KS1, JTP, G, LT, false, SP, true>), dim3(e->Bp / MZ_ROWS), dim3(256), dynh, s,
                           e->nv, e->tv, e->wstream, num_simulations, 0, e->head_prof, e->sp, 1,
                           (uint64_t)e->cfg.seed, ra);
      HIPCHECK(hipGetLastError());
      return 0;
    }
  }
#ifndef MZ_DEV_ONLY
  if constexpr (KS1 == 15 && JTP == 1 && G == 16 && LT == 2 && !SP) {
    if (e->persist_moves > 0 && e->sp.env_kind == 1) {      // whole moves of the device TicTacToe environment in this launch
      const size_t dynh = fused_head_dyn_lds(e->sims, e->NN, LT);
      const MzRootArgs ra = {e->istream, e->nst0, e->persist_moves, e->cfg.root_dirichlet_alpha,
                             e->cfg.root_exploration_fraction};
      if (!e->lds_attr_set_head) {
        HIPCHECK(hipFuncSetAttribute((const void *)k_search_fused<15, 1, 16, 2, false, false, true, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(float) * mz_fused_lds_floats(LT)));
        e->lds_attr_set_head = true;
      }
      if (e->ev_start)
        hipExtLaunchKernelGGL((k_search_fused<15, 1, 16, 2, false, false, true, true>), dim3(e->Bp / MZ_ROWS), dim3(256), dynh, s,
                              e->ev_start, e->ev_stop, 0, e->nv, e->tv, e->wstream, num_simulations, 0,
                              (unsigned long long *)nullptr, e->sp, 1, (uint64_t)e->cfg.seed, ra);
      else
        hipLaunchKernelGGL((k_search_fused<15, 1, 16, 2, false, false, true, true>), dim3(e->Bp / MZ_ROWS), dim3(256), dynh, s,
                           e->nv, e->tv, e->wstream, num_simulations, 0, e->head_prof, e->sp, 1,
                           (uint64_t)e->cfg.seed, ra);
      HIPCHECK(hipGetLastError());
      return 0;
    }
  }
#endif
  if (e->persist_moves > 0) return fail("internal: no HEAD instantiation for this configuration");
  if (e->prof_buf) {
    if (!e->lds_attr_set_prof) {
      HIPCHECK(hipFuncSetAttribute((const void *)k_search_fused<KS1, JTP, G, LT, true, SP>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(float) * mz_fused_lds_floats(LT)));
      e->lds_attr_set_prof = true;
    }
    hipLaunchKernelGGL((k_search_fused<KS1, JTP, G, LT, true, SP>), dim3(e->Bp / MZ_ROWS), dim3(256), dyn, s, e->nv,
                       e->tv, e->wstream, num_simulations, sims_done, e->prof_buf, e->sp, 0, (uint64_t)e->cfg.seed, ra0);
  } else {
    if (!e->lds_attr_set) {
      HIPCHECK(hipFuncSetAttribute((const void *)k_search_fused<KS1, JTP, G, LT, false, SP>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(float) * mz_fused_lds_floats(LT)));
      e->lds_attr_set = true;
    }
    if (e->ev_start)      // timestamps of the dispatch itself (what rocprofv3's kernel trace reports), no launch gap inside
      hipExtLaunchKernelGGL((k_search_fused<KS1, JTP, G, LT, false, SP>), dim3(e->Bp / MZ_ROWS), dim3(256), dyn, s,
                            e->ev_start, e->ev_stop, 0, e->nv, e->tv, e->wstream, num_simulations, sims_done,
                            (unsigned long long *)nullptr, e->sp, e->fuse_record ? 1 : 0, (uint64_t)e->cfg.seed, ra0);
    else
    hipLaunchKernelGGL((k_search_fused<KS1, JTP, G, LT, false, SP>), dim3(e->Bp / MZ_ROWS), dim3(256), dyn, s, e->nv,
                       e->tv, e->wstream, num_simulations, sims_done, (unsigned long long *)nullptr, e->sp,
                       e->fuse_record ? 1 : 0, (uint64_t)e->cfg.seed, ra0);
  }
  HIPCHECK(hipGetLastError());
  return 0;
}

// the split-f16 variant (mz_fused_h2.hip.h): same arguments, its own weight stream and static LDS
template <int G, int LT, bool SP>
static int launch_h2_sp(mz_engine *e, int num_simulations, int sims_done, hipStream_t s) {
  const size_t dyn = mz_fused_dyn_lds(e->sims, e->NN, LT);
  const f32x4 *ws = (const f32x4 *)e->packed_h2;
  const int maxdyn = 160 * 1024 - (int)sizeof(float) * mz_h2_lds_floats(LT);
  const MzRootArgs ra0 = {nullptr, 0, 1, 0.0, 0.0};
  if constexpr (SP) {
    if (e->persist_moves > 0) {      // the self-play loop: persist_moves whole moves in this launch
      const size_t dynh = fused_head_dyn_lds(e->sims, e->NN, LT);
      const MzRootArgs ra = {e->istream, e->nst0, e->persist_moves, e->cfg.root_dirichlet_alpha,
                             e->cfg.root_exploration_fraction};
      if (!e->lds_attr_set_head) {
        HIPCHECK(hipFuncSetAttribute((const void *)k_search_h2<G, LT, false, SP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, maxdyn));
        e->lds_attr_set_head = true;
      }
      if (e->ev_start)
        hipExtLaunchKernelGGL((k_search_h2<G, LT, false, SP, true>), dim3(e->Bp / MZ_ROWS), dim3(256), dynh, s, e->ev_start,
                              e->ev_stop, 0, e->nv, e->tv, ws, num_simulations, 0, (unsigned long long *)nullptr, e->sp, 1,
                              (uint64_t)e->cfg.seed, ra);
      else
        hipLaunchKernelGGL((k_search_h2<G, LT, false, SP, true>), dim3(e->Bp / MZ_ROWS), dim3(256), dynh, s, e->nv, e->tv, ws,
                           num_simulations, 0, (unsigned long long *)nullptr, e->sp, 1, (uint64_t)e->cfg.seed, ra);
      HIPCHECK(hipGetLastError());
      return 0;
    }
  }
  if (e->persist_moves > 0) return fail("internal: no HEAD instantiation for this configuration");
  if (e->prof_buf) {
    if (!e->lds_attr_set_prof) {
      HIPCHECK(hipFuncSetAttribute((const void *)k_search_h2<G, LT, true, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, maxdyn));
      e->lds_attr_set_prof = true;
    }
    hipLaunchKernelGGL((k_search_h2<G, LT, true, SP>), dim3(e->Bp / MZ_ROWS), dim3(256), dyn, s, e->nv, e->tv, ws,
                       num_simulations, sims_done, e->prof_buf, e->sp, 0, (uint64_t)e->cfg.seed, ra0);
  } else {
    if (!e->lds_attr_set) {
      HIPCHECK(hipFuncSetAttribute((const void *)k_search_h2<G, LT, false, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, maxdyn));
      e->lds_attr_set = true;
    }
    if (e->ev_start)
      hipExtLaunchKernelGGL((k_search_h2<G, LT, false, SP>), dim3(e->Bp / MZ_ROWS), dim3(256), dyn, s, e->ev_start,
                            e->ev_stop, 0, e->nv, e->tv, ws, num_simulations, sims_done, (unsigned long long *)nullptr,
                            e->sp, e->fuse_record ? 1 : 0, (uint64_t)e->cfg.seed, ra0);
    else
      hipLaunchKernelGGL((k_search_h2<G, LT, false, SP>), dim3(e->Bp / MZ_ROWS), dim3(256), dyn, s, e->nv, e->tv, ws,
                         num_simulations, sims_done, (unsigned long long *)nullptr, e->sp, e->fuse_record ? 1 : 0,
                         (uint64_t)e->cfg.seed, ra0);
  }
  HIPCHECK(hipGetLastError());
  return 0;
}

// -1: the split-f16 kernel does not apply (trees fit neither LDS placement beside its larger static LDS)
template <int G>
static int launch_h2(mz_engine *e, int num_simulations, int sims_done, hipStream_t s) {
  if (!e->use_lds_trees) return -2;
  const bool two = e->cfg.two_players != 0;
  if (sizeof(float) * mz_h2_lds_floats(1) + mz_fused_dyn_lds(e->sims, e->NN, 1) <= 160 * 1024)
    return two ? launch_h2_sp<G, 1, false>(e, num_simulations, sims_done, s) : launch_h2_sp<G, 1, true>(e, num_simulations, sims_done, s);
  if (e->use_lds_hybrid && sizeof(float) * mz_h2_lds_floats(2) + mz_fused_dyn_lds(e->sims, e->NN, 2) <= 160 * 1024)
    return two ? launch_h2_sp<G, 2, false>(e, num_simulations, sims_done, s) : launch_h2_sp<G, 2, true>(e, num_simulations, sims_done, s);
  return -2;
}

// single-player games (every reference environment but TicTacToe) run the instantiation without to_play handling
template <int KS1, int JTP, int G, int LT>
static int launch_fused_lt(mz_engine *e, int num_simulations, int sims_done, hipStream_t s) {
  return e->cfg.two_players ? launch_fused_sp<KS1, JTP, G, LT, false>(e, num_simulations, sims_done, s)
                            : launch_fused_sp<KS1, JTP, G, LT, true>(e, num_simulations, sims_done, s);
}

// trees in LDS when 16 of them fit beside the kernel's static LDS (160 KiB per CU); if not, at least the fields the
// descent reads (N, E, P, reward + discount * Q); else in the global pool
template <int KS1, int JTP, int G>
static int launch_fused_t(mz_engine *e, int num_simulations, int sims_done, hipStream_t s) {
  if (e->use_lds_trees) {
    if (sizeof(float) * mz_fused_lds_floats(1) + mz_fused_dyn_lds(e->sims, e->NN, 1) <= 160 * 1024)
      return launch_fused_lt<KS1, JTP, G, 1>(e, num_simulations, sims_done, s);
    if (e->use_lds_hybrid && sizeof(float) * mz_fused_lds_floats(2) + mz_fused_dyn_lds(e->sims, e->NN, 2) <= 160 * 1024)
      return launch_fused_lt<KS1, JTP, G, 2>(e, num_simulations, sims_done, s);
  }
  return launch_fused_lt<KS1, JTP, G, 0>(e, num_simulations, sims_done, s);
}

// fused-kernel instantiations: (fc1 k-steps, policy tiles, lanes per tree) by action count
// (dynamics fc1: K = 50 + A columns in k-steps of 4 -- the bias rides in the one-hot columns, fill_fc1_foldbias)
static int fused_ks1(int A) { return A <= 6 ? 14 : (A <= 10 ? 15 : (A <= 13 ? 16 : (A <= 21 ? 18 : 21))); }

static int launch_fused(mz_engine *e, int num_simulations, int sims_done, hipStream_t s) {
  const int A = e->A;
  if (e->split_f16) {
    int rc = -2;
    if (A <= 4) rc = launch_h2<4>(e, num_simulations, sims_done, s);
    else if (A <= 8) rc = launch_h2<8>(e, num_simulations, sims_done, s);
#ifndef MZ_DEV_ONLY
    else rc = launch_h2<16>(e, num_simulations, sims_done, s);
#endif
    if (rc != -2) return rc;          // (-2: trees do not fit its LDS budget -> the exact-f32 kernel)
  }
#ifdef MZ_DEV_ONLY      // kernel development: only the two bench shapes are instantiated (a quarter of the build time)
  if (A <= 4) return launch_fused_t<14, 1, 4>(e, num_simulations, sims_done, s);
  if (A >= 5 && A <= 6) return launch_fused_t<14, 1, 8>(e, num_simulations, sims_done, s);
  return fail("MZ_DEV_ONLY build: action_space %d not instantiated", A);
#else
  if (A <= 4) return launch_fused_t<14, 1, 4>(e, num_simulations, sims_done, s);
  if (A <= 6) return launch_fused_t<14, 1, 8>(e, num_simulations, sims_done, s);
  if (A <= 8) return launch_fused_t<15, 1, 8>(e, num_simulations, sims_done, s);
  if (A <= 10) return launch_fused_t<15, 1, 16>(e, num_simulations, sims_done, s);
  if (A <= 13) return launch_fused_t<16, 1, 16>(e, num_simulations, sims_done, s);
  if (A <= 16) return launch_fused_t<18, 1, 16>(e, num_simulations, sims_done, s);
  if (A <= 21) return launch_fused_t<18, 2, 32>(e, num_simulations, sims_done, s);
  return launch_fused_t<21, 2, 32>(e, num_simulations, sims_done, s);
#endif
}

// LDS placement of the trees the exact-f32 fused kernel will choose (launch_fused_t)
static int fused_lt(const mz_engine *e) {
  if (e->use_lds_trees) {
    if (sizeof(float) * mz_fused_lds_floats(1) + mz_fused_dyn_lds(e->sims, e->NN, 1) <= 160 * 1024) return 1;
    if (e->use_lds_hybrid && sizeof(float) * mz_fused_lds_floats(2) + mz_fused_dyn_lds(e->sims, e->NN, 2) <= 160 * 1024) return 2;
  }
  return 0;
}
// Can the self-play loop run as whole moves inside one launch (HEAD instantiation of the fused kernel)?
static bool selfplay_persist_ok(const mz_engine *e) {
  if (!e->use_persist || !fused_usable(e) || e->prof_buf) return false;
  if (e->sp.env_kind == 1) {
    // the device TicTacToe environment: whole moves inside the launch of the two-player <15, 1, 16> instantiation with
    // its trees compact in LDS (the exact-f32 kernel; host-given uniforms / draws work there too)
#ifdef MZ_DEV_ONLY
    return false;
#else
    return !e->split_f16 && e->cfg.two_players && e->A > 8 && e->A <= 10 && fused_lt(e) == 2 &&
           sizeof(float) * mz_fused_lds_floats(2) + fused_head_dyn_lds(e->sims, e->NN, 2) <= 160 * 1024;
#endif
  }
  if (e->cfg.two_players || e->sp.env_kind) return false;
  if (e->split_f16 && e->use_lds_trees) {      // the split-f16 kernel where it applies (launch_h2), else the exact one below
    for (int lt = 1; lt <= (e->use_lds_hybrid ? 2 : 1); ++lt)
      if (sizeof(float) * mz_h2_lds_floats(lt) + mz_fused_dyn_lds(e->sims, e->NN, lt) <= 160 * 1024)
        return sizeof(float) * mz_h2_lds_floats(lt) + fused_head_dyn_lds(e->sims, e->NN, lt) <= 160 * 1024;
  }
  const int lt = fused_lt(e);
  return lt != 0 && sizeof(float) * mz_fused_lds_floats(lt) + fused_head_dyn_lds(e->sims, e->NN, lt) <= 160 * 1024;
}

static int launch_root_priors(mz_engine *e, const int8_t *to_play, const uint8_t *legal, const double *priors,
                              hipStream_t s) {
  TREE_LAUNCH(k_tree_root_priors, s, e->tv, to_play, legal, priors);
  HIPCHECK(hipGetLastError());
  return 0;
}

static int launch_search(mz_engine *e, int num_simulations, bool selection_valid, int sims_done, hipStream_t s) {
  if (fused_usable(e) && !e->root_hidden_external) {
    if (!selection_valid) TREE_LAUNCH(k_tree_select, s, e->tv);
    return launch_fused(e, num_simulations, sims_done, s);
  }
  for (int i = 0; i < num_simulations; ++i) {
    if (!selection_valid) TREE_LAUNCH(k_tree_select, s, e->tv);
    NET_LAUNCH(k_net_recurrent_tree, e->Bp / MZ_ROWS, s, e->nv, e->tv, sims_done + i + 1);
    const int more = (i + 1 < num_simulations) ? 1 : 0;
    TREE_LAUNCH(k_tree_step, s, e->tv, more);
    selection_valid = more;
  }
  HIPCHECK(hipGetLastError());
  return 0;
}


// ---------------------------------------------------------------- ABI
extern "C" {

const char *mz_last_error(void) { return g_err.c_str(); }
int mz_version(void) { return 1; }

int mz_create(const mz_config *cfg, mz_engine **out) {
  if (!cfg || !out) return fail("mz_create: null argument");
  *out = nullptr;
  if (cfg->num_envs <= 0) return fail("mz_create: num_envs must be > 0");
  if (cfg->action_space < 1 || cfg->action_space > MZ_MAX_ACTIONS)
    return fail("mz_create: action_space %d outside [1,%d]", cfg->action_space, MZ_MAX_ACTIONS);
  if (cfg->obs_dim < 1) return fail("mz_create: obs_dim must be >= 1");
  if (cfg->num_simulations < 1 || cfg->num_simulations > 4096) return fail("mz_create: num_simulations out of range");
  const int Sv = cfg->no_support ? 1 : cfg->value_support_max - cfg->value_support_min + 1;
  const int Sr = cfg->no_support ? 1 : cfg->reward_support_max - cfg->reward_support_min + 1;
  if (Sv < 1 || Sv > MZ_MAX_SUPPORT || Sr < 1 || Sr > MZ_MAX_SUPPORT)
    return fail("mz_create: support size must be in [1,%d] (value %d, reward %d)", MZ_MAX_SUPPORT, Sv, Sr);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail("mz_create: no HIP device visible (this engine has no CPU path)");
  mz_engine *e = new mz_engine();
  if (hipGetDevice(&e->device) != hipSuccess) { delete e; return fail("mz_create: hipGetDevice failed"); }
  e->cfg = *cfg;
  e->B = cfg->num_envs;
  e->Bp = (cfg->num_envs + MZ_ROWS - 1) / MZ_ROWS * MZ_ROWS;
  e->A = cfg->action_space;
  e->O = cfg->obs_dim;
  e->sims = cfg->num_simulations;
  e->NN = 1 + (e->sims + 1) * e->A;
  e->PL = e->sims + 2;
  e->G = e->A <= 4 ? 4 : (e->A <= 8 ? 8 : (e->A <= 16 ? 16 : 32));
  e->jtp = e->A <= 16 ? 1 : 2;
  e->ks1sel = fused_ks1(e->A);
  e->use_graph = getenv("MZ_NO_GRAPH") == nullptr;
  e->use_fused = getenv("MZ_NO_FUSED") == nullptr;
  e->use_persist = getenv("MZ_NO_PERSIST") == nullptr;
  e->use_lds_trees = getenv("MZ_NO_LDS_TREES") == nullptr;
  e->use_lds_hybrid = getenv("MZ_NO_LDS_HYBRID") == nullptr;
  {
    const char *sf = getenv("MZ_SPLIT_F16");
    const bool want = cfg->split_f16 != 0 || (sf && sf[0] == '1');
    if (cfg->split_f16 != 0 && cfg->action_space > MZ_H2_MAXA) {
      delete e;
      return fail("mz_create: split_f16 supports action_space <= %d (got %d)", MZ_H2_MAXA, cfg->action_space);
    }
    e->split_f16 = want && cfg->action_space <= MZ_H2_MAXA;
  }
  TreeView &t = e->tv;
  memset(&t, 0, sizeof t);
  const size_t nb = (size_t)e->Bp;
#define DM(p, n) if (dmalloc(e, &(p), (n))) { mz_destroy(e); return -1; }
  // the node pool: 32-byte records, NS per tree (mz_common.h: MzNode)
  e->NS = (e->NN + MZ_NODE_OFF + 3) & ~3;
  MzNode *nodes;
  DM(nodes, nb * e->NS)
  e->nodes = nodes;
  t.N.base = t.W.base = t.P.base = t.R.base = t.E.base = t.TP.base = nodes;
  t.NS = e->NS;
  DM(t.legal, nb) DM(t.mn, nb) DM(t.mx, nb) DM(t.nexp, nb) DM(t.path, nb * e->PL) DM(t.plen, nb) DM(t.leaf_tp, nb)
  DM(t.leaf, nb) DM(t.slot, nb) DM(t.act, nb) DM(t.depth, nb)
  DM(t.hpool, nb * (e->sims + 1) * MZ_HS)
  DM(t.value, nb) DM(t.reward, nb) DM(t.logits, nb * e->A) DM(t.root_value, nb) DM(t.root_logits, nb * e->A)
  DM(t.noise, nb * e->A)
  double *sqrttab, *pbctab, *logtab;
  DM(sqrttab, e->sims + 2) DM(pbctab, (size_t)(e->sims + 2) * (e->sims + 2)) DM(logtab, e->sims + 2)
#undef DM
  {
    // mcts.py:116-117: math.log((N + base + 1) / base) + init and math.sqrt(N), for every N a parent
    // can have inside one search -- libm on the host, the values CPython computes.
    std::vector<double> lt(e->sims + 2), st(e->sims + 2);
    for (int i = 0; i < e->sims + 2; ++i) {
      lt[i] = log(((double)i + cfg->pb_c_base + 1) / cfg->pb_c_base) + cfg->pb_c_init;
      st[i] = sqrt((double)i);
    }
    // pb_c for every (parent visits, child visits) pair, in the reference's operation order
    // (pb_c = log(..) + init; pb_c *= sqrt(Np) / (Nc + 1)); IEEE double on the host, no contraction
    const int T = e->sims + 2;
    std::vector<double> pt((size_t)T * T);
    for (int np = 0; np < T; ++np)
      for (int nc = 0; nc < T; ++nc) {
        volatile double q = st[np] / (double)(nc + 1);
        volatile double v = lt[np] * q;
        pt[(size_t)np * T + nc] = v;
      }
    if (hipMemcpy(pbctab, pt.data(), pt.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(logtab, lt.data(), lt.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(sqrttab, st.data(), st.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
      mz_destroy(e);
      return fail("mz_create: table upload failed");
    }
  }
  t.sqrttab = sqrttab; t.pbctab = pbctab; t.logtab = logtab;
  t.B = e->B; t.A = e->A; t.sims = e->sims; t.NN = e->NN; t.PL = e->PL;
  t.two_players = cfg->two_players; t.has_min = cfg->has_min_bound; t.has_max = cfg->has_max_bound;
  t.min_bound = cfg->min_bound; t.max_bound = cfg->max_bound; t.discount = cfg->discount;
  t.init_value_score = cfg->init_value_score;
  if (build_packing(e)) { mz_destroy(e); return -1; }
  if (hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) {
    mz_destroy(e);
    return fail("mz_create: stream creation failed");
  }
  memset(&e->sp, 0, sizeof e->sp);
  *out = e;
  return 0;
}

int mz_destroy(mz_engine *e) {
  if (!e) return 0;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  for (auto &q : e->drain_q) hipEventDestroy(q.second);
  for (hipEvent_t ev : e->drain_ev_pool) hipEventDestroy(ev);
  if (e->search_graph) hipGraphExecDestroy(e->search_graph);
  for (auto &g : e->move_graph) if (g) hipGraphExecDestroy(g);
  if (e->cap_stream) hipStreamDestroy(e->cap_stream);
  for (void *p : e->allocs) hipFree(p);
  for (int w = 0; w < 2; ++w) {
    if (e->wstage[w]) hipHostFree(e->wstage[w]);
    if (e->wstage_ev[w]) hipEventDestroy(e->wstage_ev[w]);
  }
  delete e;
  return 0;
}

int mz_affine_relu(float *y, const float *scale, const float *shift, const float *residual, size_t n, int channels,
                   int hw, void *stream) {
  if (!y || !scale || !shift) return fail("mz_affine_relu: null argument");
  if (channels < 1 || hw < 4 || (hw & 3) || n % ((size_t)channels * hw)) return fail("mz_affine_relu: n %zu is not [N][%d][%d] with hw %% 4 == 0", n, channels, hw);
  if (((uintptr_t)y | (uintptr_t)residual) & 15) return fail("mz_affine_relu: tensors must be 16-byte aligned");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (residual)
    hipLaunchKernelGGL(k_affine_relu<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)y, scale, shift,
                       (const float4 *)residual, n4, channels, hw / 4);
  else
    hipLaunchKernelGGL(k_affine_relu<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)y, scale, shift,
                       (const float4 *)nullptr, n4, channels, hw / 4);
  HIPCHECK(hipGetLastError());
  return 0;
}

// ---- the elementwise ends of the learner step (mz_learner.hip.h); engine-free: device pointers and sizes only
int mz_learner_targets(const float *t_val, const float *t_rew, const float *value0, int bs, int k1, int sv, int vmin, int sr,
                       int rmin, int no_target_transform, float *sup_val, float *sup_rew, float *new_errors, void *stream) {
  if (!t_val || !t_rew || !value0 || !sup_val || !sup_rew || !new_errors) return fail("mz_learner_targets: null argument");
  if (bs < 1 || k1 < 1 || sv < 1 || sr < 1 || sv > 4096 || sr > 4096) return fail("mz_learner_targets: bad sizes");
  // (bs * k1 threads for the supports, then -- from the next 32-lane boundary on -- 32 lanes per sample for the priority refresh)
  hipLaunchKernelGGL(k_learner_targets, dim3((((bs * k1 + 31) & ~31) + 32 * bs + 127) / 128), dim3(128), 0, (hipStream_t)stream, t_val, t_rew, value0, bs,
                     k1, sv, vmin, sr, rmin, no_target_transform, sup_val, sup_rew, new_errors);
  HIPCHECK(hipGetLastError());
  return 0;
}

int mz_soft_ce_forward(const float *logits, const float *target, int positions, int bs, int bins, int64_t target_pos_stride,
                       int64_t target_row_stride, float *loss, void *stream) {
  if (!logits || !target || !loss) return fail("mz_soft_ce_forward: null argument");
  if (positions < 1 || positions > 8) return fail("mz_soft_ce_forward: 1..8 positions (num_unroll_steps + 1), got %d", positions);
  hipLaunchKernelGGL(k_soft_ce_fwd, dim3((8 * bs + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, target, positions, bs, bins,
                     target_pos_stride, target_row_stride, loss);
  HIPCHECK(hipGetLastError());
  return 0;
}

int mz_soft_ce_backward(const float *logits, const float *target, const float *grad_loss, int positions, int bs, int bins,
                        int64_t target_pos_stride, int64_t target_row_stride, float *grad_logits, void *stream) {
  if (!logits || !target || !grad_loss || !grad_logits) return fail("mz_soft_ce_backward: null argument");
  hipLaunchKernelGGL(k_soft_ce_bwd, dim3((8 * positions * bs + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, target, grad_loss,
                     positions, bs, bins, target_pos_stride, target_row_stride, grad_logits);
  HIPCHECK(hipGetLastError());
  return 0;
}

size_t mz_num_weights(const mz_engine *e) { return e ? e->n_flat : 0; }
int mz_nodes_per_tree(const mz_engine *e) { return e ? e->NN : -1; }
int mz_padded_envs(const mz_engine *e) { return e ? e->Bp : -1; }

int mz_set_weights(mz_engine *e, const float *flat, size_t n, int on_device, void *stream) {
  if (!e || !flat) return fail("mz_set_weights: null argument");
  MZ_ENTER(e);
  if (n != e->n_flat) return fail("mz_set_weights: expected %zu floats, got %zu", e->n_flat, n);
  hipStream_t s = (hipStream_t)stream;
  const float *src = flat;
  if (!on_device) {
    HIPCHECK(hipMemcpyAsync(e->flat_dev, flat, n * sizeof(float), hipMemcpyHostToDevice, s));
    src = e->flat_dev;
  }
  const int threads = 256;
  const unsigned blocks = (unsigned)((e->n_packed + threads - 1) / threads);
  if (e->split_f16) {
    // the high part of a split weight is its float16 rounding: beyond the float16 range it would be inf and poison every
    // search silently.  Checked here, loudly (one small reduction + a 4-byte read back; split_f16 is opt-in).
    if (!e->absmax_dev && dmalloc(e, &e->absmax_dev, (size_t)1)) return -1;
    HIPCHECK(hipMemsetAsync(e->absmax_dev, 0, 4, s));
    hipLaunchKernelGGL(k_absmax, dim3(64), dim3(256), 0, s, src, n, e->absmax_dev);
    unsigned bits = 0;
    HIPCHECK(hipMemcpyAsync(&bits, e->absmax_dev, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    float mx;
    memcpy(&mx, &bits, 4);
    if (!(mx <= 65504.f))
      return fail("mz_set_weights: split_f16 needs every weight finite and |w| <= 65504 (the float16 range of the high part); "
                  "max |w| = %g.  Use the exact-float32 kernel (split_f16 = 0) for these weights", (double)mx);
  }
  // the power of two the search kernel's stream is scaled by (its ReLU is a [0, 1] clamp, mz_fused.hip.h), from a bound of
  // this weight set's activations; a 16-byte read back tells the host whether one exists -- if not, the engine runs
  // the stand-alone kernels until the next weight set
  HIPCHECK(hipMemsetAsync(e->relu_bound_dev, 0, 2 * sizeof(unsigned), s));
  hipLaunchKernelGGL(k_relu_bound, dim3(64), dim3(256), 0, s, src, e->layout, e->A, e->nv.Sr, e->nv.Sv, e->relu_bound_dev);
  hipLaunchKernelGGL(k_relu_scale, dim3(1), dim3(1), 0, s, (const unsigned *)e->relu_bound_dev, e->relu_scale_dev, -1);
  hipLaunchKernelGGL(k_pack_weights, dim3(blocks), dim3(threads), 0, s, src, e->pack_idx, (const int32_t *)e->pack_idx2,
                     e->packed, e->n_packed, (const float *)e->relu_scale_dev);
  HIPCHECK(hipMemcpyAsync(e->relu_scale_host, e->relu_scale_dev, 4 * sizeof(float), hipMemcpyDeviceToHost, s));
  if (e->split_f16)
    hipLaunchKernelGGL(k_pack_weights_h2, dim3((unsigned)((e->n_packed_h2 + threads - 1) / threads)), dim3(threads), 0, s, src,
                       e->pack_idx_h2, (_Float16 *)e->packed_h2, e->n_packed_h2);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(s));     // (also: a host buffer may be freed by the caller)
  const bool scaled = e->relu_scale_host[3] != 0.f;
  if (scaled != e->stream_scaled) {      // captured graphs hold the launches of the other kernel set
    if (e->search_graph) { hipGraphExecDestroy(e->search_graph); e->search_graph = nullptr; }
    for (auto &g : e->move_graph) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  }
  e->stream_scaled = scaled;
  e->scale_host_valid = true;
  e->weights_set = true;
  return 0;
}

// The host's side of k_relu_bound / k_relu_scale: the same bound in double from a HOST copy of the weights, with tighter
// limits (a factor 2 on the activation bound, 4 on the consuming layers' largest weight), so that "1" here implies the
// device's own test passes whatever the float32 rounding of its sums.
int mz_weights_scale_ok(const float *flat, size_t n, int obs_dim, int action_space, int value_outputs, int reward_outputs) {
  if (!flat) return fail("mz_weights_scale_ok: null argument");
  if (obs_dim < 1 || action_space < 1 || value_outputs < 1 || reward_outputs < 1) return fail("mz_weights_scale_ok: bad shape");
  const FlatLayout L = flat_layout(obs_dim, action_space, value_outputs, reward_outputs);
  if (n != L.total) return fail("mz_weights_scale_ok: expected %zu floats, got %zu", L.total, n);
  const int A = action_space;
  double hb[MZ_H];
  for (int i = 0; i < MZ_H; ++i) hb[i] = 7.01 * fabs((double)flat[L.ln_w + i]) + fabs((double)flat[L.ln_b + i]);
  const size_t w1[4] = {L.rew_w1, L.tr_w1, L.val_w1, L.pol_w1}, b1[4] = {L.rew_b1, L.tr_b1, L.val_b1, L.pol_b1};
  double bound = 0.0;
  bool nan = false;
  for (int h = 0; h < 4; ++h) {
    const int K = h < 2 ? MZ_H + A : MZ_H;
    for (int r = 0; r < MZ_F; ++r) {
      const float *row = flat + w1[h] + (size_t)r * K;
      double s = 0.0, oh = 0.0;
      for (int i = 0; i < MZ_H; ++i) s += fabs((double)row[i]) * hb[i];
      for (int a = MZ_H; a < K; ++a) { const double v = fabs((double)row[a]); if (v != v) nan = true; if (v > oh) oh = v; }
      s += oh + fabs((double)flat[b1[h] + r]);
      if (s != s) nan = true;
      if (s > bound) bound = s;
    }
  }
  const size_t w2[4] = {L.rew_w2, L.tr_w2, L.val_w2, L.pol_w2};
  const int J[4] = {reward_outputs, MZ_H, value_outputs, A};
  double w2max = 0.0;
  for (int h = 0; h < 4; ++h)
    for (size_t i = 0; i < (size_t)J[h] * MZ_F; ++i) {
      const double v = fabs((double)flat[w2[h] + i]);
      if (v != v) nan = true;
      if (v > w2max) w2max = v;
    }
  bound *= 1.02;
  int k = 0;
  if (bound > 1.0) (void)frexp(bound, &k);
  return (!nan && bound < 0x1p39 && w2max < ldexp(1.0, 97 - k)) ? 1 : 0;
}

// mz_set_weights without the host waiting for anything queued on `stream`: the repack runs in stream order (the moves
// queued before it keep the old weights, everything queued after it sees the new ones), which kernel set the new weights
// run on is the caller's `scale_ok` (mz_weights_scale_ok on a host copy of the SAME weights).
int mz_set_weights_async(mz_engine *e, const float *flat, size_t n, int on_device, int scale_ok, void *stream) {
  if (!e || !flat) return fail("mz_set_weights_async: null argument");
  MZ_ENTER(e);
  if (n != e->n_flat) return fail("mz_set_weights_async: expected %zu floats, got %zu", e->n_flat, n);
  if (e->split_f16) return mz_set_weights(e, flat, n, on_device, stream);      // (its float16 range check reads back: synchronous)
  hipStream_t s = (hipStream_t)stream;
  const float *src = flat;
  if (!on_device) {
    const int w = e->wstage_next;
    e->wstage_next ^= 1;
    if (!e->wstage[w]) {
      HIPCHECK(hipHostMalloc((void **)&e->wstage[w], n * sizeof(float), hipHostMallocDefault));
      HIPCHECK(hipEventCreateWithFlags(&e->wstage_ev[w], hipEventDisableTiming));
    } else {
      HIPCHECK(hipEventSynchronize(e->wstage_ev[w]));      // the copy out of this slot, two pulls ago (long complete)
    }
    memcpy(e->wstage[w], flat, n * sizeof(float));
    HIPCHECK(hipMemcpyAsync(e->flat_dev, e->wstage[w], n * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHECK(hipEventRecord(e->wstage_ev[w], s));
    src = e->flat_dev;
  }
  const int threads = 256;
  const unsigned blocks = (unsigned)((e->n_packed + threads - 1) / threads);
  HIPCHECK(hipMemsetAsync(e->relu_bound_dev, 0, 2 * sizeof(unsigned), s));
  hipLaunchKernelGGL(k_relu_bound, dim3(64), dim3(256), 0, s, src, e->layout, e->A, e->nv.Sr, e->nv.Sv, e->relu_bound_dev);
  hipLaunchKernelGGL(k_relu_scale, dim3(1), dim3(1), 0, s, (const unsigned *)e->relu_bound_dev, e->relu_scale_dev, scale_ok ? 1 : 0);
  hipLaunchKernelGGL(k_pack_weights, dim3(blocks), dim3(threads), 0, s, src, e->pack_idx, (const int32_t *)e->pack_idx2,
                     e->packed, e->n_packed, (const float *)e->relu_scale_dev);
  HIPCHECK(hipGetLastError());
  const bool scaled = scale_ok != 0;
  if (scaled != e->stream_scaled) {      // (rare: captured graphs hold the launches of the other kernel set)
    HIPCHECK(hipStreamSynchronize(s));
    if (e->search_graph) { hipGraphExecDestroy(e->search_graph); e->search_graph = nullptr; }
    for (auto &g : e->move_graph) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  }
  e->stream_scaled = scaled;
  e->scale_host_valid = false;
  e->weights_set = true;
  return 0;
}

int mz_weight_scale(mz_engine *e, float *out, void *stream) {
  if (!e || !out) return fail("mz_weight_scale: null argument");
  MZ_ENTER(e);
  if (!e->weights_set) return fail("mz_weight_scale: weights not set (call mz_set_weights)");
  if (!e->scale_host_valid) {
    HIPCHECK(hipMemcpyAsync(e->relu_scale_host, e->relu_scale_dev, 4 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    e->scale_host_valid = true;
  }
  memcpy(out, e->relu_scale_host, sizeof(e->relu_scale_host));
  return 0;
}

#include "mz_comm.inc"

int mz_initial_inference(mz_engine *e, const float *obs, void *stream) {
  if (!e || !obs) return fail("mz_initial_inference: null argument");
  MZ_ENTER(e);
  if (!e->weights_set) return fail("mz_initial_inference: weights not set (call mz_set_weights)");
  hipStream_t s = (hipStream_t)stream;
  if (launch_root(e, obs, false, s)) return -1;
  if (e->root_hidden_external && e->search_graph) { hipGraphExecDestroy(e->search_graph); e->search_graph = nullptr; }
  e->root_hidden_external = false;
  e->root_ready = false;
  return 0;
}

int mz_root_load(mz_engine *e, const float *hidden, const float *value, const float *logits, void *stream) {
  if (!e || !value || !logits) return fail("mz_root_load: null argument");
  MZ_ENTER(e);
  hipStream_t s = (hipStream_t)stream;
  HIPCHECK(hipMemcpyAsync(e->tv.root_value, value, (size_t)e->B * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIPCHECK(hipMemcpyAsync(e->tv.root_logits, logits, (size_t)e->B * e->A * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (hidden) {
    const int n = e->B * MZ_HS;
    hipLaunchKernelGGL(k_scatter_hidden, dim3((n + 255) / 256), dim3(256), 0, s, e->tv, hidden, 1);
    HIPCHECK(hipGetLastError());
    // a hidden state from elsewhere need not be a LayerNorm output: mz_search then runs the stand-alone kernels, whose
    // weights are not scaled for the clamp ReLU (k_relu_scale)
    if (!e->root_hidden_external && e->search_graph) { hipGraphExecDestroy(e->search_graph); e->search_graph = nullptr; }
    e->root_hidden_external = true;
  }
  e->root_ready = false;
  return 0;
}

__global__ void k_copy_root_hidden(TreeView t, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.B * MZ_H) return;
  const int b = i / MZ_H, k = i % MZ_H;
  out[i] = t.hpool[(size_t)b * (t.sims + 1) * MZ_HS + k];
}

int mz_root_outputs(mz_engine *e, float *value, float *logits, float *hidden, void *stream) {
  if (!e) return fail("mz_root_outputs: null engine");
  MZ_ENTER(e);
  hipStream_t s = (hipStream_t)stream;
  if (value) HIPCHECK(hipMemcpyAsync(value, e->tv.root_value, (size_t)e->B * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (logits)
    HIPCHECK(hipMemcpyAsync(logits, e->tv.root_logits, (size_t)e->B * e->A * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (hidden) {
    const int n = e->B * MZ_H;
    hipLaunchKernelGGL(k_copy_root_hidden, dim3((n + 255) / 256), dim3(256), 0, s, e->tv, hidden);
    HIPCHECK(hipGetLastError());
  }
  return 0;
}

int mz_root_prepare(mz_engine *e, const int8_t *to_play, const uint8_t *legal, const double *noise,
                    int use_device_rng, uint64_t move_counter, void *stream) {
  if (!e) return fail("mz_root_prepare: null engine");
  MZ_ENTER(e);
  hipStream_t s = (hipStream_t)stream;
  const double *nz = noise;
  if (!noise && use_device_rng) {
    const int threads = 128;
    hipLaunchKernelGGL(k_dirichlet, dim3((e->B + threads - 1) / threads), dim3(threads), 0, s, e->tv, legal,
                       e->cfg.root_dirichlet_alpha, e->cfg.seed, move_counter, (const unsigned long long *)nullptr,
                       e->cfg.env_id_offset);
    HIPCHECK(hipGetLastError());
    nz = e->tv.noise;
  } else if (noise) {
    HIPCHECK(hipMemcpyAsync(e->tv.noise, noise, (size_t)e->B * e->A * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  TREE_LAUNCH(k_tree_root, s, e->tv, to_play, legal, nz, e->cfg.root_exploration_fraction, 1);
  HIPCHECK(hipGetLastError());
  e->sims_done = 0;
  e->selection_valid = true;
  e->root_ready = true;
  return 0;
}

int mz_search(mz_engine *e, int num_simulations, void *stream) {
  if (!e) return fail("mz_search: null engine");
  MZ_ENTER(e);
  if (!e->weights_set) return fail("mz_search: weights not set (call mz_set_weights)");
  if (!e->root_ready) return fail("mz_search: call mz_root_prepare first");
  if (num_simulations < 1 || e->sims_done + num_simulations > e->sims)
    return fail("mz_search: %d + %d simulations exceed the pool sized for num_simulations = %d", e->sims_done,
                num_simulations, e->sims);
  hipStream_t s = (hipStream_t)stream;
  const bool graphable = e->use_graph && e->selection_valid && e->sims_done == 0;
  if (graphable) {
    if (!e->search_graph || e->search_graph_sims != num_simulations) {
      if (e->search_graph) { hipGraphExecDestroy(e->search_graph); e->search_graph = nullptr; }
      hipGraph_t g = nullptr;
      HIPCHECK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
      int rc = launch_search(e, num_simulations, true, 0, e->cap_stream);
      hipError_t ce = hipStreamEndCapture(e->cap_stream, &g);
      if (rc || ce != hipSuccess) return fail("mz_search: graph capture failed");
      HIPCHECK(hipGraphInstantiate(&e->search_graph, g, nullptr, nullptr, 0));
      hipGraphDestroy(g);
      e->search_graph_sims = num_simulations;
    }
    HIPCHECK(hipGraphLaunch(e->search_graph, s));
  } else {
    if (launch_search(e, num_simulations, e->selection_valid, e->sims_done, s)) return -1;
  }
  e->sims_done += num_simulations;
  e->selection_valid = false;
  return 0;
}

// which search kernel mz_search / mz_selfplay_steps will launch for this engine right now -- out4 [host]: kind (0 the
// stand-alone kernels, 1 k_search_fused, 2 k_search_h2), LDS placement of the trees (0 global pool, 1 whole trees, 2
// compact; -1 for kind 0), fc1 k-steps of the instantiation, lanes per child group.  (Tests assert the instantiation
// they mean to exercise.)
int mz_search_kernel_info(const mz_engine *e, int *out4) {
  if (!e || !out4) return fail("mz_search_kernel_info: null argument");
  int kind = 0, lt = -1;
  if (fused_usable(e) && !e->root_hidden_external) {
    kind = 1; lt = fused_lt(e);
    if (e->split_f16 && e->use_lds_trees)
      for (int l = 1; l <= (e->use_lds_hybrid ? 2 : 1); ++l)
        if (sizeof(float) * mz_h2_lds_floats(l) + mz_fused_dyn_lds(e->sims, e->NN, l) <= 160 * 1024) { kind = 2; lt = l; break; }
  }
  out4[0] = kind; out4[1] = lt; out4[2] = e->ks1sel; out4[3] = e->G;
  return 0;
}

int mz_root_set_priors(mz_engine *e, const int8_t *to_play, const uint8_t *legal, const double *priors,
                       void *stream) {
  if (!e || !priors) return fail("mz_root_set_priors: null argument");
  MZ_ENTER(e);
  if (launch_root_priors(e, to_play, legal, priors, (hipStream_t)stream)) return -1;
  e->sims_done = 0;
  e->selection_valid = true;
  e->root_ready = true;
  return 0;
}

int mz_last_paths(mz_engine *e, int32_t *paths, int32_t *lengths, void *stream) {
  if (!e) return fail("mz_last_paths: null engine");
  MZ_ENTER(e);
  if (!e->selection_valid) return fail("mz_last_paths: no pending descent (call mz_select first)");
  hipStream_t s = (hipStream_t)stream;
  if (paths) HIPCHECK(hipMemcpyAsync(paths, e->tv.path, (size_t)e->B * e->PL * 4, hipMemcpyDeviceToDevice, s));
  if (lengths) HIPCHECK(hipMemcpyAsync(lengths, e->tv.plen, (size_t)e->B * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

int mz_search_profiled(mz_engine *e, int num_simulations, float *ms_out, void *stream) {
  if (!e || !ms_out) return fail("mz_search_profiled: null argument");
  MZ_ENTER(e);
  if (!e->weights_set) return fail("mz_search_profiled: weights not set (call mz_set_weights)");
  if (!e->root_ready || !e->selection_valid || e->sims_done != 0)
    return fail("mz_search_profiled: call right after mz_root_prepare");
  if (num_simulations < 1 || num_simulations > e->sims) return fail("mz_search_profiled: bad num_simulations");
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(2 * num_simulations + 1);
  for (auto &x : ev) HIPCHECK(hipEventCreate(&x));
  HIPCHECK(hipEventRecord(ev[0], s));
  for (int i = 0; i < num_simulations; ++i) {
    NET_LAUNCH(k_net_recurrent_tree, e->Bp / MZ_ROWS, s, e->nv, e->tv, i + 1);
    HIPCHECK(hipEventRecord(ev[2 * i + 1], s));
    const int more = (i + 1 < num_simulations) ? 1 : 0;
    TREE_LAUNCH(k_tree_step, s, e->tv, more);
    HIPCHECK(hipEventRecord(ev[2 * i + 2], s));
  }
  HIPCHECK(hipStreamSynchronize(s));
  ms_out[0] = ms_out[1] = 0.f;
  for (int i = 0; i < num_simulations; ++i) {
    float a = 0.f, b = 0.f;
    HIPCHECK(hipEventElapsedTime(&a, ev[2 * i], ev[2 * i + 1]));
    HIPCHECK(hipEventElapsedTime(&b, ev[2 * i + 1], ev[2 * i + 2]));
    ms_out[0] += a; ms_out[1] += b;
  }
  for (auto &x : ev) hipEventDestroy(x);
  e->sims_done = num_simulations;
  e->selection_valid = false;
  return 0;
}

int mz_search_timed(mz_engine *e, int num_simulations, float *ms_out, void *stream) {
  if (!e || !ms_out) return fail("mz_search_timed: null argument");
  MZ_ENTER(e);
  if (!fused_usable(e) || e->root_hidden_external) return fail("mz_search_timed: fused kernel not in use");
  if (!e->weights_set) return fail("mz_search_timed: weights not set (call mz_set_weights)");
  if (!e->root_ready || !e->selection_valid || e->sims_done != 0)
    return fail("mz_search_timed: call right after mz_root_prepare");
  if (num_simulations < 1 || num_simulations > e->sims) return fail("mz_search_timed: bad num_simulations");
  hipStream_t s = (hipStream_t)stream;
  HIPCHECK(hipEventCreate(&e->ev_start));
  HIPCHECK(hipEventCreate(&e->ev_stop));
  const int rc = launch_fused(e, num_simulations, 0, s);
  hipError_t se = hipStreamSynchronize(s);
  float ms = 0.f;
  hipError_t te = (rc == 0 && se == hipSuccess) ? hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) : hipErrorUnknown;
  hipEventDestroy(e->ev_start); hipEventDestroy(e->ev_stop);
  e->ev_start = e->ev_stop = nullptr;
  if (rc) return -1;
  if (se != hipSuccess || te != hipSuccess) return fail("mz_search_timed: %s", hipGetErrorString(se != hipSuccess ? se : te));
  *ms_out = ms;
  e->sims_done = num_simulations;
  e->selection_valid = false;
  return 0;
}

int mz_search_phase_profile(mz_engine *e, int num_simulations, unsigned long long *cycles_out, void *stream) {
  if (!e || !cycles_out) return fail("mz_search_phase_profile: null argument");
  MZ_ENTER(e);
  if (!fused_usable(e) || e->root_hidden_external) return fail("mz_search_phase_profile: fused kernel not in use");
  if (!e->root_ready || !e->selection_valid || e->sims_done != 0)
    return fail("mz_search_phase_profile: call right after mz_root_prepare");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)(e->Bp / MZ_ROWS) * 4 * MZ_NPHASE;
  unsigned long long *buf = nullptr;
  HIPCHECK(hipMalloc((void **)&buf, n * 8));
  HIPCHECK(hipMemsetAsync(buf, 0, n * 8, s));
  e->prof_buf = buf;
  int rc = launch_fused(e, num_simulations, 0, s);
  e->prof_buf = nullptr;
  if (rc) { hipFree(buf); return -1; }
  HIPCHECK(hipStreamSynchronize(s));
  std::vector<unsigned long long> h(n);
  HIPCHECK(hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost));
  hipFree(buf);
  // spread of the workgroups' totals (wave 0): the launch lasts as long as its slowest workgroup
  {
    double mn = 1e300, mx = 0, sum = 0;
    for (int g = 0; g < e->Bp / MZ_ROWS; ++g) {
      double tot = 0;
      for (int p2 = 0; p2 < MZ_NPHASE; ++p2) tot += (double)h[((size_t)g * 4) * MZ_NPHASE + p2];
      mn = tot < mn ? tot : mn; mx = tot > mx ? tot : mx; sum += tot;
    }
    e->prof_spread[0] = sum / (e->Bp / MZ_ROWS); e->prof_spread[1] = mn; e->prof_spread[2] = mx;
  }
  // average over workgroups, per wave and phase: cycles_out[4][MZ_NPHASE]
  for (int w = 0; w < 4; ++w)
    for (int p = 0; p < MZ_NPHASE; ++p) {
      unsigned long long sum = 0;
      for (int g = 0; g < e->Bp / MZ_ROWS; ++g) sum += h[((size_t)g * 4 + w) * MZ_NPHASE + p];
      cycles_out[w * MZ_NPHASE + p] = sum / (unsigned long long)(e->Bp / MZ_ROWS);
    }
  e->sims_done = num_simulations;
  e->selection_valid = false;
  return 0;
}

int mz_search_phase_spread(const mz_engine *e, double *out3) {
  if (!e || !out3) return fail("mz_search_phase_spread: null argument");
  out3[0] = e->prof_spread[0]; out3[1] = e->prof_spread[1]; out3[2] = e->prof_spread[2];
  return 0;
}

int mz_select(mz_engine *e, int32_t *leaf_node, int32_t *parent_slot, int32_t *action, int32_t *depth, void *stream) {
  if (!e) return fail("mz_select: null engine");
  MZ_ENTER(e);
  if (!e->root_ready) return fail("mz_select: call mz_root_prepare first");
  if (e->sims_done >= e->sims) return fail("mz_select: all %d simulations already done", e->sims);
  hipStream_t s = (hipStream_t)stream;
  if (!e->selection_valid) {
    TREE_LAUNCH(k_tree_select, s, e->tv);
    HIPCHECK(hipGetLastError());
    e->selection_valid = true;
  }
  if (leaf_node || parent_slot || action || depth) {      // one launch (four D2D blits cost 4 x 4 us per simulation)
    hipLaunchKernelGGL(k_export_selection, dim3((e->B + 255) / 256), dim3(256), 0, s, e->tv, leaf_node, parent_slot, action, depth);
    HIPCHECK(hipGetLastError());
  }
  return 0;
}

int mz_tree_pair_timed(mz_engine *e, const float *value, const float *reward, const float *logits, float *ms_out, void *stream) {
  if (!e || !value || !reward || !logits || !ms_out) return fail("mz_tree_pair_timed: null argument");
  MZ_ENTER(e);
  if (!e->root_ready) return fail("mz_tree_pair_timed: call mz_root_prepare first");
  if (e->sims_done >= e->sims) return fail("mz_tree_pair_timed: all %d simulations already done", e->sims);
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  for (auto &x : ev) HIPCHECK(hipEventCreate(&x));
  const bool pending = e->selection_valid;      // (the root's own first descent: mz_root_prepare leaves it selected)
  e->ev_start = ev[0]; e->ev_stop = ev[1];
  if (!pending) TREE_LAUNCH(k_tree_select, s, e->tv);
  e->ev_start = ev[2]; e->ev_stop = ev[3];
  TREE_LAUNCH(k_tree_expand_backup, s, e->tv, value, reward, logits);
  e->ev_start = e->ev_stop = nullptr;
  hipError_t le = hipGetLastError(), se = hipStreamSynchronize(s);
  int rc = 0;
  if (le != hipSuccess || se != hipSuccess) rc = fail("mz_tree_pair_timed: %s", hipGetErrorString(le != hipSuccess ? le : se));
  else if ((!pending && hipEventElapsedTime(&ms_out[0], ev[0], ev[1]) != hipSuccess) || hipEventElapsedTime(&ms_out[1], ev[2], ev[3]) != hipSuccess)
    rc = fail("mz_tree_pair_timed: event timing failed");
  if (pending) ms_out[0] = -1.f;
  for (auto &x : ev) hipEventDestroy(x);
  if (rc) return rc;
  e->sims_done += 1;
  e->selection_valid = false;
  return 0;
}

int mz_gather_hidden(mz_engine *e, float *hidden_out, void *stream) {
  if (!e || !hidden_out) return fail("mz_gather_hidden: null argument");
  MZ_ENTER(e);
  if (!e->selection_valid) return fail("mz_gather_hidden: call mz_select first");
  const int n = e->B * MZ_H;
  hipLaunchKernelGGL(k_gather_hidden, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->tv, hidden_out);
  HIPCHECK(hipGetLastError());
  return 0;
}

int mz_expand_backup(mz_engine *e, const float *value, const float *reward, const float *logits,
                     const float *hidden, void *stream) {
  if (!e || !value || !reward || !logits) return fail("mz_expand_backup: null argument");
  MZ_ENTER(e);
  if (!e->selection_valid) return fail("mz_expand_backup: call mz_select first");
  hipStream_t s = (hipStream_t)stream;
  if (hidden) {
    const int n = e->B * MZ_HS;
    hipLaunchKernelGGL(k_scatter_hidden, dim3((n + 255) / 256), dim3(256), 0, s, e->tv, hidden, 0);
  }
  TREE_LAUNCH(k_tree_expand_backup, s, e->tv, value, reward, logits);
  HIPCHECK(hipGetLastError());
  e->sims_done += 1;
  e->selection_valid = false;
  return 0;
}

int mz_expand_backup_select(mz_engine *e, const float *value, const float *reward, const float *logits, const float *hidden,
                            int32_t *leaf_node, int32_t *parent_slot, int32_t *action, int32_t *depth, void *stream) {
  if (!e || !value || !reward || !logits) return fail("mz_expand_backup_select: null argument");
  MZ_ENTER(e);
  if (!e->selection_valid) return fail("mz_expand_backup_select: call mz_select first");
  if (e->sims_done + 1 >= e->sims)        // the move's last simulation: nothing to descend for
    return mz_expand_backup(e, value, reward, logits, hidden, stream);
  hipStream_t s = (hipStream_t)stream;
  if (hidden) {
    const int n = e->B * MZ_HS;
    hipLaunchKernelGGL(k_scatter_hidden, dim3((n + 255) / 256), dim3(256), 0, s, e->tv, hidden, 0);
  }
  TREE_LAUNCH(k_tree_step_ext, s, e->tv, value, reward, logits, leaf_node, parent_slot, action, depth);
  HIPCHECK(hipGetLastError());
  e->sims_done += 1;
  e->selection_valid = true;
  return 0;
}

int mz_recurrent_inference(mz_engine *e, const float *hidden_in, const int32_t *action, int n, float *hidden_out,
                           float *reward, float *value, float *logits, void *stream) {
  if (!e || !hidden_in || !action || !hidden_out || !reward || !value || !logits)
    return fail("mz_recurrent_inference: null argument");
  MZ_ENTER(e);
  if (!e->weights_set) return fail("mz_recurrent_inference: weights not set (call mz_set_weights)");
  if (n < 1) return fail("mz_recurrent_inference: n must be >= 1");
  NET_LAUNCH(k_net_recurrent_rows, (n + MZ_ROWS - 1) / MZ_ROWS, (hipStream_t)stream, e->nv, hidden_in, action, n,
             hidden_out, reward, value, logits);
  HIPCHECK(hipGetLastError());
  return 0;
}

int mz_finalize(mz_engine *e, const double *temperature, const double *uniform, uint64_t move_counter,
                int32_t *action, double *child_visits, double *root_value, double *error, int32_t *visit_counts,
                void *stream) {
  if (!e) return fail("mz_finalize: null engine");
  MZ_ENTER(e);
  if (action && !temperature) return fail("mz_finalize: temperature is required when action is requested");
  const int threads = 128;
  hipLaunchKernelGGL(k_tree_finalize, dim3((e->B + threads - 1) / threads), dim3(threads), 0, (hipStream_t)stream,
                     e->tv, temperature, uniform, e->cfg.seed, move_counter, (const unsigned long long *)nullptr,
                     e->cfg.env_id_offset, action, child_visits, root_value, error, visit_counts);
  HIPCHECK(hipGetLastError());
  return 0;
}

int mz_export_tree(mz_engine *e, int32_t *N, double *W, double *P, float *R, int32_t *E, int8_t *TP,
                   uint32_t *legal_mask, double *minmax, double *noise, float *hidden_pool) {
  if (!e) return fail("mz_export_tree: null engine");
  MZ_ENTER(e);
  HIPCHECK(hipDeviceSynchronize());
  const TreeView &t = e->tv;
  if (N || W || P || R || E || TP) {      // the record pool -> the caller's per-field arrays [B][NN]
    std::vector<MzNode> pool((size_t)e->B * e->NS);
    HIPCHECK(hipMemcpy(pool.data(), e->nodes, pool.size() * sizeof(MzNode), hipMemcpyDeviceToHost));
    for (int b = 0; b < e->B; ++b)
      for (int k = 0; k < e->NN; ++k) {
        const MzNode &n = pool[(size_t)b * e->NS + MZ_NODE_OFF + k];
        const size_t i = (size_t)b * e->NN + k;
        if (N) N[i] = n.N;
        if (W) W[i] = n.W;
        if (P) P[i] = n.P;
        if (R) R[i] = n.R;
        if (E) E[i] = n.E;
        if (TP) TP[i] = n.TP;
      }
  }
  if (legal_mask) HIPCHECK(hipMemcpy(legal_mask, t.legal, (size_t)e->B * 4, hipMemcpyDeviceToHost));
  if (minmax) {
    std::vector<double> mn(e->B), mx(e->B);
    HIPCHECK(hipMemcpy(mn.data(), t.mn, (size_t)e->B * 8, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(mx.data(), t.mx, (size_t)e->B * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < e->B; ++b) { minmax[2 * b] = mn[b]; minmax[2 * b + 1] = mx[b]; }
  }
  if (noise) HIPCHECK(hipMemcpy(noise, t.noise, (size_t)e->B * e->A * 8, hipMemcpyDeviceToHost));
  if (hidden_pool) {
    std::vector<float> tmp((size_t)e->B * (e->sims + 1) * MZ_HS);
    HIPCHECK(hipMemcpy(tmp.data(), t.hpool, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < (size_t)e->B * (e->sims + 1); ++r)
      memcpy(hidden_pool + r * MZ_H, tmp.data() + r * MZ_HS, MZ_H * sizeof(float));
  }
  return 0;
}

#include "mz_selfplay_abi.inc"
#include "mz_fcl_abi.inc"

}  // extern "C"
