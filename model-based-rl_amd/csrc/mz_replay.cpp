// mz_replay.cpp -- libmz_replay.so: host-side prioritized replay ingest (include/mz_replay.h).
// Plain C++17, no GPU code.  Arithmetic and update order follow the reference's SumTree /
// PrioritizedReplay (replay_buffer.py) exactly; see the header for the citations.
//
// Layout.  A history slice (HistorySlice, game.py:5-16) is kept as the experience records it was made from: one row
// of rec_floats = O + A + MZR_REC_EXTRA float32 per step, exactly the device's record (include/mz_engine.h).  The bulk
// ingest therefore moves every record ONCE -- appended to its environment's open buffer, which becomes the slice when
// the game ends -- and sample_batch reads its fields straight from the rows.
// Threads.  mzr_ingest_records* splits the environments of a call over the handle's ingest threads (per-environment
// bookkeeping is independent, actors.py:160-173 runs per actor); the slices they build enter the ONE sum tree
// afterwards in (move, environment) order on the calling thread, so leaves, sums and counters are those of a
// single-threaded walk, bit for bit (replay_buffer.py:19-40 is order dependent in float64).
#include "../../include/mz_replay.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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

// One HistorySlice as record rows; leaves point into it as (history, step).  Row layout (R = OS + A + MZR_REC_EXTRA floats,
// OS = O float32 slots, or ceil(O / 4) when the observations are bytes -- mzr_config.obs_u8): obs, child_visits[A], root_value (float64, 2 slots), error (float64, 2 slots), reward, then int32 bits: action,
// flags (bit 0 done, bit 1 to_play == -1), step, env_id, episode.
struct Hist {
  int64_t n = 0;                // steps of the slice
  int64_t off = 0;              // first row of the slice inside `rows` (a finished game hands its whole buffer over)
  int64_t refs = 0;             // leaves pointing at this slice (touched by the inserting thread only)
  bool payload = false;         // rows carry observations / policies / values (mzr_save_history may omit them)
  std::vector<float> rows;
};

#define MZR_FLAG_DONE 1
#define MZR_FLAG_P2 2           // to_play == -1 (two-player games; single-player records leave it clear: to_play = +1)

// what Game/Actor keep per environment between flushes (game.py:54-77, actors.py:160-169)
struct EnvGame {
  std::vector<float> recs;      // rows from absolute history index `base` on
  int64_t base = 0;             // history index of recs[0]
  int64_t history_idx = 0;      // game.history_idx
  int64_t previous_collect_to = 0;
  bool done_at_last_flush = false;
};

// a history slice built by an ingest thread, waiting for its (move, env)-ordered tree insertion
struct Pending { int m, b; Hist *h; int64_t keep; const double *pri; bool done; };

struct Scratch {                // per ingest thread
  std::vector<Pending> pend;
  std::vector<double> pris;
  std::vector<size_t> pri_off;  // pris may reallocate while a thread works: offsets first, pointers after
};

// the slices of one ingest call in (move, env) order, with the priority buffers their Pending::pri point into
struct Job {
  std::vector<Pending> items;
  std::vector<std::vector<double>> pris;
};

// A fixed set of worker threads that run one job(tid) per call of run(); the caller is thread 0.
struct Pool {
  int T = 1;
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable go, done_cv;
  unsigned long long gen = 0;
  int left = 0;
  bool quit = false;
  std::function<void(int)> job;

  void start(int threads) {
    stop();
    T = threads < 1 ? 1 : threads;
    quit = false;
    const unsigned long long gen0 = gen;      // (a pool restarted after earlier runs: the new workers wait for the NEXT job)
    for (int i = 1; i < T; ++i)
      th.emplace_back([this, i, gen0] {
        unsigned long long seen = gen0;
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(mu);
            go.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
          }
          job(i);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--left == 0) done_cv.notify_one();
          }
        }
      });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    go.notify_all();
    for (auto &t : th) t.join();
    th.clear();
    T = 1;
  }
  void run(const std::function<void(int)> &f) {
    if (T == 1) { f(0); return; }
    {
      std::lock_guard<std::mutex> lk(mu);
      job = f;
      left = T - 1;
      ++gen;
    }
    go.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu);
    done_cv.wait(lk, [&] { return left == 0; });
  }
  ~Pool() { stop(); }
};

struct mz_replay {
  mzr_config c;
  int OS = 0;                      // float slots the observation takes in a row: O, or ceil(O / 4) for byte observations
  int R = 0;                       // floats per row
  // SumTree (replay_buffer.py:8-17)
  int64_t max_capacity, capacity_step, capacity, prev_capacity = 0, num_memories = 0, position = 0;
  std::vector<double> tree;
  std::vector<Hist *> leaf_hist;   // owned through Hist::refs
  std::vector<int32_t> leaf_step;
  int64_t frames = 0, games = 0;
  std::vector<EnvGame> envs;
  // scratch
  std::vector<double> pri, chg;
  std::vector<Scratch> scratch;
  Pool pool;
  // Deferred insertion (handles with more than one ingest thread): the sum-tree insertion of a call's slices -- the serial
  // part, replay_buffer.py:19-40 in arrival order -- runs on an inserter thread while the caller already assembles the next
  // chunk; jobs are inserted in call order, every other entry point waits for the queue to drain first, so the handle
  // behaves exactly as with immediate insertion.
  std::thread inserter;
  std::mutex qmu;
  std::condition_variable qcv, idle_cv;
  std::deque<Job> queue;
  std::vector<std::vector<double>> spare;      // recycled priority buffers
  bool inserting = false, qquit = false;
  // one caller at a time per handle, whoever it is: the actors' ingest on the handle's own Python thread and the native learner
  // loop's sampling / refresh calls (mz_fcl_run, libmz_hip.so) take turns here
  std::recursive_mutex api_mu;
  // row buffers of evicted slices, reused by the slices that arrive (mzr_ingest_slices): with the window full every new slice
  // replaces old ones, and a fresh 45 KB vector per slice is page faults under the process's one mmap lock -- the copies of four
  // ingest threads ran SLOWER than one thread's
  std::mutex spare_mu;
  std::vector<std::vector<float>> hist_spare;
};
#define MZR_LOCK(r) std::lock_guard<std::recursive_mutex> api_lock_(const_cast<mz_replay *>(r)->api_mu)

static void tree_add(mz_replay *r, const double *priorities, int64_t n, Hist *h, int64_t *positions_out);

// dst <- src for a slice's rows (tens of KB, read again only when a batch is sampled from it): streaming stores on the 32-byte
// aligned body -- a regular store first READS the destination line (read-for-ownership): a third of the memory traffic of the ONE
// replay's host, which is bound by exactly that traffic once the producing ranks assemble the slices
#include <immintrin.h>
static void copy_streaming(float *dst, const float *src, size_t n) {
  size_t i = 0;
  while (i < n && ((uintptr_t)(dst + i) & 31)) { dst[i] = src[i]; ++i; }
  for (; i + 8 <= n; i += 8) _mm256_stream_ps(dst + i, _mm256_loadu_ps(src + i));
  for (; i < n; ++i) dst[i] = src[i];
}

static void retire(mz_replay *r, Hist *h) {
  if (h->rows.capacity() > 0) {
    std::lock_guard<std::mutex> lk(r->spare_mu);
    if (r->hist_spare.size() < 8192) r->hist_spare.push_back(std::move(h->rows));
  }
  delete h;
}

static void insert_job(mz_replay *r, Job &job) {
  for (const Pending &p : job.items) {
    tree_add(r, p.pri, p.keep, p.h, nullptr);
    if (p.h->refs == 0) retire(r, p.h);
    r->frames += p.keep;
    if (p.done) r->games += 1;
  }
}

static void inserter_loop(mz_replay *r) {
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> lk(r->qmu);
      r->qcv.wait(lk, [&] { return r->qquit || !r->queue.empty(); });
      if (r->queue.empty()) return;            // quit with nothing left to insert
      job = std::move(r->queue.front());
      r->queue.pop_front();
      r->inserting = true;
    }
    insert_job(r, job);
    {
      std::lock_guard<std::mutex> lk(r->qmu);
      for (auto &v : job.pris) { v.clear(); if (r->spare.size() < 64) r->spare.push_back(std::move(v)); }
      r->inserting = false;
      r->idle_cv.notify_all();
    }
  }
}

// every entry point but the bulk ingest: wait until all deferred insertions have happened
static void drain(const mz_replay *cr) {
  mz_replay *r = const_cast<mz_replay *>(cr);
  if (!r->inserter.joinable()) return;
  std::unique_lock<std::mutex> lk(r->qmu);
  r->idle_cv.wait(lk, [&] { return r->queue.empty() && !r->inserting; });
}

static void stop_inserter(mz_replay *r) {
  if (!r->inserter.joinable()) return;
  drain(r);
  {
    std::lock_guard<std::mutex> lk(r->qmu);
    r->qquit = true;
  }
  r->qcv.notify_all();
  r->inserter.join();
  r->qquit = false;
}

static void start_threads(mz_replay *r, int threads) {
  stop_inserter(r);
  r->pool.start(threads);
  r->scratch.resize((size_t)r->pool.T);
  if (r->pool.T > 1) r->inserter = std::thread(inserter_loop, r);
}

// SumTree.update, replay_buffer.py:34-40
static inline void tree_update(mz_replay *r, int64_t idx, double priority) {
  double *t = r->tree.data();
  const double change = priority - t[idx];
  t[idx] = priority;
  while (idx != 0) {
    idx = (idx - 1) / 2;
    t[idx] += change;
  }
}

// SumTree.update (replay_buffer.py:34-40) for a run of leaves at consecutive tree indices, in arrival order.
// Every node must receive its `change` terms in arrival order for the float64 sums to equal the reference's
// one-leaf-at-a-time walk; the order between different nodes is free.  So the run is walked ONCE with one running sum
// per LEVEL: a leaf adds its change to all of them (independent add chains, one per level -- a vector add, instead of
// a store-to-load dependent walk to the root per leaf), and a level's sum is written back and re-read exactly when
// the walk leaves one of that level's nodes for the next (level L every 2^L leaves: one write-back per leaf in total).
static void tree_update_run(mz_replay *r, int64_t first_idx, const double *priorities, int64_t n) {
  double *t = r->tree.data();
  // 1-based heap numbering: node j has parent j >> 1.  Split where the leaf depth changes (j crosses a power of 2).
  for (int64_t a = 0; a < n;) {
    const uint64_t j0 = (uint64_t)(first_idx + a) + 1;
    const int depth = 63 - __builtin_clzll(j0);      // (<= 48: a window of 2^48 leaves does not fit any memory)
    int64_t b = n;
    const uint64_t next_pow = (uint64_t)1 << (depth + 1);
    if (j0 + (uint64_t)(n - a) > next_pow) b = a + (int64_t)(next_pow - j0);
    // acc[L - 1] = running sum of the level-L ancestor, L = 1 .. depth, as vectors of four levels (GCC vector extension:
    // two SSE2 or one AVX2 add per vector; the lanes beyond `depth` compute on zeros and are never stored)
    typedef double v4d __attribute__((vector_size(32)));
    v4d acc4[12];                                                // depth <= 48
    double *acc = (double *)acc4;
    const int nv = (depth + 3) / 4;
    for (int L = 1; L <= 4 * nv; ++L) acc[L - 1] = L <= depth ? t[(j0 >> L) - 1] : 0.0;
    // one leaf: add its change to every level's running sum, write back / re-read the levels whose node ends with it
    auto one_leaf = [&](int64_t i) {
      const uint64_t j = (uint64_t)(first_idx + i) + 1;
      const double c = priorities[i] - t[j - 1];
      t[j - 1] = priorities[i];
      const v4d c4 = {c, c, c, c};
      for (int v = 0; v < nv; ++v) acc4[v] += c4;
      // the ancestors this leaf is the LAST leaf of (within the run or at all): levels 1 .. number of trailing one bits of j
      int last = (i + 1 < b) ? __builtin_ctzll(~j) : depth;
      if (last > depth) last = depth;
      for (int L = 1; L <= last; ++L) {
        t[(j >> L) - 1] = acc[L - 1];
        if (i + 1 < b) acc[L - 1] = t[((j + 1) >> L) - 1];
      }
    };
    int64_t i = a;
    if (depth >= 4) {
      while (i < b && (((uint64_t)(first_idx + i) + 1) & 7)) one_leaf(i++);        // up to an 8-aligned leaf
      // aligned blocks of 8 leaves: the 4 + 2 + 1 nodes of levels 1..3 lie entirely inside a block and get their 2 / 4 / 8
      // terms straight in memory; the levels above take the block's 8 terms in order and are written back once per
      // block where a node ends -- no per-leaf branch on the (geometrically distributed) number of finished levels
      for (; i + 8 <= b; i += 8) {
        const uint64_t j = (uint64_t)(first_idx + i) + 1;                           // j % 8 == 0
        double c[8];
        for (int k = 0; k < 8; ++k) { c[k] = priorities[i + k] - t[j - 1 + k]; t[j - 1 + k] = priorities[i + k]; }
        double *l1 = t + (j >> 1) - 1, *l2 = t + (j >> 2) - 1, *l3 = t + (j >> 3) - 1;
        for (int k = 0; k < 4; ++k) l1[k] = (l1[k] + c[2 * k]) + c[2 * k + 1];
        for (int k = 0; k < 2; ++k) l2[k] = (((l2[k] + c[4 * k]) + c[4 * k + 1]) + c[4 * k + 2]) + c[4 * k + 3];
        double s3 = l3[0];
        for (int k = 0; k < 8; ++k) s3 += c[k];
        l3[0] = s3;
        {      // the levels above: eight terms in order per level, the running sums in registers (lanes of levels 1..3: unused here)
          v4d cb[8];
          for (int k = 0; k < 8; ++k) cb[k] = v4d{c[k], c[k], c[k], c[k]};
          for (int v = 0; v < nv; ++v) {
            v4d x = acc4[v];
            x += cb[0]; x += cb[1]; x += cb[2]; x += cb[3]; x += cb[4]; x += cb[5]; x += cb[6]; x += cb[7];
            acc4[v] = x;
          }
        }
        const uint64_t je = j + 7;                                                  // the block's last leaf: low 3 bits set
        const bool more = i + 8 < b;
        int last = more ? __builtin_ctzll(~je) : depth;
        if (last > depth) last = depth;
        for (int L = 4; L <= last; ++L) {
          t[(je >> L) - 1] = acc[L - 1];
          if (more) acc[L - 1] = t[((je + 1) >> L) - 1];
        }
        if (more) for (int L = 1; L <= 3; ++L) acc[L - 1] = t[((je + 1) >> L) - 1];   // for the leaf-by-leaf tail
      }
    }
    for (; i < b; ++i) one_leaf(i);
    a = b;
  }
}

// SumTree.add, replay_buffer.py:19-32, for n leaves of one history: the leaves up to the end of the current capacity
// are consecutive slots -- payload pointers, steps and the memory count are set per run instead of per leaf (the
// reference's `position = (position + 1) % capacity` and its capacity growth at the wrap, run by run).
static void tree_add(mz_replay *r, const double *priorities, int64_t n, Hist *h, int64_t *positions_out) {
  int64_t step = 0;
  while (step < n) {
    const int64_t pos = r->position;
    int64_t seg = r->capacity - pos;
    if (seg > n - step) seg = n - step;
    Hist **slot = r->leaf_hist.data() + pos;
    int32_t *ls = r->leaf_step.data() + pos;
    for (int64_t i = 0; i < seg;) {             // evicted payloads: consecutive slots mostly share one history
      Hist *old = slot[i];
      int64_t j = i + 1;
      while (j < seg && slot[j] == old) ++j;
      if (old && (old->refs -= (j - i)) == 0) retire(r, old);
      i = j;
    }
    for (int64_t i = 0; i < seg; ++i) { slot[i] = h; ls[i] = (int32_t)(step + i); }
    if (h) h->refs += seg;
    if (positions_out)
      for (int64_t i = 0; i < seg; ++i) positions_out[step + i] = pos + i;
    // `if self.position >= self.prev_capacity: self.num_memories += 1`, replay_buffer.py:26-27
    {
      const int64_t lo = pos > r->prev_capacity ? pos : r->prev_capacity;
      if (pos + seg > lo) r->num_memories += pos + seg - lo;
    }
    tree_update_run(r, pos + r->max_capacity - 1, priorities + step, seg);
    step += seg;
    r->position = pos + seg;
    if (r->position == r->capacity) {           // replay_buffer.py:29-32
      r->position = 0;
      r->prev_capacity = r->capacity;
      const int64_t next = r->capacity + r->capacity_step;
      r->capacity = next < r->max_capacity ? next : r->max_capacity;
    }
  }
}

static inline void row_put_double(float *dst, double v) { memcpy(dst, &v, sizeof v); }
static inline double row_double(const float *p) {     // a float64 stored in two float slots (4-byte aligned)
  double v;
  memcpy(&v, p, sizeof v);
  return v;
}

static int save_history(mz_replay *r, int64_t n, const double *errors, int64_t ignore, int terminal,
                        const float *obs, const float *child_visits, const double *root_values,
                        const float *rewards, const int32_t *actions, const uint8_t *dones, const int8_t *to_play) {
  // replay_buffer.py:113-119: errors[:-ignore] (python: ignore == 0 would give an empty list)
  int64_t keep = n;
  if (ignore >= 0) keep = ignore == 0 ? 0 : (n - ignore > 0 ? n - ignore : 0);
  Hist *h = nullptr;
  if (obs || child_visits || root_values || rewards || actions || dones || to_play) {
    h = new Hist();
    h->n = n;
    h->payload = obs && child_visits && root_values;
    const int O = r->c.obs_dim, OS = r->OS, A = r->c.action_space, R = r->R;
    h->rows.assign((size_t)n * R, 0.f);
    for (int64_t i = 0; i < n; ++i) {
      float *q = h->rows.data() + (size_t)i * R;
      int32_t *qi = (int32_t *)(q + OS + A + 5);
      if (obs) {
        if (r->c.obs_u8) { uint8_t *qb = (uint8_t *)q; for (int k = 0; k < O; ++k) qb[k] = (uint8_t)obs[i * O + k]; }
        else memcpy(q, obs + i * O, O * sizeof(float));
      }
      if (child_visits) memcpy(q + OS, child_visits + i * A, A * sizeof(float));
      if (root_values) row_put_double(q + OS + A, root_values[i]);
      row_put_double(q + OS + A + 2, errors[i]);
      if (rewards) q[OS + A + 4] = rewards[i];
      qi[0] = actions ? actions[i] : 0;
      qi[1] = ((dones && dones[i]) ? MZR_FLAG_DONE : 0) | ((to_play && to_play[i] < 0) ? MZR_FLAG_P2 : 0);
      qi[2] = (int32_t)i;
    }
  }
  if ((int64_t)r->pri.size() < keep) r->pri.resize(keep);
  if (r->c.alpha == 1.0)      // pow(x, 1.0) == x exactly: skip the libm call
    for (int64_t i = 0; i < keep; ++i) r->pri[i] = fabs(errors[i]) + r->c.epsilon;
  else
    for (int64_t i = 0; i < keep; ++i) r->pri[i] = pow(fabs(errors[i]) + r->c.epsilon, r->c.alpha);
  tree_add(r, r->pri.data(), keep, h, nullptr);
  if (h && h->refs == 0) delete h;         // no kept step points at it (everything was `ignore`d)
  r->frames += keep;                       // replay_buffer.py:121
  if (terminal) r->games += 1;             // replay_buffer.py:122
  return 0;
}

extern "C" {

const char *mzr_last_error(void) { return g_err.c_str(); }

int mzr_create(const mzr_config *cfg, mz_replay **out) {
  if (!cfg || !out) return fail("mzr_create: null argument");
  if (cfg->window_size < 1 || cfg->window_step < 1 || cfg->window_step > cfg->window_size)
    return fail("mzr_create: need 1 <= window_step <= window_size");
  if (cfg->obs_dim < 1 || cfg->action_space < 1) return fail("mzr_create: bad obs_dim/action_space");
  if (cfg->ingest_threads < 0 || cfg->ingest_threads > 64) return fail("mzr_create: ingest_threads must be in [0, 64]");
  mz_replay *r = new mz_replay();
  r->c = *cfg;
  r->OS = cfg->obs_u8 ? (cfg->obs_dim + 3) / 4 : cfg->obs_dim;
  r->R = r->OS + cfg->action_space + MZR_REC_EXTRA;
  r->max_capacity = cfg->window_size;
  r->capacity_step = cfg->window_step;
  r->capacity = cfg->window_step;
  r->tree.assign((size_t)(2 * cfg->window_size - 1), 0.0);
  r->leaf_hist.assign((size_t)cfg->window_size, nullptr);
  r->leaf_step.assign((size_t)cfg->window_size, 0);
  start_threads(r, cfg->ingest_threads > 1 ? cfg->ingest_threads : 1);
  *out = r;
  return 0;
}

int mzr_destroy(mz_replay *r) {
  if (r) {
    stop_inserter(r);
    r->pool.stop();
    for (int64_t i = 0; i < (int64_t)r->leaf_hist.size();) {
      Hist *h = r->leaf_hist[(size_t)i];
      int64_t j = i + 1;
      while (j < (int64_t)r->leaf_hist.size() && r->leaf_hist[(size_t)j] == h) ++j;
      if (h && (h->refs -= (j - i)) == 0) delete h;
      i = j;
    }
  }
  delete r;
  return 0;
}

int mzr_set_ingest_threads(mz_replay *r, int threads) {
  if (!r) return fail("mzr_set_ingest_threads: null");
  MZR_LOCK(r);
  if (threads < 1 || threads > 64) return fail("mzr_set_ingest_threads: threads must be in [1, 64]");
  start_threads(r, threads);
  return 0;
}
int mzr_ingest_threads(const mz_replay *r) { return r ? r->pool.T : -1; }

int mzr_priorities(const mz_replay *r, const double *errors, int64_t n, double *out) {
  if (!r || !errors || !out) return fail("mzr_priorities: null argument");
  MZR_LOCK(r);
  for (int64_t i = 0; i < n; ++i) out[i] = pow(fabs(errors[i]) + r->c.epsilon, r->c.alpha);
  return 0;
}

int mzr_tree_add(mz_replay *r, const double *priorities, int64_t n, int64_t *positions_out) {
  if (!r || !priorities) return fail("mzr_tree_add: null argument");
  MZR_LOCK(r);
  drain(r);
  tree_add(r, priorities, n, nullptr, positions_out);
  return 0;
}

int mzr_tree_update(mz_replay *r, const int64_t *idxs, const double *priorities, int64_t n) {
  if (!r || !idxs || !priorities) return fail("mzr_tree_update: null argument");
  MZR_LOCK(r);
  drain(r);
  const int64_t len = 2 * r->max_capacity - 1;
  for (int64_t i = 0; i < n; ++i)
    if (idxs[i] < 0 || idxs[i] >= len) return fail("mzr_tree_update: index %lld out of range", (long long)idxs[i]);
  if (n < 8) {
    for (int64_t i = 0; i < n; ++i) tree_update(r, idxs[i], priorities[i]);
    return 0;
  }
  // A batch (the learner's priority refresh, replay_buffer.py:200-203 -> 34-40): every node must receive its `change`
  // terms in arrival order for the float64 sums to equal the reference's leaf-by-leaf walks -- the order between different
  // nodes is free.  So: the leaves first, in order (a repeated leaf sees its predecessor's write), then level by level,
  // every level in arrival order: independent read-modify-writes whose cache misses overlap, instead of n dependent
  // walks to the root.
  // Large batches (the learner loop at batch >= 1024: 2048 random leaves x 21 levels took 215 us of its ONE host thread): the
  // entries are dealt to the handle's threads by the SUBTREE (at level L, 2^L >= threads) their leaf lies in -- different subtrees
  // share no node below level L, so every thread runs the passes above on its own entries, in arrival order, up to its subtree's
  // root; the L levels above are then walked by the caller for all entries in arrival order.  Every node still receives its
  // terms in arrival order: the sums are the same bits.
  double *t = r->tree.data();
  std::vector<double> &chg = r->chg;
  std::vector<int64_t> node((size_t)n);
  chg.resize((size_t)n);
  auto depth = [](int64_t k) { return 63 - __builtin_clzll((unsigned long long)k + 1ull); };
  const int deep = depth(len - 1);
  const bool two_depths = depth(r->max_capacity - 1) != deep;
  // entries `list` (indices into the batch, arrival order; null: all of them) up to the nodes of level `stop` (0: the root)
  auto run = [&](const int32_t *list, int64_t cnt, int stop) {
    const int64_t first_above = ((int64_t)1 << (stop + 1)) - 1;        // nodes with an index >= this have a parent at level >= stop
#define MZR_AT(q) (list ? (int64_t)list[q] : (q))
    for (int64_t q = 0; q < cnt; ++q) __builtin_prefetch(&t[idxs[MZR_AT(q)]], 1);
    for (int64_t q = 0; q < cnt; ++q) {
      const int64_t i = MZR_AT(q);
      chg[(size_t)i] = priorities[i] - t[idxs[i]];
      t[idxs[i]] = priorities[i];
      node[(size_t)i] = idxs[i];
    }
    // a capacity that is not a power of two puts the leaves at TWO depths: the deeper entries take one step up on their own
    // first (in arrival order), so that from then on every entry sits at the same depth and ALL the contributions to a node
    // arrive within one pass, in arrival order (r04 advice: the level-by-level passes alone let a shallower leaf's change
    // reach a common ancestor one pass before a deeper leaf's that came earlier in the batch)
    if (two_depths)
      for (int64_t q = 0; q < cnt; ++q) {
        int64_t &k = node[(size_t)MZR_AT(q)];
        if (k >= first_above && depth(k) == deep) {
          k = (k - 1) / 2;
          t[k] += chg[(size_t)MZR_AT(q)];
        }
      }
    for (bool any = true; any;) {
      any = false;
      for (int64_t q = 0; q < cnt; ++q)
        if (node[(size_t)MZR_AT(q)] >= first_above) __builtin_prefetch(&t[(node[(size_t)MZR_AT(q)] - 1) / 2], 1);
      for (int64_t q = 0; q < cnt; ++q) {
        const int64_t i = MZR_AT(q);
        int64_t &k = node[(size_t)i];
        if (k < first_above) continue;
        k = (k - 1) / 2;
        t[k] += chg[(size_t)i];
        any = true;
      }
    }
#undef MZR_AT
  };
  const int T = r->pool.T;
  int L = 0;
  while (((int)1 << L) < T && L < 4) ++L;
  if (T <= 1 || n < 512 || deep - 1 <= L + 1) {
    run(nullptr, n, 0);
    return 0;
  }
  const int S = 1 << L;                                      // subtrees at level L (nodes S - 1 .. 2 S - 2)
  std::vector<std::vector<int32_t>> lists((size_t)S);
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t j = (uint64_t)idxs[i] + 1;               // 1-based heap number: the ancestor at level L is j >> (depth - L)
    const int d = 63 - __builtin_clzll(j);
    lists[(size_t)((j >> (d - L)) - (uint64_t)S)].push_back((int32_t)i);
  }
  r->pool.run([&](int tid) {
    for (int sub = tid; sub < S; sub += T)
      if (!lists[(size_t)sub].empty()) run(lists[(size_t)sub].data(), (int64_t)lists[(size_t)sub].size(), L);
  });
  for (int64_t i = 0; i < n; ++i) {                          // the L levels above the subtrees' roots, in arrival order
    int64_t k = node[(size_t)i];
    while (k != 0) {
      k = (k - 1) / 2;
      t[k] += chg[(size_t)i];
    }
  }
  return 0;
}

// SumTree.get_leaf, replay_buffer.py:42-62 (returns the tree index of the leaf)
static int64_t get_leaf(const mz_replay *r, double value);
int64_t mzr_tree_get_leaf(const mz_replay *r, double value) {
  MZR_LOCK(r);
  drain(r);
  return get_leaf(r, value);
}
static int64_t get_leaf(const mz_replay *r, double value) {
  const int64_t len = 2 * r->max_capacity - 1;
  const double *t = r->tree.data();
  int64_t parent = 0;
  for (;;) {
    const int64_t left = 2 * parent + 1;
    if (left >= len) return parent;
    if (value <= t[left]) parent = left;
    else { value -= t[left]; parent = left + 1; }
  }
}

// SumTree.get_leaf's payload (replay_buffer.py:58-62): the (step, history) the leaf at tree index idx holds
int mzr_leaf_info(const mz_replay *r, int64_t idx, double *priority, int64_t *step, int64_t *n_steps, int *has_payload) {
  if (!r) return fail("mzr_leaf_info: null handle");
  MZR_LOCK(r);
  drain(r);
  const int64_t pos = idx - r->max_capacity + 1;
  if (pos < 0 || pos >= r->max_capacity) return fail("mzr_leaf_info: %lld is not a leaf index", (long long)idx);
  const Hist *h = r->leaf_hist[(size_t)pos];
  if (priority) *priority = r->tree[(size_t)idx];
  if (step) *step = h ? r->leaf_step[(size_t)pos] : -1;
  if (n_steps) *n_steps = h ? h->n : 0;
  if (has_payload) *has_payload = (h && h->payload) ? 1 : 0;
  return 0;
}

int mzr_leaf_history(const mz_replay *r, int64_t idx, float *rows_out, int64_t n_steps) {
  if (!r || !rows_out) return fail("mzr_leaf_history: null argument");
  MZR_LOCK(r);
  drain(r);
  const int64_t pos = idx - r->max_capacity + 1;
  if (pos < 0 || pos >= r->max_capacity) return fail("mzr_leaf_history: %lld is not a leaf index", (long long)idx);
  const Hist *h = r->leaf_hist[(size_t)pos];
  if (!h) return fail("mzr_leaf_history: leaf %lld is empty", (long long)idx);
  if (!h->payload) return fail("mzr_leaf_history: the history was ingested without payload");
  if (n_steps != h->n) return fail("mzr_leaf_history: the history has %lld steps, the buffer %lld", (long long)h->n, (long long)n_steps);
  memcpy(rows_out, h->rows.data() + (size_t)h->off * r->R, (size_t)h->n * r->R * sizeof(float));
  return 0;
}

double mzr_total_priority(const mz_replay *r) { MZR_LOCK(r); drain(r); return r->tree[0]; }
int64_t mzr_size(const mz_replay *r) { MZR_LOCK(r); drain(r); return r->num_memories; }
int mzr_tree_leaves(const mz_replay *r, int64_t n, double *out) {
  if (!r || !out || n > r->max_capacity) return fail("mzr_tree_leaves: bad argument");
  MZR_LOCK(r);
  drain(r);
  memcpy(out, r->tree.data() + r->max_capacity - 1, (size_t)n * sizeof(double));
  return 0;
}

int mzr_save_history(mz_replay *r, int64_t n, const double *errors, int64_t ignore, int terminal, const float *obs,
                     const float *child_visits, const double *root_values, const float *rewards,
                     const int32_t *actions, const uint8_t *dones, const int8_t *to_play) {
  if (!r || (n > 0 && !errors)) return fail("mzr_save_history: null argument");
  MZR_LOCK(r);
  drain(r);
  return save_history(r, n, errors, ignore, terminal, obs, child_visits, root_values, rewards, actions, dones, to_play);
}

int mzr_ingest_records(mz_replay *r, const float *records, int n_moves, int B, int rec_floats) {
  return mzr_ingest_records_from(r, records, n_moves, B, rec_floats, 0);
}

// actors.py:160-173 for the environments [b_lo, b_hi) of one call, environment-major: an env's bookkeeping and the
// tail of its open buffer are touched once per call instead of once per move.  The slices that fall due are built
// here (their priorities too); their insertion into the sum tree is deferred (Pending).
// what the per-environment assembly needs of a configuration: the replay's own (direct ingest) or a producing rank's (mz_assembler)
struct AsmCfg {
  int OS, A, R;
  int64_t overlap, max_history_length;
  double epsilon, alpha;
};
static AsmCfg asm_cfg_of(const mzr_config &c, int OS, int R) {
  return AsmCfg{OS, c.action_space, R, (int64_t)c.num_unroll_steps + c.td_steps, (int64_t)c.max_history_length, c.epsilon, c.alpha};
}

// env_major: the chunk is [B][n_moves][rec] (mzr_pack_env_major, made by the PRODUCING rank) instead of the device loop's
// [n_moves][B][rec]: an environment's rows of the call are one contiguous piece -- a sequential read, appended run by run
static void ingest_range(const AsmCfg &ac, EnvGame *envs, Scratch &sc, const float *records, int n_moves, int B, int env_base, int b_lo,
                         int b_hi, bool env_major) {
  const int OS = ac.OS, A = ac.A, R = ac.R;
  const int64_t overlap = ac.overlap;
  sc.pend.clear(); sc.pris.clear(); sc.pri_off.clear();
  const int PF = 6;       // envs ahead: an env's records lie n_moves strides of B * rec_floats apart (move-major ring),
  auto row = [&](int b, int m) { return env_major ? records + ((size_t)b * n_moves + m) * R : records + ((size_t)m * B + b) * R; };
  for (int b = b_lo; b < b_hi; ++b) {      // which no hardware prefetcher follows -- request them while the envs before are handled
    if (b + PF < b_hi) {
      if (env_major) {
        const char *q = (const char *)row(b + PF, 0);
        for (int k = 0; k < n_moves * R * 4; k += 64) __builtin_prefetch(q + k, 0, 1);
      } else {
        for (int m = 0; m < n_moves; ++m) {
          const char *q = (const char *)row(b + PF, m);
          __builtin_prefetch(q, 0, 1);
          __builtin_prefetch(q + 64, 0, 1);
        }
      }
      const EnvGame &gn = envs[(size_t)env_base + b + PF];      // and the tail of the open buffer the rows go to
      if (!gn.recs.empty()) {
        const char *q = (const char *)(gn.recs.data() + gn.recs.size());
        for (int k = 0; k < n_moves * R * 4; k += 64) __builtin_prefetch(q + k, 1, 1);
      }
    }
    EnvGame &g = envs[(size_t)env_base + b];
    if (g.recs.capacity() == 0) g.recs.reserve((size_t)64 * R);
    // runs of moves up to (and including) the next move at which a flush falls due (actors.py:160-169): the run's rows are
    // appended first -- in ONE piece where the chunk is environment-major -- then the flush, exactly as move by move
    for (int m0 = 0; m0 < n_moves;) {
      int e = m0;
      bool flush = false, done = false;
      int64_t hidx = g.history_idx;
      for (; e < n_moves; ++e) {
        const int32_t *ri = (const int32_t *)(row(b, e) + OS + A + 5);
        done = (ri[1] & MZR_FLAG_DONE) != 0;
        ++hidx;
        if (done || (hidx - g.previous_collect_to) == ac.max_history_length) { flush = true; break; }
      }
      const int last = flush ? e : n_moves - 1;
      if (env_major) {
        const float *src = row(b, m0);
        g.recs.insert(g.recs.end(), src, src + (size_t)(last - m0 + 1) * R);
      } else {
        for (int m = m0; m <= last; ++m) { const float *rec = row(b, m); g.recs.insert(g.recs.end(), rec, rec + R); }
      }
      g.history_idx += last - m0 + 1;
      m0 = last + 1;
      if (!flush) break;
      const int m = e;
      // actors.py:160-169
      const bool d_prev = g.previous_collect_to == 0 ? done : g.done_at_last_flush;   // dones[prev-1] (index -1 when prev == 0)
      const int64_t collect_from = d_prev ? g.previous_collect_to
                                          : (g.previous_collect_to - overlap > 0 ? g.previous_collect_to - overlap : 0);
      const int64_t n = g.history_idx - collect_from;
      const int64_t ignore = done ? -1 : overlap;
      // PrioritizedReplay.save_history (replay_buffer.py:113-122): the slice IS the env's rows
      Hist *h = new Hist();
      h->n = n;
      h->payload = true;
      int64_t keep = n;
      if (ignore >= 0) keep = ignore == 0 ? 0 : (n - ignore > 0 ? n - ignore : 0);
      const size_t off = sc.pris.size();
      sc.pris.resize(off + (size_t)n);
      double *pri = sc.pris.data() + off;
      const float *q = g.recs.data() + (size_t)(collect_from - g.base) * R;
      for (int64_t i = 0; i < n; ++i, q += R) {
        const double e_ = fabs(row_double(q + OS + A + 2)) + ac.epsilon;
        pri[i] = ac.alpha == 1.0 ? e_ : pow(e_, ac.alpha);
      }
      h->off = collect_from - g.base;
      if (done) {           // the game is over: its buffer becomes the slice (run_selfplay starts a new Game, actors.py:94-97)
        h->rows.swap(g.recs);
        g.recs.clear(); g.base = 0; g.history_idx = 0; g.previous_collect_to = 0; g.done_at_last_flush = false;
      } else {
        // the game goes on: the buffer becomes the slice too, and the new buffer starts with the rows the next slice can still reach
        // back to (the last `overlap`: 15 rows copied instead of the slice's max_history_length + overlap -- until r05 the slice was
        // copied out and the buffer trimmed: one more pass over every record on the ONE replay's host)
        g.previous_collect_to = g.history_idx;
        g.done_at_last_flush = done;
        const int64_t nb = g.history_idx - overlap > 0 ? g.history_idx - overlap : 0;
        const int64_t from = nb > g.base ? nb : g.base;
        std::vector<float> tail;
        tail.reserve((size_t)64 * R > (size_t)(g.history_idx - from + n_moves) * R ? (size_t)64 * R : (size_t)(g.history_idx - from + n_moves) * R);
        tail.assign(g.recs.begin() + (size_t)(from - g.base) * R, g.recs.end());
        h->rows.swap(g.recs);
        g.recs.swap(tail);
        g.base = from;
      }
      sc.pend.push_back(Pending{m, b, h, keep, nullptr, done});
      sc.pri_off.push_back(off);
    }
  }
  for (size_t i = 0; i < sc.pend.size(); ++i) sc.pend[i].pri = sc.pris.data() + sc.pri_off[i];
}

// [n_moves][B][rec] (what the device loop writes) -> [B][n_moves][rec], on the PRODUCING rank (distributed.ShmRing.put: it replaces the
// plain copy into the ring's slot): rank 0's ONE replay then reads every environment's rows of a chunk as one sequential piece.  Reads
// strided (prefetched a few environments ahead), writes sequential; n_moves x rec floats per environment.
void mzr_pack_env_major(const float *src, float *dst, int n_moves, int B, int rec_floats) {
  const size_t R = (size_t)rec_floats;
  const int PF = 8;
  for (int b = 0; b < B; ++b) {
    if (b + PF < B)
      for (int m = 0; m < n_moves; ++m) {
        const char *q = (const char *)(src + ((size_t)m * B + b + PF) * R);
        __builtin_prefetch(q, 0, 1);
        __builtin_prefetch(q + 64, 0, 1);
      }
    float *d = dst + (size_t)b * n_moves * R;
    for (int m = 0; m < n_moves; ++m) memcpy(d + (size_t)m * R, src + ((size_t)m * B + b) * R, R * sizeof(float));
  }
}

static int ingest_records_impl(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base, bool env_major);
int mzr_ingest_records_from(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base) {
  return ingest_records_impl(r, records, n_moves, B, rec_floats, env_base, false);
}
// the same for a chunk packed by mzr_pack_env_major ([B][n_moves][rec]); the replay's state after the call is the one
// mzr_ingest_records_from leaves on the unpacked chunk, bit for bit
int mzr_ingest_records_packed(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base) {
  return ingest_records_impl(r, records, n_moves, B, rec_floats, env_base, true);
}

static int ingest_records_impl(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base, bool env_major) {
  if (!r || !records) return fail("mzr_ingest_records: null argument");
  MZR_LOCK(r);
  if (n_moves < 0 || B < 1 || env_base < 0) return fail("mzr_ingest_records: bad shape (n_moves %d, B %d, env_base %d)", n_moves, B, env_base);
  if (rec_floats != r->R)
    return fail("mzr_ingest_records: rec_floats %d != obs slots (%d) + action_space + %d = %d", rec_floats, r->OS, MZR_REC_EXTRA, r->R);
  // what the records cannot express (include/mz_replay.h): `terminal` separate from `done`
  if (r->c.episode_life)
    return fail("mzr_ingest_records: this replay is configured with episode_life (terminal != done, game.py:90): records "
                "carry one end-of-game flag; feed such histories through mzr_save_history");
  if ((int)r->envs.size() < env_base + B) r->envs.resize((size_t)env_base + B);
  // contiguous environment ranges per thread: concatenated in thread order the slices are in environment order, and a
  // stable sort by move restores (move, env) -- the arrival order of a move-major walk, the order the reference's one
  // replay would see with actors flushing in lock-step
  const int T = r->pool.T < B ? r->pool.T : B;
  if (T <= 1) {
    ingest_range(asm_cfg_of(r->c, r->OS, r->R), r->envs.data(), r->scratch[0], records, n_moves, B, env_base, 0, B, env_major);
  } else {
    r->pool.run([&](int tid) {
      if (tid >= T) return;
      const int lo = (int)((int64_t)B * tid / T), hi = (int)((int64_t)B * (tid + 1) / T);
      ingest_range(asm_cfg_of(r->c, r->OS, r->R), r->envs.data(), r->scratch[(size_t)tid], records, n_moves, B, env_base, lo, hi, env_major);
    });
  }
  Job job;
  for (int t = 0; t < (T < 1 ? 1 : T); ++t) {
    Scratch &sc = r->scratch[(size_t)t];
    job.items.insert(job.items.end(), sc.pend.begin(), sc.pend.end());
    job.pris.push_back(std::move(sc.pris));     // the Pending::pri pointers stay valid: the buffer moves, its storage does not
    sc.pris = std::vector<double>();
  }
  std::stable_sort(job.items.begin(), job.items.end(), [](const Pending &x, const Pending &y) { return x.m < y.m; });   // b already ascending
  if (!r->inserter.joinable()) {
    insert_job(r, job);
    for (int t = 0; t < (int)job.pris.size(); ++t) { job.pris[(size_t)t].clear(); r->scratch[(size_t)t].pris = std::move(job.pris[(size_t)t]); }
    return 0;
  }
  {
    std::unique_lock<std::mutex> lk(r->qmu);
    // back-pressure: at most a few chunks' insertions pending (their slices and priorities are memory; a caller that
    // assembles faster than the tree takes them then runs at the tree's rate)
    r->idle_cv.wait(lk, [&] { return r->queue.size() < 4; });
    for (int t = 0; t < (T < 1 ? 1 : T) && !r->spare.empty(); ++t) {      // hand the ingest threads recycled buffers
      r->scratch[(size_t)t].pris = std::move(r->spare.back());
      r->spare.pop_back();
    }
    r->queue.push_back(std::move(job));
  }
  r->qcv.notify_one();
  return 0;
}

int mzr_sample_batch(const mz_replay *r, const double *draws, int bs, float *obs, int32_t *actions,
                     float *target_rewards, float *target_values, float *target_policies, int64_t *idxs,
                     double *priorities) {
  if (!r || !draws || !obs || !actions || !target_rewards || !target_values || !target_policies || !idxs || !priorities)
    return fail("mzr_sample_batch: null argument");
  MZR_LOCK(r);
  drain(r);
  const int O = r->c.obs_dim, OS = r->OS, A = r->c.action_space, K = r->c.num_unroll_steps, td = r->c.td_steps, R = r->R;
  const int TL = K + 1;
  // replay_buffer.py:81-82: discounts as float32, discount**td as a Python float
  float disc[256];
  if (K + td > 256) return fail("mzr_sample_batch: num_unroll_steps + td_steps too large");
  for (int n = 0; n < K + td; ++n) disc[n] = (float)pow(r->c.discount, (double)n);
  const double disc_td = pow(r->c.discount, (double)td);
  // blocks of 128 samples (what the passes prefetch stays in cache until it is used); a block reads the tree and the histories and
  // writes its own rows of the outputs: blocks are independent, and batches of more than one block are spread over the replay's
  // thread pool (the learner's native loop samples a batch per update on ONE thread: 33 us at batch 256, 0.5 ms at 4096 -- more
  // than the update's kernels take; block -> thread is fixed, the outputs do not depend on the thread count).  -> 0, or 1 / 2
  auto block = [&](int b0) -> int {
  const int b1 = b0 + 128 < bs ? b0 + 128 : bs;
  // The batch is bound by cache misses, not arithmetic (a window of 2e5 frames: a 32 MB tree and rows spread over the
  // heap): the work is laid out in passes of independent iterations, so that the misses of many samples are in flight at
  // once.  Pass 1: the bs tree descents (replay_buffer.py:142, SumTree.get_leaf 42-62), eight interleaved level by level
  // -- the same comparisons and subtractions per draw as get_leaf.
  {
    const int64_t len = 2 * r->max_capacity - 1;
    const double *t = r->tree.data();
    for (int i0 = b0; i0 < b1; i0 += 8) {
      const int g = b1 - i0 < 8 ? b1 - i0 : 8;
      int64_t par[8];
      double val[8];
      for (int k = 0; k < g; ++k) { par[k] = 0; val[k] = draws[i0 + k]; }
      for (bool any = true; any;) {
        any = false;
        for (int k = 0; k < g; ++k) {
          const int64_t left = 2 * par[k] + 1;
          if (left >= len) continue;
          const double tl = t[left];
          const bool go_left = val[k] <= tl;
          val[k] = go_left ? val[k] : val[k] - tl;
          par[k] = go_left ? left : left + 1;
          any = true;
        }
      }
      for (int k = 0; k < g; ++k) idxs[i0 + k] = par[k];
    }
  }
  // pass 2: the leaves' payload pointers; pass 3: the histories' headers; pass 4: the rows the targets will read
  for (int i = b0; i < b1; ++i) __builtin_prefetch(&r->leaf_hist[(size_t)(idxs[i] - r->max_capacity + 1)]);
  for (int i = b0; i < b1; ++i) {
    const Hist *h = r->leaf_hist[(size_t)(idxs[i] - r->max_capacity + 1)];
    if (h) __builtin_prefetch(h);
    __builtin_prefetch(&r->leaf_step[(size_t)(idxs[i] - r->max_capacity + 1)]);
    __builtin_prefetch(&r->tree[(size_t)idxs[i]]);
  }
  for (int i = b0; i < b1; ++i) {
    const int64_t pos = idxs[i] - r->max_capacity + 1;
    const Hist *h = r->leaf_hist[(size_t)pos];
    if (!h || !h->payload) continue;
    const float *p0 = h->rows.data() + ((size_t)h->off + (size_t)r->leaf_step[(size_t)pos]) * R;
    const int64_t left = h->n - r->leaf_step[(size_t)pos];
    const int64_t span = (left < K + td + 1 ? left : K + td + 1) * R;      // floats the targets may touch from this step on
    for (int64_t o = -(int64_t)R; o < span; o += 16) __builtin_prefetch(p0 + (o < 0 ? (r->leaf_step[(size_t)pos] > 0 ? o : 0) : o));
  }
  for (int i = b0; i < b1; ++i) {
    const int64_t idx = idxs[i];
    const int64_t pos = idx - r->max_capacity + 1;
    const Hist *h = r->leaf_hist[(size_t)pos];
    if (!h) return 1;
    const int64_t step = r->leaf_step[(size_t)pos];
    priorities[i] = r->tree[(size_t)idx];
    if (!h->payload) return 2;
    const float *rows = h->rows.data() + (size_t)h->off * R;
    auto row = [&](int64_t s) { return rows + (size_t)s * R; };
    auto reward = [&](int64_t s) { return row(s)[OS + A + 4]; };
    auto flags = [&](int64_t s) { return ((const int32_t *)(row(s) + OS + A + 5))[1]; };
    if (r->c.obs_u8) {                                                                        // 147 (bytes -> float32, as np.float32(obs))
      const uint8_t *ob = (const uint8_t *)row(step);
      for (int k = 0; k < O; ++k) obs[(size_t)i * O + k] = (float)ob[k];
    } else {
      memcpy(obs + (size_t)i * O, row(step), O * sizeof(float));
    }
    // (History: observations has one entry more than the other lists, game.py:93-96; every other list has h->n entries)
    const int64_t end_index = h->n, n_rewards = h->n;
    for (int k = 0; k < K; ++k)                                                        // 149-152
      actions[(size_t)i * K + k] = (step + k < end_index) ? ((const int32_t *)(row(step + k) + OS + A + 5))[0] : -1;
    // insert_target, replay_buffer.py:165-198
    for (int j = 0; j < TL; ++j) {
      const int64_t cur = step + j;
      const float last_reward = (cur > 0 && cur <= n_rewards) ? reward(cur - 1) : 0.f;
      float *pol = target_policies + ((size_t)i * TL + j) * A;
      if (cur < end_index) {
        const int tp = flags(cur) & MZR_FLAG_P2;
        const int64_t boot = cur + td;
        double value = boot < end_index ? row_double(row(boot) + OS + A) * disc_td : 0.0;
        const int64_t hi = boot < n_rewards ? boot : n_rewards;
        if (hi > cur) {
          float acc = 0.f;
          for (int64_t q = cur; q < hi; ++q) {
            const float rw = ((flags(q) & MZR_FLAG_P2) != tp) ? -reward(q) : reward(q);      // 187-189
            acc += rw * disc[q - cur];
          }
          value += (double)acc;
        }
        memcpy(pol, row(cur) + OS, A * sizeof(float));
        target_rewards[(size_t)i * TL + j] = last_reward;
        target_values[(size_t)i * TL + j] = (float)value;
      } else {
        for (int a = 0; a < A; ++a) pol[a] = 0.f;
        target_rewards[(size_t)i * TL + j] = last_reward;
        target_values[(size_t)i * TL + j] = 0.f;
      }
    }
  }
  return 0;
  };
  const int nblocks = (bs + 127) / 128;
  int err = 0;
  Pool &pool = const_cast<mz_replay *>(r)->pool;
  if (nblocks >= 4 && pool.T > 1) {        // (two blocks = batch 256 take 33 us on one thread: less than waking the pool costs)
    std::vector<int> errs((size_t)pool.T, 0);
    const int T = pool.T;
    pool.run([&](int tid) {
      for (int k = tid; k < nblocks && !errs[(size_t)tid]; k += T) errs[(size_t)tid] = block(128 * k);
    });
    for (int e : errs) if (e && !err) err = e;
  } else {
    for (int k = 0; k < nblocks && !err; ++k) err = block(128 * k);
  }
  if (err == 1) return fail("mzr_sample_batch: a draw hit an empty leaf (buffer smaller than the draw range?)");
  if (err == 2) return fail("mzr_sample_batch: history was ingested without payload");
  return 0;
}

// mzr_sample_batch with the stratified draws made here from the caller's generator words: `words` are 2 bs consecutive
// 32-bit Mersenne Twister outputs (random.getrandbits(64 bs), least significant word first); draw i is what
// random.uniform(seg i, seg (i + 1)) returns on them -- random.random() = ((a >> 5) 2^26 + (b >> 6)) / 2^53
// (_randommodule.c), uniform(lo, hi) = lo + (hi - lo) random() (random.py), seg = total_priority / bs
// (replay_buffer.py:136-140) -- in IEEE double, no contraction: the same doubles as the Python expressions.  Also out:
// probs [bs] = priority / total (replay_buffer.py:157), info[0] = the number of padded (-1) actions, info[1] = num_memories.
int mzr_sample_batch_words(const mz_replay *r, const uint32_t *words, int bs, float *obs, int32_t *actions, float *target_rewards,
                           float *target_values, float *target_policies, int64_t *idxs, double *probs, int64_t *info) {
  if (!r || !words || !probs || !info || bs < 1) return fail("mzr_sample_batch_words: bad argument");
  MZR_LOCK(r);
  drain(r);
  const double total = r->tree[0];
  const double seg = total / (double)bs;
  std::vector<double> draws((size_t)bs);
  for (int i = 0; i < bs; ++i) {
    const double a = (double)(words[2 * i] >> 5), b = (double)(words[2 * i + 1] >> 6);
    const double u = (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    const double lo = seg * (double)i, hi = seg * ((double)i + 1.0);
    draws[(size_t)i] = lo + (hi - lo) * u;
  }
  if (mzr_sample_batch(r, draws.data(), bs, obs, actions, target_rewards, target_values, target_policies, idxs, probs)) return -1;
  for (int i = 0; i < bs; ++i) probs[i] = probs[i] / total;
  int64_t pad = 0;
  const int K = r->c.num_unroll_steps;
  for (int i = 0; i < bs * K; ++i) pad += actions[i] < 0;
  info[0] = pad;
  info[1] = r->num_memories;
  return 0;
}

// n consecutive mzr_sample_batch_words calls in one (the learner samples a few batches ahead anyway, learners.py:124): batch j
// uses words [2 bs j, 2 bs (j + 1)) -- the generator words n consecutive sample_batch calls would consume -- and writes row
// block j of every output; info [n][2].
int mzr_sample_batches_words(const mz_replay *r, const uint32_t *words, int n, int bs, float *obs, int32_t *actions,
                             float *target_rewards, float *target_values, float *target_policies, int64_t *idxs, double *probs,
                             int64_t *info) {
  if (!r || n < 1) return fail("mzr_sample_batches_words: bad argument");
  MZR_LOCK(r);
  const size_t O = (size_t)r->c.obs_dim, A = (size_t)r->c.action_space, K = (size_t)r->c.num_unroll_steps, B = (size_t)bs;
  for (int j = 0; j < n; ++j) {
    const size_t o = (size_t)j * B;
    if (mzr_sample_batch_words(r, words + 2 * o, bs, obs + o * O, actions + o * K, target_rewards + o * (K + 1), target_values + o * (K + 1),
                               target_policies + o * (K + 1) * A, idxs + o, probs + o, info + 2 * j))
      return -1;
  }
  return 0;
}

// ---- the learner's side of the replay, natively (learners.py:115-153 calls sample_batch / update once per training step)

// PrioritizedReplay.get_priorities on the float32 errors the learner sends (learners.py:181-182: new_errors is a float32
// numpy array): numpy evaluates np.power(np.abs(errors) + epsilon, alpha) in FLOAT32 for a float32 array and Python-float
// epsilon / alpha (replay_buffer.py:110-111), so the refreshed leaves are float32 values -- not the doubles mzr_priorities
// computes for the lists of Python floats save_history passes.
int mzr_priorities_f32(const mz_replay *r, const float *errors, int64_t n, float *out) {
  if (!r || !errors || !out) return fail("mzr_priorities_f32: null argument");
  const float eps = (float)r->c.epsilon, alpha = (float)r->c.alpha;
  for (int64_t i = 0; i < n; ++i) {
    const float x = fabsf(errors[i]) + eps;
    out[i] = alpha == 1.f ? x : powf(x, alpha);
  }
  return 0;
}

// PrioritizedReplay.update (replay_buffer.py:200-203) for float32 errors: float32 priorities, leaf by leaf in arrival order
int mzr_update_errors_f32(mz_replay *r, const int64_t *idxs, const float *errors, int64_t n) {
  if (!r || !idxs || !errors || n < 0) return fail("mzr_update_errors_f32: bad argument");
  MZR_LOCK(r);
  std::vector<float> p32((size_t)n);
  std::vector<double> p64((size_t)n);
  if (mzr_priorities_f32(r, errors, n, p32.data())) return -1;
  for (int64_t i = 0; i < n; ++i) p64[(size_t)i] = (double)p32[(size_t)i];
  return mzr_tree_update(r, idxs, p64.data(), n);
}

// numpy's legacy global generator, np.random.get_state() = ('MT19937', key [624], pos, ...): 32-bit outputs
static inline uint32_t np_mt_next(uint32_t *key, int32_t *pos) {
  if (*pos >= 624) {
    int i;
    for (i = 0; i < 624 - 397; ++i) {
      const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
      key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 623; ++i) {
      const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    const uint32_t y = (key[623] & 0x80000000u) | (key[0] & 0x7fffffffu);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    *pos = 0;
  }
  uint32_t y = key[(*pos)++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// sample_batch COMPLETE for n consecutive batches (replay_buffer.py:124-163): mzr_sample_batches_words, then
//  * the padded actions (replay_buffer.py:150-151: np.random.randint(action_space) where the history ends before the
//    unroll does) drawn in element order from numpy's legacy generator, whose state the caller hands over
//    (np.random.get_state(): np_key [624], *np_pos) and takes back -- RandomState.randint(A) is a masked rejection on
//    32-bit outputs (numpy/random/_bounded_integers: buffered_bounded_masked_uint32), no draw for A = 1;
//  * beta += beta_increment up to 1 per batch (replay_buffer.py:131-132), is_weights = (N p)^-beta / max
//    (replay_buffer.py:157-159) in double.  pow() is the C library's: numpy's vectorised power may differ from it in the
//    last bit of a double (both are within an ulp; which one numpy uses depends on the host's SIMD level), the one output
//    of this call that is not bit for bit the reference's.
// is_weights [n][bs]; pads_out (may be null): total number of padded actions.
int mzr_sample_batches_full(mz_replay *r, const uint32_t *words, int n, int bs, float *obs, int32_t *actions, float *target_rewards,
                            float *target_values, float *target_policies, int64_t *idxs, double *is_weights, uint32_t *np_key,
                            int32_t *np_pos, double *beta_inout, int64_t *pads_out, uint32_t *py_key, int32_t *py_pos) {
  if (!r || (!words && (!py_key || !py_pos)) || !is_weights || !np_key || !np_pos || !beta_inout || n < 1 || bs < 1)
    return fail("mzr_sample_batches_full: bad argument");
  MZR_LOCK(r);
  // words == NULL: the generator words come from the state of Python's `random` module handed over (random.getstate()[1]:
  // key [624] + position): random.getrandbits(64 bs n) is 2 bs n consecutive MT19937 outputs, least significant word first
  std::vector<uint32_t> gen;
  if (!words) {
    gen.resize((size_t)2 * bs * n);
    for (size_t i = 0; i < gen.size(); ++i) gen[i] = np_mt_next(py_key, py_pos);
    words = gen.data();
  }
  const size_t O = (size_t)r->c.obs_dim, A = (size_t)r->c.action_space, K = (size_t)r->c.num_unroll_steps, B = (size_t)bs;
  uint32_t mask = (uint32_t)(A - 1);
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  int64_t pads = 0;
  for (int j = 0; j < n; ++j) {
    const size_t o = (size_t)j * B;
    int64_t info[2];
    double *w = is_weights + o;
    if (mzr_sample_batch_words(r, words + 2 * o, bs, obs + o * O, actions + o * K, target_rewards + o * (K + 1), target_values + o * (K + 1),
                               target_policies + o * (K + 1) * A, idxs + o, w, info))
      return -1;
    if (info[0]) {
      int32_t *act = actions + o * K;
      for (size_t i = 0; i < B * K; ++i)
        if (act[i] < 0) {
          uint32_t v = 0;
          if (A > 1) do { v = np_mt_next(np_key, np_pos) & mask; } while (v > (uint32_t)(A - 1));
          act[i] = (int32_t)v;
        }
      pads += info[0];
    }
    if (*beta_inout < 1.0) { const double b = *beta_inout + r->c.beta_increment_per_sampling; *beta_inout = b < 1.0 ? b : 1.0; }
    const double beta = *beta_inout, N = (double)info[1];
    double mx = 0.0;
    for (size_t i = 0; i < B; ++i) { w[i] = pow(N * w[i], -beta); if (w[i] > mx) mx = w[i]; }
    for (size_t i = 0; i < B; ++i) w[i] /= mx;
  }
  if (pads_out) *pads_out = pads;
  return 0;
}

// ------------------------------------------------------------------------------------------------ producer-side assembly
// One replay fed by N ranks (train --ranks N; reference train.py:71-72: every actor sends its history slices to THE replay buffer,
// actors.py:160-173): the per-ENVIRONMENT half of the ingest -- the open game buffers, the flush rules, history slicing, the
// priorities (|error| + epsilon)^alpha -- needs nothing of the replay and runs on the PRODUCING rank (mz_assembler: the very code the
// replay's own direct ingest runs); what travels through the rank's ring is the finished slices.  The replay's host then copies each
// slice once, sequentially, and inserts its leaves (until r05 it kept N x 4096 open buffers and appended 88-byte records into them
// round-robin: memory-latency bound, 7 GPUs' worth at four threads).
//
// blob: int64 count, then per slice int64 {m, b, n, keep, done, pad} | n x R float rows (padded to 8 bytes) | n double priorities.
struct mz_assembler {
  AsmCfg ac;
  int B = 0;
  std::vector<EnvGame> envs;
  Scratch sc;
  std::deque<Pending> ready;             // finished slices not yet taken, (chunk, move, env) order; pri owned below
  std::deque<std::vector<double>> ready_pri;
};

int mzr_asm_create(const mzr_config *cfg, int num_envs, mz_assembler **out) {
  if (!cfg || !out || num_envs < 1) return fail("mzr_asm_create: bad argument");
  if (cfg->episode_life) return fail("mzr_asm_create: records carry one end-of-game flag (no episode_life)");
  mz_assembler *a = new mz_assembler();
  const int OS = cfg->obs_u8 ? (cfg->obs_dim + 3) / 4 : cfg->obs_dim;
  a->ac = asm_cfg_of(*cfg, OS, OS + cfg->action_space + MZR_REC_EXTRA);
  a->B = num_envs;
  a->envs.resize((size_t)num_envs);
  *out = a;
  return 0;
}
int mzr_asm_destroy(mz_assembler *a) {
  if (!a) return 0;
  for (Pending &p : a->ready) delete p.h;
  delete a;
  return 0;
}
int mzr_asm_feed(mz_assembler *a, const float *records, int n_moves, int B, int rec_floats) {
  if (!a || !records) return fail("mzr_asm_feed: null argument");
  if (B != a->B || rec_floats != a->ac.R || n_moves < 0) return fail("mzr_asm_feed: bad shape (B %d of %d, rec %d of %d)", B, a->B, rec_floats, a->ac.R);
  ingest_range(a->ac, a->envs.data(), a->sc, records, n_moves, B, 0, 0, B, false);
  std::stable_sort(a->sc.pend.begin(), a->sc.pend.end(), [](const Pending &x, const Pending &y) { return x.m < y.m; });      // (move, env): b ascending already
  for (const Pending &p : a->sc.pend) {
    a->ready_pri.emplace_back(p.pri, p.pri + p.h->n);
    Pending q = p;
    q.pri = nullptr;                   // (the deque's vector is the owner; looked up by position)
    a->ready.push_back(q);
  }
  return 0;
}
int64_t mzr_asm_pending(const mz_assembler *a) { return a ? (int64_t)a->ready.size() : -1; }
// as many queued slices as fit `cap` bytes, oldest first, into out; -> bytes written (0: nothing queued), < 0: error
int64_t mzr_asm_take(mz_assembler *a, void *out, int64_t cap) {
  if (!a || !out || cap < 64) return fail("mzr_asm_take: bad argument");
  char *o = (char *)out;
  int64_t off = 8, count = 0;
  const size_t R = (size_t)a->ac.R;
  while (!a->ready.empty()) {
    const Pending &p = a->ready.front();
    const int64_t n = p.h->n;
    const int64_t rows_b = (int64_t)(((size_t)n * R * 4 + 7) & ~(size_t)7), need = 48 + rows_b + n * 8;
    if (off + need > cap) {
      if (count == 0) return fail("mzr_asm_take: a slice of %lld steps does not fit %lld bytes", (long long)n, (long long)cap);
      break;
    }
    int64_t hdr[6] = {p.m, p.b, n, p.keep, p.done ? 1 : 0, 0};
    memcpy(o + off, hdr, 48);
    memcpy(o + off + 48, p.h->rows.data() + (size_t)p.h->off * R, (size_t)n * R * 4);
    memcpy(o + off + 48 + rows_b, a->ready_pri.front().data(), (size_t)n * 8);
    off += need;
    ++count;
    delete p.h;
    a->ready.pop_front();
    a->ready_pri.pop_front();
  }
  if (count == 0) return 0;
  memcpy(o, &count, 8);
  return off;
}

// the slices of one blob (mzr_asm_take) enter the replay: one sequential copy per slice -- spread over the ingest threads --, then
// the (deferred) insertion of its leaves in blob order; env_base: the producing rank's first environment (kept for the error text)
int mzr_ingest_slices(mz_replay *r, const void *blob, int64_t bytes, int env_base) {
  if (!r || !blob || bytes < 8) return fail("mzr_ingest_slices: bad argument");
  if (r->c.episode_life) return fail("mzr_ingest_slices: this replay is configured with episode_life");
  // The copies run on the CALLING thread and WITHOUT the handle's lock (they touch nothing of the replay but the spare-buffer
  // list, which has a lock of its own): the host of the one replay drains several rings from several threads at once
  // (distributed.serve_rings) and only the hand-over to the inserter is serialised.  (With the copies under the lock and on the
  // handle's pool, two and four ingest threads accepted the same 118 M records/s: one caller at a time, a pool wake-up per blob.)
  const char *o = (const char *)blob;
  int64_t count = 0;
  memcpy(&count, o, 8);
  const size_t R = (size_t)r->R;
  Job job;
  job.items.resize((size_t)count);
  std::vector<const char *> rows((size_t)count);
  int64_t off = 8, total = 0;
  for (int64_t i = 0; i < count; ++i) {
    if (off + 48 > bytes) return fail("mzr_ingest_slices: truncated blob (rank with env_base %d)", env_base);
    int64_t hdr[6];
    memcpy(hdr, o + off, 48);
    const int64_t n = hdr[2];
    const int64_t rows_b = (int64_t)(((size_t)n * R * 4 + 7) & ~(size_t)7);
    if (n < 0 || off + 48 + rows_b + n * 8 > bytes) return fail("mzr_ingest_slices: truncated blob (rank with env_base %d)", env_base);
    rows[(size_t)i] = o + off + 48;
    Hist *h = new Hist();
    h->n = n; h->payload = true;
    job.items[(size_t)i] = Pending{(int)hdr[0], (int)hdr[1], h, hdr[3], nullptr, hdr[4] != 0};
    total += n;
    off += 48 + rows_b + n * 8;
  }
  job.pris.emplace_back((size_t)total);
  {
    int64_t po = 0;
    std::vector<std::vector<float>> bufs;
    {                                       // as many recycled row buffers as there are slices, in one visit of the list
      std::lock_guard<std::mutex> lk(r->spare_mu);
      const size_t take = r->hist_spare.size() < (size_t)count ? r->hist_spare.size() : (size_t)count;
      bufs.reserve(take);
      for (size_t k = 0; k < take; ++k) { bufs.push_back(std::move(r->hist_spare.back())); r->hist_spare.pop_back(); }
    }
    for (int64_t i = 0; i < count; ++i) {
      Pending &p = job.items[(size_t)i];
      const int64_t n = p.h->n;
      const int64_t rows_b = (int64_t)(((size_t)n * R * 4 + 7) & ~(size_t)7);
      std::vector<float> buf;
      if ((size_t)i < bufs.size()) buf.swap(bufs[(size_t)i]);
      if (buf.capacity() < (size_t)n * R) {
        buf.assign((const float *)rows[(size_t)i], (const float *)rows[(size_t)i] + (size_t)n * R);      // (a fresh buffer: the copy is its first touch)
      } else {
        buf.resize((size_t)n * R);
        copy_streaming(buf.data(), (const float *)rows[(size_t)i], (size_t)n * R);
      }
      p.h->rows.swap(buf);
      p.pri = job.pris[0].data() + po;
      memcpy(job.pris[0].data() + po, rows[(size_t)i] + rows_b, (size_t)n * 8);
      po += n;
    }
    _mm_sfence();                           // (this thread's streaming stores are globally visible before the slices are handed on)
  }
  MZR_LOCK(r);
  if (!r->inserter.joinable()) {
    insert_job(r, job);
    return 0;
  }
  {
    std::unique_lock<std::mutex> lk(r->qmu);
    r->idle_cv.wait(lk, [&] { return r->queue.size() < 4; });
    r->queue.push_back(std::move(job));
  }
  r->qcv.notify_one();
  return 0;
}

int64_t mzr_frames(const mz_replay *r) { MZR_LOCK(r); drain(r); return r->frames; }
int64_t mzr_games(const mz_replay *r) { MZR_LOCK(r); drain(r); return r->games; }
int mzr_add_initial_throughput(mz_replay *r, int64_t frames, int64_t games) {
  if (!r) return fail("mzr_add_initial_throughput: null");
  MZR_LOCK(r);
  drain(r);
  r->frames += frames; r->games += games;
  return 0;
}

// release store / acquire load of one int64 in shared memory: the head / tail counters of distributed.ShmRing
void mzr_store_release_i64(int64_t *p, int64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
int64_t mzr_load_acquire_i64(const int64_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

}  // extern "C"
