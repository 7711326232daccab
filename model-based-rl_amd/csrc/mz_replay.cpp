// mz_replay.cpp -- libmz_replay.so: host-side prioritized replay ingest (include/mz_replay.h).
// Plain C++17, no GPU code.  Arithmetic and update order follow the reference's SumTree /
// PrioritizedReplay (replay_buffer.py) exactly; see the header for the citations.
#include "../../include/mz_replay.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
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

// one HistorySlice (game.py:5-16) as flat arrays; leaves point into it as (history, step)
struct Hist {
  int64_t n = 0;
  int64_t refs = 0;             // leaves pointing at this slice (single-threaded handle: plain counter)
  std::vector<float> obs, child_visits, rewards;
  std::vector<double> root_values;
  std::vector<int32_t> actions;
  std::vector<uint8_t> dones;
  std::vector<int8_t> to_play;
};

// what Game/Actor keep per environment between flushes (game.py:54-77, actors.py:160-169)
struct EnvGame {
  std::vector<float> recs;      // records from absolute history index `base` on
  int64_t base = 0;             // history index of recs[0]
  int64_t history_idx = 0;      // game.history_idx
  int64_t previous_collect_to = 0;
  bool done_at_last_flush = false;
};

// a history slice built during an env-major ingest pass, waiting for its (move, env)-ordered tree insertion
struct Pending { int m, b; struct Hist *h; int64_t keep; size_t pri_off; bool done; };

struct mz_replay {
  mzr_config c;
  // SumTree (replay_buffer.py:8-17)
  int64_t max_capacity, capacity_step, capacity, prev_capacity = 0, num_memories = 0, position = 0;
  std::vector<double> tree;
  std::vector<Hist *> leaf_hist;   // owned through Hist::refs
  std::vector<int32_t> leaf_step;
  int64_t frames = 0, games = 0;
  std::vector<EnvGame> envs;
  // scratch
  std::vector<double> errs, pri, rootv, chg, pend_pri;
  std::vector<struct Pending> pend;
  std::vector<float> obs, cv, rew;
  std::vector<int32_t> act;
  std::vector<uint8_t> done;
  std::vector<int8_t> tp;
};

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
// one-leaf-at-a-time walk; the order between different nodes is free.  The upper levels, where the whole run
// sits under one node, are therefore summed in ONE pass over the leaves with one running sum per level
// (independent add chains instead of a store-to-load dependent walk per leaf); the lower levels walk runs of
// leaves that share a node with the sum in a register.
static void tree_update_run(mz_replay *r, int64_t first_idx, const double *priorities, int64_t n) {
  double *t = r->tree.data();
  if ((int64_t)r->chg.size() < n) r->chg.resize(n);
  double *chg = r->chg.data();
  for (int64_t i = 0; i < n; ++i) {
    chg[i] = priorities[i] - t[first_idx + i];
    t[first_idx + i] = priorities[i];
  }
  // 1-based heap numbering: node j has parent j >> 1.  Split where the leaf depth changes (j crosses a power of 2).
  for (int64_t a = 0; a < n;) {
    const uint64_t j0 = (uint64_t)(first_idx + a) + 1;
    const int depth = 63 - __builtin_clzll(j0);
    int64_t b = n;
    const uint64_t next_pow = (uint64_t)1 << (depth + 1);
    if (j0 + (uint64_t)(n - a) > next_pow) b = a + (int64_t)(next_pow - j0);
    const uint64_t j1 = (uint64_t)(first_idx + b - 1) + 1;      // last leaf of this sub-run
    int L = 1;
    for (; L <= depth && (j0 >> L) != (j1 >> L); ++L) {          // levels with several nodes under the run
      int64_t i = a;
      while (i < b) {
        const uint64_t node = ((uint64_t)(first_idx + i) + 1) >> L;
        double acc = t[node - 1];
        do { acc += chg[i]; ++i; } while (i < b && ((((uint64_t)(first_idx + i) + 1) >> L) == node));
        t[node - 1] = acc;
      }
    }
    if (L <= depth) {                                            // levels L..depth: one node each
      double acc[64];
      const int nl = depth - L + 1;
      for (int k = 0; k < nl; ++k) acc[k] = t[(j0 >> (L + k)) - 1];
      for (int64_t i = a; i < b; ++i) {
        const double c = chg[i];
        for (int k = 0; k < nl; ++k) acc[k] += c;
      }
      for (int k = 0; k < nl; ++k) t[(j0 >> (L + k)) - 1] = acc[k];
    }
    a = b;
  }
}

// SumTree.add, replay_buffer.py:19-32
static void tree_add(mz_replay *r, const double *priorities, int64_t n, Hist *h,
                     int64_t *positions_out) {
  int64_t run_start = 0;                       // [run_start, step) = leaves at consecutive positions, not yet summed
  int64_t run_idx = r->position + r->max_capacity - 1;
  for (int64_t step = 0; step < n; ++step) {
    {
      Hist *&slot = r->leaf_hist[r->position];
      if (slot && --slot->refs == 0) delete slot;
      slot = h;
      if (h) ++h->refs;
    }
    r->leaf_step[r->position] = (int32_t)step;
    if (positions_out) positions_out[step] = r->position;
    if (r->position >= r->prev_capacity) r->num_memories += 1;
    r->position = (r->position + 1) % r->capacity;
    if (r->position == 0) {
      tree_update_run(r, run_idx, priorities + run_start, step + 1 - run_start);
      run_start = step + 1;
      run_idx = r->max_capacity - 1;
      r->prev_capacity = r->capacity;
      const int64_t next = r->capacity + r->capacity_step;
      r->capacity = next < r->max_capacity ? next : r->max_capacity;
    }
  }
  if (run_start < n) tree_update_run(r, run_idx, priorities + run_start, n - run_start);
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
    const int O = r->c.obs_dim, A = r->c.action_space;
    if (obs) h->obs.assign(obs, obs + n * O);
    if (child_visits) h->child_visits.assign(child_visits, child_visits + n * A);
    if (root_values) h->root_values.assign(root_values, root_values + n);
    if (rewards) h->rewards.assign(rewards, rewards + n);
    if (actions) h->actions.assign(actions, actions + n);
    if (dones) h->dones.assign(dones, dones + n);
    if (to_play) h->to_play.assign(to_play, to_play + n);
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
  mz_replay *r = new mz_replay();
  r->c = *cfg;
  r->max_capacity = cfg->window_size;
  r->capacity_step = cfg->window_step;
  r->capacity = cfg->window_step;
  r->tree.assign((size_t)(2 * cfg->window_size - 1), 0.0);
  r->leaf_hist.assign((size_t)cfg->window_size, nullptr);
  r->leaf_step.assign((size_t)cfg->window_size, 0);
  *out = r;
  return 0;
}

int mzr_destroy(mz_replay *r) {
  if (r)
    for (Hist *&h : r->leaf_hist)
      if (h) { if (--h->refs == 0) delete h; h = nullptr; }
  delete r;
  return 0;
}

int mzr_priorities(const mz_replay *r, const double *errors, int64_t n, double *out) {
  if (!r || !errors || !out) return fail("mzr_priorities: null argument");
  for (int64_t i = 0; i < n; ++i) out[i] = pow(fabs(errors[i]) + r->c.epsilon, r->c.alpha);
  return 0;
}

int mzr_tree_add(mz_replay *r, const double *priorities, int64_t n, int64_t *positions_out) {
  if (!r || !priorities) return fail("mzr_tree_add: null argument");
  tree_add(r, priorities, n, nullptr, positions_out);
  return 0;
}

int mzr_tree_update(mz_replay *r, const int64_t *idxs, const double *priorities, int64_t n) {
  if (!r || !idxs || !priorities) return fail("mzr_tree_update: null argument");
  const int64_t len = 2 * r->max_capacity - 1;
  for (int64_t i = 0; i < n; ++i) {
    if (idxs[i] < 0 || idxs[i] >= len) return fail("mzr_tree_update: index %lld out of range", (long long)idxs[i]);
    tree_update(r, idxs[i], priorities[i]);
  }
  return 0;
}

// SumTree.get_leaf, replay_buffer.py:42-62 (returns the tree index of the leaf)
int64_t mzr_tree_get_leaf(const mz_replay *r, double value) {
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

double mzr_total_priority(const mz_replay *r) { return r->tree[0]; }
int64_t mzr_size(const mz_replay *r) { return r->num_memories; }
int mzr_tree_leaves(const mz_replay *r, int64_t n, double *out) {
  if (!r || !out || n > r->max_capacity) return fail("mzr_tree_leaves: bad argument");
  memcpy(out, r->tree.data() + r->max_capacity - 1, (size_t)n * sizeof(double));
  return 0;
}

int mzr_save_history(mz_replay *r, int64_t n, const double *errors, int64_t ignore, int terminal, const float *obs,
                     const float *child_visits, const double *root_values, const float *rewards,
                     const int32_t *actions, const uint8_t *dones, const int8_t *to_play) {
  if (!r || (n > 0 && !errors)) return fail("mzr_save_history: null argument");
  return save_history(r, n, errors, ignore, terminal, obs, child_visits, root_values, rewards, actions, dones, to_play);
}

static inline double rec_double(const float *p) {     // a float64 stored in two float slots (4-byte aligned)
  double v;
  memcpy(&v, p, sizeof v);
  return v;
}

int mzr_ingest_records(mz_replay *r, const float *records, int n_moves, int B, int rec_floats) {
  return mzr_ingest_records_from(r, records, n_moves, B, rec_floats, 0);
}

int mzr_ingest_records_from(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base) {
  if (!r || !records) return fail("mzr_ingest_records: null argument");
  if (n_moves < 0 || B < 1 || env_base < 0) return fail("mzr_ingest_records: bad shape (n_moves %d, B %d, env_base %d)", n_moves, B, env_base);
  const int O = r->c.obs_dim, A = r->c.action_space;
  if (rec_floats != O + A + MZR_REC_EXTRA)
    return fail("mzr_ingest_records: rec_floats %d != obs_dim+action_space+%d = %d", rec_floats, MZR_REC_EXTRA, O + A + MZR_REC_EXTRA);
  if ((int)r->envs.size() < env_base + B) r->envs.resize((size_t)env_base + B);
  const int64_t overlap = r->c.num_unroll_steps + r->c.td_steps;
  // Environment-major: an env's bookkeeping and the tail of its record buffer are touched once per call
  // instead of once per move (4096 envs x 3 cold cache lines per record were the bulk of the ingest time).  The
  // history slices are built as they fall due; their insertion into the sum tree is deferred and replayed in
  // (move, env) order, i.e. exactly the arrival order of a move-major walk.
  std::vector<Pending> &pend = r->pend;
  std::vector<double> &pris = r->pend_pri;
  pend.clear();
  pris.clear();
  const int PF = 6;       // envs ahead: an env's records lie n_moves strides of B * rec_floats apart (move-major ring),
  for (int b = 0; b < B; ++b) {      // which no hardware prefetcher follows -- request them while the envs before are handled
    if (b + PF < B)
      for (int m = 0; m < n_moves; ++m) {
        const char *q = (const char *)(records + ((size_t)m * B + b + PF) * rec_floats);
        __builtin_prefetch(q, 0, 1);
        __builtin_prefetch(q + 64, 0, 1);
      }
    EnvGame &g = r->envs[(size_t)env_base + b];
    for (int m = 0; m < n_moves; ++m) {
      const float *rec = records + ((size_t)m * B + b) * rec_floats;
      const int32_t *ri = (const int32_t *)(rec + O + A + 5);
      const bool done = ri[1] != 0;
      g.recs.insert(g.recs.end(), rec, rec + rec_floats);
      g.history_idx += 1;
      // actors.py:160-169
      const bool save = (g.history_idx - g.previous_collect_to) == r->c.max_history_length;
      if (!(save || done)) continue;
      const bool d_prev = g.previous_collect_to == 0 ? done : g.done_at_last_flush;   // dones[prev-1] (index -1 when prev == 0)
      int64_t collect_from = d_prev ? g.previous_collect_to
                                    : (g.previous_collect_to - overlap > 0 ? g.previous_collect_to - overlap : 0);
      const int64_t n = g.history_idx - collect_from;
      const int64_t ignore = done ? -1 : overlap;
      // PrioritizedReplay.save_history (replay_buffer.py:113-122) with the HistorySlice built straight from the
      // env's records (same arithmetic as save_history() above, one copy less)
      {
        Hist *h = new Hist();
        h->n = n;
        h->obs.resize((size_t)n * O); h->child_visits.resize((size_t)n * A); h->root_values.resize((size_t)n);
        h->rewards.resize((size_t)n); h->actions.resize((size_t)n); h->dones.resize((size_t)n); h->to_play.assign((size_t)n, 1);
        int64_t keep = n;
        if (ignore >= 0) keep = ignore == 0 ? 0 : (n - ignore > 0 ? n - ignore : 0);
        const size_t off = pris.size();
        pris.resize(off + (size_t)n);
        double *pri = pris.data() + off;
        const float *q = g.recs.data() + (size_t)(collect_from - g.base) * rec_floats;
        for (int64_t i = 0; i < n; ++i, q += rec_floats) {
          const int32_t *qi = (const int32_t *)(q + O + A + 5);
          memcpy(h->obs.data() + i * O, q, O * sizeof(float));
          memcpy(h->child_visits.data() + i * A, q + O, A * sizeof(float));
          h->root_values[i] = rec_double(q + O + A); h->rewards[i] = q[O + A + 4];
          h->actions[i] = qi[0]; h->dones[i] = (uint8_t)(qi[1] != 0);
          const double e = fabs(rec_double(q + O + A + 2)) + r->c.epsilon;
          pri[i] = r->c.alpha == 1.0 ? e : pow(e, r->c.alpha);
        }
        pend.push_back(Pending{m, b, h, keep, off, done});
      }
      g.previous_collect_to = g.history_idx;
      g.done_at_last_flush = done;
      if (done) {           // terminal: run_selfplay starts a new Game (actors.py:94-97)
        g.recs.clear(); g.base = 0; g.history_idx = 0; g.previous_collect_to = 0; g.done_at_last_flush = false;
      } else {              // keep only what the next slice can still reach back to
        const int64_t nb = g.history_idx - overlap > 0 ? g.history_idx - overlap : 0;
        if (nb > g.base) {
          g.recs.erase(g.recs.begin(), g.recs.begin() + (size_t)(nb - g.base) * rec_floats);
          g.base = nb;
        }
      }
    }
  }
  std::stable_sort(pend.begin(), pend.end(), [](const Pending &x, const Pending &y) { return x.m < y.m; });   // b already ascending
  for (const Pending &p : pend) {
    tree_add(r, pris.data() + p.pri_off, p.keep, p.h, nullptr);
    if (p.h->refs == 0) delete p.h;
    r->frames += p.keep;
    if (p.done) r->games += 1;
  }
  return 0;
}

int mzr_sample_batch(const mz_replay *r, const double *draws, int bs, float *obs, int32_t *actions,
                     float *target_rewards, float *target_values, float *target_policies, int64_t *idxs,
                     double *priorities) {
  if (!r || !draws || !obs || !actions || !target_rewards || !target_values || !target_policies || !idxs || !priorities)
    return fail("mzr_sample_batch: null argument");
  const int O = r->c.obs_dim, A = r->c.action_space, K = r->c.num_unroll_steps, td = r->c.td_steps;
  const int TL = K + 1;
  // replay_buffer.py:81-82: discounts as float32, discount**td as a Python float
  float disc[256];
  if (K + td > 256) return fail("mzr_sample_batch: num_unroll_steps + td_steps too large");
  for (int n = 0; n < K + td; ++n) disc[n] = (float)pow(r->c.discount, (double)n);
  const double disc_td = pow(r->c.discount, (double)td);
  for (int i = 0; i < bs; ++i) {
    const int64_t idx = mzr_tree_get_leaf(r, draws[i]);                       // replay_buffer.py:142
    const int64_t pos = idx - r->max_capacity + 1;
    const Hist *h = r->leaf_hist[(size_t)pos];
    if (!h) return fail("mzr_sample_batch: draw %d hit an empty leaf (buffer smaller than the draw range?)", i);
    const int64_t step = r->leaf_step[(size_t)pos];
    idxs[i] = idx;
    priorities[i] = r->tree[(size_t)idx];
    if (h->obs.empty() || h->child_visits.empty() || h->root_values.empty())
      return fail("mzr_sample_batch: history was ingested without payload");
    memcpy(obs + (size_t)i * O, h->obs.data() + (size_t)step * O, O * sizeof(float));   // 147
    for (int k = 0; k < K; ++k)                                                        // 149-152
      actions[(size_t)i * K + k] = (step + k < (int64_t)h->actions.size()) ? h->actions[(size_t)(step + k)] : -1;
    // insert_target, replay_buffer.py:165-198
    const int64_t end_index = (int64_t)h->root_values.size();
    const int64_t n_rewards = (int64_t)h->rewards.size();
    for (int j = 0; j < TL; ++j) {
      const int64_t cur = step + j;
      const float last_reward = (cur > 0 && cur <= n_rewards) ? h->rewards[(size_t)(cur - 1)] : 0.f;
      float *pol = target_policies + ((size_t)i * TL + j) * A;
      if (cur < end_index) {
        const int tp = h->to_play[(size_t)cur];
        const int64_t boot = cur + td;
        double value = boot < end_index ? h->root_values[(size_t)boot] * disc_td : 0.0;
        const int64_t hi = boot < n_rewards ? boot : n_rewards;
        if (hi > cur) {
          float acc = 0.f;
          for (int64_t q = cur; q < hi; ++q) {
            const float rw = (h->to_play[(size_t)q] != tp) ? -h->rewards[(size_t)q] : h->rewards[(size_t)q];
            acc += rw * disc[q - cur];
          }
          value += (double)acc;
        }
        memcpy(pol, h->child_visits.data() + (size_t)cur * A, A * sizeof(float));
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
}

int64_t mzr_frames(const mz_replay *r) { return r->frames; }
int64_t mzr_games(const mz_replay *r) { return r->games; }
int mzr_add_initial_throughput(mz_replay *r, int64_t frames, int64_t games) {
  if (!r) return fail("mzr_add_initial_throughput: null");
  r->frames += frames; r->games += games;
  return 0;
}

}  // extern "C"
