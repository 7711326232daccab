/*
 * mz_replay.h -- C ABI of the host-side prioritized replay (libmz_replay.so, plain C++17, no GPU code).
 *
 * Replaces the ingest side of the reference's PrioritizedReplay / SumTree (replay_buffer.py:6-122) and the
 * actor's history flush logic (actors.py:160-173, game.py:41-51,123-126) for experience records that arrive
 * from the GPU in bulk (mz_selfplay_drain, include/mz_engine.h) instead of as pickled HistorySlices over Ray.
 * Arithmetic follows the reference exactly: priority = (|error| + epsilon)^alpha in double, SumTree.update
 * propagates `change` leaf-to-root one leaf at a time in arrival order, so tree sums are bit-identical to the
 * reference's for the same sequence of histories.
 *
 * Returns 0 on success, <0 on error (message: mzr_last_error(), per calling thread).  One caller at a time per handle (the
 * reference's replay is a Ray actor, replay_buffer.py:69: calls are serialised): calls from different threads take turns on a
 * lock inside the handle; mzr_ingest_records* fans the environments of ONE call out over the handle's own ingest threads and
 * joins them before it returns.
 */
#ifndef MZ_REPLAY_H
#define MZ_REPLAY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mz_replay mz_replay;

typedef struct mzr_config {
  int64_t window_size;        /* --window_size  (SumTree max_capacity) */
  int64_t window_step;        /* --window_step or window_size when None (replay_buffer.py:94-98) */
  int32_t obs_dim, action_space;
  int32_t num_unroll_steps;   /* K */
  int32_t td_steps;
  int32_t max_history_length; /* --max_history_length (actors.py:160) */
  int32_t batch_size;
  double epsilon, alpha, beta, beta_increment_per_sampling;   /* replay_buffer.py:73-77 */
  double discount;
  uint64_t seed;
  int32_t two_players;        /* --two_players: histories carry to_play = +-1 and targets flip signs (replay_buffer.py:187-189) */
  int32_t episode_life;       /* --episode_life: `terminal` (end of game) differs from `done` (game.py:90) */
  int32_t ingest_threads;     /* threads mzr_ingest_records* splits the environments of a call over (0 or 1: the caller only) */
  int32_t obs_u8;             /* observations are bytes (image / -ram- frames, game.py:93-96 keeps the raw uint8 observation): a record
                               * carries them packed, 4 per float slot -- rec_floats = ceil(obs_dim / 4) + action_space + MZR_REC_EXTRA;
                               * mzr_save_history still takes float32 observations (values 0..255), mzr_sample_batch returns float32 */
} mzr_config;

const char *mzr_last_error(void);

/* PrioritizedReplay.__init__ (replay_buffer.py:71-106) */
int mzr_create(const mzr_config *cfg, mz_replay **out);
int mzr_destroy(mz_replay *r);

/* PrioritizedReplay.get_priorities (replay_buffer.py:110-111) */
int mzr_priorities(const mz_replay *r, const double *errors, int64_t n, double *out);

/* SumTree.add / update / get_leaf / total_priority (replay_buffer.py:19-66); positions_out may be NULL */
int mzr_tree_add(mz_replay *r, const double *priorities, int64_t n, int64_t *positions_out);
int mzr_tree_update(mz_replay *r, const int64_t *idxs, const double *priorities, int64_t n);
int64_t mzr_tree_get_leaf(const mz_replay *r, double value);
/* The payload SumTree.get_leaf returns beside the index and the priority (replay_buffer.py:58-62: `step, history =
 * self.buffer[buffer_index]`): mzr_leaf_info -> the leaf's priority, its step inside its history slice (-1: empty leaf),
 * the slice's length and whether it carries a payload; mzr_leaf_history copies the slice's n_steps record rows
 * ([n_steps][rec_floats], the layout of mzr_ingest_records; byte observations stay packed) to rows_out [host]. */
int mzr_leaf_info(const mz_replay *r, int64_t idx, double *priority, int64_t *step, int64_t *n_steps, int *has_payload);
int mzr_leaf_history(const mz_replay *r, int64_t idx, float *rows_out, int64_t n_steps);
double mzr_total_priority(const mz_replay *r);
int64_t mzr_size(const mz_replay *r);                  /* PrioritizedReplay.size: tree.num_memories */
int mzr_tree_leaves(const mz_replay *r, int64_t n, double *out);

/* PrioritizedReplay.save_history (replay_buffer.py:113-122) for one history slice given as arrays:
 * n steps, errors[n]; ignore < 0 means None.  The step payload (obs [n][O], child_visits [n][A], root_values,
 * rewards, actions, dones, to_play) is copied into the replay's own storage.  Any payload pointer may be NULL
 * (priorities-only ingest). */
int mzr_save_history(mz_replay *r, int64_t n, const double *errors, int64_t ignore, int terminal,
                     const float *obs, const float *child_visits, const double *root_values, const float *rewards,
                     const int32_t *actions, const uint8_t *dones, const int8_t *to_play);

/* Bulk ingest of device records (layout of mz_selfplay_drain: [n_moves][B][rec_floats], rec = obs[O],
 * child_visits[A] (float32), root_value and error as float64 in two float slots each (the reference keeps both as
 * Python floats: actors.py:147-148, game.py:112 -- priorities and value targets built from records are therefore
 * the reference's doubles), reward (float32), then int32 bits action, flags, step, env_id, episode:
 * rec_floats = obs_dim + action_space + MZR_REC_EXTRA.
 * Re-creates per environment what Actor.play_game does after each move (actors.py:160-173): histories are
 * accumulated per env and flushed to save_history when max_history_length steps were collected (with the
 * overlap/ignore rules) or the episode is done.  frames/games: PrioritizedReplay.throughput.
 * The flags word of a record carries `done` (bit 0) and the mover (bit 1 set = to_play -1; History.to_play,
 * game.py:100-101 -- n-step targets flip the sign of the other player's rewards by it, replay_buffer.py:187-189); the
 * single-player device loops leave bit 1 clear (to_play = +1), the device TicTacToe loop sets it.  `done` ends the game
 * (terminal == done): a replay created with episode_life (terminal != done, game.py:90) REFUSES records (-1,
 * mzr_last_error) -- such histories go through mzr_save_history, which takes `terminal` explicitly.
 * Threads: the environments of a call are split into contiguous ranges over mzr_config.ingest_threads threads (history
 * assembly and priorities are per environment, actors.py:160-173); the finished slices enter the one sum tree on the
 * calling thread in (move, environment) order -- leaves, sums, counters and sample batches are bit-identical for every
 * thread count (replay_buffer.py:19-40 adds in arrival order).  With more than one ingest thread that insertion is
 * DEFERRED to the handle's inserter thread (jobs in call order): a call returns once its records are copied into the
 * environments' histories -- the record buffer may be reused -- and the next chunk's assembly overlaps this chunk's
 * insertion; every other entry point first waits for the pending insertions, so nothing observable changes.
 * mzr_ingest_records_from: the B environments of this call are the replay's environments env_base .. env_base+B-1
 * (one replay fed by several actor ranks: rank r passes env_base = r * B; actors.py:169 -- every reference actor
 * sends to the ONE replay buffer, train.py:71-72). */
#define MZR_REC_EXTRA 10
int mzr_ingest_records(mz_replay *r, const float *records, int n_moves, int B, int rec_floats);
int mzr_ingest_records_from(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base);
/* One replay fed by many ranks (train --ranks N): the per-record work of the replay's host that does NOT need the replay is done
 * by the PRODUCING rank -- mzr_pack_env_major turns a chunk [n_moves][B][rec] into [B][n_moves][rec] while it is copied into the
 * shared-memory ring (distributed.ShmRing.put), and mzr_ingest_records_packed takes such a chunk: every environment's rows of the
 * call are one sequential piece, appended run by run.  Same result as mzr_ingest_records_from on the unpacked chunk, bit for bit. */
/* ... and the per-ENVIRONMENT half of the ingest itself (actors.py:160-173: the open game buffers, the flush rules, history slicing;
 * replay_buffer.py:110-111: the priorities) runs on the producing rank: an mz_assembler holds what Game / Actor keep per environment,
 * mzr_asm_feed takes the device loop's chunks ([n_moves][B][rec]), mzr_asm_take writes the slices that fell due, oldest first, as a
 * blob of at most `cap` bytes (-> bytes, 0 = none queued; mzr_asm_pending: slices still queued), and the replay's host takes a blob
 * with mzr_ingest_slices: one sequential copy per slice, then the insertion of its leaves.  The replay ends in the state
 * mzr_ingest_records_from leaves on the same chunks, bit for bit. */
typedef struct mz_assembler mz_assembler;
int mzr_asm_create(const mzr_config *cfg, int num_envs, mz_assembler **out);
int mzr_asm_destroy(mz_assembler *a);
int mzr_asm_feed(mz_assembler *a, const float *records, int n_moves, int B, int rec_floats);
int64_t mzr_asm_pending(const mz_assembler *a);
int64_t mzr_asm_take(mz_assembler *a, void *out, int64_t cap);
int mzr_ingest_slices(mz_replay *r, const void *blob, int64_t bytes, int env_base);
void mzr_pack_env_major(const float *src, float *dst, int n_moves, int B, int rec_floats);
int mzr_ingest_records_packed(mz_replay *r, const float *records, int n_moves, int B, int rec_floats, int env_base);

/* PrioritizedReplay.sample_batch (replay_buffer.py:124-163) + insert_target (165-198) for `bs` stratified
 * draws.  draws[i] is the value the reference obtains from random.uniform(seg*i, seg*(i+1)) (the caller owns
 * the RNG; seg = total_priority / bs).  Outputs: obs [bs][O]; actions [bs][K] with -1 where the reference
 * pads with np.random.randint (history shorter than K after `step`; the caller fills them in element order);
 * target_rewards / target_values [bs][K+1] float32; target_policies [bs][K+1][A] float32; idxs [bs] tree
 * indices (what `update` takes back); priorities [bs].  n-step value targets: root_values[i+td] * discount^td
 * + sum of float32 rewards, sign-flipped where to_play differs, times float32 discounts (in float32, then
 * added in double); absorbing steps past the end of the history get zero policy / value. */
int mzr_sample_batch(const mz_replay *r, const double *draws, int bs, float *obs, int32_t *actions,
                     float *target_rewards, float *target_values, float *target_policies, int64_t *idxs,
                     double *priorities);
/* mzr_sample_batch with the draws made inside from the caller's generator words (2 bs consecutive 32-bit Mersenne
 * Twister outputs = random.getrandbits(64 bs), least significant first): bit for bit the reference's
 * random.uniform(seg i, seg (i + 1)) loop (replay_buffer.py:136-140).  Also: probs [bs] = priority / total_priority
 * (replay_buffer.py:157), info[0] = number of -1 (to be padded) actions, info[1] = num_memories. */
int mzr_sample_batch_words(const mz_replay *r, const uint32_t *words, int bs, float *obs, int32_t *actions, float *target_rewards,
                           float *target_values, float *target_policies, int64_t *idxs, double *probs, int64_t *info);
/* n consecutive mzr_sample_batch_words calls in one: batch j consumes words [2 bs j, 2 bs (j + 1)) and fills row block j of
 * every output (obs [n][bs][O], ...); info [n][2]. */
int mzr_sample_batches_words(const mz_replay *r, const uint32_t *words, int n, int bs, float *obs, int32_t *actions,
                             float *target_rewards, float *target_values, float *target_policies, int64_t *idxs, double *probs,
                             int64_t *info);

/* The learner's side, complete (learners.py:115-153 calls sample_batch and update once per training step; mz_fcl_run in
 * include/mz_engine.h drives these from native code).
 * mzr_priorities_f32 / mzr_update_errors_f32: get_priorities / update (replay_buffer.py:110-111,200-203) for the FLOAT32 errors
 *   the learner sends (learners.py:181-182): numpy evaluates (|e| + epsilon) ** alpha in float32 for a float32 array, the
 *   refreshed leaves are float32 values (mzr_priorities is the float64 form of save_history's Python-float lists).
 * mzr_sample_batches_full: n consecutive sample_batch calls with everything the reference does inside them: the stratified
 *   draws from the caller's generator words (mzr_sample_batches_words), the padded actions from numpy's legacy global
 *   generator (replay_buffer.py:150-151; its state np_key [624] / *np_pos = np.random.get_state()[1:3], handed back
 *   advanced), the beta schedule (*beta_inout) and is_weights [n][bs] = (N p)^-beta / max (replay_buffer.py:131-132,157-159;
 *   the C library's pow: within one unit in the last place of numpy's, whose own power differs between hosts).  words NULL: the
 *   words are generated here from the state of Python's `random` generator (py_key [624] / *py_pos = random.getstate()[1], the same
 *   MT19937; advanced in place) -- what random.getrandbits(64 bs n) would have returned.
 * All mzr_* entry points of one handle may be called from different threads: they take turns on a lock inside the handle. */
int mzr_priorities_f32(const mz_replay *r, const float *errors, int64_t n, float *out);
int mzr_update_errors_f32(mz_replay *r, const int64_t *idxs, const float *errors, int64_t n);
int mzr_sample_batches_full(mz_replay *r, const uint32_t *words, int n, int bs, float *obs, int32_t *actions, float *target_rewards,
                            float *target_values, float *target_policies, int64_t *idxs, double *is_weights, uint32_t *np_key,
                            int32_t *np_pos, double *beta_inout, int64_t *pads_out, uint32_t *py_key, int32_t *py_pos);

/* number of ingest threads of the handle (mzr_config.ingest_threads at creation; the setter re-creates the pool) */
int mzr_set_ingest_threads(mz_replay *r, int threads);
int mzr_ingest_threads(const mz_replay *r);

int64_t mzr_frames(const mz_replay *r);   /* throughput['frames'] (replay_buffer.py:121) */
int64_t mzr_games(const mz_replay *r);    /* throughput['games'] */
int mzr_add_initial_throughput(mz_replay *r, int64_t frames, int64_t games);   /* replay_buffer.py:106-108 */

/* One int64 in memory shared between processes, written with release and read with acquire ordering: the head / tail
 * counters of the single-producer / single-consumer record rings between actor ranks and the replay rank
 * (distributed.ShmRing; the reference moves HistorySlices through Ray's object store, actors.py:169).  The producer copies a
 * chunk, then store-releases the head; the consumer load-acquires the head before it reads the chunk. */
void mzr_store_release_i64(int64_t *p, int64_t v);
int64_t mzr_load_acquire_i64(const int64_t *p);

#ifdef __cplusplus
}
#endif
#endif
