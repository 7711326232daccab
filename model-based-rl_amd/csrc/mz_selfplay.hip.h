// mz_selfplay.hip.h -- the device-resident part of Actor.play_game (reference actors.py:126-176) for B
// synthetic fixed-length environments: observation generation, Dirichlet noise, env step and the
// experience record (what Game.apply / store_search_statistics append to History, game.py:79-115).
#pragma once
#include "mz_common.h"
#include "mz_rng.h"
#include "mz_tree.hip.h"

struct SelfplayState {
  int episode_len;
  int rec_floats, ring_moves;
  int32_t *t;            // [B] env._elapsed_steps
  int32_t *episode;      // [B] episodes finished
  float *obs;            // [Bp][O] current observation (History.observations[-1])
  double *temp;          // [B] visit-softmax temperature of the env's CURRENT game (actors.py:128-129: evaluated once per game)
  double *temp_next;     // [1] temperature the next game of every env starts with (mz_selfplay_set_temperature)
  const float *obs_min, *obs_rng;   // [O] --norm_obs: network input = (obs - min) / range (actors.py:55-58,134-137); null = raw
  int obs_u8;            // synthetic observations are uint8-valued (the -ram- envs: 128 bytes of console RAM), else ~N(0,1);
                         // 2: ... and the experience record carries them as BYTES, four per float slot (obs_slots = ceil(O / 4))
  int obs_slots;         // float slots the observation takes in a record: O, or ceil(O / 4) when packed
  int export_trees;      // write the searched trees back to the global pool at the end of every move (tests / tree export)
  double *noise_log;     // [ring_moves][B][A] or null: every move's Dirichlet draw, kept per move (mz_selfplay_noise_log: the
                         // parity tests replay the moves of a whole-moves launch on the CPU with the device's own draws)
  int32_t *action;       // [B]
  double *child_visits;  // [B][A]
  double *root_value, *error;   // [B]
  unsigned long long *movecnt;   // [B] moves completed by env b (all equal; per-env so that every thread reads and
                                 // advances only its own counter): keys the RNG and the ring slot
  // ---- a real game as the environment (mz_selfplay_set_env): TicTacToe, custom_environments/tic_tac_toe.py:5-76
  int env_kind;          // 0 = synthetic fixed-length episodes (above), 1 = TicTacToe (two players, 9 cells, 9 actions)
  int8_t *board;         // [B][9] env.board: 0 empty, +1 / -1 the two players' marks
  int8_t *turn;          // [B] env.turn == game.to_play: +1 or -1, the player about to move
  uint8_t *legal;        // [B][A] legal_actions() of the current position as a mask (actors.py:141)
  int8_t *to_play;       // [B] game.to_play of the current move (root.expand's to_play, actors.py:142)
  const double *draw_uniform;   // [B] or null: the uniform select_action consumes, given by the host (parity runs,
                                // mz_selfplay_set_draws); null = the device RNG keyed (seed, env, move)
  int draws_noise;       // != 0: the Dirichlet draw of the coming moves is the one the host put into TreeView::noise
  float *ring;           // [ring_moves][B][rec_floats]
  float *host_ring;      // pinned staging for drains (optional)
  unsigned long long moves_host, drained;
  int env_offset;
  bool ready;
};

// Gamma(alpha) by Marsaglia-Tsang (alpha < 1: boost with U^(1/alpha)); counter-based draws.  float32 on the hardware's
// own log2 / exp2 / cos / sqrt: this is a noise source, not a parity quantity (parity runs pass numpy's draw in), and
// the float64 libm version of the same sampler cost 5.5 us of every move (a quarter of the root kernel).
__device__ inline float mz_u01f(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }      // [0, 1)
__device__ inline double mz_gamma(double alpha, uint64_t seed, uint32_t env, uint64_t move, uint32_t a) {
  const float al = (float)alpha;
  const float aa = al < 1.0f ? al + 1.0f : al;
  const float d = aa - 1.0f / 3.0f, c = 1.0f / __builtin_amdgcn_sqrtf(9.0f * d);
  const float LN2 = 0.69314718f;
  const uint32_t c1 = (uint32_t)move, c3 = (MZ_RNG_DIRICHLET << 24) | ((uint32_t)(move >> 32) & 0xFFFFFFu);
  float g = d;
  uint32_t ctr = 0;
  for (int it = 0; it < 64; ++it) {
    const mz_u4 r0 = mz_philox(seed, env, c1, a | (ctr++ << 8), c3);
    const float u1 = 1.0f - mz_u01f(r0.x), u2 = mz_u01f(r0.y);                       // u1 in (0,1]
    const float x = __builtin_amdgcn_sqrtf(-2.0f * LN2 * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);   // cos(2 pi u2)
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    const float u = 1.0f - mz_u01f(r0.z);
    if (LN2 * __builtin_amdgcn_logf(u) < 0.5f * x * x + d - d * v + d * LN2 * __builtin_amdgcn_logf(v)) { g = d * v; break; }
  }
  if (al < 1.0f) {
    const mz_u4 r = mz_philox(seed, env, c1, a | (0xFFFFFFu << 8), c3);
    g *= __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(1.0f - mz_u01f(r.x)) / al);
  }
  return (double)g;
}

// noise[b] ~ Dirichlet(alpha * 1_legal)  (the draw of np.random.dirichlet in mcts.py:59, from the
// device RNG instead of numpy's global stream; parity runs pass numpy's draw in instead)
static __global__ void k_dirichlet(TreeView t, const uint8_t *legal, double alpha, uint64_t seed, uint64_t move_val,
                            const unsigned long long *move_ptr, int env_offset) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  const uint64_t move = move_ptr ? (uint64_t)*move_ptr : move_val;
  const int A = t.A;
  double g[MZ_MAX_ACTIONS_K];
  double sum = 0.0;
  int nlegal = 0;
  for (int a = 0; a < A; ++a) {
    const bool ok = legal ? legal[(size_t)b * A + a] != 0 : true;
    g[a] = ok ? mz_gamma(alpha, seed, (uint32_t)(env_offset + b), move, (uint32_t)a) : 0.0;
    sum += g[a];
    nlegal += ok;
  }
  for (int a = 0; a < A; ++a) {
    const bool ok = legal ? legal[(size_t)b * A + a] != 0 : true;
    t.noise[(size_t)b * A + a] = ok ? (sum > 0.0 ? g[a] / sum : 1.0 / nlegal) : 0.0;
  }
}

// Experience record of one move: obs[O], child_visits[A] (float32), root_value and error as float64 (two float
// slots each: the reference keeps both as Python floats, actors.py:147-148, game.py:112), reward (float32), then int32
// bit patterns: action, flags (bit 0 done, bit 1 to_play == -1: game.py:100-101 appends the mover), step (pre-step),
// env_id, episode.
#ifndef MZ_REC_EXTRA
#define MZ_REC_EXTRA 10     // (include/mz_engine.h)
#endif
__device__ __forceinline__ void mz_rec_put_double(float *dst, double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  ((uint32_t *)dst)[0] = (uint32_t)u; ((uint32_t *)dst)[1] = (uint32_t)(u >> 32);
}

// float slot k of a record's observation: obs[k], or -- packed byte observations (obs_u8 == 2) -- the bytes obs[4k .. 4k + 3]
// (History keeps the raw uint8 observation, game.py:93-96; the values are exact small integers in float32)
__device__ __forceinline__ float mz_rec_obs_slot(const SelfplayState &sp, const float *obs, int O, int k) {
  if (sp.obs_u8 != 2) return obs[k];
  uint32_t w = 0;
  for (int j = 0; j < 4; ++j)
    if (4 * k + j < O) w |= ((uint32_t)obs[4 * k + j] & 0xFFu) << (8 * j);
  return __builtin_bit_cast(float, w);
}

// Game.apply (game.py:79-104) on the synthetic env + the experience record of this move.
static __global__ void k_env_step_record(TreeView tv, SelfplayState sp, int B, int O, int A, uint64_t seed) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const unsigned long long move = sp.movecnt[b];
  sp.movecnt[b] = move + 1ull;
  // Config.select_action + store_search_statistics + root error for this tree (same code as mz_finalize)
  mz_finalize_tree(tv, b, sp.temp, nullptr, seed, move, sp.env_offset, sp.action, sp.child_visits, sp.root_value,
                   sp.error, nullptr);
  float *rec = sp.ring + ((size_t)(move % (unsigned long long)sp.ring_moves) * B + b) * sp.rec_floats;
  const int t = sp.t[b], ep = sp.episode[b];
  const int OS = sp.obs_slots;
  for (int k = 0; k < OS; ++k) rec[k] = mz_rec_obs_slot(sp, sp.obs + (size_t)b * O, O, k);
  for (int a = 0; a < A; ++a) rec[OS + a] = (float)sp.child_visits[(size_t)b * A + a];
  const float reward = mz_synth_reward(seed, (uint32_t)(sp.env_offset + b), (uint32_t)ep, (uint32_t)t);
  const int done = (t + 1 >= sp.episode_len) ? 1 : 0;
  mz_rec_put_double(rec + OS + A + 0, sp.root_value[b]);
  mz_rec_put_double(rec + OS + A + 2, sp.error[b]);
  rec[OS + A + 4] = reward;
  int32_t *ri = (int32_t *)(rec + OS + A + 5);
  ri[0] = sp.action[b]; ri[1] = done; ri[2] = t; ri[3] = sp.env_offset + b; ri[4] = ep;
  if (done) {      // the next game starts: its temperature is evaluated now (actors.py:128-129)
    sp.t[b] = 0; sp.episode[b] = ep + 1; sp.temp[b] = *sp.temp_next;
  } else {
    sp.t[b] = t + 1;
  }
}


// ---- TicTacToe on the device (custom_environments/tic_tac_toe.py:5-76; the reference's only two-player environment)
// What Actor.play_game reads before a move (actors.py:134-142): observation = turn * board as float32
// (tic_tac_toe.py:24,50 + actors.py:134), the legal actions (tic_tac_toe.py:27-28) and game.to_play.
static __global__ void k_ttt_observe(SelfplayState sp, int B, int A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int turn = sp.turn[b];
  for (int k = 0; k < 9; ++k) {
    const int c = sp.board[(size_t)b * 9 + k];
    sp.obs[(size_t)b * 9 + k] = (float)(turn * c);
    sp.legal[(size_t)b * A + k] = c == 0 ? 1 : 0;
  }
  sp.to_play[b] = (int8_t)turn;
}

// Game.apply (game.py:79-104) on env.step (tic_tac_toe.py:30-51) for environment b, by ONE lane, and the tail of its
// experience record (root value, error, reward, action, flags with the mover, step, env id, episode).
__device__ __forceinline__ void mz_ttt_apply(const SelfplayState &sp, int b, int action, double root_value, double error,
                                             float *rec, int A) {
  const int O = 9;
  const int turn = sp.turn[b], t = sp.t[b], ep = sp.episode[b];
  int8_t *bd = sp.board + (size_t)b * 9;
  bd[action] = (int8_t)turn;
  const int r0 = 3 * (action / 3), c0 = action % 3;
  bool won = (bd[r0] + bd[r0 + 1] + bd[r0 + 2] == 3 * turn) || (bd[c0] + bd[c0 + 3] + bd[c0 + 6] == 3 * turn);
  if (action % 4 == 0) won = won || (bd[0] + bd[4] + bd[8] == 3 * turn);
  if (action == 2 || action == 4 || action == 6) won = won || (bd[2] + bd[4] + bd[6] == 3 * turn);
  const int done = (won || t == 8) ? 1 : 0;           // tic_tac_toe.py:37 (elapsed steps before this move)
  mz_rec_put_double(rec + O + A + 0, root_value);
  mz_rec_put_double(rec + O + A + 2, error);
  rec[O + A + 4] = won ? 1.f : 0.f;
  int32_t *ri = (int32_t *)(rec + O + A + 5);
  ri[0] = action; ri[1] = done | (turn < 0 ? 2 : 0); ri[2] = t; ri[3] = sp.env_offset + b; ri[4] = ep;
  if (done) {      // run_selfplay starts a new Game: env.reset (tic_tac_toe.py:20-25), its temperature evaluated now
    for (int k = 0; k < 9; ++k) bd[k] = 0;
    sp.turn[b] = 1; sp.t[b] = 0; sp.episode[b] = ep + 1; sp.temp[b] = *sp.temp_next;
  } else {
    sp.turn[b] = (int8_t)(-turn); sp.t[b] = t + 1;
  }
}

// End of a move: Config.select_action + store_search_statistics + root error (mz_finalize_tree), then Game.apply
// (game.py:79-104) on env.step (tic_tac_toe.py:30-51): the mover's mark goes on the board; the move wins if a line
// through it sums to +-3 (reward 1 for the mover), the game is done on a win or after the ninth move; the turn flips.
// The record carries the mover in its flags word (to_play), as History.to_play does.
static __global__ void k_ttt_step_record(TreeView tv, SelfplayState sp, int B, int A, uint64_t seed) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const unsigned long long move = sp.movecnt[b];
  sp.movecnt[b] = move + 1ull;
  mz_finalize_tree(tv, b, sp.temp, sp.draw_uniform, seed, move, sp.env_offset, sp.action, sp.child_visits, sp.root_value,
                   sp.error, nullptr);
  const int O = 9;
  float *rec = sp.ring + ((size_t)(move % (unsigned long long)sp.ring_moves) * B + b) * sp.rec_floats;
  for (int k = 0; k < O; ++k) rec[k] = sp.obs[(size_t)b * O + k];
  for (int a = 0; a < A; ++a) rec[O + a] = (float)sp.child_visits[(size_t)b * A + a];
  mz_ttt_apply(sp, b, sp.action[b], sp.root_value[b], sp.error[b], rec, A);
}
