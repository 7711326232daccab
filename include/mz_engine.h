/*
 * mz_engine.h -- C ABI of the MI355X-native MuZero self-play / MCTS engine (libmz_hip.so).
 *
 * The reference (JimOhman/model-based-rl) is pure Python with no FFI: its search path is reached by
 * duck-typed calls.  This header is the boundary a native replacement of that path
 * exports; the Python mirror in model-based-rl_amd/ (mcts.py, actors.py, ...) binds it with ctypes
 * (cffi.dlopen works on the same symbols) and keeps the reference's class surface on top.
 * Each entry point cites the reference code it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; the message is mz_last_error() (thread-local).
 *     Nothing throws across the ABI.  (Reference convention: Python exceptions, e.g. actors.py:41.)
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     asynchronous on it.  An engine is not thread-safe; use one engine per GPU per process.
 *   - pointers marked [dev] are device pointers (e.g. tensor.data_ptr()), owned by the caller and
 *     alive until the stream reaches the end of the call's work; [host] are host pointers.
 *   - B = num_envs trees are searched in lock-step; A = action_space; H = 50 (FCNetwork hidden_dim).
 *   - node numbering inside a tree: expansion index e: root 0, the leaf expanded by simulation s is
 *     s+1 (also its hidden-state slot); node index: root 0, child a of the node with expansion index
 *     e is 1 + e*A + a.  NN = 1 + (num_simulations+1)*A nodes per tree.
 */
#ifndef MZ_ENGINE_H
#define MZ_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_HIDDEN 50       /* networks.py:135 */
#define MZ_FC_WIDTH 512    /* networks.py:60,75,88,101,114 */
#define MZ_MAX_ACTIONS 32
#define MZ_MAX_SUPPORT 32

typedef struct mz_engine mz_engine;

/* The subset of the reference's Config (config.py:87-231) the search path reads. */
typedef struct mz_config {
  int32_t num_envs;             /* B; rounded up internally to a multiple of 16 */
  int32_t obs_dim;              /* prod(config.obs_space) */
  int32_t action_space;         /* config.action_space, <= MZ_MAX_ACTIONS */
  int32_t num_simulations;      /* --num_simulations */
  int32_t two_players;          /* --two_players */
  int32_t has_min_bound;        /* --known_bounds[0] is not None */
  int32_t has_max_bound;        /* --known_bounds[1] is not None */
  int32_t value_support_min, value_support_max;    /* --value_support, size <= MZ_MAX_SUPPORT */
  int32_t reward_support_min, reward_support_max;  /* --reward_support */
  int32_t no_target_transform;  /* --no_target_transform */
  double min_bound, max_bound;  /* --known_bounds */
  double discount;              /* --discount */
  double pb_c_base, pb_c_init;  /* --pb_c_base / --pb_c_init */
  double init_value_score;      /* --init_value_score */
  double root_dirichlet_alpha;  /* --root_dirichlet_alpha */
  double root_exploration_fraction; /* --root_exploration_fraction */
  uint64_t seed;                /* counter-based device RNG key (throughput mode) */
  int32_t env_id_offset;        /* global id of env 0 (rank * num_envs when actors are sharded over GPUs) */
  int32_t no_support;           /* --no_support: value / reward heads are single scalars (networks.py:135-136), returned
                                 * untransformed in eval mode (networks.py:153,161); the supports above are ignored */
  int32_t split_f16;            /* 0 (default): the FCNetwork GEMMs of mz_search / the self-play loop run on
                                 * v_mfma_f32_16x16x4_f32 -- exact float32, the reference's arithmetic type.
                                 * 1: opt-in fast path (action_space <= 13): every float32 operand is split into two
                                 * float16 parts and every product block is three v_mfma_f32_16x16x32_f16 with float32
                                 * accumulation (csrc/mz_fused_h2.hip.h): float32-level accuracy (deviation from a float64
                                 * evaluation 1-2x that of the exact path, inside the 1e-5 bound), NOT bit-identical to the
                                 * exact path.  The environment variable MZ_SPLIT_F16=1 sets it for every engine.
                                 * Range: the high part of a split value is its float16 rounding, so mz_set_weights FAILS for
                                 * weights that are not finite or exceed 65504 in magnitude; weights below 6.1e-5 in magnitude
                                 * are carried with an absolute error of ~3e-8 (float16 subnormals) instead of a relative one --
                                 * far inside the 1e-5 bound on outputs.  Activations are split at run time the same way: hidden
                                 * states are min-max free here (ReLU(LayerNorm), |x| < ~10), head pre-activations are sums of
                                 * 512 products of such values and PyTorch-initialised weights. */
} mz_config;

const char *mz_last_error(void);
int mz_version(void);

/* MCTS.__init__ (mcts.py:66-76) + Actor.__init__'s network/device set-up (actors.py:33-47).
 * Allocates every device pool (node SoA, hidden-state pool, packed weights, scratch). */
int mz_create(const mz_config *cfg, mz_engine **out);
int mz_destroy(mz_engine *e);

/* Number of float32 parameters of FCNetwork for this config (networks.py:137-144). */
size_t mz_num_weights(const mz_engine *e);

/* BaseNetwork.load_weights (networks.py:36-37; actors.py:81-85).  `flat` is the reference's
 * state_dict concatenated in its own key order: representation_head.{fc1,out}.{weight,bias},
 * value_head.{fc1,value}, policy_head.{fc1,policy}, reward_head.{fc1,reward},
 * transition_head.{fc1,out}, LN.{weight,bias}; Linear weights [out][in] row-major.
 * on_device != 0: flat is [dev] (e.g. the buffer an RCCL broadcast just filled); else [host].
 * Synchronises `stream` before it returns (a 16-byte read back decides which kernel set this weight set runs on,
 * mz_weight_scale below; a host buffer may be released as soon as the call returns). */
int mz_set_weights(mz_engine *e, const float *flat, size_t n, int on_device, void *stream);

/* Actor.sync_weights (actors.py:81-85) without draining the launch-ahead pipeline: mz_set_weights whose host side waits for
 * NOTHING queued on `stream`.  The repack runs in stream order -- moves queued before the call keep the old weights, everything
 * queued after it sees the new ones; a [host] source is copied into the engine's pinned staging before the call returns
 * (the caller's buffer is free again), a [dev] source must stay valid until the stream has passed the call (e.g. the buffer
 * mz_broadcast_weights filled on the same stream).  No read back: which kernel set the weights run on is the caller's
 * `scale_ok` = mz_weights_scale_ok(...) evaluated on a HOST copy of the SAME weights (the learner rank has one; it travels
 * with the training step).  Nothing in it synchronises or allocates after the first call with a [host] source: with a
 * [dev] source the call can be captured into a graph.  split_f16 engines fall back to mz_set_weights (range check).
 * mz_weights_scale_ok: 1 if this weight set admits the power-of-two scale of the search kernel's stream (mz_weight_scale),
 * 0 if not, < 0 on error; host arithmetic only, no GPU.  value_outputs / reward_outputs: the support sizes (1 with
 * --no_support). */
int mz_set_weights_async(mz_engine *e, const float *flat, size_t n, int on_device, int scale_ok, void *stream);
int mz_weights_scale_ok(const float *flat_host, size_t n, int obs_dim, int action_space, int value_outputs, int reward_outputs);

/* The path's ONE collective (reference: every actor pulls the pickled state_dict from the storage actor,
 * actors.py:81-85, shared_storage.py:12-18, learners.py:85-86): the flat float32 weights from the learner rank to every
 * actor rank as an RCCL broadcast issued on the caller's HIP stream -- ordered by the stream with the repack that follows
 * (mz_set_weights_async(on_device = 1) on the same stream), no host wait in between.  librccl is bound at run time:
 * mz_comm_load(path) (NULL / "": the copy the process already holds, else the loader's search path); mz_comm_unique_id
 * (rank 0) -> 128 bytes, handed to every rank by the caller (torch.distributed / any channel); mz_comm_create on every rank
 * (collective: ncclCommInitRank on the calling thread's current device); mz_broadcast_weights: flat [dev][n] in place,
 * from rank `root`. */
typedef struct mz_comm mz_comm;
int mz_comm_load(const char *librccl_path);
int mz_comm_unique_id(void *out128);
int mz_comm_create(int rank, int world, const void *unique_id128, mz_comm **out);
int mz_comm_destroy(mz_comm *c);
/* ranks the communicator spans, as RCCL reports it (ncclCommCount) */
int mz_comm_count(const mz_comm *c, int *ranks_out);
int mz_broadcast_weights(mz_comm *c, float *flat, size_t n, int root, void *stream);

/* Diagnostic: how the last mz_set_weights packed the search kernel's weight stream.  out [host][4] =
 * {1, 2^-k, 2^k, chosen}: with chosen = 1 the stream carries the four 512-wide hidden layers of the search
 * (networks.py:70-93,96-119: reward / transition / value / policy fc1) times 2^-k and the layers reading their
 * activations times 2^k -- bit-neutral (powers of two), it lets the kernel apply nn.ReLU as a [0, 1] clamp on two
 * elements per instruction; 2^k exceeds a bound of every activation computed from this weight set.  chosen = 0
 * (non-finite or absurdly large weights): unscaled stream, ordinary maximum.  Synchronises `stream` after an
 * mz_set_weights_async (the four floats are read back on demand). */
int mz_weight_scale(mz_engine *e, float *out, void *stream);

/* BaseNetwork.initial_inference (networks.py:26-29) for B observations, actors.py:139.
 * obs [dev][B][obs_dim] float32 (already normalised, actors.py:134-137).  Fills hidden slot 0, the
 * root value and the root policy logits inside the engine. */
int mz_initial_inference(mz_engine *e, const float *obs, void *stream);

/* Same, but with network outputs computed elsewhere (any torch network): hidden [dev][B][50] or NULL,
 * value [dev][B], logits [dev][B][A]. */
int mz_root_load(mz_engine *e, const float *hidden, const float *value, const float *logits, void *stream);

/* Read back what initial inference produced: value [dev][B], logits [dev][B][A], hidden [dev][B][50]
 * (any may be NULL). */
int mz_root_outputs(mz_engine *e, float *value, float *logits, float *hidden, void *stream);

/* Node(0) + root.expand + root.add_exploration_noise + MinMaxStats.reset
 * (actors.py:132,141-143; mcts.py:47-61,79).
 * to_play [dev][B] int8 or NULL (= +1);  legal [dev][B][A] uint8 or NULL (= all legal);
 * noise [dev][B][A] float64: the Dirichlet draw scattered to the legal positions (parity mode, drawn
 * by the host's numpy in the reference's order), or NULL with use_device_rng != 0 to draw it on the
 * device from the counter-based RNG keyed (seed, env, move), or NULL with use_device_rng == 0 for no
 * noise at all (evaluation, evaluate.py). */
int mz_root_prepare(mz_engine *e, const int8_t *to_play, const uint8_t *legal, const double *noise,
                    int use_device_rng, uint64_t move_counter, void *stream);

/* Root from priors the caller already holds (its own Node.expand + add_exploration_noise, mcts.py:47-61):
 * priors [dev][B][A] float64 (entries of illegal actions ignored).  Used by the batch-1 MCTS.run front-end. */
int mz_root_set_priors(mz_engine *e, const int8_t *to_play, const uint8_t *legal, const double *priors,
                       void *stream);

/* The search path of the pending descent (after mz_select): paths [dev][B][num_simulations+2] node indices,
 * lengths [dev][B].  (MCTS.run returns these as `search_paths`, mcts.py:101-102.) */
int mz_last_paths(mz_engine *e, int32_t *paths, int32_t *lengths, void *stream);

/* MCTS.run (mcts.py:78-102) for all B trees with the engine's own FCNetwork kernels:
 * num_simulations x { select_child descent, recurrent_inference, expand, backpropagate },
 * no host synchronisation. */
int mz_search(mz_engine *e, int num_simulations, void *stream);




/* The same loop opened up for an external network (MuZeroNetwork/TinyNetwork through PyTorch, or
 * recorded outputs in the parity tests):
 *   mz_select        = the descent of mcts.py:83-94; outputs (any may be NULL):
 *                      leaf_node, parent_slot (hidden slot of search_path[-2]), action, depth [dev][B] int32
 *   mz_gather_hidden = hidden_out[b] = hidden pool[b][parent_slot[b]]  ([dev][B][50])
 *   mz_expand_backup = node.expand (mcts.py:97) + backpropagate (mcts.py:99,126-143) with
 *                      value/reward [dev][B], logits [dev][B][A], hidden [dev][B][50] or NULL. */
int mz_select(mz_engine *e, int32_t *leaf_node, int32_t *parent_slot, int32_t *action, int32_t *depth,
              void *stream);
int mz_gather_hidden(mz_engine *e, float *hidden_out, void *stream);
int mz_expand_backup(mz_engine *e, const float *value, const float *reward, const float *logits,
                     const float *hidden, void *stream);
/* mz_expand_backup followed by the next simulation's mz_select (mcts.py:97-99 then 83-94 of the next iteration) in ONE launch:
 * the descent reads the lines the backup has just written from cache, and a simulation of an external network costs one tree
 * launch instead of three.  Same arguments as mz_expand_backup, then mz_select's outputs (any may be null).  At a move's
 * last simulation there is no next descent: it is mz_expand_backup alone and the outputs are left untouched. */
int mz_expand_backup_select(mz_engine *e, const float *value, const float *reward, const float *logits, const float *hidden,
                            int32_t *leaf_node, int32_t *parent_slot, int32_t *action, int32_t *depth, void *stream);

/* BaseNetwork.recurrent_inference (networks.py:31-34) on arbitrary rows, outside any tree:
 * hidden_in [dev][n][50], action [dev][n] int32 -> hidden_out [dev][n][50], reward/value [dev][n],
 * logits [dev][n][A].  n <= num_envs.  (Network parity tests; evaluate.py-style callers.) */
int mz_recurrent_inference(mz_engine *e, const float *hidden_in, const int32_t *action, int n,
                           float *hidden_out, float *reward, float *value, float *logits, void *stream);

/* End of a move: Config.select_action (config.py:70-81), Game.store_search_statistics
 * (game.py:106-115), root error (actors.py:147-148).
 * temperature [dev][B] float64; uniform [dev][B] float64 in [0,1) (the draw np.random.choice would
 * consume; NULL = device RNG keyed (seed, env, move)).  temperature 0 picks the
 * floor(u*n_ties)-th arg-max child.  Outputs (any may be NULL): action [dev][B] int32,
 * child_visits [dev][B][A] float64, root_value [dev][B] float64, error [dev][B] float64
 * (root value - initial value), visit_counts [dev][B][A] int32. */
int mz_finalize(mz_engine *e, const double *temperature, const double *uniform, uint64_t move_counter,
                int32_t *action, double *child_visits, double *root_value, double *error,
                int32_t *visit_counts, void *stream);

/* Raw tree dump (Node objects of mcts.py:28-45 in SoA form) to [host] arrays, synchronous.
 * Each per-node array is [B][NN]; minmax [B][2]; legal_mask [B] (bit a = root child a exists);
 * noise [B][A] (the Dirichlet draw last mixed in).  Any pointer may be NULL. */
int mz_export_tree(mz_engine *e, int32_t *N, double *W, double *P, float *R, int32_t *E, int8_t *TP,
                   uint32_t *legal_mask, double *minmax, double *noise, float *hidden_pool);

/* Epilogue of a residual block of the conv networks in GPU inference (reference networks.py:393-410: BatchNorm2d in eval
 * mode, the skip connection, ReLU), for the torch-network path (BASELINE configs[4]; the convolutions themselves stay
 * MIOpen's): in place over a contiguous NCHW float32 tensor y [dev][n] = [N][channels][hw], hw % 4 == 0,
 *   y = relu(y * scale[c] + shift[c] (+ residual)),   scale = weight / sqrt(running_var + eps), shift = bias - mean * scale
 * (scale, shift [dev][channels]; residual [dev][n] or NULL).  One pass instead of PyTorch's 2-3 elementwise kernels. */
int mz_affine_relu(float *y, const float *scale, const float *shift, const float *residual, size_t n, int channels,
                   int hw, void *stream);

/* Introspection for tests/bench. */
int mz_nodes_per_tree(const mz_engine *e);
int mz_padded_envs(const mz_engine *e);

/* ---- on-device self-play loop (actors.py:126-176 for B synthetic fixed-length envs) ----------
 * Synthetic env (gym/ALE are not installed on either box): observation
 * obs_t[i] = Irwin-Hall(4) approximation of N(0,1) from Philox4x32-10 keyed (seed, env, episode, t),
 * reward_t = U(-1,1) keyed likewise, all actions legal, done at t == episode_len.
 * (mz_selfplay_set_obs: byte-valued observations and --norm_obs for the -ram- shapes.)
 * mz_selfplay_steps runs `moves` complete moves: obs -> initial inference -> root -> search ->
 * select_action -> env.step -> experience record appended to a device ring.
 * Each record is rec_floats() = obs_dim + action_space + MZ_REC_EXTRA float32 slots: obs[O] (raw, as History keeps
 * it), child_visits[A], root_value and error as float64 in two slots each (the reference's Python floats,
 * actors.py:147-148, game.py:112), reward, then as int32 bit patterns: action, flags (bit 0 = done, bit 1 = the mover was
 * player -1: History.to_play, game.py:100-101; always clear for the single-player environments), step, env_id, episode.
 * mz_selfplay_drain copies the records produced since the last drain into `out` [host, pinned
 * preferred] asynchronously on `stream` and returns their count through *n_records after the
 * stream is synchronised by the caller (records are laid out move-major: [moves][B]).  `stream` may be a
 * copy stream other than the one mz_selfplay_steps ran on, provided the caller orders it behind those steps
 * (event): later moves then overlap the copy.  The ring normally keeps them in different slots; where it cannot
 * (records so large that the ring holds few moves), mz_selfplay_steps makes its stream wait for the event the
 * drain recorded behind its copy before it launches moves into slots the copy may still be reading.
 * mz_selfplay_steps plays up to 16 moves per launch: single-player games with their trees in LDS run as whole moves
 * inside ONE launch of the search kernel (root, simulations and end of every move; no launch, no grid-wide drain and
 * no weight reload between moves); every other configuration as one hipGraph of 2 kernels per move (root, search +
 * end of move).  Both produce the same records bit for bit (MZ_NO_PERSIST=1 at mz_create selects the graph).
 * stagger != 0: env i starts its first episode at t0 = hash(env id) % episode_len (uniform episode ends).
 * temperature: Config.visit_softmax_temperature at reset; mz_selfplay_set_temperature changes what every env's
 * NEXT game starts with (actors.py:128-129: evaluated once per game), games in progress keep theirs. */
#define MZ_REC_EXTRA 10
int mz_selfplay_reset(mz_engine *e, int episode_len, double temperature, int stagger, void *stream);
int mz_selfplay_set_temperature(mz_engine *e, double temperature, void *stream);
/* The move counter of every environment (Actor's count of moves played, actors.py:87-124: keys the device RNG by (seed, env, move)
 * and places the experience records).  For a caller that wants a resumed run to continue its count instead of replaying the random
 * stream of its first moves (the reference reseeds on resume: Actor.load_state does not call this), and for the tests, which start
 * close to 2^30 and 2^32.  The device ring must be drained; synchronous. */
int mz_selfplay_set_moves(mz_engine *e, unsigned long long moves);
/* The environment the loop plays (call before mz_selfplay_reset).  0 (default): the synthetic fixed-length episodes above.
 * 1: TicTacToe with the reference's rules (custom_environments/tic_tac_toe.py:5-76: observation turn * board, legal =
 * empty cells, reward 1 for the winning move, done on a win or after nine moves, players alternate) entirely on the device;
 * needs obs_dim 9, action_space 9, two_players.  mz_selfplay_reset's episode_len / stagger are ignored (real games).
 * mz_selfplay_set_draws (game environments only): the Dirichlet draw (noise [dev][B][A] float64 at the legal positions,
 * mcts.py:59) and / or the uniform of select_action (uniform [dev][B] float64, config.py:77) of the following moves come
 * from the caller -- numpy's stream in the reference's order -- instead of the device RNG; NULL, NULL switches back. */
int mz_selfplay_set_env(mz_engine *e, int kind);
int mz_selfplay_set_draws(mz_engine *e, const double *noise, const double *uniform, void *stream);
/* Observation format of the synthetic env (call before mz_selfplay_reset, with the ring drained; synchronous).  uint8_obs
 * 1: observations are bytes 0..255 (the -ram- envs), one per float slot of the record; 2: the same observations, and the
 * record carries them as BYTES, four per float slot -- rec_floats = ceil(obs_dim / 4) + action_space + MZ_REC_EXTRA
 * (mz_selfplay_rec_floats), the layout mz_replay.h's obs_u8 replay ingests (game.py:93-96 keeps the raw uint8 observation):
 * 208 instead of 592 bytes per env-step on the Pong-ram shapes.  obs_min / obs_range [host][obs_dim], both or neither: --norm_obs
 * (actors.py:55-58,134-137): the network input is (obs - obs_min) / obs_range in float32, the record keeps the raw
 * observation (the learner normalises its own batches, learners.py:167-168). */
int mz_selfplay_set_obs(mz_engine *e, int uint8_obs, const float *obs_min, const float *obs_range);
/* The elementwise ends of the learner step (reference learners.py:164-230) as single launches -- the step is launch-bound,
 * and PyTorch spells these as ~45 (targets) and ~11 (each categorical loss) tiny kernels.  Engine-free: device pointers,
 * sizes, a stream; capturable into a graph (no synchronisation, no allocation).
 * mz_learner_targets (learners.py:176-189; config.py:27-33,51-68): t_val, t_rew [bs][k1] as sample_batch returns them (k1 = K
 *   + 1 unroll positions), value0 [bs][sv] the value logits of the initial inference ->  sup_val [k1][bs][sv], sup_rew
 *   [k1][bs][sr]: two-hot supports of h(target) (h skipped with no_target_transform), position-major; new_errors [bs] =
 *   inverse_transform(value0) - t_val[:, 0], the priority refresh.
 * mz_soft_ce_forward / _backward (utils.py:53-60, summed over the unroll positions as learners.py:191-203 does): logits
 *   [positions][bs][bins], target element (p, b, s) at target[p * target_pos_stride + b * target_row_stride + s] -> loss [bs]
 *   = sum_p sum_s -target log_softmax(logits); backward: grad_logits = grad_loss[b] (softmax(logits) sum_s target - target). */
int mz_learner_targets(const float *t_val, const float *t_rew, const float *value0, int bs, int k1, int sv, int vmin, int sr,
                       int rmin, int no_target_transform, float *sup_val, float *sup_rew, float *new_errors, void *stream);
int mz_soft_ce_forward(const float *logits, const float *target, int positions, int bs, int bins, int64_t target_pos_stride,
                       int64_t target_row_stride, float *loss, void *stream);
int mz_soft_ce_backward(const float *logits, const float *target, const float *grad_loss, int positions, int bs, int bins,
                        int64_t target_pos_stride, int64_t target_row_stride, float *grad_logits, void *stream);
/* ---- The FCNetwork learner step as two launches at batches up to 256 (four at 512, five to six above or with gradient clipping; csrc/mz_fcl.hip.h): reference learners.py:164-230 (update_weights: K-step
 * unroll, categorical targets, soft cross-entropy, gradient hooks 0.5 and 1 / K, importance weights, clip_grad_norm_,
 * optimizer.step) with networks.py:135-180 (FCNetwork), config.py:27-33,51-68, utils.py:53-60,73-83 (Adam / AdamW,
 * eps 1.5e-4).  No GEMM library, no autograd tape: forward chain, heads + losses + their backward, backward chain, weight
 * gradients (deterministic: no atomics), gradient norm, optimiser.  Every pointer below is a DEVICE pointer unless it says
 * host; nothing synchronises or allocates after mz_fcl_create, so a step can be captured into a graph.
 * mz_fcl_create: batch (a multiple of 16), unroll steps K (1..7), FCNetwork sizes (observations <= 1024
 *   features, actions <= 14, supports <= 64 bins).
 * mz_fcl_bind: the flat float32 parameter vector (engine.WEIGHT_ORDER = mz_set_weights' order, mz_fcl_num_params
 *   floats), Adam's exp_avg / exp_avg_sq of the same shape, `nsteps` float32 step counters (torch keeps one per parameter:
 *   all are incremented, the first is read) and the learning rate as a device float; builds the packed weight copies.
 * mz_fcl_repack: after anything else wrote the parameter vector (load_state_dict).
 * mz_fcl_step: obs [batch][obs_dim] (normalised), actions int64 or int32 [batch][K], target_rewards / target_values [batch][K + 1],
 *   target_policies [batch][K + 1][actions], is_weights [batch] (float64 or float32) as replay_buffer.sample_batch returns
 *   them -> the parameters, optimiser state and step counters updated in place (unless no_update: gradients only),
 *   new_errors [batch] = inverse_transform(value_0) - target_values[:, 0] (the priority refresh, learners.py:181-182),
 *   loss_sums float64 [3] += the weighted means of the reward, value and policy losses (learners.py:228-230).
 * mz_fcl_read_grad: the last step's gradient (after the sum over positions, before clipping) into a HOST buffer; waits. */
typedef struct mz_fcl mz_fcl;
int mz_fcl_create(int batch, int unroll_steps, int obs_dim, int action_space, int value_support_min, int value_support_max,
                  int reward_support_min, int reward_support_max, int no_target_transform, mz_fcl **out);
int mz_fcl_destroy(mz_fcl *c);
size_t mz_fcl_num_params(mz_fcl *c);
int mz_fcl_bind(mz_fcl *c, float *params, float *exp_avg, float *exp_avg_sq, float *steps, int nsteps, const float *lr,
                void *stream);
int mz_fcl_repack(mz_fcl *c, void *stream);
int mz_fcl_step(mz_fcl *c, const float *obs, const void *actions, int actions_are_i32, const float *target_rewards, const float *target_values,
                const float *target_policies, const void *is_weights, int weights_are_f64, double beta1, double beta2, double eps,
                double weight_decay, double clip_grad, int adamw, int no_update, float *new_errors, double *loss_sums, void *stream);
/* mz_fcl_update: mz_fcl_step from HOST arrays (what replay_buffer.sample_batch returns, learners.py:165-180): staged through
 * pinned memory (two slots), one host-to-device copy, the step, the new errors copied back -- nothing is waited for except
 * the slot's own use mz_fcl_slots() updates ago (6 staging slots).  mz_fcl_errors(slot): the new errors of that update into host_out [batch], waiting for
 * their copy (the priority refresh goes to the replay one update behind, as the reference's fire-and-forget refresh does). */
int mz_fcl_update(mz_fcl *c, const float *obs, const void *actions, int actions_are_i32, const float *target_rewards,
                  const float *target_values, const float *target_policies, const void *is_weights, int weights_are_f64, double beta1,
                  double beta2, double eps, double weight_decay, double clip_grad, int adamw, double *loss_sums, void *stream,
                  int *slot_out);
int mz_fcl_errors(mz_fcl *c, int slot, float *host_out);
/* Learner.learn's loop body in native code (learners.py:115-131: sample_batch -> update_weights -> replay_buffer.update),
 * n_updates training steps per call: Python only at the boundaries where the reference's loop does something else
 * (send_weights, save_state, logging; learners.py:132-153).  The replay is reached through the caller's function table
 * (include/mz_replay.h: sample = mzr_sample_batches_full, refresh = mzr_update_errors_f32, last_error = mzr_last_error;
 * libmz_hip.so does not link libmz_replay.so).  words [n_updates][2 batch]: the generator words of the stratified draws
 * (random.getrandbits(64 batch n_updates), least significant word first) -- or NULL with py_key [624] / *py_pos = the state of
 * Python's `random` generator (random.getstate()[1]), from which the same words are generated per update and which comes back
 * advanced; np_key [624] / *np_pos: numpy's legacy generator
 * state for the padded actions, advanced in place; *beta_inout: the replay's beta (schedule applied per batch); obs_min /
 * obs_range [host][obs_dim] or NULL: --norm_obs; lrs [host][n_updates] or NULL: the learning rate of every update (a
 * scheduler's values; NULL: the device float bound by mz_fcl_bind); loss_sums [dev][3] as mz_fcl_step; *pads_out (may be
 * NULL): padded actions drawn.  Batch i is sampled while update i - 1 runs on the GPU; the refresh of update i reaches the
 * replay before batch i + 2 is drawn (the reference's own lag is up to batches_per_fetch = 15 batches, learners.py:124).
 * Returns with the last (up to mz_fcl_slots() - 1) updates still in flight: the next call hands their refreshes over as their staging slots come
 * up; n_updates = 0 flushes -- waits for them and hands the refreshes over (before anything else touches the replay's
 * priorities or this handle's staging: mz_fcl_update, a checkpoint, the end of Learner.learn). */
typedef struct mz_fcl_source {
  void *replay;
  int (*sample)(void *replay, const uint32_t *words, int n, int bs, float *obs, int32_t *actions, float *target_rewards,
                float *target_values, float *target_policies, int64_t *idxs, double *is_weights, uint32_t *np_key, int32_t *np_pos,
                double *beta_inout, int64_t *pads_out, uint32_t *py_key, int32_t *py_pos);
  int (*refresh)(void *replay, const int64_t *idxs, const float *errors, int64_t n);
  const char *(*last_error)(void);
} mz_fcl_source;
int mz_fcl_run(mz_fcl *c, const mz_fcl_source *src, int n_updates, const uint32_t *words, uint32_t *np_key, int32_t *np_pos,
               double *beta_inout, const float *obs_min, const float *obs_range, double beta1, double beta2, double eps,
               double weight_decay, double clip_grad, int adamw, const float *lrs, double *loss_sums, void *stream,
               int64_t *pads_out, uint32_t *py_key, int32_t *py_pos);
/* staging slots of this handle: how many updates mz_fcl_update / mz_fcl_run keep in flight before they wait for the oldest */
int mz_fcl_slots(mz_fcl *c);
int mz_selfplay_steps(mz_engine *e, int moves, void *stream);
/* mz_selfplay_steps with the records written straight into host_records [host, PAGE-LOCKED: hipHostMalloc /
 * torch pin_memory][moves][B][rec_floats] by the kernels' own stores through the buffer's device mapping (0.6 GB/s of
 * posted PCIe writes at the benched rate), instead of the device ring + mz_selfplay_drain's D2H copy.  The buffer is
 * complete once work enqueued on `stream` behind this call has completed (event / synchronise); it must stay untouched
 * until then.  The device ring must hold no undrained moves; these moves count as drained.  This is the hand-off
 * Actor.run_selfplay uses (actors.py:160-169's history hand-off): no copy stream, no cross-stream dependency for the
 * runtime to track -- that dependency kept a runtime thread of every rank spinning (profiles/r05_host_threads.txt).
 * Fails when host_records is not page-locked. */
int mz_selfplay_steps_into(mz_engine *e, int moves, float *host_records, void *stream);
/* 16 where mz_selfplay_steps plays whole moves inside one launch of the search kernel (at most that many per launch),
 * 0 where a move is a hipGraph node pair (root kernel, search kernel). */
int mz_selfplay_moves_per_launch(const mz_engine *e);
int mz_selfplay_rec_floats(const mz_engine *e);
int mz_selfplay_ring_moves(const mz_engine *e);
int mz_selfplay_drain(mz_engine *e, float *out, int max_moves, int *n_moves, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MZ_ENGINE_H */
