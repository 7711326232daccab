/* mz_engine_debug.h -- instrumentation, profiling and test hooks of libmz_hip.so.  NOT part of the drop-in boundary
 * (include/mz_engine.h is what a maintainer binds): these entry points exist for the parity tests (tests/), the benchmark's clocks
 * (bench.py: HIP events on the kernels' own dispatches) and kernel development (scripts/).  Same conventions as mz_engine.h: plain
 * C, int return codes + mz_last_error(), [dev] / [host] say where a pointer lives. */
#ifndef MZ_ENGINE_DEBUG_H
#define MZ_ENGINE_DEBUG_H
#include "mz_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic twin of mz_search (call right after mz_root_prepare): the same launches, eagerly, with a
 * hipEvent between every pair of launches on `stream`; synchronous.  ms_out[0] = summed time of the
 * num_simulations recurrent-inference launches, ms_out[1] = summed time of the tree-step launches. */
int mz_search_profiled(mz_engine *e, int num_simulations, float *ms_out, void *stream);

/* mz_search with HIP events bracketing the dispatch of the fused search kernel itself (hipExtLaunchKernelGGL start /
 * stop events on `stream`, the same timestamps rocprofv3's kernel trace reports); call right after mz_root_prepare;
 * synchronous.  ms_out[0] = duration of the launch in milliseconds.  bench.py's roofline figure comes from here. */
int mz_search_timed(mz_engine *e, int num_simulations, float *ms_out, void *stream);

/* Diagnostic build of the fused search kernel with in-kernel s_memtime stamps (never used for timing
 * claims): cycles_out [host][4 waves][14 phases] = per-wave cycle totals over num_simulations, averaged
 * over workgroups.  Phases: 0 gather, 1 barrier, 2 dynamics fc1, 3 dynamics fc2, 4 combine, 5 LN/reward,
 * 6 store + prediction fc1, 7 prediction fc2, 8 combine, 9 value/logits, 10 expand, 11 backup, 12 descent, 13 rest of the tree step. */
int mz_search_phase_profile(mz_engine *e, int num_simulations, unsigned long long *cycles_out, void *stream);

/* of the last mz_search_phase_profile: out3 [host] = mean, minimum, maximum over the workgroups of a workgroup's total
 * cycles (the launch lasts as long as its slowest workgroup) */
int mz_search_phase_spread(const mz_engine *e, double *out3);

/* One simulation's tree work with a clock on each kernel: mz_select (no outputs) + mz_expand_backup (no hidden state),
 * hipExtLaunchKernelGGL start / stop events around each of the two dispatches (the timestamps rocprofv3's kernel trace
 * reports); synchronous.  ms_out [host][2] = duration of k_tree_select (-1 where the descent was already pending: the
 * first simulation after mz_root_prepare, which selects inside the root kernel), of k_tree_expand_backup, in milliseconds.
 * The roofline clock of `bench.py --workload tree` (the tree kernels against HBM / cache bandwidth). */
int mz_tree_pair_timed(mz_engine *e, const float *value, const float *reward, const float *logits, float *ms_out,
                       void *stream);

/* keep != 0: every move of the self-play loop also writes its searched tree back to the node pool (so that
 * mz_export_tree after mz_selfplay_steps shows the last move's tree).  Default 0: the loop never reads it. */
int mz_selfplay_export_trees(mz_engine *e, int keep);

/* keep != 0: every move also stores its Dirichlet draw (mcts.py:59; drawn on the device in this loop) in a per-move log
 * as long as the experience ring; mz_selfplay_read_noise copies the draw of move `move` (0 = first move after
 * mz_selfplay_reset, one of the last ring_moves moves) to out [host][B][A] float64, synchronously.  Test
 * instrumentation: any move of a whole-moves launch can be replayed on the CPU by the parity tests with the device's own draw
 * and the observation its record carries. */
int mz_selfplay_noise_log(mz_engine *e, int keep);
int mz_selfplay_read_noise(mz_engine *e, uint64_t move, double *out);

/* Test instrumentation of the fused search kernels' TREE code (the launch mz_search and mz_selfplay_steps run; the
 * stand-alone kernels get their network outputs through mz_expand_backup anyway).  The tree step of a simulation --
 * Node.expand, MCTS.backpropagate, the next select_child descent: mcts.py:47-55,83-92,104-143 -- consumes exactly three
 * things from the network: the value and reward scalars (networks.py:153-154,161-162) and the A policy logits.
 * buf [dev][keep_moves][num_envs][num_simulations + 1][2 + A] float32 is the caller's and must outlive the mode.
 *   mode 1, log:    every simulation s of every tree stores (value, reward, logits[A]) in slot 1 + s of row
 *                   (move % keep_moves, tree); the root of a self-play move stores (value, 0, logits[A]) in slot 0;
 *                   mz_search uses row 0.  The parity tests replay the logged outputs through the CPU oracle's tree and
 *                   demand every tree identical -- no network evaluation on the checker's side, hence no tie margin.
 *   mode 2, inject: mz_search's simulations READ slot 1 + s (row 0, keep_moves = 1) instead of their own network
 *                   outputs: the reference's recorded outputs (tests/golden) reach the fused kernels' own tree code.
 *   mode 0, off:    the production state (a null pointer in the kernels' arguments; buf ignored).
 * Synchronous; drops captured graphs. */
int mz_sim_io(mz_engine *e, int mode, float *buf, int keep_moves);

/* development hook: mz_fcl_run's host time since the handle was created (or the last reset): out [host][6] = seconds waiting for
 * a staging slot's previous update, in priority refreshes, in sampling, in launching, in the calls as a whole; number of updates */
int mz_fcl_run_stats(mz_fcl *c, double *out6, int reset);

/* test hook: the assembled gradient of the last mz_fcl_step (flat, engine.WEIGHT_ORDER) into a HOST buffer; waits for the device */
int mz_fcl_read_grad(mz_fcl *c, float *host_out, size_t n);

/* test hook: tape `which` of the last step into a HOST buffer (0 chain inputs, 1 chain fc1 activations, 2 LayerNorm x-hat, 3 rstd,
 * 4 hidden states, 5 / 6 chain deltas, 7 head fc1 activations, 8 / 9 head deltas, 10 d loss / d hidden state per head, 11 per-sample
 * losses; [position][row / 16][feature][16 rows] each, heads [head][position]...); host_out null: returns the float count. */
long long mz_fcl_read_tape(mz_fcl *c, int which, float *host_out, size_t n);

/* development hook: s_memtime stamps (shader clock) at the phase boundaries of k_fcl_heads, workgroup 0 of every head at unroll
 * position 1 (3 x 16 slots), and of k_fcl_chain_fwd4's position 2 (slots 48..53); slots 12..15, 28..31, 44..47, 59..66: the two-launch
 * step's timeline on the constant 100 MHz clock (chain workgroup 0's passes, its last two positions' value units, the last unit / job /
 * chain workgroup to end; scripts/fcl_heads_phases.py names them): enable = 1 arms it for the following steps, enable = 0 reads the stamps
 * of the last step into host_out [96]. */
int mz_fcl_heads_profile(mz_fcl *c, int enable, unsigned long long *host_out);

/* Which kernel mz_search / mz_selfplay_steps launch for this engine right now: out4 [host] = kind (0 the stand-alone
 * kernels, 1 k_search_fused, 2 k_search_h2), LDS placement of the trees (0 node pool, 1 whole trees in LDS, 2 compact in
 * LDS; -1 for kind 0), dynamics-fc1 k-steps of the instantiation, lanes per child group.  For tests: they assert the
 * instantiation they mean to exercise. */
int mz_search_kernel_info(const mz_engine *e, int *out4);

/* mz_selfplay_steps with one pair of HIP events around every search-kernel dispatch (hipExtLaunchKernelGGL start / stop
 * events on `stream`: the timestamps rocprofv3's kernel trace reports).  The k moves are launched eagerly, back to back,
 * with no synchronisation in between (the state of the timed loop); synchronous at the end.  ms_out [host][k] =
 * duration of each search launch in milliseconds.  bench.py's roofline figure is the mean of these. */
int mz_selfplay_steps_timed(mz_engine *e, int k, float *ms_out, void *stream);

/* Diagnostic: `moves` (<= 16) whole moves in ONE launch of the persistent self-play kernel with s_memtime stamps between
 * the phases of a move; cycles_out [host][8] = shader cycles per move, mean over all waves, in the order of a move:
 * root first stage (observation, obs_dim+1 -> 512), representation out + LayerNorm, prediction, root tree part (Dirichlet
 * draw, root.expand, first descent), resident weight steps + tree set-up, ring priming + barrier, all simulations, end of
 * the move (select_action, env step, record).  Synchronous; the moves' records land in the ring like any others.
 * Exact-f32 kernel only; fails where the self-play loop does not run as whole moves in one launch (two-player games,
 * trees in the global pool, MZ_NO_PERSIST) and for split_f16. */
int mz_selfplay_phase_profile(mz_engine *e, int moves, double *cycles_out, void *stream);

/* observation the synthetic env would emit for (env, episode, t): [host] out[obs_dim]; and reward */
int mz_synth_obs(const mz_engine *e, int env, int episode, int t, float *out_obs, float *out_reward);

#ifdef __cplusplus
}
#endif
#endif
