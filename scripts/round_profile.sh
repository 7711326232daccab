#!/bin/bash
# One gpurun call that refreshes every measured artifact of a round:
#   scripts/round_profile.sh <tag>      (run on the GPU box from the repo root; outputs under gpurun_out/; every command bounded: timeout 900)
# bench line, rocprofv3 kernel stats of the same command, the two PMC passes (separate runs, no trace domains), traffic.json
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 python3 bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -c 1500 $O/bench_$TAG.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 bench.py --steps 128 --warmup 16 --runs 1 --no-cpu-baseline --no-live-traffic > $O/bench_prof_$TAG.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -- python3 bench.py --steps 32 --warmup 16 --runs 1 --no-cpu-baseline --no-live-traffic > $O/pmc_fetch_$TAG.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -- python3 bench.py --steps 32 --warmup 16 --runs 1 --no-cpu-baseline --no-live-traffic > $O/pmc_write_$TAG.log 2>&1
timeout 900 python3 scripts/make_traffic.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/traffic_$TAG.json > /dev/null
# matrix-pipe utilisation as a counter (its own pass, counters only)
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_mfma_$TAG -- python3 bench.py --steps 32 --warmup 16 --runs 1 --no-cpu-baseline --no-live-traffic > $O/pmc_mfma_$TAG.log 2>&1
timeout 900 python3 scripts/make_mfma_util.py $O/pmc_mfma_$TAG $O/mfma_util_$TAG.json > /dev/null
timeout 900 python3 scripts/phase_profile.py 8 4 30 ${TAG}_lunar > $O/phase_${TAG}_lunar.txt 2>&1
timeout 900 python3 scripts/phase_profile.py 128 6 50 ${TAG}_pong > $O/phase_${TAG}_pong.txt 2>&1
cp profiles/phase_cycles_${TAG}_*.json $O/ 2>/dev/null
timeout 900 python3 bench.py --workload pong --no-cpu-baseline > $O/bench_pong_$TAG.json 2>> $O/bench_$TAG.err
# secondary lines: the opt-in split-f16 search kernel on both shapes (+ its phase tables), MuZeroNetwork through PyTorch-ROCm
timeout 900 python3 bench.py --split-f16 --no-cpu-baseline > $O/bench_split_$TAG.json 2>> $O/bench_$TAG.err
timeout 900 python3 bench.py --workload pong --split-f16 --no-cpu-baseline > $O/bench_pong_split_$TAG.json 2>> $O/bench_$TAG.err
MZ_SPLIT_F16=1 timeout 900 python3 scripts/phase_profile.py 8 4 30 ${TAG}_lunar_split > $O/phase_${TAG}_lunar_split.txt 2>&1
MZ_SPLIT_F16=1 timeout 900 python3 scripts/phase_profile.py 128 6 50 ${TAG}_pong_split > $O/phase_${TAG}_pong_split.txt 2>&1
cp profiles/phase_cycles_${TAG}_*split.json $O/ 2>/dev/null
timeout 900 python3 bench.py --workload breakout > $O/bench_breakout_$TAG.json 2>> $O/bench_$TAG.err
# two-player games on the device (TicTacToe, reference rules), and the stand-alone tree kernels against the HBM / cache roofs
timeout 900 python3 bench.py --workload tictactoe --no-cpu-baseline > $O/bench_tictactoe_$TAG.json 2>> $O/bench_$TAG.err
timeout 900 python3 bench.py --workload tree > $O/bench_tree_$TAG.json 2>> $O/bench_$TAG.err
# HBM-side traffic of the stand-alone tree kernels (separate counter passes), then the tree line again with it
(cd /tmp; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_tree_fetch_$TAG -- python3 $R/bench.py --workload tree --steps 4 --warmup 1 > /dev/null 2>&1
 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_tree_write_$TAG -- python3 $R/bench.py --workload tree --steps 4 --warmup 1 > /dev/null 2>&1)
timeout 900 python3 scripts/make_traffic.py $O/pmc_tree_fetch_$TAG $O/pmc_tree_write_$TAG $O/tree_traffic_$TAG.json > /dev/null
rm -rf $O/pmc_tree_fetch_$TAG $O/pmc_tree_write_$TAG
# the learner line: Learner.launch on the handles train.launch builds (mz_fcl_run takes the loop body), PyTorch graph beside it;
# per-kernel times of the native loop (rocprofv3 kernel trace of the same command, native variant only)
timeout 900 python3 bench.py --workload learner > $O/bench_learner_$TAG.json 2>> $O/bench_$TAG.err
(cd /tmp; MZ_LEARNER_ONLY=native timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner_$TAG -- python3 $R/bench.py --workload learner --steps 400 --runs 1 > /dev/null 2>&1)
f=$(find $O/prof_learner_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/learner_kernels_$TAG.csv; rm -rf $O/prof_learner_$TAG
# the product's own entry point: train --selfplay_only at the bench's size (frames accepted by the replay after priming / wall seconds)
timeout 900 python3 -m model_based_rl_amd.train --environment LunarLander-v2 --num_envs 4096 --num_simulations 30 --seed 0 --selfplay_only --max_moves 4864 --prime_moves 768 --window_size 2097152 --weight_sync_frequency 128 --runs_dir /tmp/mz_runs > $O/train_selfplay_$TAG.txt 2>&1
# the driver's form of the headline, and the weight pull with / without its host wait (A/B on this box)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_$TAG.json 2>> $O/bench_$TAG.err
for i in 1 2 3; do for S in 0 1; do MZ_SYNC_WEIGHTS=$S timeout 900 python3 bench.py --no-cpu-baseline --no-live-traffic --steps 1024 --runs 3 --sync-every 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MZ_SYNC_WEIGHTS=$S', round(d['value']), d['config']['weight_sync'][:60])"; done; done > $O/weight_sync_ab_$TAG.txt 2>&1
# full-grid parity report (value error distribution included)
MZ_PARITY_REPORT=$O/parity_full_grid_$TAG.txt timeout 900 python3 -m pytest tests/test_gpu_bench_parity.py -q -k full_grid_vs_oracle > /dev/null 2>&1
# bench.py --gpus 2 launching its own ranks (two ranks on this one GPU over gloo: the path of the driver's N > 1 runs)
MZ_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --envs 2048 --no-cpu-baseline > $O/bench_selflaunch_2ranks_1gpu_$TAG.json 2>> $O/bench_$TAG.err
# ... and eight ranks on this one GPU (512 environments each): shards, host cores of all ranks, seven rings into one replay
MZ_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 8 --envs 512 --min-seconds 2 --runs 3 --no-cpu-baseline > $O/bench_selflaunch_8ranks_1gpu_$TAG.json 2>> $O/bench_$TAG.err
# ... and at the FULL size (4096 environments per rank): the host side of an 8-GPU run on the one box there is
MZ_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 8 --min-seconds 2 --runs 2 --no-cpu-baseline > $O/bench_8ranks_fullsize_1gpu_$TAG.json 2>> $O/bench_$TAG.err
# host cores per thread of one rank (records stored into pinned memory / the copy-stream path), and what spins in the runtime
(echo "== scripts/experiments/runtime_spin_probe.py: busiest three threads (cores, tid) per activity; main tid printed last"
 python3 scripts/experiments/runtime_spin_probe.py 2>&1 | grep -v amdgpu
 echo; echo "== scripts/experiments/actor_thread_cpu.py, MZ_RECORD_COPY=1 (device ring + D2H copy on a copy stream)"
 MZ_RECORD_COPY=1 python3 scripts/experiments/actor_thread_cpu.py 2>&1 | grep -v "amdgpu\|online"
 echo; echo "== scripts/experiments/actor_thread_cpu.py, default (mz_selfplay_steps_into: the kernels store the records into pinned memory)"
 python3 scripts/experiments/actor_thread_cpu.py 2>&1 | grep -v "amdgpu\|online") > $O/host_threads_$TAG.txt 2>&1
# the world-size-1 RCCL branch of the bench (process group over nccl, device-side weight broadcasts)
MZ_BENCH_FORCE_DIST=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29591 bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_world1_$TAG.json 2>> $O/bench_$TAG.err
# host side: ingest thread scaling, the one-replay path of train --ranks 8 with synthetic producers
timeout 900 python3 scripts/ingest_bench.py --json $O/ingest_bench_$TAG.json > $O/ingest_bench_$TAG.txt 2>&1
timeout 900 python3 scripts/one_replay_bench.py --ranks 8 --chunks 200 --threads 4 --json $O/one_replay_8ranks_4threads_$TAG.json > /dev/null 2>&1
timeout 900 python3 scripts/one_replay_bench.py --ranks 8 --chunks 200 --threads 8 --json $O/one_replay_8ranks_8threads_$TAG.json > /dev/null 2>&1
# ... both record shapes, the r05 hand-off (record chunks, all work on rank 0) beside the r06 one (slices assembled by the producers)
bash scripts/one_replay_ab.sh one_replay_$TAG > $O/one_replay_shapes_$TAG.txt 2>&1
# the learner step, kernels only: batch sweep on HIP events + per-kernel durations at batch 256 and 2048 (rocprofv3 kernel trace)
bash scripts/fcl_sweep.sh fcl_$TAG > /dev/null 2>&1
cp $O/fcl_$TAG/sweep.json $O/learner_step_sweep_$TAG.json; cp $O/fcl_$TAG/kernels_256.csv $O/learner_step_kernels_256_$TAG.csv; cp $O/fcl_$TAG/kernels_2048.csv $O/learner_step_kernels_2048_$TAG.csv
# ... and the timeline of its batch-256 launch (s_memrealtime stamps: chain workgroup 0's passes, its units, the last unit / job / chain workgroup)
timeout 300 python3 scripts/fcl_heads_phases.py 2>&1 | grep -v amdgpu.ids > $O/learner_timeline_$TAG.txt
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); head -8 "$f"
