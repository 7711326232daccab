mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_learner.py -x -q -m gpu -k "launch_structures or benched or native_step_gradients" 2>&1 | tail -3
timeout 300 python scripts/fcl_heads_phases.py > gpurun_out/r06/fcl_phases_w.txt 2>&1; head -7 gpurun_out/r06/fcl_phases_v.txt
bash scripts/fcl_sweep.sh r06/sweep_w 256 | head -6
MZ_LEARNER_ONLY=native MZ_LEARNER_NO_SWEEP=1 timeout 900 python bench.py --workload learner --runs 3 > gpurun_out/r06/bench_learner_m.json 2> gpurun_out/r06/bench_learner_m.err; tail -2 gpurun_out/r06/bench_learner_m.err
python -c "
import json; l=json.load(open('gpurun_out/r06/bench_learner_m.json')); print(l['value'], l['roofline']['us_per_update'], l['config']['native_loop_host_us_per_update'])"
