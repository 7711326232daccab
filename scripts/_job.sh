timeout 400 python -m pytest tests/test_learner.py -x -q -m gpu -k "hand_offs" 2>&1 | tail -4
timeout 900 python3 bench.py --workload learner > gpurun_out/bench_learner_r06_a.json 2> gpurun_out/bench_learner_r06_a.err; tail -c 600 gpurun_out/bench_learner_r06_a.json
