timeout 500 python -m pytest tests/test_learner.py -x -q -m gpu 2>&1 | tail -3
timeout 100 python scripts/fcl_step_sweep.py 256,512,1024 300 2>&1 | grep batch
