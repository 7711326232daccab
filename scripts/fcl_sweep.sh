#!/bin/bash
# native learner step, kernels only: batch sweep (HIP events) + per-kernel durations at batch 256 and 2048 (rocprofv3 kernel trace)
# usage: fcl_sweep.sh <tag> [batches] [tests]      environment: MZ_HIP_LIB (another build of the library, A/B)
TAG=$1
BATCHES=${2:-256,512,1024,2048,4096}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$3" == "tests" ]; then (timeout 1200 python -m pytest tests/test_learner.py -x -q -m gpu -k "native or sweep or golden or benched" 2>&1 | tail -8) > $O/tests.log; cat $O/tests.log; fi
(timeout 600 python3 scripts/fcl_step_sweep.py $BATCHES 200 $O/sweep.json 2>&1 | tail -12) > $O/sweep.log
for B in 256 2048; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$B -o p -- python3 scripts/fcl_step_sweep.py $B 100 > $O/prof$B.log 2>&1
  python3 scripts/rocpd_kernels.py $O/prof$B/p_results.db fcl > $O/kernels_$B.csv
  rm -rf $O/prof$B
done
cat $O/sweep.log; cat $O/kernels_256.csv; cat $O/kernels_2048.csv
