#!/bin/bash
# native learner step: tests of the step, updates per second, per-kernel times (rocprofv3 kernel trace of the same loop)
# usage: fcl_profile.sh <tag> [notests]
TAG=$1
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then (timeout 900 python -m pytest tests/test_learner.py -x -q -m gpu -k native 2>&1 | tail -5) > $O/tests.log; fi
(MZ_LS_ONLY=native timeout 900 python scripts/learner_graph_speed.py $O/speed.json 2>&1 | tail -20) > $O/speed.log
MZ_LS_ONLY=native rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 scripts/learner_graph_speed.py > $O/prof.log 2>&1
python3 scripts/rocpd_kernels.py $O/prof/p_results.db fcl > $O/kernels.csv
rm -rf $O/prof
cat $O/tests.log; grep -E "updates_per_second|graph_replay|sample_batch|batch_wait|update_call" $O/speed.log; cat $O/kernels.csv
