#!/bin/bash
# A/B of two builds of libmz_hip.so on ONE box (boxes of the pool differ by ~1 %): alternates the two libraries.
#   scripts/ab_bench.sh <libA.so> <libB.so> [rounds] [extra bench.py arguments, e.g. "--workload pong"]
A=$1; B=$2; N=${3:-3}; X=${4:-}
for i in $(seq $N); do
  for L in $A $B; do
    MZ_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 1024 $X 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['roofline'].get('us_per_move', d['roofline']['us_per_launch']), 2))"
  done
done
