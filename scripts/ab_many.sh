#!/bin/bash
# several builds of libmz_hip.so alternating on ONE box:  scripts/ab_many.sh <rounds> lib1.so lib2.so ...
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    MZ_HIP_LIB=$L python bench.py --no-cpu-baseline --no-live-traffic --steps 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['roofline'].get('us_per_move', d['roofline']['us_per_launch']), 2))"
  done
done
