#!/bin/bash
# the ONE replay's capacity (scripts/one_replay_bench.py) with and without the producer-side packing, both record shapes
# usage: one_replay_ab.sh <tag>
O=gpurun_out/$1
mkdir -p $O
for shape in lunar pong; do
  for t in 4 8; do
    for nopack in 1 0; do
      MZ_RING_NO_PACK=$nopack timeout 600 python3 scripts/one_replay_bench.py --ranks 8 --chunks 100 --threads $t --shape $shape 2>/dev/null | tail -1 > $O/one_replay_${shape}_t${t}_nopack${nopack}.json
      python3 -c "
import json; l=json.load(open('$O/one_replay_${shape}_t${t}_nopack${nopack}.json'))
print('$shape threads $t %s %6.1f M records/s = %5.1f GPUs worth' % ('r05 hand-off (plain copy)  ' if $nopack else 'packed by the producer    ', l['records_per_s']/1e6, l['gpus_worth']))"
    done
  done
done
