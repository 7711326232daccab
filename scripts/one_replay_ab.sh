#!/bin/bash
# the ONE replay's capacity (scripts/one_replay_bench.py): the r05 hand-off (record chunks through the rings, everything on rank 0),
# and the r06 one (every producing rank assembles its history slices itself; rank 0 copies slices and inserts leaves), both shapes
# usage: one_replay_ab.sh <tag>
O=gpurun_out/$1
mkdir -p $O
for shape in lunar pong; do
  for t in 2 4 8; do
    for raw in 1 0; do
      MZ_RING_RAW=$raw MZ_RING_NO_PACK=1 timeout 600 python3 scripts/one_replay_bench.py --ranks 8 --chunks 100 --threads $t --shape $shape 2>/dev/null | tail -1 > $O/one_replay_${shape}_t${t}_raw${raw}.json
      python3 -c "
import json; l=json.load(open('$O/one_replay_${shape}_t${t}_raw${raw}.json'))
print('$shape threads $t %s %6.1f M records/s = %5.1f GPUs worth' % ('record chunks, all work on rank 0 (r05)' if $raw else 'slices assembled by the producers     ', l['records_per_s']/1e6, l['gpus_worth']))"
    done
  done
done
