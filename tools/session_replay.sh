#!/bin/bash
# exact replay + parity campaigns on the current kernels (one gpurun call): gpurun_out/replay/
cd "$(dirname "$0")/.."
OUT=gpurun_out/replay; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1500 python tools/exact_replay_check.py 250 4e6; echo "-- amp"; timeout 600 python tools/exact_replay_check.py 60 4e6 amp; echo "-- tail only"; EXACT_TAIL=1 timeout 900 python tools/exact_replay_check.py 150 4e6) > $OUT/exact_replay.txt 2>&1
tail -4 $OUT/exact_replay.txt
for shape in toy mid amp; do
  n=600; [ $shape = mid ] && n=300; [ $shape = amp ] && n=120
  mc=25000; [ $shape = mid ] && mc=60000; [ $shape = amp ] && mc=60000
  timeout 1500 python tools/parity_campaign.py --n3 $n --n2 0 --shape $shape --seconds 1200 --max-candidates $mc > $OUT/parity_campaign_$shape.json 2> $OUT/parity_$shape.err
  tail -c 400 $OUT/parity_campaign_$shape.json; echo
done
