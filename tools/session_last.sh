#!/bin/bash
# parity evidence on the FINAL kernels of a round, scaled to ~12 minutes of one gpurun call (tools/session_replay.sh is the 17-minute form),
# then the riders and their rocprofv3 kernel stats  ->  gpurun_out/last/
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=gpurun_out/last; mkdir -p $OUT
export TMPDIR=/tmp
(echo "python tools/exact_replay_check.py 150 4e6"; timeout 300 python tools/exact_replay_check.py 150 4e6 | tail -3
 echo "python tools/exact_replay_check.py 60 4e6 amp"; timeout 120 python tools/exact_replay_check.py 60 4e6 amp | tail -3
 echo "EXACT_TAIL=1 python tools/exact_replay_check.py 80 4e6"; EXACT_TAIL=1 timeout 200 python tools/exact_replay_check.py 80 4e6 | tail -3
 echo "python tools/exact_replay_check.py bench 8 3.3e7"; timeout 200 python tools/exact_replay_check.py bench 8 3.3e7 | tail -9) > $OUT/exact_replay.txt 2>&1
tail -6 $OUT/exact_replay.txt
for shape in toy mid; do
  n=300; [ $shape = mid ] && n=120
  mc=25000; [ $shape = mid ] && mc=60000
  timeout 200 python tools/parity_campaign.py --n3 $n --n2 0 --shape $shape --seconds 150 --max-candidates $mc > $OUT/parity_campaign_$shape.json 2> $OUT/parity_$shape.err
  tail -c 300 $OUT/parity_campaign_$shape.json; echo
done
cd /tmp
timeout 300 python $ROOT/tools/riders.py > $ROOT/$OUT/riders.json 2> $ROOT/$OUT/riders.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/kr -o kr -- python $ROOT/tools/riders.py > /dev/null 2> $ROOT/$OUT/kr.err
cp $(find $ROOT/$OUT/kr -name '*kernel_stats.csv' | head -1) $ROOT/$OUT/riders_kernel_stats.csv 2>/dev/null
rm -rf $ROOT/$OUT/kr
tail -c 300 $ROOT/$OUT/riders.json
