#!/bin/bash
# kernel trace of the N = 1 box driven through the multi-GPU host loop (MOLLYHIP_FORCE_DOMAIN=1) next to mhip_vv_run
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/fd_trace; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for mode in domain single; do
  if [ $mode = domain ]; then export MOLLYHIP_FORCE_DOMAIN=1; else unset MOLLYHIP_FORCE_DOMAIN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python $ROOT/bench.py --steps 1000 --warmup 100 --equil 500 --profile-steps 10 --no-cpu-baseline > $OUT/$mode.json 2> $OUT/$mode.err
  cp $(find $OUT/$mode -name "*kernel_stats.csv" | head -1) $OUT/${mode}_kernel_stats.csv
  rm -rf $OUT/$mode
  python -c "import json; d=json.loads(open('$OUT/$mode.json').read().strip().splitlines()[-1]); print('$mode', d['ms_per_step'])"
  head -14 $OUT/${mode}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
