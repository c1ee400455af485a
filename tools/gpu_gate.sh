#!/bin/bash
# GPU gate of a round: the multi-process test that faulted in round 1 (repeated, then traced if it still faults), the whole `-m gpu`
# suite without -x, and the driver-form bench line.  Run on the GPU box:  bash tools/gpu_gate.sh [tag]
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
fail=0
for k in 1 2 3; do
  timeout 400 python -m pytest tests/test_gpu_domain.py -q -x -k "8-f64-0.2-40" > $out/${tag}_dom8_$k.log 2>&1
  rc=$?; echo "rc $rc" >> $out/${tag}_dom8_$k.log; [ $rc -ne 0 ] && fail=1
done
if [ $fail -ne 0 ]; then
  MOLLYHIP_TRACE=1 timeout 400 python -m pytest tests/test_gpu_domain.py -q -x -k "8-f64-0.2-40" > $out/${tag}_dom8_trace.log 2>&1
  echo "rc $?" >> $out/${tag}_dom8_trace.log
fi
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/${tag}_gputest.log 2>&1
echo "rc $?" >> $out/${tag}_gputest.log
tail -5 $out/${tag}_gputest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_20_5.json 2> $out/${tag}_bench_20_5.err
timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 500 --no-cpu-baseline > $out/${tag}_bench_2000.json 2> $out/${tag}_bench_2000.err
cat $out/${tag}_bench_20_5.json | cut -c1-400
cat $out/${tag}_bench_2000.json | cut -c1-400
