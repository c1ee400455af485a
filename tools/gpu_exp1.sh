#!/bin/bash
out=gpurun_out; mkdir -p $out
python tools/dbg_cadence.py > $out/x1_dbg_cadence.log 2>&1
for s in 200 100 75 50; do
  MOLLYHIP_INNER_SKIN_PM=$s timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x1_skin$s.json 2> $out/x1_skin$s.err
done
MOLLYHIP_INNER_SKIN_PM=100 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/x1_skin100_20.json 2>/dev/null
timeout 600 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider --deselect "tests/test_gpu_cadence.py::test_set_state_then_forces_every_step_keeps_the_lists[charged-float32]" > $out/x1_gputest.log 2>&1; echo "rc $?" >> $out/x1_gputest.log
tail -3 $out/x1_gputest.log
