#!/bin/bash
# the multi-rank paths on the one GPU: the domain tests, the N = 1 box through the domain loop against mhip_vv_run, two ranks sharing the GPU (engine loop against
# host loop), and a kernel trace of a two-rank run (how many dispatches a plain ghosted step is)
out=gpurun_out; mkdir -p $out; tag=${1:-r06}
timeout 1500 python -m pytest tests/test_gpu_domain.py tests/test_gpu_bench_cli.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/${tag}_domaintest.log 2>&1; echo "rc $?" >> $out/${tag}_domaintest.log
tail -5 $out/${tag}_domaintest.log
for fd in 0 1; do
  if [ $fd = 1 ]; then export MOLLYHIP_FORCE_DOMAIN=1; else unset MOLLYHIP_FORCE_DOMAIN; fi
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --traffic file --steps 2000 --warmup 200 > $out/${tag}_lj1m_fd$fd.json 2> $out/${tag}_lj1m_fd$fd.err
  python - <<PY
import json
try:
    d = json.load(open("$out/${tag}_lj1m_fd$fd.json")); print("force_domain=$fd", round(d["ms_per_step"], 4), d["config"].get("parallelism"), {k: round(v, 4) for k, v in d["roofline"].get("stage_ms_per_step", {}).items() if v})
except Exception as e:
    print("force_domain=$fd FAILED", e, open("$out/${tag}_lj1m_fd$fd.err").read()[-1500:])
PY
done
unset MOLLYHIP_FORCE_DOMAIN
bash tools/gpu_dom2.sh ${2:-lj256k} 2>&1 | tee $out/${tag}_two_ranks_one_gpu.txt
