#!/bin/bash
# profiles/pmc_quick.sh <tag> <cmd...> : one SQ counter pass + one kernel-trace pass for an arbitrary command
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$@" > "$OUT/out.txt" 2> "$OUT/trace.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- "$@" > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- "$@" > /dev/null 2> "$OUT/pmc_sq2.err"
python "$ROOT/profiles/summarize.py" "$OUT" "$TAG" > "$OUT/summary.json" 2> "$OUT/summarize.err"
python - "$OUT/summary.json" <<'PY'
import json,sys
s=json.load(open(sys.argv[1]))
for k,v in sorted(s['kernels'].items(), key=lambda kv:-kv[1]['total_ns'])[:8]: print(k, v['calls'], round(v['avg_us'],1))
for kern in ('k_build','k_forces','k_filter'):
    if kern in s['pmc']:
        print(kern, {c: round(v['per_launch']) for c,v in sorted(s['pmc'][kern].items())})
PY
rm -rf "$OUT/trace" "$OUT/pmc_sq" "$OUT/pmc_sq2"
