#!/bin/bash
# profiles/collect.sh <workload> <tag> [steps]
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace/stats pass plus separate PMC passes, exactly as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters
# are collected without any sys/hip/hsa trace domain).  Raw output → gpurun_out/prof_<tag>/, the summary that
# is committed → gpurun_out/prof_<tag>/summary.json (copy it to profiles/).
set -u
WL=${1:-lj1m}; TAG=${2:-r02_$WL}; STEPS=${3:-200}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload $WL --steps $STEPS --warmup 50 --profile-steps 50 --no-cpu-baseline --no-secondary --traffic file"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $CMD > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $CMD > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- $CMD > /dev/null 2> "$OUT/pmc_sq2.err"
python "$ROOT/profiles/summarize.py" "$OUT" "$WL" "${TAG%%_*}" > "$OUT/summary.json" 2> "$OUT/summarize.err"      # also writes <round>_traffic_<workload>{,_prune,_build}.json
# keep the summaries, drop the per-dispatch raw tables (gpurun_out/ is capped at 64 MiB)
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_sq2"
cat "$OUT/summary.json"
