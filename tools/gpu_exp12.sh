#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x12_default_2000.json 2>/dev/null
python -c "
import json; d=json.load(open('$out/x12_default_2000.json')); print('lj1m', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['stage_ms_per_step'])"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/cluster_sweep.hip -o /tmp/cluster_sweep 2>/dev/null && /tmp/cluster_sweep > $out/x12_cluster_sweep.txt 2>&1
cat $out/x12_cluster_sweep.txt
export TMPDIR=/tmp; (cd /tmp; rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cs_pmc -- /tmp/cluster_sweep > /dev/null 2> /dev/null; python $GRAFT_REPO_ROOT/tools/pmc_kernels.py /tmp/cs_pmc k_cluster > $GRAFT_REPO_ROOT/$out/x12_cs_pmc.txt 2>&1)
cat $out/x12_cs_pmc.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/x12_gputest.log 2>&1; echo "rc $?" >> $out/x12_gputest.log
tail -4 $out/x12_gputest.log
timeout 300 python bench.py --workload 6mrr_pme --steps 2000 --warmup 300 --no-cpu-baseline > $out/x12_6mrr.json 2>/dev/null
python -c "
import json; d=json.load(open('$out/x12_6mrr.json')); print('6mrr', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
