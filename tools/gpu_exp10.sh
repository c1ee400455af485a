#!/bin/bash
out=gpurun_out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/cluster_sweep.hip -o /tmp/cluster_sweep 2>/dev/null && /tmp/cluster_sweep > $out/x10_cluster_sweep.txt 2>&1
cat $out/x10_cluster_sweep.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cs_pmc -- /tmp/cluster_sweep > /dev/null 2> $GRAFT_REPO_ROOT/$out/x10_cs_pmc.err
python $GRAFT_REPO_ROOT/tools/pmc_kernels.py /tmp/cs_pmc k_cluster > $GRAFT_REPO_ROOT/$out/x10_cs_pmc.txt 2>&1
cat $GRAFT_REPO_ROOT/$out/x10_cs_pmc.txt
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x10_default_2000.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $out/x10_default_20_5.json 2>/dev/null
cut -c1-300 $out/x10_default_2000.json
