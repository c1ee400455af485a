#!/bin/bash
# round 5, call l: atoms per lane and staging round of the packed pair pass (4 -> 6, 8): A/B on lj256k and lj1m; PMC of the pruning pass by full kernel name
out=gpurun_out; mkdir -p $out; R=$PWD
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree ab/lib_sb6.so ab/lib_sb8.so tree 2>&1 | cut -c1-330; done | tee $out/r05_l_sb_ab.txt
CMD="python $R/bench.py --workload lj1m --steps 150 --warmup 20 --profile-steps 20 --equil 300 --no-cpu-baseline --no-secondary"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$out/pmc_k1 -- $CMD > /dev/null 2> $R/$out/pmc_k1.err
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$out/pmc_k2 -- $CMD > /dev/null 2> $R/$out/pmc_k2.err
cd $R
for d in pmc_k1 pmc_k2; do PMC_NAME_CHARS=90 python tools/pmc_kernels.py $out/$d k_forces 2>&1 | cut -c1-600; done | tee $out/r05_l_pmc_lj1m.txt
rm -rf $out/pmc_k1 $out/pmc_k2
