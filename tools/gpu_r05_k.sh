#!/bin/bash
# round 5, call k: hipGraph against plain launches (micro), time stamps of the lj256k pair pass, PMC of the lj1m pruning pass
out=gpurun_out; mkdir -p $out; R=$PWD
( cd tools/micro && ./graph_launch ) 2>&1 | tee $out/r05_graph_launch.txt
MOLLYHIP_DBG_TIMES=400 MOLLYHIP_LIB_AB=$R/ab/lib_stamps.so timeout 600 python tools/force_ab.py --child --workload lj256k --steps 1200 2>&1 | grep "mhip dbg\|AB_RESULT" | cut -c1-400 | tail -4 | tee $out/r05_k_stamps_lj256k.txt
MOLLYHIP_DBG_TIMES=400 MOLLYHIP_LIB_AB=$R/ab/lib_stamps.so timeout 600 python tools/force_ab.py --child --workload lj1m --steps 800 2>&1 | grep "mhip dbg\|AB_RESULT" | cut -c1-400 | tail -3 | tee $out/r05_k_stamps_lj1m.txt
CMD="python $R/bench.py --workload lj1m --steps 150 --warmup 20 --profile-steps 20 --equil 300 --no-cpu-baseline --no-secondary"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$out/pmc_k1 -- $CMD > /dev/null 2> $R/$out/pmc_k1.err
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$out/pmc_k2 -- $CMD > /dev/null 2> $R/$out/pmc_k2.err
cd $R
for d in pmc_k1 pmc_k2; do python tools/pmc_kernels.py $out/$d k_forces k_build 2>&1 | cut -c1-600; done | tee $out/r05_k_pmc_lj1m.txt
rm -rf $out/pmc_k1 $out/pmc_k2
