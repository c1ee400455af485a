#!/bin/bash
# (record of an experiment: the kernels it timed — k_pme_xy, k_pme_yx / k_pme_yz and their MOLLYHIP_PME_PLANES / MOLLYHIP_PME_DEBUG switches — were removed afterwards, profiles/r04_force_ab.txt §15)
# round 4, call x: stages of k_pme_yx (MOLLYHIP_PME_DEBUG=11..13 stops the kernel behind a stage; timing only)
out=gpurun_out; mkdir -p $out; R=$PWD
for d in 11 12 13 0; do
  cd /tmp && export TMPDIR=/tmp && MOLLYHIP_PME_DEBUG=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x$d -o x -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 200 --equil 0 > /dev/null 2>&1; cd $R
  f=$(find gpurun_out/prof_x$d -name "*kernel_stats.csv" | head -1)
  echo "== PME_DEBUG=$d" | tee -a $out/r04_x_yx_stages.txt
  grep -E "k_pme_y[xz]|k_pme_z_r2c" $f | cut -d, -f1-5 | cut -c1-200 | tee -a $out/r04_x_yx_stages.txt
  rm -rf gpurun_out/prof_x$d
done
