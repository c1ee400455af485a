#!/bin/bash
# round 4, call u: stages of k_build on 6mrr (MOLLYHIP_BUILD_DEBUG=n stops the kernel behind stage n; the lists are garbage then — only the kernel's time counts)
out=gpurun_out; mkdir -p $out; R=$PWD
for n in 0 1 2 3 4 7; do
  cd /tmp && export TMPDIR=/tmp
  MOLLYHIP_BUILD_DEBUG=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_u$n -o u -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 30 --equil 0 > /dev/null 2>&1
  cd $R
  f=$(find gpurun_out/prof_u$n -name "*kernel_stats.csv" | head -1)
  echo "== BUILD_DEBUG=$n" | tee -a $out/r04_u_build_stages.txt
  grep -E "k_build<|k_cell_keys|k_permute|Sort|sort|scan|Scan" $f | cut -d, -f1-6 | cut -c1-260 | tee -a $out/r04_u_build_stages.txt
  rm -rf gpurun_out/prof_u$n
done
