out=gpurun_out; mkdir -p $out
bash profiles/collect.sh lj1m r02_lj1m 300 > $out/collect_lj1m.log 2>&1
bash profiles/collect.sh lj256k r02_lj256k 300 > $out/collect_lj256k.log 2>&1
bash profiles/collect.sh 6mrr_pme r02_6mrr_pme 400 > $out/collect_6mrr.log 2>&1
ls -la $out/prof_r02_lj1m
