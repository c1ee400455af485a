#!/bin/bash
# kernel trace of two ranks sharing the one GPU through mhip_domain_run: which dispatches make a plain ghosted step (profiles/rNN_timeline_two_ranks.txt)
out=gpurun_out; mkdir -p $out; tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
rm -rf $out/prof_dom2
MOLLYHIP_ENGINE_LOOP=1 MOLLYHIP_DIST_BACKEND=gloo MOLLYHIP_FORCE_DEVICE=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_dom2 -- \
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --workload ${2:-lj256k} --steps 100 --warmup 20 --equil 200 --profile-steps 20 \
   > $out/${tag}_trace_bench.json 2> $out/${tag}_trace_bench.err
python tools/dom_trace_summary.py $out/prof_dom2 | tee $out/${tag}_timeline_two_ranks.txt
rm -rf $out/prof_dom2
