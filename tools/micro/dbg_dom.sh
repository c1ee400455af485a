out=gpurun_out; mkdir -p $out
MOLLYHIP_DEBUG=1 MOLLYHIP_ENGINE_LOOP=1 MOLLYHIP_DIST_BACKEND=gloo MOLLYHIP_FORCE_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29507 \
     bench.py --gpus 2 --workload lj256k --steps 100 --warmup 100 --equil 300 > $out/dbg_dom.json 2> $out/dbg_dom.err
grep -n 'mhip\|Error' $out/dbg_dom.err | grep -A40 'step 263' | cut -c1-200 | head -60
