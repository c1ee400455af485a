#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -3
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree tree 2>&1 | tee $out/r04_g_ab_6mrr.txt
