#!/bin/bash
# round 5, call ae: parity of the staging re-order across the generic variants (charged fluids, 6mrr, triclinic), then 6mrr_pme A/B against ab/lib_base.so
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_implementations.py tests/test_gpu_6mrr.py tests/test_gpu_triclinic.py tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
timeout 600 python tools/force_ab.py --workload 6mrr_pme --steps 2000 ab/lib_base.so tree ab/lib_base.so tree ab/lib_base.so tree 2>&1 | cut -c1-330 | tee $out/r05_ae_6mrr.txt
echo finished
