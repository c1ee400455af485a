#!/bin/bash
# round 4: PME on a TriclinicBoundary — parity on the GPU
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pme.py tests/test_gpu_triclinic.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -15 | tee $out/r04_tri_tests.log
