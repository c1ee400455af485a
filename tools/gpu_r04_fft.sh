#!/bin/bash
# round 4: PME meshes through the FFT library (axes beyond 512 points) — parity on the GPU
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pme.py tests/test_gpu_triclinic.py tests/test_gpu_6mrr.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -15 | tee $out/r04_fft_tests.log
