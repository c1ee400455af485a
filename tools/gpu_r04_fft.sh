#!/bin/bash
# round 4: PME meshes through the FFT library (axes beyond 512 points), PME on a TriclinicBoundary — parity on the GPU
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider -x -k "triclinic or fft" 2>&1 | tail -15 | tee $out/r04_fft_tests.log
