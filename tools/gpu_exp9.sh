#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_triclinic.py -m gpu -q --timeout 600 -p no:cacheprovider > $out/x9_tri.log 2>&1; echo "rc $?" >> $out/x9_tri.log
tail -5 $out/x9_tri.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_triclinic.py > $out/x9_rest.log 2>&1; echo "rc $?" >> $out/x9_rest.log
tail -4 $out/x9_rest.log
