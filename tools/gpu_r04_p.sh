#!/bin/bash
# (record of an experiment: the epilogue integrator and MOLLYHIP_VV_EPILOGUE were removed afterwards, profiles/r04_force_ab.txt §12)
# round 4, call p: the integrator in the epilogue of the plain force pass (k_forces<…, VV>) — parity, A/B at 1M and 256k atoms
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_vv_epilogue.py tests/test_gpu_cadence.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -12 | tee $out/r04_p_tests.log
timeout 900 python tools/force_ab.py --workload lj1m --steps 1500 tree:MOLLYHIP_VV_EPILOGUE=0 tree tree:MOLLYHIP_VV_EPILOGUE=0 tree 2>&1 | tee $out/r04_p_ab_lj1m.txt
timeout 900 python tools/force_ab.py --workload lj256k --steps 3000 tree:MOLLYHIP_VV_EPILOGUE=0 tree tree:MOLLYHIP_VV_EPILOGUE=0 tree 2>&1 | tee $out/r04_p_ab_lj256k.txt
