#!/bin/bash
cd $GRAFT_REPO_ROOT
PMPC_LIB=polympc_amd/_variants/lib_lscompact.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "robot or config_A or smoke or full_size or sqp_batch" 2>&1 | tail -3
for i in 1 2 3; do
  for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_lscompact.so; do
    echo "== A, D: $L"; PMPC_LIB=$L REPS=20 python tests/tools_config_bench.py A D 2>&1 | grep config | cut -c1-100
  done
done
for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_lscompact.so; do echo "== R 16 nodes / 11 nodes: $L"; PMPC_LIB=$L REPS=10 BA=2048 python tests/tools_config_bench.py R 2>&1 | grep config | cut -c1-100; PMPC_LIB=$L REPS=10 BA=4096 P=5 S=2 python tests/tools_config_bench.py R 2>&1 | grep config | cut -c1-100; done
