#!/bin/bash
# round 6 (late): the rank-m update's operands A(j, .) from the node blocks + a transposed D~ table in LDS (lib_atabB / lib_atabR) against loads from the workspace (the shipped library)
cd $GRAFT_REPO_ROOT
PMPC_LIB=polympc_amd/_variants/lib_atabB.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cstr or config_B or full_size" 2>&1 | tail -2
PMPC_LIB=polympc_amd/_variants/lib_atabR.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "robot or fixture or sixteen or mirroring or filter" 2>&1 | tail -2
for i in 1 2; do
  for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_atabB.so; do echo "== B: $L"; PMPC_LIB=$L REPS=8 python tests/tools_config_bench.py B 2>&1 | grep config | cut -c1-110; done
  for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_atabR.so; do echo "== R 16 nodes / 11 nodes: $L"; PMPC_LIB=$L REPS=10 BA=2048 python tests/tools_config_bench.py R 2>&1 | grep config | cut -c1-110; PMPC_LIB=$L REPS=10 BA=4096 P=5 S=2 python tests/tools_config_bench.py R 2>&1 | grep config | cut -c1-110; done
done
