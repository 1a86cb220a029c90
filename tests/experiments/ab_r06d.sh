#!/bin/bash
cd $GRAFT_REPO_ROOT
PMPC_LIB=polympc_amd/_variants/lib_condsmall.so python tests/experiments/condsmall_probe.py
PMPC_POISON=1 PMPC_LIB=polympc_amd/_variants/lib_condsmall.so python tests/experiments/condsmall_probe.py | head -3
for i in 1 2 3; do
  for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_condsmall.so; do
    echo "== A, D: $L"; PMPC_LIB=$L REPS=20 python tests/tools_config_bench.py A D 2>&1 | grep config | cut -c1-100
  done
done
for BA in 1 64 512; do for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_condsmall.so; do echo "BA=$BA $L"; BA=$BA PMPC_LIB=$L REPS=20 python tests/tools_config_bench.py A 2>&1 | grep config | cut -c1-100; done; done
