#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in polympc_amd/libpolympc_amd.so polympc_amd/_variants/lib_rc19.so polympc_amd/_variants/lib_rc28.so; do
    echo "== A: $L"; PMPC_LIB=$L REPS=20 python tests/tools_config_bench.py A 2>&1 | grep config | cut -c1-100
  done
done
echo "== phase profile, lone instance"; PMPC_PHASE_PROFILE=1 B=1 python tests/tools_phase_profile.py
echo "== phase profile, 4096"; PMPC_PHASE_PROFILE=1 B=4096 python tests/tools_phase_profile.py
