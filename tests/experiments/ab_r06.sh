#!/bin/bash
# same-box A/B: the round-5 library (polympc_amd/_variants/lib_r05.so) against the current one, config A and D alternating; then the mid-size team-kernel question
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in polympc_amd/_variants/lib_r05.so polympc_amd/libpolympc_amd.so; do
    echo "== $L"; PMPC_ABI_ANY=1 PMPC_LIB=$L REPS=20 python tests/tools_config_bench.py A D 2>&1 | grep config | cut -c1-120
  done
done
echo "== mid-size team kernel: robot 21 nodes (168 KKT rows)"
for BA in 1 64 256 512; do for W in 0 1; do echo "BA=$BA PMPC_BIG_WG4=$W"; BA=$BA P=5 S=4 PMPC_BIG_WG4=$W REPS=10 python tests/tools_config_bench.py R 2>&1 | grep config | cut -c1-110; done; done
