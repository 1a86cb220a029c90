#!/bin/bash
# same-box A/B, round 6: config C (large-instance kernel, condensed mode) with H x of the dual residual from the KKT identity (current library) against the round-5 library;
# then the launch timeline of the headline kernel (library with -DPMPC_EXPERIMENT_WG_STAMPS)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for L in polympc_amd/_variants/lib_r05.so polympc_amd/libpolympc_amd.so; do
    echo "== C: $L"; PMPC_ABI_ANY=1 PMPC_LIB=$L REPS=4 python tests/tools_config_bench.py C 2>&1 | grep config | cut -c1-110
    echo "== C lone / 256: $L"; for BC in 1 256; do PMPC_ABI_ANY=1 PMPC_LIB=$L REPS=4 BC=$BC python tests/tools_config_bench.py C 2>&1 | grep config | cut -c1-110; done
  done
done
echo "== timeline"
PMPC_LIB=polympc_amd/_variants/lib_stamps.so python tests/experiments/launch_timeline.py
