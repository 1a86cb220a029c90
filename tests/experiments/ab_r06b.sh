#!/bin/bash
# same-box A/B, round 6: (1) the headline kernel with the KKT build reading the LOWER triangle of H only (-DPMPC_EXPERIMENT_FORCE_SYMLOWER=1: the read-side cost of any half /
# packed storage of the BFGS matrix) against the round-5 library; (2) the condensed register kernels with H x of the dual residual from the KKT identity (current library)
# against the round-5 library on configs B and R
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in polympc_amd/_variants/lib_r05.so polympc_amd/_variants/lib_symlower.so; do
    echo "== A: $L"; PMPC_ABI_ANY=1 PMPC_LIB=$L REPS=20 python tests/tools_config_bench.py A 2>&1 | grep config | cut -c1-100
  done
done
for i in 1 2; do
  for L in polympc_amd/_variants/lib_r05.so polympc_amd/libpolympc_amd.so; do
    echo "== B, R: $L"; PMPC_ABI_ANY=1 PMPC_LIB=$L REPS=5 BA=2048 python tests/tools_config_bench.py B R 2>&1 | grep config | cut -c1-100
  done
done
