#!/bin/bash
# round 6, VERDICT item 3: the miscompiled hook build of the small condensed kernel (-DPMPC_EXPERIMENT_SMALL_POL=2 -DPMPC_EXPERIMENT_CND_WITH_RUIZ) with the four forms of
# pivot_lane_setup (-DPMPC_PIVOT_SETUP_FORM=0..3: shipped / asm volatile / EXEC saved and restored / v_cndmask selects), each under the device-poisoning harness
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in 0 1 2 3; do
  PMPC_LIB=polympc_amd/_variants/lib_faulty_f$f.so PMPC_POISON=1 RUNS=2 timeout 300 python tests/experiments/pol_small_probe.py > gpurun_out/exec_probe_f$f.log 2>&1
  echo "== form $f: rc $?"; grep "^run" gpurun_out/exec_probe_f$f.log | cut -c1-200; grep "columns" gpurun_out/exec_probe_f$f.log | head -1 | cut -c1-300
done
