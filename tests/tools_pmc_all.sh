#!/bin/bash
# Developer tool: the round's full set of rocprofv3 summaries (kernel stats + PMC passes, tests/tools_pmc.sh) on the GPU box:
#   tests/tools_pmc_all.sh r06   ->  gpurun_out/r06_cfg{A,B,B_blockbfgs,C,C_lone_wg4,R128,R128_blockbfgs}_{kernel_stats.csv,pmc_summary.json}
R=$GRAFT_REPO_ROOT
T=${1:-r06}
[ -x $R/tests/experiments/hbm_counter_calibration ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tests/experiments/hbm_counter_calibration $R/tests/experiments/hbm_counter_calibration.hip
bash $R/tests/tools_pmc.sh ${T}_cfgA > /dev/null 2>&1
bash $R/tests/tools_pmc.sh ${T}_cfgB "python $R/tests/tools_config_bench.py B" > /dev/null 2>&1
HESSIAN_UPDATE=1 bash $R/tests/tools_pmc.sh ${T}_cfgB_blockbfgs "python $R/tests/tools_config_bench.py B" > /dev/null 2>&1
bash $R/tests/tools_pmc.sh ${T}_cfgC "python $R/tests/tools_config_bench.py C" > /dev/null 2>&1
BC=1 bash $R/tests/tools_pmc.sh ${T}_cfgC_lone_wg4 "python $R/tests/tools_config_bench.py C" > /dev/null 2>&1
BA=2048 bash $R/tests/tools_pmc.sh ${T}_cfgR128 "python $R/tests/tools_config_bench.py R" > /dev/null 2>&1
BA=2048 HESSIAN_UPDATE=1 bash $R/tests/tools_pmc.sh ${T}_cfgR128_blockbfgs "python $R/tests/tools_config_bench.py R" > /dev/null 2>&1
cd $R/gpurun_out
for c in A B B_blockbfgs C C_lone_wg4 R128 R128_blockbfgs; do
  echo "== $c"; python - <<P
import json
try:
    d=json.load(open("${T}_cfg${c}_pmc_summary.json"))
    t=d.get("traffic") or {}; l2=d.get("l2") or {}
    print("traffic GB", (t.get("bytes_per_launch") or 0)/1e9, "fetch", (t.get("fetch_bytes_per_launch") or 0)/1e9, "write", (t.get("write_bytes_per_launch") or 0)/1e9, "L2 hit", l2.get("hit_rate"), "build", d.get("library_build_id"))
except Exception as e: print("missing", e)
P
  head -3 ${T}_cfg${c}_kernel_stats.csv 2>/dev/null | cut -c1-200
done
# keep the merged output small: the raw counter CSVs stay on the box
rm -rf $R/gpurun_out/pmc_${T}_* $R/gpurun_out/prof_${T}_*
