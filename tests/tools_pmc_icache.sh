#!/bin/bash
# Developer tool: instruction-cache counters of the bench kernel (run via gpurun). Output under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
BENCH="python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_icache -o icache -- $BENCH > $R/gpurun_out/pmc_${TAG}_icache.log 2>&1
tail -3 $R/gpurun_out/pmc_${TAG}_icache.log
python - <<PY
import csv,glob,collections
fs=glob.glob("$R/gpurun_out/pmc_${TAG}_icache/*counter_collection.csv")
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "sqp_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, sum(v)/len(v))
PY
