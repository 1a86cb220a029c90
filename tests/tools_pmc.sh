#!/bin/bash
# Developer tool: rocprofv3 evidence for a command on the GPU box (run via gpurun). PMC passes are separate from each other and use
# --kernel-trace only (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots). Output under gpurun_out/.
#   tests/tools_pmc.sh TAG                       -> the bench line (config A):   python bench.py --steps 10 --warmup 2 ...
#   tests/tools_pmc.sh TAG "python tests/tools_config_bench.py C"   -> any other command (config B / C / D kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
CMD=${2:-"python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 --configs= --no-replay --detail-out= "}
CMDT=${2:-"python $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 --configs= --no-replay --detail-out= "}
cd $R
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -o $name -- $CMD > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; }
# ONLY=tcc tests/tools_pmc.sh TAG [CMD]: just the L2 hit / miss pass (added to an existing collection of the same tag)
if [ "$ONLY" = "tcc" ]; then run tcc TCC_HIT_sum TCC_MISS_sum; exit 0; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- $CMDT > $R/gpurun_out/prof_$TAG.log 2>&1
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum   # L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2 section); FETCH_SIZE counts the fabric requests behind it, Infinity-Cache hits included
# counter calibration on a known byte count with the same access width (8 B / lane)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_calfetch -o cal -- $R/tests/experiments/hbm_counter_calibration > $R/gpurun_out/pmc_${TAG}_calfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_calwrite -o cal -- $R/tests/experiments/hbm_counter_calibration > $R/gpurun_out/pmc_${TAG}_calwrite.log 2>&1
python $R/tests/tools_pmc_summary.py $TAG > $R/gpurun_out/pmc_${TAG}_summary.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \;
tail -5 $R/gpurun_out/pmc_${TAG}_summary.log
