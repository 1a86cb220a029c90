#!/bin/bash
# Developer tool: rocprofv3 evidence for the bench kernel on the GPU box (run via gpurun). PMC passes are separate from each
# other and use --kernel-trace only (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots). Output under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
BENCH="python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/prof_$TAG.log 2>&1
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -o $name -- $BENCH > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
# counter calibration on a known byte count with the same access width (8 B / lane)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_calfetch -o cal -- $R/tests/experiments/hbm_counter_calibration > $R/gpurun_out/pmc_${TAG}_calfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_calwrite -o cal -- $R/tests/experiments/hbm_counter_calibration > $R/gpurun_out/pmc_${TAG}_calwrite.log 2>&1
ls $R/gpurun_out | head -40
