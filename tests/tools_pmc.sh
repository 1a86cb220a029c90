cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o $name -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $R/gpurun_out/pmc_sq1
