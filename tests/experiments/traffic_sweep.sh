cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for B in 1024 2048 4096; do
for C in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/ts_${B}_$C -o x -- python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 --configs= --no-replay --batch $B > $R/gpurun_out/ts_${B}_$C.log 2>&1
done; done
python - <<'P'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
for B in (1024,2048,4096):
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        fs=glob.glob(f'{R}/gpurun_out/ts_{B}_{C}/**/*counter_collection.csv',recursive=True)
        tot=0;n=0
        for f in fs:
            for r in csv.DictReader(open(f)):
                if 'sqp_kernel' in r['Kernel_Name'] and r['Counter_Name']==C:
                    tot+=float(r['Counter_Value']);n+=1
        print(B,C,n,tot/max(n,1))
P
