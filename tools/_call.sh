mkdir -p gpurun_out/r03x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03x/kernels_trace -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/r03x/per_kernel.json 2> $R/gpurun_out/r03x/per_kernel.err; echo "trace rc=$?"
f=$(find $R/gpurun_out/r03x/kernels_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r03x/kernel_stats.csv; head -30 "$f" | cut -c1-130
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r03x/small_sq -- python3 $R/tools/probe_small.py > $R/gpurun_out/r03x/small_sq.log 2>&1; echo "sq rc=$?"
python3 - <<'PY'
import csv,glob,os,json
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03x/small_sq/**/*counter_collection.csv',recursive=True)[0]
acc={}
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-60:]
    if 'stats' in k or 'noise_obs' in k:
        a=acc.setdefault(k,{}).setdefault(r['Counter_Name'],[0,0]); a[0]+=float(r['Counter_Value']); a[1]+=1
out={k:{c:round(v[0]/v[1],1) for c,v in d.items()} for k,d in acc.items()}
json.dump(out,open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03x/stats_noise_sq.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
