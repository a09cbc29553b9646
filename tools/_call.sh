mkdir -p gpurun_out/r03d
python tools/probe_tile_map.py > gpurun_out/r03d/tile_map.json 2> gpurun_out/r03d/tile_map.txt; echo "tile_map rc=$?"; cat gpurun_out/r03d/tile_map.txt | tail -9
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03d/small_trace -- python3 $GRAFT_REPO_ROOT/tools/probe_small.py > $GRAFT_REPO_ROOT/gpurun_out/r03d/small_trace.log 2>&1; echo "trace rc=$?"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03d/small_trace -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03d/small_sq -- python3 $GRAFT_REPO_ROOT/tools/probe_small.py > $GRAFT_REPO_ROOT/gpurun_out/r03d/small_sq.log 2>&1; echo "sq rc=$?"
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03d/small_sq/**/*counter_collection.csv',recursive=True)[0]
acc={}
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-60:]
    if 'stats' in k or 'noise_obs' in k:
        a=acc.setdefault((k,r['Counter_Name']),[0,0]); a[0]+=float(r['Counter_Value']); a[1]+=1
for k,v in sorted(acc.items()): print(k, round(v[0]/v[1],1))
PY
