cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03z/cull2 -- python3 $GRAFT_REPO_ROOT/tools/probe_cull.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03z/cull2 -name "*kernel_stats.csv" | head -1); grep "k_uf_" "$f" | cut -c1-160
