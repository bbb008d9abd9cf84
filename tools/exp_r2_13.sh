cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
DFX_TRACE_B=1 bash tools/run_trace.sh > gpurun_out/r2/exp13.log 2>&1
