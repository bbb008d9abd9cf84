set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
bash tools/run_trace.sh
bash tools/run_trace.sh -DDFX_GELU_POLY
bash tools/ab.sh "" "-DDFX_GELU_POLY"
bash tools/ab.sh "-DDFX_SWP" "-DDFX_SWP -DDFX_GELU_POLY"
} > gpurun_out/r2/exp1.log 2>&1
