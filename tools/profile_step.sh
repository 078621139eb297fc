# one decode step, kernel by kernel (rocprofv3 --kernel-trace of a short serial bench run + tools/step_trace.py)
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-step}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -o run -- python3 $R/bench.py --steps 1 --warmup 1 --replicas 1 --no-cpu-baseline --no-codec --no-config0 --no-lm --max-new-tokens 48 > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
cd $R
python3 tools/step_trace.py $(find gpurun_out/$TAG -name "*kernel_trace.csv" | head -1) 5 > gpurun_out/${TAG}_trace.txt
cat gpurun_out/${TAG}_trace.txt
rm -rf gpurun_out/$TAG
