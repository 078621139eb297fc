set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03g_kt -o run -- python3 $R/bench.py --steps 1 --warmup 1 --replicas 1 --no-cpu-baseline --no-codec --no-config0 --no-lm --max-new-tokens 8 > $R/gpurun_out/r03g_kt.log 2>&1
cd $R
f=$(find gpurun_out/r03g_kt -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-160
