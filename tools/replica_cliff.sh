# Kernel traces of the pipelined bench at 3 and 4 replicas (the 3 -> 4 cliff of profiles/r02_schedule_sweeps.txt): per-queue timelines.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for N in 3 4; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cliff_r$N -o run -- python3 $R/bench.py --steps $((2*N)) --warmup 1 --replicas $N --no-codec --no-lm --no-cpu-baseline --no-config0 --max-new-tokens 96 > $R/gpurun_out/cliff_r$N.json 2> $R/gpurun_out/cliff_r$N.err
  python3 -c "import json;d=json.loads(open('$R/gpurun_out/cliff_r$N.json').read().strip().splitlines()[-1]);print('replicas',$N,'value',d['value'],'ms/pass',d['ms_per_step'])"
  python3 $R/tools/replica_trace.py $(find $R/gpurun_out/cliff_r$N -name "*kernel_trace.csv" | head -1)
done
