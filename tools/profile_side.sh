set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_codecs -o run -- python3 $R/tools/bench_codecs.py > $R/gpurun_out/r03_codecs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_orpheus -o run -- python3 $R/tools/bench_orpheus.py orpheus-3b 210 64 > $R/gpurun_out/r03_orpheus.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_cosy -o run -- python3 $R/tools/bench_cosyvoice2.py 15 > $R/gpurun_out/r03_cosy.log 2>&1
tail -1 $R/gpurun_out/r03_codecs.log | cut -c1-200; tail -1 $R/gpurun_out/r03_orpheus.log | cut -c1-300; tail -1 $R/gpurun_out/r03_cosy.log | cut -c1-300
