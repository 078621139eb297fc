set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 1 --warmup 1 --replicas 1 --no-cpu-baseline --no-codec --no-config0 --no-lm"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f_kt -o run -- $B --max-new-tokens 32 > $R/gpurun_out/r03f_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03f_pmc_fetch -o run -- $B --max-new-tokens 32 > $R/gpurun_out/r03f_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03f_pmc_write -o run -- $B --max-new-tokens 32 > $R/gpurun_out/r03f_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r03f_pmc_mfma -o run -- $B --max-new-tokens 4 > $R/gpurun_out/r03f_pmc_mfma.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/r03f_pmc_fetch gpurun_out/r03f_pmc_write gpurun_out/r03f_pmc_summary.json
python3 tools/pmc_mfma_summary.py gpurun_out/r03f_pmc_mfma gpurun_out/r03f_pmc_mfma_summary.json | head -8
