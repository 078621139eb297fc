# rocprofv3 kernel trace of the LM decode step on bf16 and on packed q4 weights (Orpheus-3B).  Run as `gpurun -- 'bash tools/profile_lm.sh TAG'`.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for MODE in bf16 q4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_lm_$MODE -o run -- python3 $R/tools/lm_step_profile.py orpheus-3b $MODE 64 > $R/gpurun_out/${TAG}_lm_$MODE.log 2>&1
  tail -1 $R/gpurun_out/${TAG}_lm_$MODE.log
done
cd $R
for MODE in bf16 q4; do
  f=$(ls gpurun_out/${TAG}_lm_$MODE/*/run_kernel_stats.csv 2>/dev/null | head -1 || true)
  [ -z "$f" ] && f=$(find gpurun_out/${TAG}_lm_$MODE -name "*kernel_stats.csv" | head -1)
  echo "== $MODE: $f"; head -14 "$f" | cut -c1-200
done
