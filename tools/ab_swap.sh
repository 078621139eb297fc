# Same-box A/B of two builds of lib/libmia.so (box-to-box variance on the pool is +-3 %, more than most kernel changes):
#   here:    cp mlx-swift-audio_amd/lib/libmia.so mlx-swift-audio_amd/lib/libmia_prev.so ; <edit, rebuild> ;
#            cp mlx-swift-audio_amd/lib/libmia.so mlx-swift-audio_amd/lib/libmia_new.so
#   then:    gpurun -- 'bash tools/ab_swap.sh prev new'
# Alternates the named builds (each run a fresh process: serial passes of the headline workload) and prints the stage times.
# Delete the libmia_*.so copies afterwards (they are git-ignored but travel with every gpurun snapshot).
set -e
L=mlx-swift-audio_amd/lib
for i in 1 2; do
  for v in "$@"; do
    cp $L/libmia_$v.so $L/libmia.so
    timeout -k 10 200 python bench.py --steps 2 --warmup 1 --replicas 1 --no-cpu-baseline --no-codec --no-config0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('$v', 'ms/pass', d['ms_per_step'], 'dec ms/step', d['roofline']['avg_launch_ms'], 'enc_gemm', s['enc_gemm']['ms_per_pass'], 'enc_att', s['enc_attention']['ms_per_pass'], 'norm', s['enc_norm']['ms_per_pass'])"
  done
done
