# Decode-step regression check on the MI355X box: Whisper / LM parity tests, then serial passes of the headline workload (decode ms per
# step).  Run as `gpurun -- 'bash tools/ab_decode.sh'`; prefix a bench line with VAR=1 to A/B an experimental switch.
set -e
timeout -k 10 900 python -m pytest tests/test_whisper_gpu.py tests/test_timing_gpu.py tests/test_fullsize_gpu.py tests/test_lm_gpu.py -m gpu -q -x > gpurun_out/ab_tests.log 2>&1 || { tail -40 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --replicas 1 --no-cpu-baseline --no-codec --no-config0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/pass', d['ms_per_step'], 'decode', d['stages']['decode']['ms_per_pass'], 'ms/step', d['roofline']['avg_launch_ms'])"
