# decode-step check on the MI355X box: Whisper parity tests, then serial passes of the headline workload (decode ms per step)
set -e
true

for cfg in "MIA_X=1" "MIA_DEC_W_NT=1"; do
  env $cfg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --replicas 1 --no-cpu-baseline --no-codec --no-config0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms/pass', d['ms_per_step'], 'decode', d['stages']['decode']['ms_per_pass'], 'ms/step', d['roofline']['avg_launch_ms'])"
done
