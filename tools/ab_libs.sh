# Same-box A/B of two builds of libmia.so (round 3's most useful measurement pattern: boxes differ by +-1.5 %, so two builds are only
# comparable inside ONE gpurun call).
#
#   here (CPU container):   bash tools/ab_libs.sh prepare <git-ref>     # builds <git-ref> -> ab_libs/libmia_base.so, the working tree -> ab_libs/libmia_new.so
#   on the GPU box:         gpurun -- 'bash tools/ab_libs.sh run "python bench.py --no-lm --no-codec --no-config0 --no-cpu-baseline" 2'
#
# `run` alternates base / new N times, prints value, serial-step ms and serial value of each bench line, and leaves the NEW library in place.
# ab_libs/ is git-ignored but travels with the gpurun snapshot.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
case "$1" in
  prepare)
    ref=${2:?git ref of the base build}
    mkdir -p ab_libs
    python -c "import __graft_entry__ as g; g.build()" > /dev/null
    cp mlx-swift-audio_amd/lib/libmia.so ab_libs/libmia_new.so
    tmp=$(mktemp -d)
    git worktree add -q --detach "$tmp/base" "$ref"
    (cd "$tmp/base" && python -c "import __graft_entry__ as g; g.build()" > /dev/null && cp mlx-swift-audio_amd/lib/libmia.so "$R/ab_libs/libmia_base.so")
    git worktree remove --force "$tmp/base"
    echo "ab_libs/: base = $ref, new = working tree"
    ;;
  run)
    cmd=${2:?bench command}
    n=${3:-2}
    mkdir -p gpurun_out
    for i in $(seq 1 "$n"); do
      for v in base new; do
        cp ab_libs/libmia_$v.so mlx-swift-audio_amd/lib/libmia.so
        $cmd > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
        python - "$v" gpurun_out/ab_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[1], d.get("value"), r.get("avg_launch_ms"), (r.get("execution") or {}).get("value"))
PY
      done
    done
    cp ab_libs/libmia_new.so mlx-swift-audio_amd/lib/libmia.so
    ;;
  *) echo "usage: ab_libs.sh prepare <git-ref> | run \"<bench command>\" [repeats]"; exit 2 ;;
esac
