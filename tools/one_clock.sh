#!/usr/bin/env bash
# tools/one_clock.sh <tag> [equalize|clahe] -- see tools/one_clock.py.  The program after `--` is python3 itself; kernel-trace only.
set -uo pipefail
TAG=${1:-rXX}; OP=${2:-equalize}
OUT=gpurun_out/${TAG}_one_clock_${OP}
mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS=(--op "$OP" --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-second-resolution)
python3 bench.py "${ARGS[@]}" > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || { tail -5 "$OUT/bench_plain.err"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py "${ARGS[@]}" > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || { tail -5 "$OUT/bench_under_rocprof.err"; exit 1; }
python3 tools/one_clock.py "$OUT" "$OP" "gpurun_out/${TAG}_one_clock_${OP}.json" | tee "gpurun_out/${TAG}_one_clock_${OP}.txt"
rm -rf "$OUT/trace"
