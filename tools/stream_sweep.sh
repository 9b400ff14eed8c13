#!/usr/bin/env bash
# Host frame in -> host frame out throughput of the worker pool (nv12_stream, unpaced) and the paced 60 fps case.
#   tools/stream_sweep.sh [outfile]        run from the repo root on a GPU box
set -uo pipefail
OUT=${1:-gpurun_out/stream_sweep.txt}
BIN=opencv-opencl_amd/lib/nv12_stream
: > "$OUT"
run() { echo "### $*" >> "$OUT"; timeout -k 10 120 "$BIN" "$@" 2>&1 | grep -E "^nv12_stream|^done|^latency|error" >> "$OUT"; }
for pol in host device; do
  for w in 1 2 4; do
    run --width 3840 --height 2160 --frames 2000 --workers $w --uv-policy $pol
  done
done
run --width 3840 --height 2160 --frames 2000 --workers 1 --depth 2
run --width 3840 --height 2160 --frames 2000 --workers 1 --depth 8
run --width 3840 --height 2160 --frames 2000 --workers 1 --no-pin
run --width 3840 --height 2160 --frames 2000 --workers 2 --no-pin
run --width 3840 --height 2160 --frames 2000 --workers 1 --uv copy
run --width 3840 --height 2160 --frames 2000 --workers 1 --uv copy --uv-policy device
run --width 3840 --height 2160 --frames 2000 --workers 1 --op clahe
run --width 3840 --height 2160 --frames 1000 --workers 1 --op channels
run --width 1920 --height 1080 --frames 4000 --workers 1
run --width 3840 --height 2160 --frames 512 --workers 1 --paced --fps 60
cat "$OUT"
