#!/usr/bin/env bash
# Host frame in -> host frame out throughput of the worker pool (nv12_stream, unpaced) and the paced 60 fps case.
#   tools/stream_sweep.sh [outfile]        run from the repo root on a GPU box
set -uo pipefail
OUT=${1:-gpurun_out/stream_sweep.txt}
BIN=opencv-opencl_amd/lib/nv12_stream
: > "$OUT"
run() { echo "### $*" >> "$OUT"; timeout -k 10 120 "$BIN" "$@" 2>&1 | grep -E "^nv12_stream|^workers:|^placement|^done|^worker time|^latency|error" >> "$OUT"; }
for w in 1 1 2 2; do
  run --width 3840 --height 2160 --frames 3000 --workers $w
done
run --width 3840 --height 2160 --frames 3000 --workers 4                           # capped at two per GPU
run --width 3840 --height 2160 --frames 3000 --workers 4 --max-workers-per-gpu 0   # uncapped: what the cap avoids
run --width 3840 --height 2160 --frames 3000 --workers 8 --max-workers-per-gpu 0
echo "### MI_LUMAEQ_PIPE_COPY_STREAMS=1 (one copy stream per direction, as in round 2) --workers 1" >> "$OUT"
MI_LUMAEQ_PIPE_COPY_STREAMS=1 timeout -k 10 120 "$BIN" --width 3840 --height 2160 --frames 3000 --workers 1 2>&1 | grep -E "^done" >> "$OUT"
echo "### MI_LUMAEQ_PIPE_FLAT_PRIORITY=1 (all pipe streams at one priority, as in round 2) --workers 2" >> "$OUT"
MI_LUMAEQ_PIPE_FLAT_PRIORITY=1 timeout -k 10 120 "$BIN" --width 3840 --height 2160 --frames 3000 --workers 2 2>&1 | grep -E "^done" >> "$OUT"
run --width 3840 --height 2160 --frames 3000 --workers 1 --uv-policy device
run --width 3840 --height 2160 --frames 3000 --workers 1 --depth 2
run --width 3840 --height 2160 --frames 3000 --workers 1 --depth 8
run --width 3840 --height 2160 --frames 2000 --workers 1 --no-pin
run --width 3840 --height 2160 --frames 2000 --workers 2 --no-pin
run --width 3840 --height 2160 --frames 3000 --workers 1 --no-numa-bind
run --width 3840 --height 2160 --frames 3000 --workers 1 --uv copy
run --width 3840 --height 2160 --frames 3000 --workers 1 --op clahe
run --width 3840 --height 2160 --frames 1000 --workers 1 --op channels
run --width 1920 --height 1080 --frames 6000 --workers 1
run --width 3840 --height 2160 --frames 512 --workers 1 --paced --fps 60
cat "$OUT"
