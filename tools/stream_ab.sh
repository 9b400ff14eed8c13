#!/usr/bin/env bash
# A/B of the streaming pipeline's knobs (nv12_stream, unpaced, 4K): copy streams per direction x workers x NUMA binding.
#   tools/stream_ab.sh [outfile]        run from the repo root on a GPU box
set -uo pipefail
OUT=${1:-gpurun_out/stream_ab.txt}
BIN=opencv-opencl_amd/lib/nv12_stream
: > "$OUT"
run() { echo "### $*" >> "$OUT"; timeout -k 10 120 env "${ENVV[@]}" "$BIN" "${ARGS[@]}" 2>&1 | grep -E "^nv12_stream|^placement|^done|^latency|error" >> "$OUT"; }
for rep in 1 2; do
  for cs in 1 2; do
    for w in 1 2 4; do
      ENVV=(MI_LUMAEQ_PIPE_COPY_STREAMS=$cs); ARGS=(--width 3840 --height 2160 --frames 3000 --workers $w)
      echo "### copy_streams=$cs workers=$w" >> "$OUT"
      timeout -k 10 120 env "${ENVV[@]}" "$BIN" "${ARGS[@]}" 2>&1 | grep -E "^done|error" >> "$OUT"
    done
  done
done
for cs in 1 2; do
  echo "### copy_streams=$cs workers=1 --no-numa-bind" >> "$OUT"
  timeout -k 10 120 env MI_LUMAEQ_PIPE_COPY_STREAMS=$cs "$BIN" --width 3840 --height 2160 --frames 3000 --workers 1 --no-numa-bind 2>&1 | grep -E "^placement|^done|error" >> "$OUT"
  echo "### copy_streams=$cs workers=1 --no-pin" >> "$OUT"
  timeout -k 10 120 env MI_LUMAEQ_PIPE_COPY_STREAMS=$cs "$BIN" --width 3840 --height 2160 --frames 2000 --workers 1 --no-pin 2>&1 | grep -E "^done|error" >> "$OUT"
  echo "### copy_streams=$cs workers=2 --no-pin" >> "$OUT"
  timeout -k 10 120 env MI_LUMAEQ_PIPE_COPY_STREAMS=$cs "$BIN" --width 3840 --height 2160 --frames 2000 --workers 2 --no-pin 2>&1 | grep -E "^done|error" >> "$OUT"
  echo "### copy_streams=$cs workers=1 1080p" >> "$OUT"
  timeout -k 10 120 env MI_LUMAEQ_PIPE_COPY_STREAMS=$cs "$BIN" --width 1920 --height 1080 --frames 6000 --workers 1 2>&1 | grep -E "^done|error" >> "$OUT"
  echo "### copy_streams=$cs workers=1 paced 60 fps" >> "$OUT"
  timeout -k 10 120 env MI_LUMAEQ_PIPE_COPY_STREAMS=$cs "$BIN" --width 3840 --height 2160 --frames 512 --workers 1 --paced --fps 60 2>&1 | grep -E "^placement|^done|^latency|error" >> "$OUT"
done
cat "$OUT"
