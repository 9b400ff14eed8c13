#!/usr/bin/env bash
# Quick A/B of the streaming pool (nv12_stream, unpaced): 4K with 1 and 2 workers, 1080p and 720p with one.
#   tools/stream_ab.sh [outfile]        run from the repo root on a GPU box; knobs come from the environment (include/mi_lumaeq_tuning.h)
set -uo pipefail
OUT=${1:-gpurun_out/stream_ab.txt}
BIN=opencv-opencl_amd/lib/nv12_stream
: > "$OUT"
for w in 1 2; do
  echo "### 4K workers=$w" >> "$OUT"
  for rep in 1 2 3; do timeout -k 10 60 "$BIN" --width 3840 --height 2160 --frames 3000 --workers $w 2>&1 | grep -E "^done|error" | cut -c1-70 >> "$OUT"; done
done
echo "### 1080p workers=1" >> "$OUT"
for rep in 1 2; do timeout -k 10 60 "$BIN" --width 1920 --height 1080 --frames 8000 --workers 1 2>&1 | grep -E "^done|error" | cut -c1-70 >> "$OUT"; done
echo "### 720p workers=1" >> "$OUT"
for rep in 1 2; do timeout -k 10 60 "$BIN" --width 1280 --height 720 --frames 12000 --workers 1 2>&1 | grep -E "^done|error" | cut -c1-70 >> "$OUT"; done
cat "$OUT"
