#!/usr/bin/env bash
# What a streaming worker costs the HOST (nv12_stream prints process CPU time per frame, getrusage): paced 4K60 and unpaced, with the
# polling wait of round 3 (MI_LUMAEQ_PIPE_WAIT=spin), the blocking wait (sync) and the default poll-with-back-off; registered and
# pageable frame rings; one and two workers.
#   tools/stream_cpu.sh [outfile]        run from the repo root on a GPU box
set -uo pipefail
OUT=${1:-gpurun_out/stream_cpu.txt}
BIN=opencv-opencl_amd/lib/nv12_stream
: > "$OUT"
run() { echo "### MI_LUMAEQ_PIPE_WAIT=${MI_LUMAEQ_PIPE_WAIT:-backoff(default)} $*" >> "$OUT"
        timeout -k 10 120 "$BIN" "$@" 2>&1 | grep -E "^done|^worker time|^host CPU|^latency|error" >> "$OUT"; }
for mode in spin sync backoff; do
  export MI_LUMAEQ_PIPE_WAIT=$mode
  run --width 3840 --height 2160 --frames 600 --workers 1 --paced --fps 60
  run --width 3840 --height 2160 --frames 3000 --workers 1
  run --width 3840 --height 2160 --frames 3000 --workers 1
  run --width 3840 --height 2160 --frames 3000 --workers 2
done
unset MI_LUMAEQ_PIPE_WAIT
for spin in 0 5 50; do
  echo "### MI_LUMAEQ_PIPE_WAIT_SPIN_US=$spin" >> "$OUT"
  MI_LUMAEQ_PIPE_WAIT_SPIN_US=$spin run --width 3840 --height 2160 --frames 3000 --workers 1
  MI_LUMAEQ_PIPE_WAIT_SPIN_US=$spin run --width 3840 --height 2160 --frames 600 --workers 1 --paced --fps 60
done
run --width 3840 --height 2160 --frames 600 --workers 1 --paced --fps 60 --no-pin
run --width 3840 --height 2160 --frames 2000 --workers 1 --no-pin
run --width 3840 --height 2160 --frames 2000 --workers 2 --no-pin
# 1080p: the default (300 us of polling, then sleeps), sleeping from the first poll, and never sleeping
run --width 1920 --height 1080 --frames 6000 --workers 1
MI_LUMAEQ_PIPE_WAIT=backoff run --width 1920 --height 1080 --frames 6000 --workers 1
MI_LUMAEQ_PIPE_WAIT=spin run --width 1920 --height 1080 --frames 6000 --workers 1
MI_LUMAEQ_PIPE_WAIT_SPIN_US=1000000 run --width 1920 --height 1080 --frames 6000 --workers 1
run --width 1280 --height 720 --frames 8000 --workers 1
MI_LUMAEQ_PIPE_WAIT_SPIN_US=1000000 run --width 1280 --height 720 --frames 8000 --workers 1
run --width 3840 --height 2160 --frames 600 --workers 1 --paced --fps 60 --op clahe
cat "$OUT"
