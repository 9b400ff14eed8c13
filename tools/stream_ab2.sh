BIN=opencv-opencl_amd/lib/nv12_stream
for w in 1 2; do
  echo "### pageable ring workers=$w"
  for rep in 1 2; do timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 2000 --workers $w --no-pin 2>&1 | grep "^done\|error\|^worker time" | cut -c1-130; done
done
echo "### pinned ring workers=1"; timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 3000 --workers 1 2>&1 | grep "^done" | cut -c1-70
