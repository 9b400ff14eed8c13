BIN=opencv-opencl_amd/lib/nv12_stream
for w in 3 4 8; do
  echo "### uncapped workers=$w"
  for rep in 1 2 3; do timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 3000 --workers $w --max-workers-per-gpu 0 2>&1 | grep "^done\|error" | cut -c1-70; done
done
echo "### workers=2"; timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 3000 --workers 2 2>&1 | grep "^done" | cut -c1-70
