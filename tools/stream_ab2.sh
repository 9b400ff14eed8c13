BIN=opencv-opencl_amd/lib/nv12_stream
for w in 1 2 4 8; do
  echo "### workers=$w"
  for rep in 1 2 3 4; do timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 3000 --workers $w 2>&1 | grep "^done\|error" | cut -c1-60; done
done
