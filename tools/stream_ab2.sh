BIN=opencv-opencl_amd/lib/nv12_stream
for d in 4 8 12 16; do
  echo "### workers=1 depth=$d"
  for rep in 1 2; do timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 3000 --workers 1 --depth $d 2>&1 | grep "^done\|error" | cut -c1-70; done
done
for d in 4 8; do
  echo "### workers=2 depth=$d"
  for rep in 1 2; do timeout -k 10 60 $BIN --width 3840 --height 2160 --frames 3000 --workers 2 --depth $d 2>&1 | grep "^done\|error" | cut -c1-70; done
done
for w in 1 2; do
  echo "### 1080p workers=$w"
  for rep in 1 2; do timeout -k 10 60 $BIN --width 1920 --height 1080 --frames 8000 --workers $w 2>&1 | grep "^done\|error\|^worker time" | cut -c1-130; done
done
