#!/usr/bin/env bash
# Re-creates the files under profiles/ on a GPU box (run from the repo root through gpurun).
#   tools/collect_profiles.sh <tag> [equalize|clahe|clahe16]        e.g.  tools/collect_profiles.sh r02_a clahe
# Counter passes are separate runs with --kernel-trace only (never combined with trace domains), FETCH_SIZE and
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; the program after `--` is python3 itself.
# The profiled runs pass --no-second-resolution: only the 64 x 4K launches of the headline workload are in the statistics
# (round 2's r02_q files also averaged the 256 x 1080p launches of the second-resolution leg in).
set -euo pipefail
TAG=${1:-rXX}
OP=${2:-equalize}
OUT=gpurun_out/${TAG}_${OP}
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$OP" = clahe16 ]; then
  PROG=(python3 tools/prof16.py 16 12bit)
  "${PROG[@]}" > "$OUT/prof16.txt" 2> "$OUT/prof16.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "${PROG[@]}" > /dev/null 2> "$OUT/stats.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- "${PROG[@]}" > /dev/null 2> "$OUT/pmc_$c.err"
  done
else
  python3 bench.py --op "$OP" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --op "$OP" --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-second-resolution > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- python3 bench.py --op "$OP" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-second-resolution > /dev/null 2> "$OUT/pmc_$c.err"
  done
fi
python3 tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_summary.json" > /dev/null
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/stats/**/*_kernel_stats.csv", recursive=True)[0])))
with open(out + "/kernel_stats_mi.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        if "mi::" in r["Name"]:
            w.writerow(r)
print(open(out + "/kernel_stats_mi.csv").read())
PY
[ -f "$OUT/bench_n1.json" ] && cat "$OUT/bench_n1.json"
rm -rf "$OUT/stats" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"      # raw traces are large; the summaries above are what gets committed
echo "copy $OUT/{bench_n1.json,kernel_stats_mi.csv,pmc_summary.json} into profiles/ (named per round) and update profiles/traffic.json"
