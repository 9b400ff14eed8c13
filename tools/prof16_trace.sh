#!/usr/bin/env bash
# per-kernel durations of the 16-bit CLAHE by content kind (rocprofv3 --kernel-trace --stats): tools/prof16_trace.sh <tag> [wide option]
set -uo pipefail
TAG=${1:-r06}; WIDE=${2:-2}
export TMPDIR=/tmp
for kind in 12bit 14bit full ramp hot; do
  out=gpurun_out/${TAG}_trace_${kind}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/prof16_trace.py $kind 16 $WIDE > /dev/null 2> "$out.err"
  python3 - "$out" "$kind" <<'PY'
import csv, glob, sys
out, kind = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*_kernel_stats.csv", recursive=True)
if not f: print(kind, "no stats"); sys.exit(0)
print(f"== {kind}")
for r in csv.DictReader(open(f[0])):
    if "mi::" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e3:10.1f} us')
PY
  rm -rf "$out"
done
