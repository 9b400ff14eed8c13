#!/usr/bin/env bash
# SQ counters per launch of our kernels (separate --pmc pass, --kernel-trace only): LDS bank conflicts and wait share.
#   tools/sq_counters.sh <out.json> -- <program ...>      e.g.  tools/sq_counters.sh gpurun_out/sq_clahe.json -- python3 bench.py --op clahe --steps 5 --warmup 2 --no-cpu-baseline --no-extras
set -euo pipefail
OUT=$1; shift; shift
export TMPDIR=/tmp
D=$(mktemp -d gpurun_out/sq_XXXX)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$D" -- "$@" > /dev/null 2> "$D.err"
python3 - "$D" "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if n.startswith("void "): n = n[5:]
        if not n.startswith("mi::"): continue
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"_method": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace (own pass); averages per launch"}
for k, cs in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches"] = len(next(iter(cs.values())))
    if e.get("SQ_LDS_IDX_ACTIVE"): e["lds_conflict_share"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 4)
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_share"] = round(e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"], 4)
        e["issue_stall_share"] = round(e.get("SQ_WAIT_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 4)
        e["active_share"] = round(e.get("SQ_ACTIVE_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 4)
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "share" in kk} for k, v in res.items() if k != "_method"}, indent=1))
PY
rm -rf "$D" "$D.err"
