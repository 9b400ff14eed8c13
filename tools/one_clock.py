"""One clock for the roofline: for the SAME launches of one bench.py run under rocprofv3, the library's HIP-event
durations (hipExtLaunchKernelGGL start / stop events, what roofline.frac is computed from) beside rocprofv3's kernel-trace durations;
and the HIP-event figure of a plain, unprofiled run on the same box.   python3 tools/one_clock.py <dir> <op> <out.json>
<dir> holds bench_under_rocprof.json, bench_plain.json and trace/ (rocprofv3 --kernel-trace --stats --output-format csv)."""
import csv, glob, json, sys
d, op, out = sys.argv[1], sys.argv[2], sys.argv[3]
def line(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"no JSON line in {path}")
under, plain = line(f"{d}/bench_under_rocprof.json"), line(f"{d}/bench_plain.json")
steps = under["steps"]
trace = glob.glob(f"{d}/trace/**/*_kernel_trace.csv", recursive=True)[0]
per = {}
for r in csv.DictReader(open(trace)):
    name = r["Kernel_Name"]
    if "mi::" not in name:
        continue
    short = name.split("mi::")[1].split("(")[0].split("<")[0]
    per.setdefault(short, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rec = {"what": __doc__.split("\n")[0], "op": op, "steps": steps, "warmup": under["warmup"], "workload": under["config"]["workload"], "kernels": {}}
print(f"{'kernel':28s} {'launches':>8s} {'HIP events, plain run':>22s} {'HIP events, under rocprofv3':>28s} {'rocprofv3, same launches':>25s} {'rocprofv3, all launches':>24s}")
for k, e in under["kernels_rank0"].items():
    if k not in per:
        continue
    lst = sorted(per[k])
    per_step = len(lst) // (steps + under["warmup"]) if (steps + under["warmup"]) else 1
    same = lst[-e["launches"]:]                                      # the timed region's launches: the last `launches` of this kernel
    same_ms = sum(b - a for a, b in same) / len(same) / 1e6
    all_ms = sum(b - a for a, b in lst) / len(lst) / 1e6
    pl = plain["kernels_rank0"].get(k, {}).get("avg_ms")
    rec["kernels"][k] = {"launches": e["launches"], "hip_events_plain_run_ms": pl, "hip_events_under_rocprofv3_ms": e["avg_ms"],
                         "rocprofv3_same_launches_ms": round(same_ms, 5), "rocprofv3_all_launches_ms": round(all_ms, 5), "launches_per_step": per_step}
    print(f"{k:28s} {e['launches']:8d} {pl if pl is not None else float('nan'):22.5f} {e['avg_ms']:28.5f} {same_ms:25.5f} {all_ms:24.5f}")
rec["value_plain_run"], rec["value_under_rocprofv3"] = plain["value"], under["value"]
json.dump(rec, open(out, "w"), indent=1)
print(f"frames/s: plain {plain['value']}, under rocprofv3 {under['value']}  ->  {out}")
