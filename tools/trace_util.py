"""Copy-engine and kernel occupancy from a rocprofv3 trace directory (--memory-copy-trace --kernel-trace --hip-trace, csv):
per direction: busy time / wall time of the steady state, mean copy duration, time per frame; HIP API time by function.
    python tools/trace_util.py gpurun_out/trace_dir"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
mc = [r for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
kt = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
api = [r for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
for direction in ("MEMORY_COPY_HOST_TO_DEVICE", "MEMORY_COPY_DEVICE_TO_HOST"):
    cp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in mc if r["Direction"] == direction)
    big = [(s, e) for s, e in cp if e - s > 50_000]               # whole-plane copies only
    if len(big) < 20:
        print(direction, "too few large copies", len(big)); continue
    big = big[len(big) // 5:]                                       # steady state
    wall = big[-1][1] - big[0][0]
    busy = 0; cur_s, cur_e = big[0]
    for s, e in big[1:]:                                            # union of intervals
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    durs = [e - s for s, e in big]
    print(f"{direction}: {len(big)} plane copies, wall {wall/1e3:.0f} us, engine busy {100*busy/wall:.1f} %, mean copy {sum(durs)/len(durs)/1e3:.1f} us, "
          f"per frame {wall/len(big)/1e3:.1f} us")
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kt]
if ks:
    print("kernels:", len(ks), "mean %.1f us" % (sum(e - s for s, e in ks) / len(ks) / 1e3))
tot = defaultdict(lambda: [0, 0])
threads = set()
for r in api:
    t = tot[r["Function"]]; t[0] += 1; t[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); threads.add(r["Thread_Id"])
print("threads calling HIP:", len(threads))
for f, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {f:28s} {n:7d} calls, {ns/n/1e3:8.1f} us each, {ns/1e6:8.1f} ms total")
