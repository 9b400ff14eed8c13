#!/bin/bash
# tools/gpu_suite.sh <label> [pytest args...] -- the GPU test suite, ONE pytest process, with whatever a GPU fault leaves behind
# brought home.  The HSA runtime writes `gpucore.<pid>` into the working directory when a GPU memory fault ends the process; gpurun
# only carries gpurun_out/ back, which is how round 3's gpucore.790 was lost.  After the run every core found is examined ON THE
# BOX with rocgdb (agents, queues, dispatches, waves with their PCs: enough to tell a wave of one of this library's kernels from a
# copy-engine access) and copied into gpurun_out/ when it fits gpurun's 64 MiB limit.  Never loops, never retries.
label=${1:?usage: tools/gpu_suite.sh <label> [pytest args...]}
shift
cd "$(dirname "$0")/.." || exit 2
mkdir -p gpurun_out
rm -f gpucore.*
python -m pytest tests -m gpu -x -q "$@" > "gpurun_out/${label}_gpu_tests.txt" 2>&1
rc=$?
tail -n 6 "gpurun_out/${label}_gpu_tests.txt"
for core in gpucore.*; do
    [ -e "$core" ] || continue
    out="gpurun_out/${label}_${core}.rocgdb.txt"
    ls -l "$core" > "$out"
    timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex "info agents" -ex "info queues" -ex "info dispatches" -ex "info threads" \
        -ex "thread apply all bt 8" "$(command -v python3)" "$core" >> "$out" 2>&1
    gzip -1 -c "$core" > "/tmp/${core}.gz" && [ "$(stat -c %s "/tmp/${core}.gz")" -le $((40 << 20)) ] && cp "/tmp/${core}.gz" "gpurun_out/${label}_${core}.gz"
    echo "GPU core $core examined: $out"
done
exit $rc
