"""GPU tests of the library's behaviour under failure and contention (run with -m gpu): error exits that drain their streams,
pipes that stay in step with their callers, MI_ERR_BUSY rules, demotion of the fused path after repeated repairs, lifetime of the
fused kernel's ticket stamps.  The injected failures come from libmi_lumaeq_test.so (the product sources + -DMI_TEST_HOOKS);
every byte is still compared with the CPU oracle.  Reference for what "error handling" replaces: the accelerator worker that
catches a type nobody throws and pushes the frame on regardless, /root/reference OpenCLequalHist.cpp:346-367."""
import time

import numpy as np
import pytest

import mi_lumaeq
import oracle
from mi_lumaeq import synth, xfer

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    """host array -> device tensor WITHOUT handing pageable memory to the runtime (mi_lumaeq.xfer says why)"""
    return xfer.to_device(a)


host = xfer.to_host                                     # device tensor -> numpy, likewise through pinned staging


def hooks_ctx():
    c = mi_lumaeq.Context(0, lib=mi_lumaeq.test_lib())
    c.set_option("two_kernel_max_frames", 0)      # these tests are about the fused kernel's hand-off: one- and two-frame calls must take it too
    return c


def test_product_library_knows_no_test_hook():
    assert "+test-hooks" in mi_lumaeq.test_lib().mi_version().decode() and "+test-hooks" not in mi_lumaeq.version()
    with mi_lumaeq.Context(0) as c:
        for name in ("fused_fault_inject", "fused_timeout_us", "hip_fail_after", "host_direct"):
            with pytest.raises(mi_lumaeq.MiError) as e:
                c.set_option(name, 1)
            assert e.value.status == 1


def test_fused_path_is_demoted_after_repeated_repairs_and_probed_again():
    """A context whose fused launches keep losing their hand-off (here: an injected lost producer, 300 us bound) must stop
    paying the stall: three repaired launches, then the three-kernel path -- the oracle's bytes on every call -- then ONE probe after
    the demotion period, a second demotion for twice as long when the probe is repaired too, and the fused path back for good once
    a probe window stays clean."""
    w, h = 1920, 1080
    ys = [synth.y_plane(w, h, synth.DISTS[k % 5], 900 + k) for k in range(4)]
    want = [oracle.equalize_hist(y) for y in ys]
    c = hooks_ctx()
    try:
        c.set_option("fused_reprobe_ms", 400)
        c.set_option("fused_timeout_us", 300)
        c.set_option("fused_fault_inject", 1)
        stall, quick = [], []
        for k in range(12):
            t0 = time.perf_counter()
            got = c.equalize_hist(ys[k % 4])
            (stall if k < 3 else quick).append(time.perf_counter() - t0)
            assert np.array_equal(got, want[k % 4]), k
        assert c.get_stat("fused_fallbacks") == 3, "launches 4..12 must not have been fused"
        assert c.get_stat("fused_demotions") == 1 and c.get_stat("fused_demoted") == 1
        assert c.get_stat("fused_hard_errors") == 0
        # the device forms obey the same verdict without any synchronisation of their own
        d_in = dev(np.stack([synth.nv12_frame(w, h, "D2", 5 + k) for k in range(3)]))
        d_out = torch.zeros_like(d_in)
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, 3, 0)
        c.synchronize()
        assert c.get_stat("fused_fallbacks") == 3
        assert np.array_equal(host(d_out[2]), oracle.nv12_frame(host(d_in[2]), w, h, uv_mode=0, op=0))
        # after the period: one probe (repaired: the fault is still injected) -> demoted again, for twice as long
        time.sleep(0.45)
        assert np.array_equal(c.equalize_hist(ys[0]), want[0])
        assert c.get_stat("fused_fallbacks") == 4
        assert np.array_equal(c.equalize_hist(ys[1]), want[1])
        assert c.get_stat("fused_demotions") == 2 and c.get_stat("fused_demoted") == 1 and c.get_stat("fused_fallbacks") == 4
        time.sleep(0.45)                                            # 400 ms have passed, 800 have not
        assert np.array_equal(c.equalize_hist(ys[2]), want[2])
        assert c.get_stat("fused_demoted") == 1 and c.get_stat("fused_fallbacks") == 4
        # the GPU is free again: the next probe stays clean and the fused path is back
        c.set_option("fused_fault_inject", 0)
        c.set_option("fused_timeout_ms", 50)
        time.sleep(0.45)
        for k in range(40):
            assert np.array_equal(c.equalize_hist(ys[k % 4]), want[k % 4]), k
        assert c.get_stat("fused_demoted") == 0 and c.get_stat("fused_demotions") == 2 and c.get_stat("fused_fallbacks") == 4
        prof_before = c.profile_read(reset=True)
        c.set_profiling(1)
        c.equalize_hist(ys[0])
        c.set_profiling(0)
        assert c.profile_read()["equalize_fused_kernel"]["launches"] == 1      # really the fused kernel again
        # "fused_demote_after" = 0 switches the mechanism off
        c.set_option("fused_demote_after", 0)
        c.set_option("fused_timeout_us", 300)
        c.set_option("fused_fault_inject", 1)
        for k in range(5):
            assert np.array_equal(c.equalize_hist(ys[k % 4]), want[k % 4])
        assert c.get_stat("fused_fallbacks") == 9 and c.get_stat("fused_demotions") == 2
    finally:
        c.close()


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
def test_host_form_error_exits_drain_the_stream_and_leave_the_context_usable(pinned):
    """Every checked HIP call of a host-pointer form is made to REPORT a failure once, after it was issued (option "hip_fail_after"
    of the test library): the call must return MI_ERR_HIP, must not return before the copies it queued on the caller's planes have
    finished (statistic "error_drains"), and the next call on the same context must produce the oracle's bytes."""
    w, h = 1280, 720
    y = synth.y_plane(w, h, "D2", 31)
    want_eq, want_cl = oracle.equalize_hist(y), oracle.clahe(y, 2.0, 8, 8)
    c = hooks_ctx()
    try:
        if pinned:
            src_t, dst_t = torch.from_numpy(y.copy()).pin_memory(), torch.empty((h, w), dtype=torch.uint8).pin_memory()
            src, dst = src_t.numpy(), dst_t.numpy()
        else:
            src, dst = y.copy(), np.empty_like(y)
        assert np.array_equal(c.equalize_hist(src, dst), want_eq)   # sizes every buffer: the sweep below meets steady-state calls
        failures = 0
        for op in ("equalize", "clahe"):
            for n in range(1, 60):
                c.set_option("hip_fail_after", n)
                drains0 = c.get_stat("error_drains")
                try:
                    got = c.equalize_hist(src, dst) if op == "equalize" else c.clahe(src, 2.0, 8, 8, dst)
                except mi_lumaeq.MiError as e:
                    assert e.status == 3, e
                    failures += 1
                    assert c.get_stat("error_drains") >= drains0   # (>=: a failure before anything was queued has nothing to drain)
                    c.set_option("hip_fail_after", 0)
                    dst[:] = 0
                    got = c.equalize_hist(src, dst) if op == "equalize" else c.clahe(src, 2.0, 8, 8, dst)
                    assert np.array_equal(got, want_eq if op == "equalize" else want_cl), (op, n)
                    continue
                c.set_option("hip_fail_after", 0)
                assert np.array_equal(got, want_eq if op == "equalize" else want_cl), (op, n)
                break                                               # n exceeds the number of checked calls: the sweep is complete
            else:
                pytest.fail("a host form makes more than 60 checked HIP calls?")
        assert failures >= 8                                        # copies, launches, events, synchronisations were all hit
        assert c.get_stat("error_drains") >= failures // 2
        # the less travelled host forms share stage_in / stage_out
        nv = synth.nv12_frame(w, h, "D1", 8)
        want_nv = oracle.nv12_bgr_equalize(nv, w, h)
        for n in range(1, 12):
            c.set_option("hip_fail_after", n)
            try:
                c.nv12_bgr_equalize(nv, w, h)
            except mi_lumaeq.MiError as e:
                assert e.status == 3
            c.set_option("hip_fail_after", 0)
            assert np.array_equal(c.nv12_bgr_equalize(nv, w, h), want_nv), n
        # 16-bit CLAHE: a failure reported between its launches (the tile histograms ran, the LUT kernel or the interpolation did
        # not) must leave the per-frame arrival words of the histogram kernel at zero for the next call
        y16 = np.random.default_rng(9).integers(0, 4096, (h, w), dtype=np.uint16)
        want16 = oracle.clahe16(y16, 2.0, 8, 8)
        hit = 0
        for n in range(1, 16):
            c.set_option("hip_fail_after", n)
            try:
                c.clahe16(y16, 2.0, 8, 8)
            except mi_lumaeq.MiError as e:
                assert e.status == 3
                hit += 1
            c.set_option("hip_fail_after", 0)
            assert np.array_equal(c.clahe16(y16, 2.0, 8, 8), want16), n
        assert hit >= 4
    finally:
        c.close()


def test_pipe_errors_keep_the_caller_and_the_pipe_in_step():
    """mi_pipe_submit that fails occupies no slot and leaves nothing in flight; mi_pipe_wait that fails has still retired the oldest
    frame.  Either way the frames around the failed one come back under their own tags with the oracle's bytes."""
    w, h = 1280, 720
    frames = [synth.nv12_frame(w, h, synth.DISTS[k % 5], 60 + k) for k in range(6)]
    want = [oracle.nv12_frame(f, w, h, uv_mode=0, op=0) for f in frames]
    c = hooks_ctx()
    try:
        with mi_lumaeq.Pipe(c, w, h, depth=3) as pipe:
            outs = [np.zeros_like(f) for f in frames]
            assert pipe.submit(frames[0], outs[0], 100)
            for n in range(1, 30):                                  # every checked call of a submit fails once
                c.set_option("hip_fail_after", n)
                try:
                    ok = pipe.submit(frames[1], outs[1], 101)
                except mi_lumaeq.MiError as e:
                    assert e.status == 3
                    assert pipe.pending == 1                        # the failed frame took no slot
                    continue
                finally:
                    c.set_option("hip_fail_after", 0)
                assert ok and pipe.pending == 2
                break
            else:
                pytest.fail("mi_pipe_submit makes more than 29 checked HIP calls?")
            assert c.get_stat("error_drains") >= 1
            assert pipe.submit(frames[2], outs[2], 102)
            tag, out = pipe.wait()
            assert tag == 100 and np.array_equal(out, want[0])
            # a failing wait: its frame is reported lost, the slot is retired, the NEXT wait returns the next tag
            c.set_option("hip_fail_after", 1)
            with pytest.raises(mi_lumaeq.MiError) as e:
                pipe.wait()
            assert e.value.status == 3 and pipe.pending == 1
            c.set_option("hip_fail_after", 0)
            tag, out = pipe.wait()
            assert tag == 102 and np.array_equal(out, want[2])
            assert pipe.pending == 0
            # the pipe works on as if nothing had happened
            for k in range(3, 6):
                assert pipe.submit(frames[k], outs[k], 200 + k)
            for k in range(3, 6):
                tag, out = pipe.wait()
                assert tag == 200 + k and np.array_equal(out, want[k]), k
    finally:
        c.close()


def test_pipe_and_context_busy_rules():
    """One pipe per context; while frames are pending the context's other compute entry points answer MI_ERR_BUSY (they would race the
    pipe's streams for the context's scratch); with nothing pending they work, and a device-form call on a caller's stream is
    ordered in front of the next frame's kernels."""
    w, h = 1920, 1080
    f = synth.nv12_frame(w, h, "D2", 3)
    y = f[: w * h].reshape(h, w)
    want = oracle.nv12_frame(f, w, h, uv_mode=0, op=0)
    with mi_lumaeq.Context(0) as c:
        with mi_lumaeq.Pipe(c, w, h, depth=2) as pipe:
            with pytest.raises(mi_lumaeq.MiError) as e:
                mi_lumaeq.Pipe(c, w, h, depth=2)
            assert e.value.status == mi_lumaeq.ERR_BUSY
            o = np.zeros_like(f)
            assert pipe.submit(f, o, 1)
            d = dev(f[None])
            for call in (lambda: c.equalize_hist(y), lambda: c.clahe(y, 2.0, 8, 8), lambda: c.equalize_hist_nv12_batch_dev(d, d, w, h, 1, 0),
                         lambda: c.analyze_diff(y, y), lambda: c.nv12_bgr_equalize(f, w, h)):
                with pytest.raises(mi_lumaeq.MiError) as e:
                    call()
                assert e.value.status == mi_lumaeq.ERR_BUSY
            assert c.get_stat("fused_fallbacks") == 0               # statistics and options stay available
            assert pipe.wait()[0] == 1 and np.array_equal(o, want)
            # nothing pending: a long device-form batch on torch's stream, then a frame through the pipe straight away.  Both use the
            # context's hand-off block; the pipe's compute stream must wait for the batch (event recorded when the call returned).
            n = 48
            batch = np.stack([synth.nv12_frame(w, h, synth.DISTS[k % 5], 400 + k) for k in range(4)])
            d_in = dev(batch).repeat(n // 4, 1)
            d_out = torch.zeros_like(d_in)
            for rep in range(3):
                c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 1, stream=torch.cuda.current_stream().cuda_stream)
                o2 = np.zeros_like(f)
                assert pipe.submit(f, o2, 7 + rep)
                assert pipe.wait()[0] == 7 + rep and np.array_equal(o2, want)
                torch.cuda.synchronize()
                got = host(d_out)
                for k in (0, 1, 2, 3, n - 1):
                    assert np.array_equal(got[k], oracle.nv12_frame(batch[k % 4], w, h, uv_mode=1, op=0)), (rep, k)
            assert c.get_stat("fused_fallbacks") == 0
        # the pipe is gone: a new one may be created
        with mi_lumaeq.Pipe(c, w, h, depth=2) as pipe:
            o = np.zeros_like(f)
            assert pipe.submit(f, o, 9) and pipe.wait()[0] == 9 and np.array_equal(o, want)


def test_ticket_stamps_share_the_lifetime_of_their_hand_off_block():
    """A captured graph keeps replaying into the hand-off block it was recorded with -- its own sequence numbers, its own ticket stamps
    -- after a larger batch made the context allocate a new block whose epochs start over.  Replays and eager launches are
    interleaved, both with injected hand-off failures, so both blocks run their stamp-driven repair again and again: with shared
    stamps an eager launch's epoch could meet a replay's stamps (or the reverse) and a ticket that was never written would count
    as done.  Every output byte is compared."""
    w, h, n = 1280, 720, 3
    c = hooks_ctx()
    try:
        c.set_option("fused_demote_after", 0)
        c.set_option("fused_timeout_us", 400)
        d_in = synth.nv12_batch_torch(w, h, n, "D2", "cuda:0", seed=5)
        d_out = torch.zeros_like(d_in)
        c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0)
        c.synchronize()
        c.set_option("fused_fault_inject", 2)                       # frames left partly written: the repair goes by the stamps
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            c.equalize_hist_nv12_batch_dev(d_in, d_out, w, h, n, 0, stream=torch.cuda.current_stream().cuda_stream)
        # regrow: 80 frames > the first block's 64-frame layout -> a new block (the old one is kept for the graph)
        n2 = 80
        e_src = [synth.nv12_frame(w, h, synth.DISTS[k % 5], 40 + k) for k in range(4)]
        e_in = dev(np.stack(e_src)).repeat(n2 // 4, 1)
        e_out = torch.zeros_like(e_in)
        e_want = [oracle.nv12_frame(f, w, h, uv_mode=1, op=0) for f in e_src]
        for rep in range(6):
            # eager launch into the NEW block (its epochs restarted at 1), alternately with 3 frames (same ticket range as the graph) and 80
            m = n if rep % 2 == 0 else n2
            e_out.zero_()
            c.equalize_hist_nv12_batch_dev(e_in, e_out, w, h, m, 1)
            # replay into the OLD block
            d_in.copy_(synth.nv12_batch_torch(w, h, n, synth.DISTS[rep % 5], "cuda:0", seed=300 + rep))
            d_out.zero_()
            g.replay()
            torch.cuda.synchronize()
            c.synchronize()
            got = host(e_out)
            for k in range(m if m == n else 8):
                assert np.array_equal(got[k], e_want[k % 4]), ("eager", rep, k)
            assert np.array_equal(got[m - 1], e_want[(m - 1) % 4]), ("eager", rep, m - 1)
            src, out = host(d_in), host(d_out)
            for k in range(n):
                assert np.array_equal(out[k], oracle.nv12_frame(src[k], w, h, uv_mode=0, op=0)), ("replay", rep, k)
        assert c.get_stat("fused_fallbacks") >= 6 and c.get_stat("fused_hard_errors") == 0
        del g
        torch.cuda.synchronize()
    finally:
        c.close()


def test_numa_placement_entry_points():
    """mi_device_pci_bus_id / mi_thread_bind_near_device (what every per-GPU worker calls before it creates its context): the PCI address
    has sysfs shape, binding reports the node and never fails hard, the thread ends up on a subset of the CPUs it had, and
    MI_LUMAEQ_NUMA_BIND=0 turns it into a report."""
    import os
    import re
    import subprocess
    import sys
    bdf = mi_lumaeq.device_pci_bus_id(0)
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", bdf), bdf
    with pytest.raises(mi_lumaeq.MiError):
        mi_lumaeq.device_pci_bus_id(99)
    # in a child process: the binding is per thread and would otherwise stick to the test session's main thread
    code = ("import os, sys, json\n"
            "sys.path.insert(0, 'opencv-opencl_amd/python')\n"
            "import mi_lumaeq\n"
            "before = sorted(os.sched_getaffinity(0))\n"
            "b = mi_lumaeq.bind_thread_near_device(0)\n"
            "after = sorted(os.sched_getaffinity(0))\n"
            "print(json.dumps({'b': b, 'before': len(before), 'after': len(after), 'subset': set(after) <= set(before)}))\n")
    import json
    root = str(__import__("pathlib").Path(__file__).resolve().parents[1])
    for env_off in (False, True):
        env = dict(os.environ)
        if env_off:
            env["MI_LUMAEQ_NUMA_BIND"] = "0"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["subset"] and d["after"] >= 1
        if env_off:
            assert d["b"]["cpus"] == 0 and d["after"] == d["before"] and "not applied" in d["b"]["why"]
        elif d["b"]["node"] >= 0 and d["b"]["cpus"] > 0:
            assert d["after"] == d["b"]["cpus"] <= d["before"] and "NUMA node" in d["b"]["why"]
        else:
            assert d["after"] == d["before"] and "not bound" in d["b"]["why"]


def test_pool_starts_at_most_two_workers_per_gpu_and_prints_its_placement():
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parents[1] / "opencv-opencl_amd" / "lib" / "nv12_stream"
    r = subprocess.run([str(exe), "--width", "1280", "--height", "720", "--frames", "200", "--workers", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    n_gpu = mi_lumaeq.device_count()
    started = min(5, 2 * n_gpu)
    if started < 5:
        assert f"workers: 5 requested, {started} started" in r.stdout
    assert r.stdout.count("placement: worker ") == started and "placement: submitting thread" in r.stdout
    assert "done: 200 frames" in r.stdout and "errors=0" in r.stdout
    r = subprocess.run([str(exe), "--width", "1280", "--height", "720", "--frames", "100", "--workers", "3", "--max-workers-per-gpu", "0", "--no-numa-bind"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("placement: worker ") == 3 and "NUMA binding off" in r.stdout and "errors=0" in r.stdout


def test_pinned_staged_transfer_helpers_round_trip():
    """mi_lumaeq.xfer: what tests, bench.py and smoke() move data with instead of torch's pageable .to() / .cpu()."""
    rng = np.random.default_rng(3)
    for shape, dt in (((3, 1080 * 1920 * 3 // 2), np.uint8), ((2, 64, 48), np.int16), ((7,), np.int32), ((0,), np.uint8)):
        a = rng.integers(0, 200, shape).astype(dt)
        d = xfer.to_device(a)
        assert d.is_cuda and tuple(d.shape) == a.shape
        back = xfer.to_host(d)
        assert back.dtype == a.dtype and np.array_equal(back, a) and back.flags.writeable
    v = xfer.to_device(np.arange(100, dtype=np.uint8).reshape(10, 10))[::2, 1:5]      # a non-contiguous device view
    assert np.array_equal(xfer.to_host(v), np.arange(100, dtype=np.uint8).reshape(10, 10)[::2, 1:5])


def test_host_forms_randomized_memory_kinds_pitches_and_options():
    """Differential run over the host-pointer forms' staging machinery: random sizes (below and above the helper thread's and the
    chunking's thresholds), row pitches, pinned / unpinned / registered planes in every combination, in place, one or two copy streams,
    one or two copying threads -- every result against the oracle."""
    rng = np.random.default_rng(2026)
    sizes = [(1, 1), (17, 3), (640, 360), (1279, 719), (1920, 1080), (2048, 1024), (4096, 8), (8, 4096), (3840, 2160)]
    with mi_lumaeq.Context(0) as c:
        for case in range(48):
            w, h = sizes[case % len(sizes)] if case < 2 * len(sizes) else (int(rng.integers(1, 2500)), int(rng.integers(1, 1500)))
            pad_s, pad_d = int(rng.integers(0, 3)) * 16 + int(rng.integers(0, 2)) * 5, int(rng.integers(0, 3)) * 32
            kind_s, kind_d = rng.integers(0, 3, 2)                # 0 ordinary numpy, 1 pinned torch memory, 2 registered numpy
            if (w + max(pad_s, pad_d)) * h > 12_000_000:
                pad_s = pad_d = 0

            def plane(kind, pad, fill):
                shape = (h, w + pad)
                if kind == 1:
                    t = torch.empty(shape, dtype=torch.uint8).pin_memory()
                    a = t.numpy()
                    keep = t
                else:
                    a = np.empty(shape, np.uint8)
                    keep = a
                    if kind == 2 and a.nbytes >= (1 << 20):        # (small arrays share heap pages with their neighbours: leave them alone)
                        mi_lumaeq.host_register(a)
                        registered.append(a)
                a[:] = fill
                return a, keep
            registered = []
            src_full, keep_s = plane(kind_s, pad_s, 0)
            dst_full, keep_d = plane(kind_d, pad_d, 0xEE)
            try:
                y = rng.integers(0, 256, (h, w), dtype=np.uint8) if case % 3 else synth.y_plane(w, h, synth.DISTS[case % 5], case)
                src_full[:, :w] = y
                src, dst = src_full[:, :w], dst_full[:, :w]
                c.set_option("host_copy_streams", int(rng.integers(1, 3)))
                c.set_option("host_copy_threads", int(rng.integers(1, 3)))
                op = case % 4
                if op == 0:
                    got, want = c.equalize_hist(src, dst), oracle.equalize_hist(y)
                elif op == 1:
                    cfg = (float(rng.choice([0.0, 2.0, 40.0])), int(rng.integers(1, 9)), int(rng.integers(1, 9)))
                    got, want = c.clahe(src, *cfg, dst=dst), oracle.clahe(y, *cfg)
                elif op == 2:                                      # in place on the source view
                    want = oracle.equalize_hist(y)
                    got = c.equalize_hist(src, src)
                else:                                              # whole NV12 frame (tight), UV copy or fill
                    w2, h2 = max(2, w & ~1), max(2, h & ~1)
                    f = synth.nv12_frame(w2, h2, synth.DISTS[case % 5], case)
                    uv = int(rng.integers(0, 2))
                    got, want = c.equalize_hist_nv12(f, w2, h2, uv), oracle.nv12_frame(f, w2, h2, uv_mode=uv, op=0)
                assert np.array_equal(got, want), (case, w, h, op, int(kind_s), int(kind_d), pad_s, pad_d)
                if op in (0, 1) and pad_d:
                    assert (dst_full[:, w:] == 0xEE).all(), (case, "bytes beyond the view were written")
            finally:
                for a in registered:
                    mi_lumaeq.host_unregister(a)
        assert c.get_stat("error_drains") == 0 and c.get_stat("fused_hard_errors") == 0


def test_pipe_randomized_sessions():
    """Random pipe sessions: op, UV mode and policy, depth, pinned / unpinned / mixed frame memory, random interleaving of submit and
    wait (including BUSY answers when the pipe is full) -- every frame comes back under its own tag, in order, with the oracle's bytes."""
    rng = np.random.default_rng(77)
    for session in range(10):
        w, h = [(640, 360), (1280, 720), (322, 182), (1920, 1080)][session % 4]
        op = [mi_lumaeq.OP_EQUALIZE, mi_lumaeq.OP_CLAHE, mi_lumaeq.OP_EQUALIZE, mi_lumaeq.OP_CHANNELS][session % 4]
        uv = int(rng.integers(0, 2))
        policy = [mi_lumaeq.PIPE_UV_AUTO, mi_lumaeq.PIPE_UV_HOST, mi_lumaeq.PIPE_UV_DEVICE][session % 3]
        depth = int(rng.integers(2, 7))
        n = 14
        frames = [synth.nv12_frame(w, h, synth.DISTS[k % 5], 1000 * session + k) for k in range(n)]
        if op == mi_lumaeq.OP_CHANNELS:
            want = [oracle.nv12_bgr_equalize(f, w, h) for f in frames]
        else:
            want = [oracle.nv12_frame(f, w, h, uv_mode=uv, op=1 if op == mi_lumaeq.OP_CLAHE else 0, clip_limit=2.0, tiles_x=4, tiles_y=2) for f in frames]
        outs = [np.zeros_like(f) for f in frames]
        registered = []
        for k in range(n):                                         # a random subset of the buffers is registered (pinned)
            for a in (frames[k], outs[k]):
                if a.nbytes >= (1 << 19) and rng.integers(0, 2):
                    mi_lumaeq.host_register(a)
                    registered.append(a)
        c = mi_lumaeq.Context(0)
        try:
            with mi_lumaeq.Pipe(c, w, h, op=op, uv_mode=uv, clip_limit=2.0, tiles_x=4, tiles_y=2, depth=depth, uv_policy=policy) as pipe:
                nxt, done = 0, 0
                while done < n:
                    if nxt < n and (pipe.pending == 0 or rng.integers(0, 3)):
                        if pipe.submit(frames[nxt], outs[nxt], 5000 + nxt):
                            nxt += 1
                            continue
                        assert pipe.pending == pipe.depth          # BUSY only when full
                    tag, out = pipe.wait()
                    assert tag == 5000 + done and out is outs[done]
                    assert np.array_equal(out, want[done]), (session, done, op, uv, policy, depth)
                    done += 1
                assert pipe.pending == 0
        finally:
            c.close()
            for a in registered:
                mi_lumaeq.host_unregister(a)


def test_registering_and_unregistering_while_two_pipes_stream():
    """The pin registry under real traffic (host/pin_registry.hpp, round 5): two threads stream frames through their own pipes on
    registered rings while a third keeps registering and unregistering OTHER buffers (each unregister may wait for the device, and
    since round 5 does so without the registry's lock) and keeps asking to unregister the ring buffers that are in flight.  Every such
    request must be answered BUSY or, between two frames, succeed and be re-registered -- never crash, never corrupt a frame: every
    delivered frame is compared with the oracle.  (The reference's accelerator worker maps, enqueues on and unmaps caller buffers with
    no guard at all: OpenCLequalHist.cpp:307-367.)"""
    import threading
    import time
    w, h = 1280, 720
    stream_s, min_frames = 3.0, 600                                # a fixed wall time of streaming, not a frame count
    fb = w * h * 3 // 2
    base = [synth.nv12_frame(w, h, synth.DISTS[k % 5], 8800 + k) for k in range(4)]
    want = [oracle.nv12_frame(f, w, h, uv_mode=0, op=0) for f in base]
    errors, busy_answers, reregistered, churned, streamed = [], [0], [0], [0], [0, 0]
    stop = threading.Event()
    rings = [None, None]
    first_submit = [threading.Event(), threading.Event()]
    go = threading.Event()                                         # set by the churner once BOTH streams have a frame in flight

    def streamer(idx):
        try:
            ins = [base[k % 4].copy() for k in range(6)]
            outs = [np.zeros(fb, np.uint8) for _ in range(6)]
            for a in ins + outs:
                mi_lumaeq.host_register(a)
            rings[idx] = ins + outs
            with mi_lumaeq.Context(0) as c:
                with mi_lumaeq.Pipe(c, w, h, depth=3) as pipe:
                    sub = done = 0
                    t_end = None
                    while True:
                        if t_end is None and go.is_set():
                            t_end = time.monotonic() + stream_s    # the clock starts when the churner starts asking
                        more = t_end is None or time.monotonic() < t_end or done < min_frames
                        while more and sub - done < 3:
                            assert pipe.submit(ins[sub % 6], outs[sub % 6], sub)
                            sub += 1
                            first_submit[idx].set()
                        if sub == done:
                            break
                        tag, out = pipe.wait()
                        assert tag == done
                        if not np.array_equal(out, want[(done % 6) % 4]):
                            errors.append(f"streamer {idx}: frame {done} differs from the oracle")
                        out[:64] = 0                               # the slot's next frame must write it again
                        done += 1
                    streamed[idx] = done
        except Exception as e:                                     # noqa: BLE001 -- reported by the main thread
            errors.append(f"streamer {idx}: {e!r}")
        finally:
            first_submit[idx].set()                                # never leave the churner waiting for a stream that died

    def churner():
        try:
            for ev in first_submit:                                # both rings published, both streams have submitted a frame
                ev.wait(timeout=120)
            assert all(r is not None for r in rings), "a streamer never published its ring"
            go.set()
            scratch = [np.zeros(1 << 20, np.uint8) for _ in range(3)]
            while not stop.is_set():
                for a in scratch:
                    mi_lumaeq.host_register(a)
                for a in scratch:
                    mi_lumaeq.host_unregister(a)
                churned[0] += 1
                for ring in rings:
                    for a in ring[::5]:
                        if stop.is_set():
                            break
                        try:
                            mi_lumaeq.host_unregister(a)
                        except mi_lumaeq.MiError as e:
                            assert e.status == mi_lumaeq.ERR_BUSY, e
                            busy_answers[0] += 1
                        else:
                            mi_lumaeq.host_register(a)             # it was idle at that instant: pin it again for its next frame
                            reregistered[0] += 1
        except Exception as e:                                     # noqa: BLE001
            errors.append(f"churner: {e!r}")
            go.set()

    ts = [threading.Thread(target=streamer, args=(i,)) for i in range(2)]
    ch = threading.Thread(target=churner)
    for t in ts:
        t.start()
    ch.start()
    for t in ts:
        t.join(timeout=300)
    stop.set()
    ch.join(timeout=60)
    try:
        assert not errors, errors[:5]
        assert not any(t.is_alive() for t in ts) and not ch.is_alive()
        # the scenario of the docstring really ran: the third thread worked beside the streams, requests hit ring buffers with a DMA in
        # flight (BUSY) AND idle ones (unregistered and pinned again, while frames kept flowing through them)
        assert min(streamed) >= min_frames, streamed
        assert churned[0] >= 3, churned
        assert busy_answers[0] > 0, (busy_answers, reregistered, churned, streamed)
        assert reregistered[0] > 0, (busy_answers, reregistered, churned, streamed)
        print(f"two pipes x {streamed} frames; churner rounds {churned[0]}, BUSY {busy_answers[0]}, unregistered + re-registered {reregistered[0]}")
    finally:
        for ring in rings:
            for a in ring or []:
                try:
                    mi_lumaeq.host_unregister(a)
                except mi_lumaeq.MiError:
                    pass


def test_host_unregister_is_refused_while_a_pipe_dma_is_pending():
    """A frame the caller registered is DMA'd as it is, asynchronously: between mi_pipe_submit and the mi_pipe_wait that retires it the
    copy engines own its pages, and unpinning them then is a GPU access to an ordinary heap address (the round-3 memory fault was one).
    mi_host_unregister answers MI_ERR_BUSY in that window -- for the input and the output buffer -- and works once the frame has been
    waited for, or once the pipe has been destroyed with the frame never waited for.  Unregistered frames are staged: never guarded."""
    w, h = 1920, 1080
    f = synth.nv12_frame(w, h, "D2", 31).copy()
    o, o2 = np.zeros_like(f), np.zeros_like(f)
    plain_in, plain_out = f.copy(), np.zeros_like(f)
    want = oracle.nv12_frame(f, w, h, uv_mode=0, op=0)
    for a in (f, o, o2):
        mi_lumaeq.host_register(a)
    registered = {id(a): a for a in (f, o, o2)}                    # (by identity: `in` / remove on a list would compare array contents)

    def unregister(a):
        mi_lumaeq.host_unregister(a)
        del registered[id(a)]
    try:
        with mi_lumaeq.Context(0) as c:
            with mi_lumaeq.Pipe(c, w, h, depth=3) as pipe:
                assert pipe.submit(f, o, 1)
                for a in (f, o):
                    with pytest.raises(mi_lumaeq.MiError) as e:
                        mi_lumaeq.host_unregister(a)
                    assert e.value.status == mi_lumaeq.ERR_BUSY
                assert pipe.submit(plain_in, plain_out, 2)             # staged both ways: the DMA never sees these arrays
                assert pipe.wait()[0] == 1 and np.array_equal(o, want)
                unregister(o)                                          # frame 1 retired: its output buffer is free
                with pytest.raises(mi_lumaeq.MiError):
                    mi_lumaeq.host_unregister(plain_in)                # never registered: BAD_ARG as before, pending or not
                assert pipe.wait()[0] == 2 and np.array_equal(plain_out, want)
                assert pipe.submit(f, o2, 3)                           # left in flight on purpose
                with pytest.raises(mi_lumaeq.MiError) as e:
                    mi_lumaeq.host_unregister(o2)
                assert e.value.status == mi_lumaeq.ERR_BUSY
            # the pipe is gone (its streams were drained): nothing is pending any more, the buffers are idle -- and that is ALL
            # include/mi_lumaeq.h promises for a frame that was never waited for: o2's content is undefined (here: UV never written)
            unregister(o2)
            unregister(f)
            assert c.get_stat("host_planes_direct") >= 4 and c.get_stat("host_planes_staged") >= 2
            # second variant: the SAME frame waited for before the pipe goes -- now the whole frame is there, UV half included
            mi_lumaeq.host_register(f)
            registered[id(f)] = f
            o3 = np.zeros_like(f)
            mi_lumaeq.host_register(o3)
            registered[id(o3)] = o3
            with mi_lumaeq.Pipe(c, w, h, depth=3) as pipe:
                assert pipe.submit(f, o3, 3)
                with pytest.raises(mi_lumaeq.MiError) as e:
                    mi_lumaeq.host_unregister(o3)
                assert e.value.status == mi_lumaeq.ERR_BUSY
                assert pipe.wait()[0] == 3 and pipe.pending == 0
                assert np.array_equal(o3, want)                        # full frame: Y through the kernels, UV = 128 written by the wait
                unregister(o3)                                         # retired: free, although the pipe still exists
            unregister(f)
    finally:
        for a in list(registered.values()):
            mi_lumaeq.host_unregister(a)


def test_memory_the_caller_pinned_with_hiphostregister_is_dmad_directly(ctx):
    """include/mi_lumaeq.h promises that memory the caller pinned by its own means -- hipHostMalloc (a pinned torch tensor: tested
    above) OR plain hipHostRegister -- is recognised and DMA'd as it is.  The one-allocation rule asks hipMemGetAddressRange about
    both ends of the plane; this checks that the runtime does describe a hipHostRegister'ed range that way (if it did
    not, such planes would silently be staged -- 2.3 k instead of 5.4 k frames/s -- and nothing would fail)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
    w, h = 2048, 512
    raw_in, raw_out = np.zeros(w * h + 8192, np.uint8), np.zeros(w * h + 8192, np.uint8)
    planes = []
    for raw in (raw_in, raw_out):
        off = (-raw.ctypes.data) % 4096
        planes.append(raw[off: off + w * h].reshape(h, w))
    y, dst = planes
    y[:] = synth.y_plane(w, h, "D2", 77)
    want = oracle.equalize_hist(y)
    pinned = []
    try:
        for a in (y, dst):
            assert hip.hipHostRegister(a.ctypes.data, a.nbytes, 0) == 0
            pinned.append(a)
        s0, d0 = ctx.get_stat("host_planes_staged"), ctx.get_stat("host_planes_direct")
        assert np.array_equal(ctx.equalize_hist(y, dst), want)
        assert ctx.get_stat("host_planes_direct") == d0 + 2 and ctx.get_stat("host_planes_staged") == s0, "a hipHostRegister'ed plane was staged"
        # a sub-plane of the registration is still inside ONE allocation
        assert np.array_equal(ctx.equalize_hist(y[: h // 2], dst[: h // 2]), oracle.equalize_hist(y[: h // 2]))
        assert ctx.get_stat("host_planes_direct") == d0 + 4
    finally:
        for a in pinned:
            assert hip.hipHostUnregister(a.ctypes.data) == 0
    # unpinned behind the library's back: the verdict was never cached, the plane is staged from now on
    s1, d1 = ctx.get_stat("host_planes_staged"), ctx.get_stat("host_planes_direct")
    assert np.array_equal(ctx.equalize_hist(y, dst), want)
    assert ctx.get_stat("host_planes_staged") == s1 + 2 and ctx.get_stat("host_planes_direct") == d1


def test_a_plane_is_only_dmad_directly_when_it_lies_in_one_pinned_allocation(ctx):
    """Both ends of a plane pinned is not enough: two separately pinned blocks with ordinary memory between them, or a registration
    that covers part of the plane, must be packed through the library's staging like any unpinned plane (host_range_pinned asks the
    runtime for the allocation around each end).  A pinned torch tensor -- one allocation -- is still DMA'd as it is."""
    w, h = 2048, 512                                                # 1 MiB plane, page-aligned inside a larger array
    raw = np.zeros(w * h + 8192, np.uint8)
    off = (-raw.ctypes.data) % 4096
    y = raw[off: off + w * h].reshape(h, w)
    y[:] = synth.y_plane(w, h, "D2", 12)
    want = oracle.equalize_hist(y)
    head, tail = y[: h // 4], y[3 * h // 4:]                         # first and last quarter pinned, the middle half pageable
    mi_lumaeq.host_register(head)
    mi_lumaeq.host_register(tail)
    try:
        s0, d0 = ctx.get_stat("host_planes_staged"), ctx.get_stat("host_planes_direct")
        got = ctx.equalize_hist(y)
        assert np.array_equal(got, want)
        assert ctx.get_stat("host_planes_staged") == s0 + 2 and ctx.get_stat("host_planes_direct") == d0      # source AND fresh destination staged
        assert np.array_equal(ctx.equalize_hist(head), oracle.equalize_hist(head))                            # a plane inside ONE registration
        assert ctx.get_stat("host_planes_direct") == d0 + 1
    finally:
        mi_lumaeq.host_unregister(head)
        mi_lumaeq.host_unregister(tail)
    pin_in, pin_out = torch.from_numpy(y.copy()).pin_memory(), torch.empty((h, w), dtype=torch.uint8).pin_memory()
    d1 = ctx.get_stat("host_planes_direct")
    got = ctx.equalize_hist(pin_in.numpy(), pin_out.numpy())
    assert np.array_equal(got, want) and ctx.get_stat("host_planes_direct") == d1 + 2


def test_pipe_default_depth_follows_the_frame_size():
    """depth = 0 asks for the library's default: three 4K frames in flight (what one submitting-and-waiting thread drives fastest),
    six frames of 1080p and below; an explicit depth is kept (clamped to 2..16).  Frames still come back in order with the oracle's bytes."""
    with mi_lumaeq.Context(0) as c:
        for (w, h, want) in ((3840, 2160, 3), (1920, 1080, 6), (640, 360, 6)):
            with mi_lumaeq.Pipe(c, w, h) as pipe:
                assert pipe.depth == want, (w, h, pipe.depth)
                f = synth.nv12_frame(w, h, "D2", 5)
                outs = [np.zeros_like(f) for _ in range(want)]
                for k in range(want):
                    assert pipe.submit(f, outs[k], k)
                assert not pipe.submit(f, np.zeros_like(f), 99)          # full at its depth: MI_ERR_BUSY
                assert [pipe.wait()[0] for _ in range(want)] == list(range(want))
                ref = oracle.nv12_frame(f, w, h, uv_mode=0, op=0)
                assert all(np.array_equal(o, ref) for o in outs)
        for asked, got in ((1, 2), (5, 5), (40, 16)):
            with mi_lumaeq.Pipe(c, 640, 360, depth=asked) as pipe:
                assert pipe.depth == got
