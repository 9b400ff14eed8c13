// equalize_fused.hip.h -- KF fused single-read equalizeHist (persistent, ticketed, register-resident slices)
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "common.hip.h"
#include "equalize.hip.h"

namespace mi {
// =============================================================================================
// KF  fused single-read equalizeHist (+ NV12 UV): histogram, CDF/LUT and LUT apply in ONE launch,
// the Y plane read from HBM once.  (SURVEY 8a rows A2+A3+A4+A7.)
//
// MI355X-first design: a 4K Y plane (8.3 MB) does not fit a CU, but it fits the chip: the frame is
// cut into 80 KiB slices, a workgroup keeps its slice in REGISTERS (256 threads x 20 x 16 B) from the
// histogram pass to the apply pass, and the ~102 workgroups holding one frame's slices meet once:
//   1. ticket = atomicAdd(work) -- persistent workgroups take (frame, slice) tickets in order, so the
//      slices of the oldest unfinished frame are always held by running workgroups (no deadlock for
//      any dispatch order as long as >= T workgroups are co-resident; the host guarantees T <= CUs/2);
//   2. slice histogram in LDS (bank-replicated, as K1) -> non-zero bins added to ghist[frame] with
//      agent-scope atomics; every wave waits vmcnt(0); one lane takes an arrival number;
//   3. the LAST arriver exchanges the 256 counts out (returning atomics: coherent by construction),
//      checks sum == W*H (an exact integrity test of the hand-off; retried, bounded), computes the LUT,
//      publishes it with write-through (sc1) stores + checksum, then sets ready[frame];
//   4. the others poll ready[frame] from one lane (relaxed sc1 loads + s_sleep), then ONE agent acquire,
//      then load the 256-byte LUT with sc1 loads and verify the checksum (cdna_hip_programming.md
//      Guideline 16: release on the producer side is replaced by write-through stores drained with
//      vmcnt(0); the consumer keeps the acquire);
//   5. everybody applies the LUT to its registers, streams the result out and stamps its ticket's flag.
// UV planes are extra tickets (pure fill / copy).  HBM traffic per NV12 frame: read W*H, write
// W*H (+ UV) instead of reading W*H twice.
//
// Failure semantics (fail SOFT).  Every spin is bounded (s_memrealtime): on a timeout the workgroup sets
// *status and leaves without stamping its ticket, everybody else follows, so the grid always drains.  The
// launch is ALWAYS followed, on the same stream, by fused_finish_kernel, which (a) in the normal case only
// resets the ticket counter and advances the launch sequence number, and (b) when *status is set redoes every
// ticket whose flag was not stamped -- with the frame's published LUT where one exists, from a fresh histogram
// otherwise -- with no inter-workgroup dependency at all, then cleans the hand-off block and counts the event
// in the sticky statistics words.  The caller's stream therefore always carries correct output; nothing has
// to be polled on the host.  All per-launch state (ticket counter, epoch) lives in the block itself, so a
// captured launch pair replays from a hipGraph unchanged.
// =============================================================================================
constexpr int kVPT = 20;                            // default: 16-byte vectors a thread keeps in registers (80 KiB slices)
constexpr int kLutPubWords = 128;                   // per frame: 64 LUT dwords + checksum, padded to 512 B
constexpr int kFlagStride = 32;                     // one 128-B line per frame flag / counter
// control words at the head of the hand-off block (u32 indices; each group on its own 128-B line)
constexpr int kFusedWork = 0;                       // u64 ticket dispenser, 0 at every launch (reset by the finish kernel)
constexpr int kFusedStatus = 32;                    // != 0: a bounded wait expired in the launch in flight
constexpr int kFusedSeq = 40;                       // launch sequence number; epoch = 2*seq + 1
constexpr int kFusedFin = 48;                       // arrival counter of the finish kernel
constexpr int kFusedStats = 64;                     // sticky: [0] launches repaired, [1] frames repaired, [2] unrecoverable frames, [3] last status
constexpr int kFusedCtlWords = 128;

struct FusedJob {
    const uint8_t* src; uint8_t* dst;               // Y plane of frame 0 (16-B aligned)
    long long src_frame, dst_frame;                 // bytes between frames (multiples of 16)
    long long nvec;                                 // W*H / 16 (exact)
    int total;                                      // W*H
    int n_frames;
    int T, U;                                       // Y tickets / UV tickets per frame
    int slice_vecs;                                 // 16-byte vectors per Y ticket (kThreads * VPT)
    int acquire;                                    // 1: consumers issue an agent acquire before reading the LUT
#ifdef MI_TEST_HOOKS
    int fault_inject;                               // libmi_lumaeq_test.so only, see equalize_fused_kernel
#endif
    unsigned long long timeout_ticks;               // bound of every wait, in 100 MHz ticks
    UVJob uv;
    uint32_t* ctl;                                  // control words (kFused*)
    uint32_t* ghist;                                // [cap][256]        drained by the last arriver of a frame
    uint32_t* cnt;                                  // [cap][kFlagStride] reset by the last arriver of a frame
    uint32_t* ready;                                // [cap][kFlagStride] stamped with the launch epoch
    uint32_t* lutpub;                               // [cap][kLutPubWords] checksum carries the launch epoch
    uint32_t* sflag;                                // [n_frames * (T+U)]  ticket k done <=> sflag[k] == epoch (part of the same block)
    uint32_t* host_repaired;                        // pinned host word: "launches repaired" of this block, written by the finish kernel
    uint32_t* host_hard;                            // pinned host word: "unrecoverable frames" of this block, likewise
};

// The three injected failures exist in libmi_lumaeq_test.so only (-DMI_TEST_HOOKS): the shipping kernel has no such branch.
#ifdef MI_TEST_HOOKS
#define MI_FAULT(j, n) ((j).fault_inject == (n))
#else
#define MI_FAULT(j, n) false
#endif

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Opaque identity: stops LICM/CSE from keeping hundreds of derived values (LDS addresses, extracted pixel
// bytes) alive across the phases of the persistent loop -- without it the kernel spills ~200 VGPRs.
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void launder(u32x4& v) { asm volatile("" : "+v"(v)); }

struct FusedShared {
    EqLutShared eq;
    unsigned long long ticket;
    uint32_t lut_words[64];
    uint32_t red[4];
    uint32_t epoch;
    int last, ok, timeout;
};

__device__ __forceinline__ uint32_t lut_checksum(uint32_t wave_total, uint32_t epoch) { return wave_total + 0x5EED0001u + epoch; }

// Zeroes the hand-off block when it is (re)allocated.
__global__ __launch_bounds__(kThreads) void zero_words_kernel(uint32_t* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) p[i] = 0;
}

// fault_inject (libmi_lumaeq_test.so only, option "fused_fault_inject"):
//   1  the last arriver of frame 0 leaves without publishing its LUT (lost producer: the consumers' waits expire);
//   2  the workgroups holding slices 0 and 1 of frame min(1, n-1), unless they are the last arriver, receive the LUT, then
//      raise *status and leave without writing (a frame left partly written, its LUT published);
//   3  the last arriver of frame 0 publishes a LUT whose checksum never matches and leaves (status 2 on the consumers).
template <int VPT>
__global__ __launch_bounds__(kThreads, 4) void equalize_fused_kernel(FusedJob j)
{
    constexpr int kSliceVecs = kThreads * VPT;
    __shared__ uint32_t lds[256 * kCopies];          // slice histogram, then the replicated LUT
    __shared__ FusedShared sh;
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const unsigned long long P = (unsigned long long)(j.T + j.U);
    const unsigned long long total_tickets = P * (unsigned long long)j.n_frames;
    unsigned long long* const work = reinterpret_cast<unsigned long long*>(j.ctl + kFusedWork);
    uint32_t* const status = j.ctl + kFusedStatus;
    if (t == 0) sh.epoch = ld_agent(j.ctl + kFusedSeq) * 2u + 1u;   // constant during the launch: only the finish kernel advances it
    for (;;) {
        __syncthreads();
        if (t == 0) sh.ticket = __hip_atomic_fetch_add(work, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.epoch);
        unsigned long long k = sh.ticket;                            // make it provably wave-uniform (SGPRs): all the
        k = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(k >> 32)) << 32) |   // per-ticket address math then
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)k);                              // stays scalar (guide T20)
        if (k >= total_tickets) break;
        const int f = (int)(k / P);
        const int r = (int)(k - (unsigned long long)f * P);
        if (r >= j.T) {                               // UV ticket (A7): 64 KiB of plain fill / copy
            uv_flat(j.uv.src ? j.uv.src + (long long)f * j.uv.src_frame : nullptr, j.uv.dst + (long long)f * j.uv.dst_frame,
                    j.uv.bytes, j.uv.mode, r - j.T, j.U);
            if (t == 0) st_agent(j.sflag + k, epoch);
            continue;
        }
        // ---- 1. slice -> registers (loads issued first, LDS zeroing overlaps their latency)
        const long long v0 = (long long)r * kSliceVecs;
        const long long rem = j.nvec - v0;            // vectors of this slice that exist (> 0)
        const int rem32 = (int)(rem < (long long)kSliceVecs ? rem : (long long)kSliceVecs);
        // buffer descriptors over exactly this slice: 32-bit lane offset + scalar offset, and the hardware range
        // check drops the lanes beyond a short last slice (loads return 0, stores are discarded) -- no predicates
        const auto srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(j.src + (long long)f * j.src_frame + v0 * 16), 0, rem32 * 16, 0x00020000);
        const auto drsrc = __builtin_amdgcn_make_buffer_rsrc(j.dst + (long long)f * j.dst_frame + v0 * 16, 0, rem32 * 16, 0x00020000);
        const int toff = t * 16;
        u32x4 q[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) q[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, toff, i * (kThreads * 16), 0);
        for (int i = t; i < 256 * kCopies; i += kThreads) lds[i] = 0;
        __syncthreads();
        // ---- 2. slice histogram, publish with agent-scope atomics
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            if (i * kThreads + t < rem32) hist_add_vec(lds, q[i], copy);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        {
            const uint32_t c = lds_hist_bin(lds, launder(t));
            if (c) __hip_atomic_fetch_add(j.ghist + (size_t)f * 256 + t, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's atomics have been performed
        __syncthreads();
        if (t == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(j.cnt + (size_t)f * kFlagStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.last = (arrived == (uint32_t)(j.T - 1));
            sh.ok = 1;
        }
        __syncthreads();
        uint8_t my_lut;
        if (sh.last && MI_FAULT(j, 1) && f == 0) break;         // test hook 1: a lost producer (the others must time out)
        if (sh.last) {
            // ---- 3. last arriver: collect, verify, compute and publish the LUT
            uint32_t h = 0;
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                h += __hip_atomic_exchange(j.ghist + (size_t)f * 256 + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t ws = wave_sum(h);
                __syncthreads();
                if ((t & 63) == 0) sh.red[t >> 6] = ws;
                if (t == 0) sh.timeout = (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks);   // one decision for the block
                __syncthreads();
                if (sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3] == (uint32_t)j.total) break;
                if (sh.timeout) {
                    if (t == 0) { sh.ok = 0; st_agent(status, 1u); }
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            __syncthreads();
            if (!sh.ok) break;
            my_lut = equalize_lut_value(h, j.total, &sh.eq);
            reinterpret_cast<uint8_t*>(sh.lut_words)[t] = my_lut;
            __syncthreads();
            if (t < 64) {
                const uint32_t w = sh.lut_words[t];
                uint32_t* pub = j.lutpub + (size_t)f * kLutPubWords;
                st_agent(pub + t, w);
                const uint32_t sum = lut_checksum(wave_sum(w), epoch) + (MI_FAULT(j, 3) && f == 0 ? 1u : 0u);   // test hook 3
                if (t == 0) {
                    st_agent(pub + 64, sum);
                    st_agent(j.cnt + (size_t)f * kFlagStride, 0u);  // all T arrivals are in: leave the counter clean for the next launch
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // write-through stores have left this CU
                if (t == 0) st_agent(j.ready + (size_t)f * kFlagStride, epoch);
            }
            if (MI_FAULT(j, 3) && f == 0) break;               // test hook 3: nobody can use this frame's LUT
        } else {
            // ---- 4. wait for the frame's LUT
            if (t == 0) {
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                const uint32_t* flag = j.ready + (size_t)f * kFlagStride;
                while (ld_agent(flag) != epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks || ld_agent(status) != 0u) { sh.ok = 0; st_agent(status, 1u); break; }
                }
                if (j.acquire) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            if (!sh.ok) break;
            if (t < 64) {
                const uint32_t* pub = j.lutpub + (size_t)f * kLutPubWords;
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const uint32_t w = ld_agent(pub + t);
                    const uint32_t want = ld_agent(pub + 64);
                    if (lut_checksum(wave_sum(w), epoch) == want) { sh.lut_words[t] = w; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks || ld_agent(status) != 0u) { if (t == 0) { sh.ok = 0; st_agent(status, 2u); } break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
            if (!sh.ok) break;
            if (MI_FAULT(j, 2) && r <= 1 && f == (j.n_frames > 1 ? 1 : 0)) {   // test hook 2: leave a frame partly written
                if (t == 0) st_agent(status, 1u);
                break;
            }
            my_lut = reinterpret_cast<const uint8_t*>(sh.lut_words)[t];
        }
        // ---- 5. replicated LUT in LDS, apply to the registers, stream out
        __syncthreads();                                            // everyone is done with the histogram in lds[]
        {
            const uint32_t v = my_lut;
            const int tl = launder(t);
#pragma unroll
            for (int c = 0; c < kCopies; ++c) lds[(tl << kCopyShift) + ((c + tl) & (kCopies - 1))] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            launder(q[i]);                                          // re-extract the bytes here instead of keeping 256 of them live
            __builtin_amdgcn_raw_buffer_store_b128(lut_vec(lds, q[i], copy), drsrc, toff, i * (kThreads * 16), 0);
            __builtin_amdgcn_sched_barrier(0);                      // keep the bodies apart: the slice already owns 4*VPT VGPRs
        }
        if (t == 0) st_agent(j.sflag + k, epoch);                   // this ticket's output is on its way (complete at kernel end)
    }
}

// ---------------------------------------------------------------------------------------------------------
// Finish kernel: runs after EVERY equalize_fused_kernel launch, on the same stream.  grid = min(n_frames, 4*CUs).
// Normal case (*status == 0): one load per workgroup; the last workgroup to arrive resets the ticket counter and
// advances the sequence number (=> a new epoch for the next launch: flags, LUT checksums and ticket stamps of this
// launch can never be mistaken for the next one's).
// Failure case: each workgroup repairs whole frames on its own (no inter-workgroup dependency, so it cannot stall
// whatever else runs on the GPU): a ticket is redone iff its stamp is missing.  Source pixels of a missing Y ticket are
// intact even when the call is in place, because a slice is only overwritten by the ticket that owns it.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void fused_finish_kernel(FusedJob j)
{
    __shared__ uint32_t lds[256 * kCopies];
    __shared__ FusedShared sh;
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    uint32_t* const stats = j.ctl + kFusedStats;
    const uint32_t status = ld_agent(j.ctl + kFusedStatus);         // written by the fused launch only: uniform over the grid
    const uint32_t seq = ld_agent(j.ctl + kFusedSeq);
    const uint32_t epoch = seq * 2u + 1u;
    if (status != 0) {
        const int P = j.T + j.U;
        for (int f = blockIdx.x; f < j.n_frames; f += gridDim.x) {
            const uint32_t* fl = j.sflag + (size_t)f * P;
            j.ghist[(size_t)f * 256 + t] = 0;                       // whatever the broken hand-off left behind
            if (t == 0) j.cnt[(size_t)f * kFlagStride] = 0;
            int undone = 0, done = 0;
            for (int r = t; r < j.T; r += kThreads) { const bool d = ld_agent(fl + r) == epoch; undone |= !d; done |= d; }
            const int any_undone = __syncthreads_or(undone);
            const int any_done = __syncthreads_or(done);
            const uint8_t* src = j.src + (long long)f * j.src_frame;
            uint8_t* dst = j.dst + (long long)f * j.dst_frame;
            if (any_undone) {
                // the frame's published LUT, if this launch got that far (stamped + checksummed with this launch's epoch)
                if (t == 0) sh.ok = 0;
                __syncthreads();
                if (t < 64 && ld_agent(j.ready + (size_t)f * kFlagStride) == epoch) {
                    const uint32_t* pub = j.lutpub + (size_t)f * kLutPubWords;
                    const uint32_t w = ld_agent(pub + t);
                    if (lut_checksum(wave_sum(w), epoch) == ld_agent(pub + 64)) { sh.lut_words[t] = w; if (t == 0) sh.ok = 1; }
                }
                __syncthreads();
                const bool have_lut = sh.ok != 0;
                bool skip = false;
                uint8_t my_lut = 0;
                if (have_lut) {
                    my_lut = reinterpret_cast<const uint8_t*>(sh.lut_words)[t];
                } else if (any_done && src == dst) {
                    // cannot happen under the protocol (a slice is written only after a valid LUT was received); never guess
                    if (t == 0) __hip_atomic_fetch_add(stats + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    skip = true;
                } else {
                    lds_hist_zero(lds);
                    hist_flat(lds, src, (long long)j.total, 0, 1);
                    __syncthreads();
                    my_lut = equalize_lut_value(lds_hist_bin(lds, t), j.total, &sh.eq);
                }
                __syncthreads();
                if (!skip) {
                    const uint32_t v = my_lut;
#pragma unroll
                    for (int c = 0; c < kCopies; ++c) lds[(t << kCopyShift) + ((c + t) & (kCopies - 1))] = v;
                    __syncthreads();
                    for (int r = 0; r < j.T; ++r) {
                        if (ld_agent(fl + r) == epoch) continue;    // same address for every lane: uniform branch
                        const long long v0 = (long long)r * j.slice_vecs;
                        const long long nv = j.nvec - v0 < (long long)j.slice_vecs ? j.nvec - v0 : (long long)j.slice_vecs;
                        lut_flat(lds, src + v0 * 16, dst + v0 * 16, nv * 16, 0, 1);
                    }
                    if (t == 0) __hip_atomic_fetch_add(stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
            for (int u = 0; u < j.U; ++u) {
                if (ld_agent(fl + j.T + u) == epoch) continue;
                uv_flat(j.uv.src ? j.uv.src + (long long)f * j.uv.src_frame : nullptr, j.uv.dst + (long long)f * j.uv.dst_frame,
                        j.uv.bytes, j.uv.mode, u, j.U);
            }
        }
        (void)copy;
    }
    // last one out: the ticket counter, the status word and the epoch are ready for the next launch
    __syncthreads();
    if (t == 0) {
        const uint32_t arrived = __hip_atomic_fetch_add(j.ctl + kFusedFin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            if (status != 0) {
                const uint32_t repaired = __hip_atomic_fetch_add(stats + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                st_agent(stats + 3, status);
                // the host decides about demoting the fused path from this word, without a copy or a synchronisation
                if (j.host_repaired) __hip_atomic_store(j.host_repaired, repaired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // every workgroup's count of refused frames was added before it arrived here (the barrier above drains its atomics)
                const uint32_t hard = __hip_atomic_load(stats + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (j.host_hard && hard) __hip_atomic_store(j.host_hard, hard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            st_agent(j.ctl + kFusedStatus, 0u);
            st_agent(j.ctl + kFusedWork, 0u); st_agent(j.ctl + kFusedWork + 1, 0u);
            st_agent(j.ctl + kFusedSeq, seq + 1u);
            st_agent(j.ctl + kFusedFin, 0u);
        }
    }
}

}  // namespace mi
