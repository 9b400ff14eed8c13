// clahe_cell.hip.h -- KC single-read CLAHE by CELLS (docs/experiments.md R5.4-R5.6): histogram, clip / redistribute / LUT and the
// bilinear blend in ONE launch, the Y plane read from HBM once.  Part of the gfx950 kernel set of libmi_lumaeq (see
// ../lumaeq_kernels.hip.h for the design notes).  Reference call being replaced: clahevideo.cpp:195, clahe1frame.cpp:92-95
// (CLAHE::apply); arithmetic: SURVEY.md App. A.2, the same device helpers as clahe.hip.h.
//
// A CELL is the intersection of one tile with one interpolation band and one horizontal tile pair -- a quadrant of a tile.  Every
// pixel of a cell (a) counts into ONE tile histogram and (b) is blended from the SAME four tile LUTs {r1, r2} x {ta, tb}
// (clahe.cpp CLAHE_Interpolation_Body: ty1 / tx1 are constant over the cell), one of which is the cell's own tile.  A 256-thread
// workgroup keeps the cell's pixels in REGISTERS (8 x 16 B per lane) from the histogram to the blend; its 32 KiB of LDS hold first the
// bank-replicated histogram and then the cell's ONE table as f32x4[256][8] -- eight replicas, lane uses copy (lane & 7), so the eight
// lanes a ds_read_b128 serves per cycle hit eight different bank quads whatever the pixel values are.
//
// Hand-off (the fused equalizeHist kernel's, equalize_fused.hip.h, per TILE instead of per frame):
//   1. persistent workgroups draw cell tickets from EIGHT dispensers, workgroup b from dispenser b % 8.  Dispenser x hands out the
//      cells of cell rows x, x + 8, ... frame after frame, left to right: all cells of a cell row are worked on by workgroups of one
//      XCD (workgroups are dealt round-robin over the XCDs), whose L2 then serves the 128-byte lines that the 240-byte cell rows of
//      horizontal neighbours share -- stores 201 -> 124 us, loads 154 -> 124 us per 64 4K frames (profiles/r05_c_*).  Correctness
//      never depends on where a workgroup really runs;
//   2. cell histogram in LDS -> its non-zero bins added to the TILE's global histogram with agent-scope atomics, then one arrival on
//      the tile's counter.  Nobody waits for the adds to land: step 3 validates itself;
//   3. the LAST of a tile's four arrivers exchanges the 256 counts out (returning atomics: coherent by construction), again and
//      again until their sum is the tile's area -- an exact test that every add of all four cells has arrived (bounded) -- then clips,
//      redistributes, scans, and stores the tile LUT, its salted checksum and the tile's flag (epoch) without waiting in between;
//   4. every cell waits for the flags of its four tiles (bounded) and loads the four LUTs, retrying until the checksum holds: a flag
//      that overtook its LUT, or a stale line of an earlier launch, cannot satisfy it.  Every word read after a wait is read with an
//      agent-scope (sc1) load or a returning atomic, so no acquire fence is issued -- `buffer_inv sc1` four times per cell cost
//      240 us per 64 4K frames (profiles/r05_d_*);
//   5. builds its table, blends from the registers, streams out, stamps its ticket.
// What bounds the kernel is how long a cell is HELD: time = cells x hold time / resident workgroups (16 384 x 31 us / 1024 = 499 us
// with the first hand-off, which waited for vmcnt(0) three times on the critical path and read four partials one after the other).
// Hence: nothing on the chain waits for a store to land, and the workgroup's scalars live INSIDE the 32 KiB histogram / table array
// (they are only needed while it holds neither), so that five workgroups fit a CU's 160 KiB.
// No deadlock for any dispatch order as long as cells_x workgroups per dispenser are co-resident: a cell waits only for tiles of tile
// rows <= its own + 1, i.e. for cells at most two cell rows further on, and consecutive cell rows belong to different dispensers, so
// of everything the oldest unfinished cell waits for, a dispenser owes at most ONE cell row -- cells_x tickets, all of them drawn
// (every ticket before them is finished) and none of them waiting before its histogram is out.  The host launches 8 n workgroups
// with n >= cells_x inside its co-residency allowance.
// In place is safe: a cell is overwritten only after its OWN tile's LUT has been published, i.e. after all four cells of the tile have
// been counted; no other tile's histogram reads those pixels.
//
// Failure semantics (fail SOFT, as the fused equalizeHist kernel): every wait is bounded (s_memrealtime); on expiry the workgroup sets
// *status and leaves without stamping, everybody follows, the grid drains.  The launch is ALWAYS followed by clahe_cell_finish_kernel
// on the same stream: housekeeping in the normal case; when *status is set it redoes every unstamped cell with no inter-workgroup
// dependency -- tile LUTs taken from what was published where the checksum holds, recomputed from the source otherwise (a tile whose
// LUT was never published has had no cell written, so its pixels are intact, in place included).
//
// Only "regular" geometries take this path (the host checks, clahe_cell_geometry): no padding, tile_w a multiple of 32, and the
// band / pair boundaries -- found with the reference's own float expressions -- at the same offset in every tile, the column one on a
// multiple of 16.  Everything else runs the two-pass kernels of clahe.hip.h.
#pragma once
#include "clahe.hip.h"
#include "equalize_fused.hip.h"      // ld_agent / st_agent / wave_sum / launder

namespace mi {

constexpr int kCellVPT = 8;            // 16-byte vectors (rows) a lane of a 256-thread workgroup keeps per cell (stage-1 kernel)
constexpr int kCell2Threads = 512;     // the fused kernel: 512 threads per workgroup, TWO cells in flight, ...
constexpr int kCell2VPT = 4;           // ... four vectors per lane and cell: 32 VGPRs of pixels in all
constexpr int kCellRep = 8;            // replicas of the cell's f32x4 table: 256 x 8 x 16 B = 32 KiB, the histogram's LDS
constexpr int kCellQueues = 8;         // ticket dispensers = XCDs
constexpr int kCellLutWords = 72;      // a tile's published LUT: 64 dwords + checksum, padded
// control words at the head of the hand-off block (u32 indices)
constexpr int kCellWork = 0;           // 8 x u64 ticket dispensers, 0 at every launch (reset by the finish kernel)
constexpr int kCellStatus = 32;        // != 0: a bounded wait expired in the launch in flight
constexpr int kCellSeq = 40;           // launch sequence number; epoch = 2 * seq + 1
constexpr int kCellFin = 48;           // arrival counter of the finish kernel
constexpr int kCellStats = 64;         // sticky: [0] launches repaired, [1] cells repaired, [2] last status
constexpr int kCellCtlWords = 128;

struct CellGeom {
    int cells_x, cells_y;              // 2 * tiles_x, 2 * tiles_y
    int groups, phases;                // 16-pixel column groups of a cell (tile_w / 32); row phases = 256 / groups
    int ysplit;                        // rows of a tile that belong to the band above (ty1 = ty - 1): rows [0, ysplit); the rest: ty1 = ty
    int variant;                       // measurement switches of the stage-1 kernel (option "clahe_cell_variant"), 0 in production
};

struct CellJob {
    PlaneBatch p;
    ClaheGeom g;
    CellGeom cg;
    UVJob uv;
    int n_frames;
    int rows_per_queue;                // cell rows a dispenser owns per frame = ceil(cells_y / 8)
    int acquire;                       // 1: consumers issue an agent acquire before reading published data
#ifdef MI_TEST_HOOKS
    int fault_inject;                  // libmi_lumaeq_test.so only, see clahe_cell_fused_kernel
#endif
    unsigned long long timeout_ticks;  // bound of every wait, in 100 MHz ticks
    uint32_t* ctl;                     // control words (kCell*)
    uint32_t* tcnt;                    // [frames * tiles]  arrivals of a tile's four cells; reset by the last arriver
    uint32_t* tready;                  // [frames * tiles]  stamped with the launch epoch when the tile's LUT is out
    uint32_t* lutpub;                  // [frames * tiles][kCellLutWords]
    uint32_t* ghist;                   // [frames * tiles][256]  drained (exchanged to zero) by the last arriver of a tile
    uint32_t* sflag;                   // [frames * cells]  cell done <=> sflag == epoch
    uint8_t* luts_fix;                 // [frames * tiles][256] plain LUT bytes: scratch of the repair pass only
    uint32_t* host_repaired;           // pinned host word: "launches repaired", written by the finish kernel
};

#ifdef MI_TEST_HOOKS
#define MI_CELL_FAULT(j, n) ((j).fault_inject == (n))
#else
#define MI_CELL_FAULT(j, n) false
#endif

// 16 pixels of one row from the cell's replicated table: clahe_vec16_f32 (clahe.hip.h) with ONE table and a lane-fixed replica
template <bool FMA>
__device__ __forceinline__ u32x4 cell_vec16(const f32x4* quadf, u32x4 q, uint32_t rep, const f32x2* xw, float ya, float ya1)
{
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    const f32x2 yv = {ya1, ya};
    uint32_t ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 e[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) e[b] = quadf[(((w[k] >> (8 * b)) & 0xffu) << 3) + rep];
        f32x2 tb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = k * 4 + b;
            const f32x2 ac = {e[b].x, e[b].y}, bd = {e[b].z, e[b].w};
            if (FMA) tb[b] = pk_fma_bcast_lo(ac, xw[j], pk_mul_bcast_hi(bd, xw[j]));
            else tb[b] = (pk_mul_bcast_lo(ac, xw[j]) + pk_mul_bcast_hi(bd, xw[j])) * yv;
        }
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float r = FMA ? __fmaf_rn(tb[b].x, ya1, __fmul_rn(tb[b].y, ya)) : __fadd_rn(tb[b].x, tb[b].y);
            acc = __builtin_amdgcn_cvt_pk_u8_f32(rintf(r), b, acc);
        }
        ow[k] = acc;
    }
    u32x4 o; o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
    return o;
}

// Where a cell lies and which lanes own which of its rows (wave-uniform except grp / phase / my_rows)
struct CellPlace {
    int tx, ty, hx, hy, pr, bnd;       // tile, quadrant, pair (unclamped tx1 = pr - 1), band (unclamped ty1 = bnd - 1)
    int x0, y0, nrows, cw;
    int grp, phase, phases, my_rows;   // phases = row phases of THIS workgroup = blockDim.x / groups (256 or 512 threads)
    __device__ __forceinline__ CellPlace(int cell, const ClaheGeom& g, const CellGeom& cg)
    {
        const int cy = cell / cg.cells_x, cx = cell - cy * cg.cells_x;
        tx = cx >> 1; hx = cx & 1; ty = cy >> 1; hy = cy & 1;
        pr = tx + hx; bnd = ty + hy;
        cw = g.tile_w >> 1;
        x0 = tx * g.tile_w + hx * cw;
        y0 = ty * g.tile_h + hy * cg.ysplit;
        nrows = hy ? g.tile_h - cg.ysplit : cg.ysplit;
        grp = (int)threadIdx.x % cg.groups; phase = (int)threadIdx.x / cg.groups;
        phases = (int)blockDim.x / cg.groups;
        my_rows = phase < phases ? nrows : 0;                    // rows this lane may touch: phase, phase + phases, ... < my_rows
    }
};

// salted checksum: a stale LUT of an earlier launch, or of another tile, must not pass
__device__ __forceinline__ uint32_t cell_lut_checksum(uint32_t wave_total, uint32_t epoch, uint32_t tile_id) { return wave_total + 0x5EED0C1Au + epoch * 0x85EBCA6Bu + tile_id; }

// The workgroup's scalars.  In the fused kernel this struct is OVERLAID on the first words of the 32 KiB LDS array: it is used only
// while the array holds neither the histogram nor the table (between the fold and the table build, and around the ticket draw).
struct CellShared {
    unsigned long long ticket;
    uint32_t lutw[4][64];              // the four tile LUTs of this cell: {r1,ta}, {r1,tb}, {r2,ta}, {r2,tb}
    uint32_t red[4];
    uint32_t s_wave[4];
    uint32_t epoch;
    int last, ok, timeout;
};

// the cell's table entry of value v = threadIdx.x from the four LUTs held as bytes in sh.lutw: {a, c, b, d} = {r1,ta}, {r2,ta}, {r1,tb}, {r2,tb}
__device__ __forceinline__ f32x4 cell_table_entry(const CellShared& sh)
{
    const int tl = launder((int)threadIdx.x);
    return f32x4{(float)reinterpret_cast<const uint8_t*>(sh.lutw[0])[tl], (float)reinterpret_cast<const uint8_t*>(sh.lutw[2])[tl],
                 (float)reinterpret_cast<const uint8_t*>(sh.lutw[1])[tl], (float)reinterpret_cast<const uint8_t*>(sh.lutw[3])[tl]};
}
// ... written eight times (the caller has made sure nobody still reads what the table overwrites -- sh itself, in the fused kernel)
__device__ __forceinline__ void cell_write_table(f32x4* quadf, f32x4 e)
{
    const int tl = launder((int)threadIdx.x);
#pragma unroll
    for (int r = 0; r < kCellRep; ++r) quadf[(tl << 3) + ((r + tl) & (kCellRep - 1))] = e;
}

// column weights {xa1, xa} of this lane's 16 columns (all in pair cp.pr by construction of the cell)
template <bool FMA>
__device__ __forceinline__ void cell_col_weights(f32x2* xw, const CellPlace& cp, const ClaheGeom& g)
{
#pragma unroll
    for (int jx = 0; jx < kInterpPx; ++jx) {
        const float txf = tile_coord<FMA>(cp.x0 + cp.grp * 16 + jx, g.inv_tw);
        const float xa = __fsub_rn(txf, (float)(cp.pr - 1));
        xw[jx].x = __fsub_rn(1.0f, xa); xw[jx].y = xa;
    }
}

// ---------------------------------------------------------------------------------------------------------
// KC  the fused kernel.  grid = 8 n persistent workgroups.
// fault_inject (libmi_lumaeq_test.so only, option "fused_fault_inject"):
//   1  the last arriver of tile 0 of frame 0 leaves without publishing its LUT (lost producer: the consumers' waits expire);
//   2  the workgroup holding cell 1 of frame min(1, n-1) receives its LUTs, then raises *status and leaves without writing;
//   3  the last arriver of tile 0 of frame 0 publishes a LUT whose checksum never matches.
// ---------------------------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(kCell2Threads, 4) void clahe_cell_fused_kernel(CellJob j)
{
    constexpr int NT = kCell2Threads;
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * kCopies];   // EXACTLY 32 KiB
    static_assert(sizeof(CellShared) <= 2048, "CellShared is overlaid on the head of lds[]");
    CellShared& sh = *reinterpret_cast<CellShared*>(lds);          // valid only while lds[] holds neither histogram nor table
    f32x4* const quadf = reinterpret_cast<f32x4*>(lds);
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const bool lower = __builtin_amdgcn_readfirstlane(t >> 6) < kThreads / 64;     // waves 0..3: the 256 threads of the fold / scan / LUT stages
    const int queue = (int)(blockIdx.x & (kCellQueues - 1));
    const int tiles = j.g.tiles_x * j.g.tiles_y, cells = j.cg.cells_x * j.cg.cells_y;
    const int per_frame = j.rows_per_queue * j.cg.cells_x;        // tickets a dispenser hands out per frame
    const unsigned long long total_tickets = (unsigned long long)per_frame * (unsigned long long)j.n_frames;
    unsigned long long* const work = reinterpret_cast<unsigned long long*>(j.ctl + kCellWork) + queue;
    uint32_t* const status = j.ctl + kCellStatus;
    // constant during the launch: only the finish kernel advances the sequence number (every lane loads the same word)
    const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ld_agent(j.ctl + kCellSeq) * 2u + 1u));
    const int phases = NT / j.cg.groups;
    const int sstride = phases * (int)j.p.src_step, dstride = phases * (int)j.p.dst_step;
    // TWO cells per workgroup (stage 3, R5.8).  Cell A was loaded, counted and published in the previous iteration and still waits for
    // its four tile LUTs; cell B is this iteration's ticket.  B's load + histogram + publication (no wait in it) runs BEFORE A's wait,
    // so by the time A asks for its LUTs they are several microseconds old: the hand-off latency that stage 2 could not hide (R5.6)
    // is covered by the other cell's work.  512 threads, so that two cells are 2 x 4 vectors per lane.
    bool have_a = false;
    int f_a = 0, cell_a = 0;
    u32x4 qa[kCell2VPT];
    for (;;) {
        __syncthreads();                                            // cell A's table of the previous iteration is no longer read: lds[] is free
        if (t == 0) sh.ticket = __hip_atomic_fetch_add(work, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        unsigned long long k = sh.ticket;
        k = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(k >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)k);
        const bool drained = k >= total_tickets;
        if (drained && !have_a) break;
        int f_b = 0, cell_b = 0;
        bool have_b = false;
        if (!drained) {
            f_b = (int)(k / (unsigned long long)per_frame);
            const int r = (int)(k - (unsigned long long)f_b * per_frame);
            const int cy = (r / j.cg.cells_x) * kCellQueues + queue;
            have_b = cy < j.cg.cells_y;                              // cells_y not a multiple of 8: a dispenser's last row may not exist
            cell_b = cy * j.cg.cells_x + (r % j.cg.cells_x);
        }
        __syncthreads();                                            // everybody has read its ticket: sh may go under the histogram
        u32x4 qb[kCell2VPT];
        if (have_b) {
            // ---- 1. cell B -> registers.  Buffer descriptors over exactly the cell's span: rows beyond the cell and idle lanes fall
            // outside the range (loads return 0, stores are dropped), so neither needs a predicate.
            const CellPlace cp(cell_b, j.g, j.cg);
            const int span = (cp.nrows - 1) * (int)j.p.src_step + cp.cw;
            const auto srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(j.p.src + (long long)f_b * j.p.src_frame + (long long)cp.y0 * j.p.src_step + cp.x0), 0, span, 0x00020000);
            const int soff = cp.phase * (int)j.p.src_step + cp.grp * 16 + (cp.phase < phases ? 0 : 0x40000000);
#pragma unroll
            for (int i = 0; i < kCell2VPT; ++i) qb[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, soff, i * sstride, 0);
            for (int i = t; i < 256 * kCopies; i += NT) lds[i] = 0;
            __syncthreads();
            // ---- 2. B's histogram; non-zero bins -> the tile's global histogram; arrive on the tile
#pragma unroll
            for (int i = 0; i < kCell2VPT; ++i) {
                if (cp.phase + i * phases < cp.my_rows) hist_add_vec(lds, qb[i], copy);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            const size_t tile_id = (size_t)f_b * tiles + cp.ty * j.g.tiles_x + cp.tx;
            if (lower) {
                const uint32_t c = lds_hist_bin(lds, launder(t));
                if (c && !(j.cg.variant & 128)) __hip_atomic_fetch_add(j.ghist + tile_id * 256 + t, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                                        // every bin has been folded: lds[] is free, sh is valid again
            if (t == 0) {
                sh.ok = 1;
                if (j.cg.variant & 128) sh.last = 0;                 // (128: measurement only -- nothing published, nobody arrives)
                else sh.last = (__hip_atomic_fetch_add(j.tcnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3u);
            }
            __syncthreads();
            if (sh.last && MI_CELL_FAULT(j, 1) && tile_id == 0) break;   // test hook 1: a lost producer
            if (sh.last) {
                // ---- 3. last arriver of the tile: drain the tile's histogram until it is complete, compute and publish the tile LUT
                // (256-thread stages: waves 4..7 only keep the barriers company)
                uint32_t h = 0;
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    uint32_t ws = 0;
                    if (lower) {
                        h += __hip_atomic_exchange(j.ghist + tile_id * 256 + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ws = wave_sum(h);
                    }
                    __syncthreads();
                    if (lower && (t & 63) == 0) sh.red[t >> 6] = ws;
                    if (t == 0) sh.timeout = (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks);   // one decision for the block
                    __syncthreads();
                    if (sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3] == (uint32_t)(j.g.tile_w * j.g.tile_h)) break;   // every add of all four cells is in
                    if (sh.timeout) {
                        if (t == 0) { sh.ok = 0; st_agent(status, 1u); }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                __syncthreads();
                if (!sh.ok) break;
                if (lower) {
                    const uint8_t lv = tile_lut_value(h, j.g, sh.s_wave);    // two barriers per scan inside
                    reinterpret_cast<uint8_t*>(sh.lutw[0])[t] = lv;
                } else {
                    if (j.g.clip > 0) { __syncthreads(); __syncthreads(); }
                    __syncthreads(); __syncthreads();
                }
                __syncthreads();
                if (t < 64) {
                    const uint32_t w = sh.lutw[0][t];
                    uint32_t* pub = j.lutpub + tile_id * kCellLutWords;
                    st_agent(pub + t, w);
                    const uint32_t sum = cell_lut_checksum(wave_sum(w), epoch, (uint32_t)tile_id) + (MI_CELL_FAULT(j, 3) && tile_id == 0 ? 1u : 0u);   // test hook 3
                    if (t == 0) {
                        st_agent(pub + 64, sum);
                        st_agent(j.tcnt + tile_id, 0u);             // all four arrivals are in: leave the counter clean for the next launch
                        st_agent(j.tready + tile_id, epoch);        // not ordered behind the LUT: the consumers' checksum loop covers that
                    }
                }
                __syncthreads();
            }
        } else {
            if (t == 0) sh.ok = 1;
            __syncthreads();
        }
        if (have_a) {
            // ---- 4. cell A's four tile LUTs (published while B was being counted): wave w < 4 fetches LUT w = {r1,ta}, {r1,tb}, {r2,ta}, {r2,tb}
            const CellPlace cp(cell_a, j.g, j.cg);
            if (lower) {
                const int wv = t >> 6, lane = t & 63;
                const int ta = max(cp.pr - 1, 0), tb = min(cp.pr, j.g.tiles_x - 1);
                const int r1 = max(cp.bnd - 1, 0), r2 = min(cp.bnd, j.g.tiles_y - 1);
                const int want_tile = ((wv & 2) ? r2 : r1) * j.g.tiles_x + ((wv & 1) ? tb : ta);
                const size_t wid = (size_t)f_a * tiles + want_tile;
                if (j.cg.variant & 64) {                             // measurement only: LUTs of a preceding tile-histogram pass, nobody waits
                    sh.lutw[wv][lane] = reinterpret_cast<const uint32_t*>(j.luts_fix + wid * 256)[lane];
                } else {
                    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                    int wave_ok = 1;
                    if (lane == 0) {
                        const uint32_t* flag = j.tready + wid;
                        while (ld_agent(flag) != epoch) {
                            __builtin_amdgcn_s_sleep(4);
                            if (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks || ld_agent(status) != 0u) { wave_ok = 0; sh.ok = 0; st_agent(status, 1u); break; }
                        }
                    }
                    wave_ok = __builtin_amdgcn_readfirstlane(wave_ok);   // lane 0's verdict for the wave
                    const uint32_t* pub = j.lutpub + wid * kCellLutWords;
                    while (wave_ok) {
                        const uint32_t w = ld_agent(pub + lane);
                        const uint32_t want = ld_agent(pub + 64);
                        if (cell_lut_checksum(wave_sum(w), epoch, (uint32_t)wid) == want) { sh.lutw[wv][lane] = w; break; }
                        if (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks || ld_agent(status) != 0u) { if (lane == 0) { sh.ok = 0; st_agent(status, 2u); } break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
            }
            __syncthreads();
            if (!sh.ok) break;
            if (MI_CELL_FAULT(j, 2) && cell_a == 1 && f_a == (j.n_frames > 1 ? 1 : 0)) {   // test hook 2: leave a frame partly written
                if (t == 0) st_agent(status, 1u);
                break;
            }
            // ---- 5. A's table (over sh: every lane takes its entry first; lane t writes value t & 255, replicas 4 (t >> 8) .. + 3),
            // blend from the registers, stream out, stamp
            const int v = launder(t & 255);
            const f32x4 entry = {(float)reinterpret_cast<const uint8_t*>(sh.lutw[0])[v], (float)reinterpret_cast<const uint8_t*>(sh.lutw[2])[v],
                                 (float)reinterpret_cast<const uint8_t*>(sh.lutw[1])[v], (float)reinterpret_cast<const uint8_t*>(sh.lutw[3])[v]};
            __syncthreads();
            {
                const int r0 = (t >> 8) * (kCellRep / 2);
#pragma unroll
                for (int r = 0; r < kCellRep / 2; ++r) quadf[(v << 3) + ((r0 + r + v) & (kCellRep - 1))] = entry;
            }
            f32x2 xw[kInterpPx];                                    // (computed here, not earlier: 32 VGPRs that need not live across the wait)
            cell_col_weights<FMA>(xw, cp, j.g);
            __syncthreads();
            const int dspan = (cp.nrows - 1) * (int)j.p.dst_step + cp.cw;
            const auto drsrc = __builtin_amdgcn_make_buffer_rsrc(j.p.dst + (long long)f_a * j.p.dst_frame + (long long)cp.y0 * j.p.dst_step + cp.x0, 0, dspan, 0x00020000);
            const int doff = cp.phase * (int)j.p.dst_step + cp.grp * 16 + (cp.phase < phases ? 0 : 0x40000000);
            const uint32_t rep = t & (kCellRep - 1);
#pragma unroll
            for (int i = 0; i < kCell2VPT; ++i) {
                if (cp.phase + i * phases < cp.my_rows) {
                    launder(qa[i]);
                    const float tyf = tile_coord<FMA>(cp.y0 + cp.phase + i * phases, j.g.inv_th);
                    const float ya = __fsub_rn(tyf, (float)(cp.bnd - 1)), ya1 = __fsub_rn(1.0f, ya);
                    __builtin_amdgcn_raw_buffer_store_b128(cell_vec16<FMA>(quadf, qa[i], rep, xw, ya, ya1), drsrc, doff, i * dstride, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j.uv.bytes > 0 && lower)                              // (uv_flat strides by 256 threads)
                uv_flat(j.uv.src ? j.uv.src + (long long)f_a * j.uv.src_frame : nullptr, j.uv.dst + (long long)f_a * j.uv.dst_frame, j.uv.bytes, j.uv.mode, cell_a, cells);
            if (t == 0) st_agent(j.sflag + (size_t)f_a * cells + cell_a, epoch);   // this cell's output is on its way (complete at kernel end)
        }
        // B becomes A
        have_a = have_b; f_a = f_b; cell_a = cell_b;
#pragma unroll
        for (int i = 0; i < kCell2VPT; ++i) qa[i] = qb[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Finish kernel: runs after EVERY clahe_cell_fused_kernel launch, on the same stream.  grid = min(n_frames, 4 * CUs).
// Normal case (*status == 0): one load per workgroup; the last workgroup to arrive resets the dispensers and advances the sequence
// number (a new epoch: flags, checksums and stamps of this launch can never be mistaken for the next one's).
// Failure case: a workgroup repairs whole frames on its own, with no inter-workgroup dependency: (a) every tile's LUT as plain bytes
// into luts_fix -- the published one where its flag and checksum carry this launch's epoch, recomputed from the source pixels
// otherwise; (b) every cell without a stamp: table from luts_fix, blend, store, its share of the UV plane.
// ---------------------------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(kThreads) void clahe_cell_finish_kernel(CellJob j)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * kCopies];
    __shared__ CellShared sh;
    f32x4* const quadf = reinterpret_cast<f32x4*>(lds);
    const int t = threadIdx.x;
    uint32_t* const stats = j.ctl + kCellStats;
    const uint32_t status = ld_agent(j.ctl + kCellStatus);          // written by the fused launch only: uniform over the grid
    const uint32_t seq = ld_agent(j.ctl + kCellSeq);
    const uint32_t epoch = seq * 2u + 1u;
    const int tiles = j.g.tiles_x * j.g.tiles_y, cells = j.cg.cells_x * j.cg.cells_y;
    if (status != 0) {
        for (int f = blockIdx.x; f < j.n_frames; f += gridDim.x) {
            const uint32_t* fl = j.sflag + (size_t)f * cells;
            int undone = 0;
            for (int c = t; c < cells; c += kThreads) undone |= (ld_agent(fl + c) != epoch);
            // whatever the broken hand-off left behind in this frame's arrival counters and tile histograms
            for (int tile = t; tile < tiles; tile += kThreads) st_agent(j.tcnt + (size_t)f * tiles + tile, 0u);
            for (size_t i = t; i < (size_t)tiles * 256; i += kThreads) st_agent(j.ghist + (size_t)f * tiles * 256 + i, 0u);
            if (!__syncthreads_or(undone)) continue;
            const uint8_t* src = j.p.src + (long long)f * j.p.src_frame;
            // (a) the frame's tile LUTs as plain bytes
            for (int tile = 0; tile < tiles; ++tile) {
                const size_t tile_id = (size_t)f * tiles + tile;
                __syncthreads();
                if (t == 0) sh.ok = 0;
                __syncthreads();
                if (t < 64 && ld_agent(j.tready + tile_id) == epoch) {
                    const uint32_t* pub = j.lutpub + tile_id * kCellLutWords;
                    const uint32_t w = ld_agent(pub + t);
                    if (cell_lut_checksum(wave_sum(w), epoch, (uint32_t)tile_id) == ld_agent(pub + 64)) { sh.lutw[0][t] = w; if (t == 0) sh.ok = 1; }
                }
                __syncthreads();
                uint8_t lv;
                if (sh.ok) {
                    lv = reinterpret_cast<const uint8_t*>(sh.lutw[0])[t];
                } else {
                    // never published: no cell of this tile has been written (a cell needs its own tile's LUT), its pixels are intact
                    const int ty = tile / j.g.tiles_x, tx = tile - ty * j.g.tiles_x;
                    lds_hist_zero(lds);
                    hist_rows<kThreads>(lds, src + (long long)ty * j.g.tile_h * j.p.src_step + (long long)tx * j.g.tile_w, j.p.src_step, j.g.tile_w, j.g.tile_h);
                    __syncthreads();
                    const uint32_t c = lds_hist_bin(lds, t);
                    __syncthreads();
                    lv = tile_lut_value(c, j.g, sh.s_wave);
                }
                __hip_atomic_store(j.luts_fix + tile_id * 256 + t, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // (b) the cells without a stamp
            for (int cell = 0; cell < cells; ++cell) {
                if (ld_agent(fl + cell) == epoch) continue;          // same address for every lane: uniform branch
                const CellPlace cp(cell, j.g, j.cg);
                const int ta = max(cp.pr - 1, 0), tb = min(cp.pr, j.g.tiles_x - 1);
                const int r1 = max(cp.bnd - 1, 0), r2 = min(cp.bnd, j.g.tiles_y - 1);
                __syncthreads();                                     // the previous cell's table is no longer read
                {
                    const int wv = t >> 6, lane = t & 63;
                    const int want_tile = ((wv & 2) ? r2 : r1) * j.g.tiles_x + ((wv & 1) ? tb : ta);
                    sh.lutw[wv][lane] = __hip_atomic_load(reinterpret_cast<const uint32_t*>(j.luts_fix + ((size_t)f * tiles + want_tile) * 256) + lane,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                cell_write_table(quadf, cell_table_entry(sh));
                f32x2 xw[kInterpPx];
                cell_col_weights<FMA>(xw, cp, j.g);
                __syncthreads();
                const uint32_t rep = t & (kCellRep - 1);
                const uint8_t* srow = src + (long long)cp.y0 * j.p.src_step + cp.x0 + cp.grp * 16;
                uint8_t* drow = j.p.dst + (long long)f * j.p.dst_frame + (long long)cp.y0 * j.p.dst_step + cp.x0 + cp.grp * 16;
                for (int row = cp.phase; row < cp.my_rows; row += cp.phases) {
                    const u32x4 qv = *reinterpret_cast<const u32x4_u*>(srow + (long long)row * j.p.src_step);
                    const float tyf = tile_coord<FMA>(cp.y0 + row, j.g.inv_th);
                    const float ya = __fsub_rn(tyf, (float)(cp.bnd - 1)), ya1 = __fsub_rn(1.0f, ya);
                    *reinterpret_cast<u32x4_u*>(drow + (long long)row * j.p.dst_step) = cell_vec16<FMA>(quadf, qv, rep, xw, ya, ya1);
                }
                if (j.uv.bytes > 0)
                    uv_flat(j.uv.src ? j.uv.src + (long long)f * j.uv.src_frame : nullptr, j.uv.dst + (long long)f * j.uv.dst_frame, j.uv.bytes, j.uv.mode, cell, cells);
                if (t == 0) __hip_atomic_fetch_add(stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
    // last one out: dispensers, status word and epoch are ready for the next launch
    __syncthreads();
    if (t == 0) {
        const uint32_t arrived = __hip_atomic_fetch_add(j.ctl + kCellFin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            if (status != 0) {
                const uint32_t repaired = __hip_atomic_fetch_add(stats + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                st_agent(stats + 2, status);
                if (j.host_repaired) __hip_atomic_store(j.host_repaired, repaired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            st_agent(j.ctl + kCellStatus, 0u);
            for (int w = 0; w < 2 * kCellQueues; ++w) st_agent(j.ctl + kCellWork + w, 0u);
            st_agent(j.ctl + kCellSeq, seq + 1u);
            st_agent(j.ctl + kCellFin, 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stage-1 measurement kernel (R5.5): all of a cell's per-workgroup work, the tile LUTs taken from a preceding tile_hist_kernel
// launch; option "clahe_single_read" = 2.  grid = (cells, frames).  variant: 1 no histogram, 2 no blend (pixels copied), 4 no partial
// store, 8 XCD-aware cell order, 16 no stores, 32 no loads.
// ---------------------------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(kThreads, 4) void clahe_cell_kernel(PlaneBatch p, ClaheGeom g, CellGeom cg, const uint8_t* __restrict__ luts,
                                                                uint32_t* __restrict__ cellhist, UVJob uv)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * kCopies];
    f32x4* const quadf = reinterpret_cast<f32x4*>(lds);
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const int f = (int)gridDim.y - 1 - (int)blockIdx.y;
    int cell = blockIdx.x;
    if ((cg.variant & 8) && cg.cells_y % 8 == 0) {                  // slot i -> XCD i % 8 -> cell row 8 * (i / (8 * cells_x)) + i % 8
        const int i = blockIdx.x, per = 8 * cg.cells_x;
        cell = (8 * (i / per) + (i & 7)) * cg.cells_x + ((i >> 3) % cg.cells_x);
    }
    const CellPlace cp(cell, g, cg);
    const int span = (cp.nrows - 1) * (int)p.src_step + cp.cw, dspan = (cp.nrows - 1) * (int)p.dst_step + cp.cw;
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.src + (long long)f * p.src_frame + (long long)cp.y0 * p.src_step + cp.x0), 0, span, 0x00020000);
    const auto drsrc = __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)f * p.dst_frame + (long long)cp.y0 * p.dst_step + cp.x0, 0, dspan, 0x00020000);
    const int idle = cp.phase < cg.phases ? 0 : 0x40000000;
    const int soff = cp.phase * (int)p.src_step + cp.grp * 16 + idle, doff = cp.phase * (int)p.dst_step + cp.grp * 16 + idle;
    const int sstride = cg.phases * (int)p.src_step, dstride = cg.phases * (int)p.dst_step;
    u32x4 q[kCellVPT];
#pragma unroll
    for (int k = 0; k < kCellVPT; ++k) q[k] = (cg.variant & 32) ? u32x4{0u, 0u, 0u, 0u} : __builtin_amdgcn_raw_buffer_load_b128(srsrc, soff, k * sstride, 0);
    for (int i = t; i < 256 * kCopies; i += kThreads) lds[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCellVPT; ++k) {
        if (cp.phase + k * cg.phases < cp.my_rows && !(cg.variant & 1)) hist_add_vec(lds, q[k], copy);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (!(cg.variant & 4)) cellhist[((size_t)f * gridDim.x + cell) * 256 + t] = lds_hist_bin(lds, launder(t));
    __syncthreads();
    {
        const int ta = max(cp.pr - 1, 0), tb = min(cp.pr, g.tiles_x - 1);
        const int r1 = max(cp.bnd - 1, 0), r2 = min(cp.bnd, g.tiles_y - 1);
        const uint8_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * 256 + t;
        const f32x4 e = {(float)lf[(size_t)(r1 * g.tiles_x + ta) * 256], (float)lf[(size_t)(r2 * g.tiles_x + ta) * 256],
                         (float)lf[(size_t)(r1 * g.tiles_x + tb) * 256], (float)lf[(size_t)(r2 * g.tiles_x + tb) * 256]};
        const int tl = launder(t);
#pragma unroll
        for (int r = 0; r < kCellRep; ++r) quadf[(tl << 3) + ((r + tl) & (kCellRep - 1))] = e;
    }
    f32x2 xw[kInterpPx];
    cell_col_weights<FMA>(xw, cp, g);
    __syncthreads();
    const uint32_t rep = t & (kCellRep - 1);
#pragma unroll
    for (int k = 0; k < kCellVPT; ++k) {
        if (cp.phase + k * cg.phases < cp.my_rows) {
            launder(q[k]);
            const float tyf = tile_coord<FMA>(cp.y0 + cp.phase + k * cg.phases, g.inv_th);
            const float ya = __fsub_rn(tyf, (float)(cp.bnd - 1)), ya1 = __fsub_rn(1.0f, ya);
            if (cg.variant & 16) { if (q[k].x == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(q[k], drsrc, doff, k * dstride, 0); }
            else if (cg.variant & 2) __builtin_amdgcn_raw_buffer_store_b128(q[k], drsrc, doff, k * dstride, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(cell_vec16<FMA>(quadf, q[k], rep, xw, ya, ya1), drsrc, doff, k * dstride, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (uv.bytes > 0)
        uv_flat(uv.src ? uv.src + (long long)f * uv.src_frame : nullptr, uv.dst + (long long)f * uv.dst_frame, uv.bytes, uv.mode, blockIdx.x, gridDim.x);
}

}  // namespace mi
