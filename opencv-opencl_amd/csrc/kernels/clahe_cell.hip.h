// clahe_cell.hip.h -- KC single-read CLAHE by CELLS (docs/experiments.md R5.4): histogram and interpolation of a tile quadrant from
// the same registers.  Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
//
// A CELL is the intersection of one tile with one interpolation band and one horizontal tile pair -- a quadrant of a tile.  Every
// pixel of a cell (a) counts into ONE tile histogram and (b) is blended from the SAME four tile LUTs {r1, r2} x {ta, tb}
// (clahe.cpp CLAHE_Interpolation_Body: ty1 / tx1 are constant over the cell), so a 256-thread workgroup keeps the cell's pixels in
// registers (8 x 16 B per lane) from the histogram to the blend, and its 32 KiB of LDS hold first the bank-replicated histogram and
// then the cell's one table as f32x4[256][8] -- eight replicas, lane uses copy (lane & 7): the eight lanes a ds_read_b128 serves per
// cycle hit eight different bank quads whatever the pixel values are.  Reference call being replaced: clahevideo.cpp:195,
// clahe1frame.cpp:92-95 (CLAHE::apply); arithmetic: SURVEY.md App. A.2, identical to clahe.hip.h (same helpers).
//
// Only "regular" geometries take this path (the host checks, clahe_cell_geometry): no padding, tile_w a multiple of 32, and the
// band / pair boundaries -- found with the reference's own float expressions -- at the same offset in every tile, the column one on a
// multiple of 16.  Everything else runs the two-pass kernels of clahe.hip.h.
#pragma once
#include "clahe.hip.h"
#include "equalize_fused.hip.h"      // launder(), the hand-off helpers

namespace mi {

constexpr int kCellVPT = 8;            // 16-byte vectors (rows) a lane keeps in registers
constexpr int kCellRep = 8;            // replicas of the cell's f32x4 table: 256 x 8 x 16 B = 32 KiB, the histogram's LDS

struct CellGeom {
    int cells_x, cells_y;              // 2 * tiles_x, 2 * tiles_y
    int groups, phases;                // 16-pixel column groups of a cell (tile_w / 32); row phases = 256 / groups
    int ysplit;                        // rows of a tile that belong to the band above (ty1 = ty - 1): rows [0, ysplit); the rest: ty1 = ty
    int variant;                       // stage-1 ablations (option "clahe_cell_variant"): 1 no histogram, 2 no blend (pixels copied), 4 no partial store
};

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

// Stage 1 (R5.4): all of a cell's work, the tile LUTs taken from a preceding tile_hist_kernel launch.  grid = (cells, frames).
// `cellhist` receives the cell's 256-bin partial (what the hand-off of stage 2 publishes).
template <bool FMA>
__global__ __launch_bounds__(kThreads, 4) void clahe_cell_kernel(PlaneBatch p, ClaheGeom g, CellGeom cg, const uint8_t* __restrict__ luts,
                                                                uint32_t* __restrict__ cellhist, UVJob uv)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * kCopies];
    f32x4* const quadf = reinterpret_cast<f32x4*>(lds);
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const int f = (int)gridDim.y - 1 - (int)blockIdx.y;           // last frame first: the tail of what the LUT pass streamed is still cached
    // XCD-aware order (speed only; variant bit 8): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Horizontal
    // neighbours share the 128-byte lines their common edge cuts (cell rows are 240 bytes at 4K 8x8), so all cells of a cell ROW go to
    // one XCD: slot i -> XCD i % 8 -> cell row 8 * (i / (8 * cells_x)) + i % 8, column (i / 8) % cells_x.
    int cell = blockIdx.x;
    if ((cg.variant & 8) && cg.cells_y % 8 == 0) {
        const int i = blockIdx.x, per = 8 * cg.cells_x;
        cell = (8 * (i / per) + (i & 7)) * cg.cells_x + ((i >> 3) % cg.cells_x);
    }
    const int cy = cell / cg.cells_x, cx = cell - cy * cg.cells_x;
    const int tx = cx >> 1, hx = cx & 1, ty = cy >> 1, hy = cy & 1;
    const int cw = g.tile_w >> 1;
    const int x0 = tx * g.tile_w + hx * cw;
    const int y0 = ty * g.tile_h + hy * cg.ysplit;
    const int nrows = hy ? g.tile_h - cg.ysplit : cg.ysplit;
    const int grp = t % cg.groups, phase = t / cg.groups;
    const int my_rows = phase < cg.phases ? nrows : 0;             // rows this lane may touch: phase, phase + phases, ... < my_rows
    // ---- 1. the cell -> registers.  Buffer descriptors over exactly the cell's span: rows beyond the cell and idle lanes fall outside
    // the range (loads return 0, stores are dropped), so neither needs a predicate.
    const int span = (nrows - 1) * (int)p.src_step + cw;
    const int dspan = (nrows - 1) * (int)p.dst_step + cw;
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.src + (long long)f * p.src_frame + (long long)y0 * p.src_step + x0), 0, span, 0x00020000);
    const auto drsrc = __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)f * p.dst_frame + (long long)y0 * p.dst_step + x0, 0, dspan, 0x00020000);
    const int idle = phase < cg.phases ? 0 : 0x40000000;          // an offset no cell reaches
    const int soff = phase * (int)p.src_step + grp * 16 + idle;
    const int doff = phase * (int)p.dst_step + grp * 16 + idle;
    const int sstride = cg.phases * (int)p.src_step, dstride = cg.phases * (int)p.dst_step;
    u32x4 q[kCellVPT];
#pragma unroll
    for (int k = 0; k < kCellVPT; ++k) q[k] = (cg.variant & 32) ? u32x4{0u, 0u, 0u, 0u} : __builtin_amdgcn_raw_buffer_load_b128(srsrc, soff, k * sstride, 0);
    for (int i = t; i < 256 * kCopies; i += kThreads) lds[i] = 0;
    __syncthreads();
    // ---- 2. the cell's histogram
#pragma unroll
    for (int k = 0; k < kCellVPT; ++k) {
        if (phase + k * cg.phases < my_rows && !(cg.variant & 1)) hist_add_vec(lds, q[k], copy);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (!(cg.variant & 4)) cellhist[((size_t)f * gridDim.x + cell) * 256 + t] = lds_hist_bin(lds, launder(t));
    __syncthreads();                                              // the histogram has been read: its LDS becomes the table
    // ---- 3. the cell's ONE table: {a, c, b, d} = LUT[r1][ta], LUT[r2][ta], LUT[r1][tb], LUT[r2][tb] of value t, eight replicas
    const int pr = tx + hx, bnd = ty + hy;                        // unclamped tx1 = pr - 1, ty1 = bnd - 1
    {
        const int ta = max(pr - 1, 0), tb = min(pr, g.tiles_x - 1);
        const int r1 = max(bnd - 1, 0), r2 = min(bnd, g.tiles_y - 1);
        const uint8_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * 256 + t;
        const f32x4 e = {(float)lf[(size_t)(r1 * g.tiles_x + ta) * 256], (float)lf[(size_t)(r2 * g.tiles_x + ta) * 256],
                         (float)lf[(size_t)(r1 * g.tiles_x + tb) * 256], (float)lf[(size_t)(r2 * g.tiles_x + tb) * 256]};
        const int tl = launder(t);
#pragma unroll
        for (int r = 0; r < kCellRep; ++r) quadf[(tl << 3) + ((r + tl) & (kCellRep - 1))] = e;
    }
    // column weights of this lane's 16 columns (all in pair pr by construction of the cell)
    f32x2 xw[kInterpPx];
#pragma unroll
    for (int j = 0; j < kInterpPx; ++j) {
        const float txf = tile_coord<FMA>(x0 + grp * 16 + j, g.inv_tw);
        const float xa = __fsub_rn(txf, (float)(pr - 1));
        xw[j].x = __fsub_rn(1.0f, xa); xw[j].y = xa;
    }
    __syncthreads();
    // ---- 4. blend from the registers, stream out
    const uint32_t rep = t & (kCellRep - 1);
#pragma unroll
    for (int k = 0; k < kCellVPT; ++k) {
        if (phase + k * cg.phases < my_rows) {
            launder(q[k]);
            const float tyf = tile_coord<FMA>(y0 + phase + k * cg.phases, g.inv_th);
            const float ya = __fsub_rn(tyf, (float)(bnd - 1)), ya1 = __fsub_rn(1.0f, ya);
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
