// kernels_matrix.hip -- the matrix path: scans of ANY order up to RF_MAX_ORDER = 32 in their direct form, every stage of
// the tiled algorithm a small dense f32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain).
//
// RecFilter::add_filter takes any order (lib/recfilter.cpp:260-343) and the reference's own app sweeps one scan of order
// 1, 3, .. 29 (apps/audio/audio_filter_high_order.cpp:14,38-42).  The fused kernels keep an order <= 3 recurrence in
// registers; above that the recurrence itself stops being the cheap part and the tiling algebra of lib/split.cpp turns into
// what north_star reserves the matrix cores for.  Per scan, tile of T = 32 NB samples, sub-blocks of 32 samples:
//
//   pass 1   tail extraction (extract_tails_from_each_scan, lib/split.cpp:256-499): the k-sample tail of the tile-local scan
//            is LINEAR in the tile -- tails[k x units] = H[k x T] . tile[T x units]: one GEMM, no recurrence runs.
//   chain    cross-tile carry recurrence (create_complete_tail_term, lib/split.cpp:743-867): c_t = l_t + A c_(t-1), A = k x k:
//            a chain of GEMMs [k x k] . [k x 32 columns], blocked over chunks of 16 tiles (levels: chunk exits are a shorter
//            sequence with the transfer matrix A^16, and so on), then propagated down with the tabulated powers of A.
//   pass 2   final pass (add_residuals_to_final_result, lib/split.cpp:1647-1780): the tile is recomputed sub-block by
//            sub-block, y_b = G x_b + R y_(b-1), G = 32 x 32 impulse-response (Toeplitz, triangular) matrix of the scan,
//            R = 32 x 32 effect of the previous sub-block's outputs (k non-zero columns); the first sub-block takes the
//            neighbouring tile's completed tail in y_(-1)'s place.
//
// Lanes are UNITS (a line's tile): lane l = 32 h + u holds column u of every 32 x 32 operand, and the 32-sample direction of
// a sub-block is the K index of the MFMA.  The accumulator layout of a 32 x 32 result puts row (t>>2)*8 + 4h + (t&3) in
// register t of lane half h -- so the K index is ASSIGNED in that order (step t of the k loop <-> that row pair): then a
// result is the next product's B operand with no lane movement (y_(b-1) in pass 2, c_(t-1) in the chain), and the A
// operands (the constant matrices) are stored pre-permuted as fragments frag[t][lane] = Mat[lane & 31][row(t, lane >> 5)].
//
// The image goes through LDS both ways (coalesced 16-byte accesses; a 128-unit x T-sample block per workgroup), so one
// kernel body serves scans along x (lane = line, or lane = tile for 1-D signals) and along y / z (lane = column).
// A clamped border is the zero-border operator plus a rank-one term in the scan's first sample (what the clamped prologue of
// lib/recfilter.cpp:330-336 adds is linear in x_0): dG / dH, applied by the lanes whose tile is where the scan enters the image.
#include <mutex>
#include <set>
#include <utility>

#include "kernels_matrix.h"

namespace rf {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kMxThreads = 64 * kMxWaves;
constexpr int kMxPitchY = kMxUnits + 8;      // LDS pitch of a [sample][unit] block: the two lane halves (rows 4 apart) hit disjoint banks

__device__ __forceinline__ int mx_row(int t, int h) { return ((t >> 2) << 3) + (h << 2) + (t & 3); }

__device__ __forceinline__ floatx16 mx_zero() {
    floatx16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}

// geometry of the workgroup's block: rows x cols of the LDS image, where it lies in the plane
struct MxBlock {
    int64_t gbase, gpitch;     // element offset of (row 0, col 0), elements between rows
    int rows, cols;            // extent of the LDS image
    int rows_valid, cols_valid;
    int pitch;                 // LDS pitch in floats
    int tile;                  // MX_XL / MX_Y: the tile of the block
    int64_t first;             // first unit (MX_X1), line (MX_XL) or line of column 0 (MX_Y)
    int64_t s0;                // MX_XL / MX_Y: the sample the block's tile starts at (tile * T - off: may be negative)
    int lo, hi;                // MX_XL / MX_Y: the samples [lo, hi) of the block's tile exist
    int last_hi;               // how many samples of the LAST tile exist
};

// Where element (row, chunk column c) of the block lies in the plane and whether it exists.  Tiles need not divide the extent:
// tile t covers the samples [t T - off, (t + 1) T - off), and whatever falls outside [0, N) loads as zeros and is never stored.
// The padding is always on the side where the scan LEAVES the image (off = 0 for a causal scan, M T - N for an anticausal one):
// zeros behind the last sample change nothing for the samples in front of them, and the tile where the scan enters is whole.
template <bool XM>
__device__ __forceinline__ bool mx_element(const MxPassArgs &a, const MxBlock &b, int row, int c, int64_t &offset) {
    // samples [lo, hi) of a tile exist: everything but the first tile starts at 0, everything but the last one ends at T
    if constexpr (XM) {
        if (a.mode == MX_X1) {
            const int64_t U = b.first + row;
            int64_t tile = U;
            offset = b.gbase + (int64_t)(row * a.T + c);                       // a 1-D signal: its tiles follow one another in memory
            if (a.lines != 1) {
                const int64_t line = U / a.M;
                tile = U - line * a.M;
                offset = line * a.N + tile * a.T - a.off + c;
            }
            const int lo = tile == 0 ? (int)a.off : 0, hi = tile == a.M - 1 ? b.last_hi : a.T;
            return row < b.rows_valid && c >= lo && c < hi;
        }
        offset = b.gbase + (int64_t)row * b.gpitch + c;
        return row < b.rows_valid && c >= b.lo && c < b.hi;
    } else {
        offset = b.gbase + (int64_t)row * b.gpitch + c;
        return c < b.cols_valid && row >= b.lo && row < b.hi;
    }
}

template <bool XM>
__device__ __forceinline__ MxBlock mx_block(const MxPassArgs &a) {
    MxBlock b;
    b.last_hi = (int)(a.N + a.off - (int64_t)(a.M - 1) * a.T);
    b.lo = 0; b.hi = a.T;
    if constexpr (XM) {
        b.rows = kMxUnits; b.cols = a.T; b.pitch = a.T + 4; b.cols_valid = a.T;
        if (a.mode == MX_X1) {
            const int64_t U0 = (int64_t)blockIdx.x * kMxUnits;
            b.gbase = U0 * a.T - a.off; b.gpitch = a.T; b.tile = 0; b.first = U0; b.s0 = 0;          // (per row: mx_element)
            const int64_t left = a.units - U0;
            b.rows_valid = left < kMxUnits ? (int)left : kMxUnits;
        } else {
            const int64_t L0 = (int64_t)blockIdx.x * kMxUnits;
            b.tile = (int)blockIdx.y;
            b.s0 = (int64_t)b.tile * a.T - a.off;
            b.lo = b.tile == 0 ? (int)a.off : 0; b.hi = b.tile == a.M - 1 ? b.last_hi : a.T;
            b.gbase = L0 * a.N + b.s0; b.gpitch = a.N; b.first = L0;
            const int64_t left = a.lines - L0;
            b.rows_valid = left < kMxUnits ? (int)left : kMxUnits;
        }
    } else {
        const int64_t c0 = (int64_t)blockIdx.x * kMxUnits, outer = blockIdx.z;
        b.tile = (int)blockIdx.y;
        b.rows = a.T; b.cols = kMxUnits; b.pitch = kMxPitchY; b.rows_valid = a.T;
        b.s0 = (int64_t)b.tile * a.T - a.off;
        b.lo = b.tile == 0 ? (int)a.off : 0; b.hi = b.tile == a.M - 1 ? b.last_hi : a.T;
        b.gbase = (outer * a.N + b.s0) * a.inner + c0; b.gpitch = a.inner;
        b.first = outer * a.inner + c0;
        const int64_t left = a.inner - c0;
        b.cols_valid = left < kMxUnits ? (int)left : kMxUnits;
    }
    return b;
}

// Every thread moves 4 NB chunks of 16 bytes; all of them are requested before the first one is stored to LDS (a loop
// that loads and stores chunk by chunk is a chain of 4 NB dependent round trips to HBM per workgroup).
template <bool XM, int NB>
__device__ __forceinline__ void mx_load_block(const float *__restrict__ src, float *lds, const MxPassArgs &a, const MxBlock &b) {
    constexpr int W4 = XM ? 8 * NB : kMxUnits / 4;        // 16-byte chunks per row of the LDS image
    float4 v[4 * NB];
    if (!a.ragged) {          // tiles that divide the extent (wave-uniform): rows and columns of the block exist or not as a whole
#pragma unroll
        for (int it = 0; it < 4 * NB; it++) {
            const int f = (int)threadIdx.x + kMxThreads * it;
            const int row = f / W4, c = (f - row * W4) << 2;
            v[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (row < b.rows_valid && c < b.cols_valid) v[it] = *reinterpret_cast<const float4 *>(src + b.gbase + (int64_t)row * b.gpitch + c);
        }
    } else {
#pragma unroll
        for (int it = 0; it < 4 * NB; it++) {
            const int f = (int)threadIdx.x + kMxThreads * it;
            const int row = f / W4, c = (f - row * W4) << 2;
            // (branch-free: an element that does not exist reads the plane's first chunk and is replaced by zeros)
            int64_t o;
            const bool ok = mx_element<XM>(a, b, row, c, o);
            const float4 got = *reinterpret_cast<const float4 *>(src + (ok ? o : 0));
            v[it] = ok ? got : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
#pragma unroll
    for (int it = 0; it < 4 * NB; it++) {
        const int f = (int)threadIdx.x + kMxThreads * it;
        const int row = f / W4, c = (f - row * W4) << 2;
        *reinterpret_cast<float4 *>(lds + row * b.pitch + c) = v[it];
    }
}

template <bool XM, int NB>
__device__ __forceinline__ void mx_store_block(float *__restrict__ dst, const float *lds, const MxPassArgs &a, const MxBlock &b) {
    constexpr int W4 = XM ? 8 * NB : kMxUnits / 4;
    if (!a.ragged) {
#pragma unroll
        for (int it = 0; it < 4 * NB; it++) {
            const int f = (int)threadIdx.x + kMxThreads * it;
            const int row = f / W4, c = (f - row * W4) << 2;
            if (row < b.rows_valid && c < b.cols_valid)
                *reinterpret_cast<float4 *>(dst + b.gbase + (int64_t)row * b.gpitch + c) = *reinterpret_cast<const float4 *>(lds + row * b.pitch + c);
        }
    } else {
#pragma unroll
        for (int it = 0; it < 4 * NB; it++) {
            const int f = (int)threadIdx.x + kMxThreads * it;
            const int row = f / W4, c = (f - row * W4) << 2;
            int64_t o;
            if (mx_element<XM>(a, b, row, c, o)) *reinterpret_cast<float4 *>(dst + o) = *reinterpret_cast<const float4 *>(lds + row * b.pitch + c);
        }
    }
}

// the 32 samples of sub-block sb of this lane's unit, in K order: x[t] = sample row(t, h)
template <bool XM>
__device__ __forceinline__ void mx_read_sub(const float *lds, int pitch, int mine, int h, int sb, float (&x)[16]) {
    if constexpr (XM) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(lds + mine * pitch + 32 * sb + 8 * q + 4 * h);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++) x[t] = lds[(32 * sb + mx_row(t, h)) * pitch + mine];
    }
}

template <bool XM>
__device__ __forceinline__ void mx_write_sub(float *lds, int pitch, int mine, int h, int sb, const floatx16 &y) {
    if constexpr (XM) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4 *>(lds + mine * pitch + 32 * sb + 8 * q + 4 * h) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++) lds[(32 * sb + mx_row(t, h)) * pitch + mine] = y[t];
    }
}

// this lane's unit
struct MxLane {
    bool valid, enters;        // enters: the scan enters the slab (the image, unless it is sharded) in this tile
    bool border;               // ... and that is where it enters the IMAGE: the clamped-border term applies, no carry comes in
    int64_t line;
    int tile;
    int64_t tidx, prev_tidx;   // index of the unit's tail / of the tail it takes its carry from (scan direction)
};

template <bool XM>
__device__ __forceinline__ MxLane mx_lane(const MxPassArgs &a, const MxBlock &b, int mine) {
    MxLane l;
    int64_t line;
    if (XM && a.mode == MX_X1) {
        const int64_t U = b.first + mine;
        l.valid = mine < b.rows_valid;
        const int64_t Uc = l.valid ? U : 0;
        line = Uc / a.M;
        l.tile = (int)(Uc - line * a.M);
        l.tidx = Uc;
        l.prev_tidx = a.causal ? Uc - 1 : Uc + 1;
    } else {
        l.valid = XM ? mine < b.rows_valid : mine < b.cols_valid;
        line = l.valid ? b.first + mine : 0;
        l.tile = b.tile;
        l.tidx = (int64_t)l.tile * a.lines + line;
        l.prev_tidx = l.tidx + (a.causal ? -a.lines : a.lines);
    }
    l.enters = a.causal ? l.tile == 0 : l.tile == a.M - 1;
    l.border = l.enters && (a.causal ? a.slab_first != 0 : a.slab_last != 0);
    l.line = line;
    return l;
}

// ---- pass 1: tails[k x units] = H[k x T] . tile[T x units] ------------------------------------------------------------
// Tails are stored [unit][KP], KP = k rounded up to 8 (the rows k .. KP-1 are zeros): registers 4q .. 4q+3 of a lane are
// four consecutive rows, so every tail access of the path is a 16-byte access.
template <bool XM, int NB>
__global__ void __launch_bounds__(kMxThreads)
mx_pass1_kernel(const float *__restrict__ src, MxPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mx_lds[];
    const MxBlock blk = mx_block<XM>(a);
    mx_load_block<XM, NB>(src, mx_lds, a, blk);
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    const int mine = 32 * w + u;
    const MxLane ln = mx_lane<XM>(a, blk, mine);
    float Hf[NB][16];                                      // requested while the block is on its way
#pragma unroll
    for (int sb = 0; sb < NB; sb++)
#pragma unroll
        for (int t = 0; t < 16; t++) Hf[sb][t] = a.H[(sb * 16 + t) * 64 + lane];
    __syncthreads();
    floatx16 acc = mx_zero();
#pragma unroll
    for (int sb = 0; sb < NB; sb++) {
        float x[16];
        mx_read_sub<XM>(mx_lds, blk.pitch, mine, h, sb, x);
#pragma unroll
        for (int t = 0; t < 16; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Hf[sb][t], x[t], acc, 0, 0, 0);
    }
    if (a.clamped && ln.valid && ln.border) {
        const int m0 = a.causal ? 0 : a.T - 1;
        const float x0 = XM ? mx_lds[mine * blk.pitch + m0] : mx_lds[m0 * blk.pitch + mine];
#pragma unroll
        for (int t = 0; t < 16; t++) acc[t] = fmaf(a.dH[mx_row(t, h)], x0, acc[t]);
    }
    if (ln.valid) {
        const int KP = 8 * ((a.k + 7) >> 3);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r0 = 8 * q + 4 * h;
            if (r0 < KP) *reinterpret_cast<float4 *>(a.tails + ln.tidx * KP + r0) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

// The tile-local tails of the NEXT scan from the finished block in LDS (MxPassArgs::next): that scan's pass 1, without its read
// of the image.  XV: how the next scan sees the block -- its samples along a row of the LDS image (true) or down a column.
template <bool XV, int NB>
__device__ __forceinline__ void mx_next_tails(const MxPassArgs &a, const float *lds, int pitch, int mine, int h, int lane, bool valid,
                                              int tile, int tiles, int64_t tidx) {
    floatx16 acc = mx_zero();
#pragma unroll
    for (int sb = 0; sb < NB; sb++) {
        float x[16];
        mx_read_sub<XV>(lds, pitch, mine, h, sb, x);
        const float *Hf = a.next_H + (size_t)sb * 16 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 16; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Hf[t * 64], x[t], acc, 0, 0, 0);
    }
    const bool enters = (a.next_causal ? tile == 0 : tile == tiles - 1) && (a.next_causal ? a.slab_first != 0 : a.slab_last != 0);
    if (a.clamped && valid && enters) {
        const int m0 = a.next_causal ? 0 : 32 * NB - 1;
        const float x0 = XV ? lds[mine * pitch + m0] : lds[m0 * pitch + mine];
#pragma unroll
        for (int t = 0; t < 16; t++) acc[t] = fmaf(a.next_dH[mx_row(t, h)], x0, acc[t]);
    }
    if (valid) {
        const int KP = 8 * ((a.next_k + 7) >> 3);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r0 = 8 * q + 4 * h;
            if (r0 < KP) *reinterpret_cast<float4 *>(a.next_tails + tidx * KP + r0) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

// ---- pass 2: y_b = G x_b + R y_(b-1) ----------------------------------------------------------------------------------
template <bool XM, int NB>
__global__ void __launch_bounds__(kMxThreads)
mx_pass2_kernel(const float *__restrict__ src, float *__restrict__ dst, MxPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mx_lds[];
    const MxBlock blk = mx_block<XM>(a);
    mx_load_block<XM, NB>(src, mx_lds, a, blk);
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    const int mine = 32 * w + u;
    const MxLane ln = mx_lane<XM>(a, blk, mine);
    float Gf[16], Rf[16];
#pragma unroll
    for (int t = 0; t < 16; t++) { Gf[t] = a.G[t * 64 + lane]; Rf[t] = a.R[t * 64 + lane]; }
    // K steps of R that are not all zero: the k most recent rows of the previous sub-block (wave-uniform)
    const int nl = 4 * ((a.k + 7) >> 3), KP = 2 * nl, t_lo = a.causal ? 16 - nl : 0, t_hi = t_lo + nl;
    // the completed tail of the neighbouring tile, laid out as the rows of a sub-block that precedes the tile:
    // row i of it is tail 31 - i (causal: the most recent output is the last row) or tail i (anticausal)
    floatx16 prev = mx_zero();
    if (ln.valid && (!ln.enters || (a.incoming != nullptr && !ln.border))) {
        // (the first tile of a slab that is not the image's: what the slabs before it hand over)
        const float *tp = ln.enters ? a.incoming + ln.line * KP : a.tails + ln.prev_tidx * KP;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r_lo = a.causal ? 28 - 8 * q - 4 * h : 8 * q + 4 * h;
            if (r_lo < KP) {
                const float4 v = *reinterpret_cast<const float4 *>(tp + r_lo);
                if (a.causal) { prev[4 * q] = v.w; prev[4 * q + 1] = v.z; prev[4 * q + 2] = v.y; prev[4 * q + 3] = v.x; }
                else          { prev[4 * q] = v.x; prev[4 * q + 1] = v.y; prev[4 * q + 2] = v.z; prev[4 * q + 3] = v.w; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        const int sb = a.causal ? bi : NB - 1 - bi;
        float x[16];
        mx_read_sub<XM>(mx_lds, blk.pitch, mine, h, sb, x);
        floatx16 c = mx_zero();
        if (bi == 0 && a.clamped && ln.valid && ln.border) {
            const int m0 = a.causal ? 0 : a.T - 1;
            const float x0 = XM ? mx_lds[mine * blk.pitch + m0] : mx_lds[m0 * blk.pitch + mine];
#pragma unroll
            for (int t = 0; t < 16; t++) c[t] = a.dG[mx_row(t, h)] * x0;
        }
#pragma unroll
        for (int t = 0; t < 16; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Gf[t], x[t], c, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (t >= t_lo && t < t_hi) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Rf[t], prev[t], c, 0, 0, 0);
        prev = c;
        mx_write_sub<XM>(mx_lds, blk.pitch, mine, h, sb, c);
    }
    __syncthreads();
    mx_store_block<XM, NB>(dst, mx_lds, a, blk);
    if (a.next == 1) mx_next_tails<XM, NB>(a, mx_lds, blk.pitch, mine, h, lane, ln.valid, ln.tile, a.M, ln.tidx);
    if constexpr (XM) {
        if (a.next == 2) {
            // the block seen by the y scans: lane = column mine of x tile blk.tile, samples = the block's 128 rows = y tile blockIdx.x
            const int ytile = (int)blockIdx.x, MY = (int)(a.lines / a.T);
            const int64_t column = (int64_t)blk.tile * a.T + mine;
            mx_next_tails<false, NB>(a, mx_lds, blk.pitch, mine, h, lane, true, ytile, MY, (int64_t)ytile * a.N + column);
        }
    }
}

// ---- the carry chain: x_j = s_j + A x_(j-1) over the steps of a chunk, 32 columns per wave ----------------------------
// Elements are [KP] rows, rows fastest: NLQ = KP / 8 sixteen-byte pieces per lane and element.
struct MxCol {
    bool valid;
    int64_t off, eoff, eprev;      // element indices: the column's first element, its exit, the exit of the chunk before it
    int len, chunk;
};

__device__ __forceinline__ MxCol mx_col(const MxChainArgs &a, int64_t c) {
    MxCol col;
    col.valid = c < a.ncols;
    const int64_t cc = col.valid ? c : 0;
    const int64_t c_hi = cc / a.cdiv, c_lo = cc - c_hi * a.cdiv;
    const int64_t chunk = a.chunk_is_lo ? c_lo : c_hi;
    col.chunk = (int)chunk;
    const int64_t left = a.Mtot - chunk * a.C;
    col.len = left < a.C ? (int)left : a.C;
    col.off = a.base + c_hi * a.s_hi + c_lo * a.s_lo;
    col.eoff = c_hi * a.e_hi + c_lo * a.e_lo;
    col.eprev = a.enter_fixed ? c_lo * a.e_lo : col.eoff - (a.chunk_is_lo ? a.e_lo : a.e_hi);
    return col;
}

constexpr int kMxAhead = 4;        // elements of the chain requested ahead of the step that consumes them

template <int NLQ>
__global__ void __launch_bounds__(kMxThreads)
mx_chain_kernel(MxChainArgs a) {
    constexpr int NL = 4 * NLQ, KP = 8 * NLQ;
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    const MxCol col = mx_col(a, ((int64_t)blockIdx.x * kMxWaves + w) * 32 + u);
    float Af[NL];
#pragma unroll
    for (int t = 0; t < NL; t++) Af[t] = a.A[t * 64 + lane];
    auto elem = [&](int j) { return a.seq + (col.off + (int64_t)j * a.s_j) * KP + 4 * h; };
    float4 ring[kMxAhead][NLQ];
    auto request = [&](int slot, int j) {
#pragma unroll
        for (int q = 0; q < NLQ; q++) {
            ring[slot][q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (col.valid && j < col.len) ring[slot][q] = *reinterpret_cast<const float4 *>(elem(j) + 8 * q);
        }
    };
#pragma unroll
    for (int d = 0; d < kMxAhead; d++) request(d, d);
    floatx16 x = mx_zero();
    for (int j0 = 0; j0 < a.C; j0 += kMxAhead) {
#pragma unroll
        for (int d = 0; d < kMxAhead; d++) {
            const int j = j0 + d;
            floatx16 c = mx_zero();
#pragma unroll
            for (int q = 0; q < NLQ; q++) { c[4 * q] = ring[d][q].x; c[4 * q + 1] = ring[d][q].y; c[4 * q + 2] = ring[d][q].z; c[4 * q + 3] = ring[d][q].w; }
            request(d, j + kMxAhead);                       // before this step's arithmetic and stores
#pragma unroll
            for (int t = 0; t < NL; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[t], x[t], c, 0, 0, 0);
            if (col.valid && j < col.len) {
#pragma unroll
                for (int q = 0; q < NLQ; q++) *reinterpret_cast<float4 *>(elem(j) + 8 * q) = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
                x = c;
            }
        }
    }
    if (a.exits != nullptr && col.valid) {
#pragma unroll
        for (int q = 0; q < NLQ; q++)
            *reinterpret_cast<float4 *>(a.exits + col.eoff * KP + 8 * q + 4 * h) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    }
}

// ---- propagation: element j of a chunk += (A^(j+1)) . (completed exit of the chunk before it) --------------------------
template <int NLQ>
__global__ void __launch_bounds__(kMxThreads)
mx_apply_kernel(MxChainArgs a) {
    constexpr int NL = 4 * NLQ, KP = 8 * NLQ;
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    const MxCol col = mx_col(a, ((int64_t)blockIdx.x * kMxWaves + w) * 32 + u);
    const int j = (int)blockIdx.y;
    const bool on = col.valid && (col.chunk >= 1 || a.enter_fixed != 0) && j < col.len;
    floatx16 c = mx_zero(), e = mx_zero();
    float *mine = a.seq + (col.off + (int64_t)j * a.s_j) * KP + 4 * h;
    if (on) {
#pragma unroll
        for (int q = 0; q < NLQ; q++) {
            const float4 ev = *reinterpret_cast<const float4 *>(a.exits + col.eprev * KP + 8 * q + 4 * h);
            const float4 cv = *reinterpret_cast<const float4 *>(mine + 8 * q);
            e[4 * q] = ev.x; e[4 * q + 1] = ev.y; e[4 * q + 2] = ev.z; e[4 * q + 3] = ev.w;
            c[4 * q] = cv.x; c[4 * q + 1] = cv.y; c[4 * q + 2] = cv.z; c[4 * q + 3] = cv.w;
        }
    }
    const float *Pf = a.P + (size_t)j * 16 * 64 + lane;
#pragma unroll
    for (int t = 0; t < NL; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Pf[t * 64], e[t], c, 0, 0, 0);
    if (on) {
#pragma unroll
        for (int q = 0; q < NLQ; q++) *reinterpret_cast<float4 *>(mine + 8 * q) = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    }
}

size_t mx_lds_bytes(const MxPassArgs &a) {
    return a.mode == MX_Y ? (size_t)a.T * kMxPitchY * sizeof(float) : (size_t)kMxUnits * (a.T + 4) * sizeof(float);
}

dim3 mx_grid(const MxPassArgs &a) {
    if (a.mode == MX_X1) return dim3((unsigned)((a.units + kMxUnits - 1) / kMxUnits));
    if (a.mode == MX_XL) return dim3((unsigned)((a.lines + kMxUnits - 1) / kMxUnits), (unsigned)a.M);
    return dim3((unsigned)((a.inner + kMxUnits - 1) / kMxUnits), (unsigned)a.M, (unsigned)(a.lines / a.inner));
}

int mx_check(const MxPassArgs &a) {
    if (a.T != 32 * a.NB || a.NB < 1 || a.NB > kMxMaxNB || a.k < 1 || a.k > 32 || a.M < 1 || a.off < 0 || a.off >= a.T ||
        (int64_t)a.M * a.T < a.N + a.off || (int64_t)(a.M - 1) * a.T >= a.N + a.off || (a.mode != MX_Y && (a.N % 4 != 0 || a.off % 4 != 0))) {
        set_error("matrix path: bad tile geometry (T %d, NB %d, M %d, k %d)", a.T, a.NB, a.M, a.k);
        return RF_ERR_INVALID_ARG;
    }
    const dim3 g = mx_grid(a);
    if (g.y > 65535u || g.z > 65535u || (a.mode == MX_Y && (a.inner % 4 != 0 || a.lines % a.inner != 0))) {
        set_error("matrix path: extents out of range");
        return RF_ERR_UNSUPPORTED;
    }
    return RF_OK;
}

// more than 64 KiB of dynamic LDS: opt in, once per kernel and device
template <typename K>
int mx_allow_lds(K kern, size_t lds) {
    if (lds <= 64 * 1024) return RF_OK;
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> opted;
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    const std::pair<const void *, int> key(reinterpret_cast<const void *>(kern), dev);
    std::lock_guard<std::mutex> lock(mu);
    if (opted.count(key)) return RF_OK;
    RF_HIP_CHECK(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    opted.insert(key);
    return RF_OK;
}

}  // namespace

#define RF_MX_PASS(KERNEL, XMODE, ...)                                                                                        \
    switch (a.NB) {                                                                                                          \
        case 1: if (int rc = mx_allow_lds(KERNEL<XMODE, 1>, lds)) return rc; hipLaunchKernelGGL((KERNEL<XMODE, 1>), mx_grid(a), dim3(kMxThreads), lds, stream, __VA_ARGS__); break; \
        case 2: if (int rc = mx_allow_lds(KERNEL<XMODE, 2>, lds)) return rc; hipLaunchKernelGGL((KERNEL<XMODE, 2>), mx_grid(a), dim3(kMxThreads), lds, stream, __VA_ARGS__); break; \
        case 3: if (int rc = mx_allow_lds(KERNEL<XMODE, 3>, lds)) return rc; hipLaunchKernelGGL((KERNEL<XMODE, 3>), mx_grid(a), dim3(kMxThreads), lds, stream, __VA_ARGS__); break; \
        default: if (int rc = mx_allow_lds(KERNEL<XMODE, 4>, lds)) return rc; hipLaunchKernelGGL((KERNEL<XMODE, 4>), mx_grid(a), dim3(kMxThreads), lds, stream, __VA_ARGS__); break; \
    }

int launch_mx_pass1(const float *src, const MxPassArgs &a, hipStream_t stream) {
    if (int rc = mx_check(a)) return rc;
    const size_t lds = mx_lds_bytes(a);
    if (a.mode == MX_Y) { RF_MX_PASS(mx_pass1_kernel, false, src, a) }
    else { RF_MX_PASS(mx_pass1_kernel, true, src, a) }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

int launch_mx_pass2(const float *src, float *dst, const MxPassArgs &a, hipStream_t stream) {
    if (int rc = mx_check(a)) return rc;
    const size_t lds = mx_lds_bytes(a);
    if (a.mode == MX_Y) { RF_MX_PASS(mx_pass2_kernel, false, src, dst, a) }
    else { RF_MX_PASS(mx_pass2_kernel, true, src, dst, a) }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
#undef RF_MX_PASS

int launch_mx_chain(const MxChainArgs &a, hipStream_t stream) {
    if (a.ncols <= 0 || a.C <= 0) return RF_OK;
    const int64_t blocks = (a.ncols + kMxUnits - 1) / kMxUnits;
    if (blocks >= (1ll << 31)) { set_error("matrix path: too many chain columns"); return RF_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)blocks), block(kMxThreads);
    switch ((a.k + 7) >> 3) {
        case 1: hipLaunchKernelGGL(mx_chain_kernel<1>, grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL(mx_chain_kernel<2>, grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL(mx_chain_kernel<3>, grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL(mx_chain_kernel<4>, grid, block, 0, stream, a); break;
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

int launch_mx_apply(const MxChainArgs &a, hipStream_t stream) {
    if (a.ncols <= 0 || a.C <= 0) return RF_OK;
    const int64_t blocks = (a.ncols + kMxUnits - 1) / kMxUnits;
    if (blocks >= (1ll << 31)) { set_error("matrix path: too many chain columns"); return RF_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)blocks, (unsigned)a.C), block(kMxThreads);
    switch ((a.k + 7) >> 3) {
        case 1: hipLaunchKernelGGL(mx_apply_kernel<1>, grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL(mx_apply_kernel<2>, grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL(mx_apply_kernel<3>, grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL(mx_apply_kernel<4>, grid, block, 0, stream, a); break;
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

}  // namespace rf
