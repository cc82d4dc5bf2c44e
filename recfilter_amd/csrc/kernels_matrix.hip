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
// The image goes through LDS both ways (coalesced 16-byte accesses on the HBM side, operand layout on the other), so one kernel
// body serves scans along x (lane = line, or lane = tile for 1-D signals) and along y / z (lane = column).  The passes STREAM: a
// wave owns 32 units and walks their tile sub-block by sub-block through a private 32 x 32 staging buffer (see "the streaming
// passes" below).
// A causal scan and the anticausal scan behind it share ONE stage where their tails are short (a PAIR: mx_pass2p_kernel and
// MxPassArgs::pair -- pass 1 forms both scans' tails in one contraction, the final pass keeps the tile's causal result in registers).
// A clamped border is the zero-border operator plus a rank-one term in the scan's first sample (what the clamped prologue of
// lib/recfilter.cpp:330-336 adds is linear in x_0): dG / dH, applied by the lanes whose tile is where the scan enters the image.
#include <cstdlib>

#include "kernels_matrix.h"

namespace rf {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kMxThreads = 64 * kMxWaves;

__device__ __forceinline__ int mx_row(int t, int h) { return ((t >> 2) << 3) + (h << 2) + (t & 3); }

__device__ __forceinline__ floatx16 mx_zero() {
    floatx16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}

// Geometry of the workgroup's block -- 128 units x one tile; rows = units and columns = samples for scans along x, rows = samples
// and columns = units for y / z -- and where it lies in the plane.  Tiles need not divide the extent: tile t covers the samples
// [t T - off, (t + 1) T - off), and whatever falls outside [0, N) loads as zeros and is never stored.  The padding is always on
// the side where the scan LEAVES the image (off = 0 for a causal scan, M T - N for an anticausal one): zeros behind the last
// sample change nothing for the samples in front of them, and the tile where the scan enters is whole.
struct MxBlock {
    int64_t gbase, gpitch;     // element offset of (row 0, col 0), elements between rows
    int rows_valid, cols_valid;
    int tile;                  // MX_XL / MX_Y: the tile of the block
    int64_t first;             // first unit (MX_X1), line (MX_XL) or line of column 0 (MX_Y)
    int64_t s0;                // MX_XL / MX_Y: the sample the block's tile starts at (tile * T - off: may be negative)
    int lo, hi;                // MX_XL / MX_Y: the samples [lo, hi) of the block's tile exist
    int last_hi;               // how many samples of the LAST tile exist
};

template <bool XM>
__device__ __forceinline__ MxBlock mx_block(const MxPassArgs &a) {
    MxBlock b;
    b.last_hi = (int)(a.N + a.off - (int64_t)(a.M - 1) * a.T);
    b.lo = 0; b.hi = a.T;
    if constexpr (XM) {
        b.cols_valid = a.T;
        if (a.mode == MX_X1) {
            const int64_t U0 = (int64_t)blockIdx.x * kMxUnits;
            b.gbase = U0 * a.T - a.off; b.gpitch = a.T; b.tile = 0; b.first = U0; b.s0 = 0;          // (per row: mx_element)
            const int64_t left = a.units - U0;
            b.rows_valid = left < kMxUnits ? (int)left : kMxUnits;
        } else {
            // (consecutive workgroups walk ALONG the lines: with the line block as the fast index, a launch of 16384 lines keeps a
            // 4 KiB window open in every one of them at once -- more DRAM pages than the memory holds open)
            const int64_t lb = (int64_t)blockIdx.x / a.M;
            const int64_t L0 = lb * kMxUnits;
            b.tile = (int)((int64_t)blockIdx.x - lb * a.M);
            b.s0 = (int64_t)b.tile * a.T - a.off;
            b.lo = b.tile == 0 ? (int)a.off : 0; b.hi = b.tile == a.M - 1 ? b.last_hi : a.T;
            b.gbase = L0 * a.N + b.s0; b.gpitch = a.N; b.first = L0;
            const int64_t left = a.lines - L0;
            b.rows_valid = left < kMxUnits ? (int)left : kMxUnits;
        }
    } else {
        const int64_t c0 = (int64_t)blockIdx.x * kMxUnits, outer = blockIdx.z;
        b.tile = (int)blockIdx.y;
        b.rows_valid = a.T;
        b.s0 = (int64_t)b.tile * a.T - a.off;
        b.lo = b.tile == 0 ? (int)a.off : 0; b.hi = b.tile == a.M - 1 ? b.last_hi : a.T;
        b.gbase = (outer * a.N + b.s0) * a.inner + c0; b.gpitch = a.inner;
        b.first = outer * a.inner + c0;
        const int64_t left = a.inner - c0;
        b.cols_valid = left < kMxUnits ? (int)left : kMxUnits;
    }
    return b;
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

// ---- the STREAMING passes ---------------------------------------------------------------------------------------------
// Until round 5 the passes staged a whole 128-unit x T-sample block per workgroup: 66 KiB of LDS, two workgroups (two waves per
// SIMD) per CU, and a wave's time was its memory time PLUS its matrix time -- at f32 the 40 MFMAs of a sub-block are no small part
// of it (16384^2, order 12: 0.27 ms of matrix-core time against 0.37 ms of HBM time per final pass, 0.58-0.66 ms measured).  The
// streaming passes walk the tile sub-block by sub-block instead: every WAVE owns 32 units and a private 32 x 32 staging buffer
// (4.5-5 KiB), has the next two sub-blocks requested (one 128-byte run per unit: 16 bytes per lane, four loads each) while it
// works on this one, and stores y_b as soon as it exists.  No workgroup barrier inside the walk, three workgroups = 12 waves per
// CU (tools/microbench/mx_stream.hip is the gate that was measured first: 0.41-0.47 ms with 40 MFMAs per sub-block, 0.38-0.40
// with 24; the passes: 0.38 ms without a hand-over, 0.46-0.50 with one, 0.24 for pass 1).
constexpr int kMxStagePitchX = 36;     // [unit][32 samples + 4]
constexpr int kMxStagePitchY = 40;     // [sample][32 units + 8]: the two lane halves (rows 4 apart) hit disjoint banks
template <bool XM> constexpr int kMxStageFloats = 32 * (XM ? kMxStagePitchX : kMxStagePitchY);

__device__ __forceinline__ void mx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Where a lane's four 16-byte chunks of a sub-block lie.  X modes: chunk i = samples 4 lc .. 4 lc + 3 of unit lr + 8 i of the wave;
// y / z: chunk i = units 4 lc .. of sample lr + 8 i.  Chunk i of sub-block sb sits at base[i] + sb * step and exists when its
// position along the tile, 32 sb + p0[i], falls into [lo[i], hi[i]) (hi = 0: never).
template <bool XM>
struct MxStream {
    int64_t base[4], step;
    int lo[4], hi[4], p0[4];
    int slot[4];               // float index of the chunk in a staging buffer
};

template <bool XM>
__device__ __forceinline__ MxStream<XM> mx_stream(const MxPassArgs &a, const MxBlock &b, int w, int lane) {
    MxStream<XM> st;
    const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = lr + 8 * i;
        if constexpr (XM) {
            const int row = 32 * w + r;
            const bool valid = row < b.rows_valid;
            st.p0[i] = 4 * lc;
            st.slot[i] = r * kMxStagePitchX + 4 * lc;
            if (a.mode == MX_X1) {
                const int64_t U = b.first + row;
                int64_t tile = U;
                st.base[i] = b.gbase + (int64_t)row * a.T + 4 * lc;
                if (a.lines != 1) {
                    const int64_t line = U / a.M;
                    tile = U - line * a.M;
                    st.base[i] = line * a.N + tile * a.T - a.off + 4 * lc;
                }
                st.lo[i] = tile == 0 ? (int)a.off : 0;
                st.hi[i] = !valid ? 0 : tile == a.M - 1 ? b.last_hi : a.T;
            } else {
                st.base[i] = b.gbase + (int64_t)row * b.gpitch + 4 * lc;
                st.lo[i] = b.lo;
                st.hi[i] = valid ? b.hi : 0;
            }
        } else {
            const int col = 32 * w + 4 * lc;
            st.p0[i] = r;
            st.slot[i] = r * kMxStagePitchY + 4 * lc;
            st.base[i] = b.gbase + (int64_t)r * b.gpitch + col;
            st.lo[i] = b.lo;
            st.hi[i] = col < b.cols_valid ? b.hi : 0;
        }
    }
    st.step = XM ? 32 : 32 * b.gpitch;
    return st;
}

// The requests of a sub-block.  Nothing here may LOOK at what was loaded: a select on the loaded value right behind the load makes
// the compiler wait for it on the spot, and a walk whose every request is waited for before the next one is issued has no
// requests in flight (a chunk that does not exist reads the plane's first chunk; mx_stage_put replaces it by zeros).
// NT: non-temporal accesses (the image is read once and written once: 2-5 % at 16384^2) -- but only when every 128-byte run is a
// whole cache line.  Rows that are not multiples of 128 bytes make neighbouring sub-blocks share a line, and a line fetched with
// the hint is gone before the next sub-block asks for its other half (16380^2, order 12: 5.07 ms with the hint, 3.4 without).
template <bool XM, bool NT>
__device__ __forceinline__ void mx_request(const float *__restrict__ src, const MxStream<XM> &st, int sb, floatx4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pos = 32 * sb + st.p0[i];
        const bool ok = pos >= st.lo[i] && pos < st.hi[i];
        const floatx4 *from = reinterpret_cast<const floatx4 *>(src + (ok ? st.base[i] + sb * st.step : 0));
        if constexpr (NT) v[i] = __builtin_nontemporal_load(from);
        else v[i] = *from;
    }
}

template <bool XM>
__device__ __forceinline__ void mx_stage_put(float *stage, const MxStream<XM> &st, int sb, const floatx4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pos = 32 * sb + st.p0[i];
        const bool ok = pos >= st.lo[i] && pos < st.hi[i];
        *reinterpret_cast<floatx4 *>(stage + st.slot[i]) = ok ? v[i] : floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

template <bool XM>
__device__ __forceinline__ void mx_stage_read(const float *stage, int u, int h, float (&x)[16]) {
    if constexpr (XM) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(stage + u * kMxStagePitchX + 8 * q + 4 * h);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++) x[t] = stage[mx_row(t, h) * kMxStagePitchY + u];
    }
}

template <bool XM>
__device__ __forceinline__ void mx_stage_write(float *stage, int u, int h, const floatx16 &y) {
    if constexpr (XM) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4 *>(stage + u * kMxStagePitchX + 8 * q + 4 * h) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++) stage[mx_row(t, h) * kMxStagePitchY + u] = y[t];
    }
}

// sample m (0 .. 31) of this lane's unit in a staged sub-block
template <bool XM>
__device__ __forceinline__ float mx_stage_at(const float *stage, int u, int m) {
    return XM ? stage[u * kMxStagePitchX + m] : stage[m * kMxStagePitchY + u];
}

__device__ __forceinline__ void mx_store_tail(float *tails, int64_t tidx, int k, int h, const floatx16 &acc) {
    const int KP = 8 * ((k + 7) >> 3);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r0 = 8 * q + 4 * h;
        if (r0 < KP) *reinterpret_cast<float4 *>(tails + tidx * KP + r0) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// The constant operands a wave needs per sub-block (H of pass 1, the next stage's H in pass 2) and the border vectors sit in LDS,
// copied once per workgroup: a global load inside the walk would be waited for IN ORDER, behind the requests for the next
// sub-block that were issued before it -- the HBM latency the walk exists to hide.
// Every load of constants is ISSUED before the first request for image data and only then waited for, so that the wait does not
// include the image's latency.
struct MxConsts {
    float4 frag[kMxMaxNB];
    float vec;
};

__device__ __forceinline__ MxConsts mx_consts_request(const float *frag, int nb, const float *vec) {
    MxConsts c;
#pragma unroll
    for (int i = 0; i < kMxMaxNB; i++) {
        c.frag[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < nb) c.frag[i] = reinterpret_cast<const float4 *>(frag)[(int)threadIdx.x + kMxThreads * i];
    }
    c.vec = threadIdx.x < 32 ? vec[threadIdx.x] : 0.0f;
    return c;
}

__device__ __forceinline__ void mx_consts_store(float *lds, const MxConsts &c, int nb) {
#pragma unroll
    for (int i = 0; i < kMxMaxNB; i++)
        if (i < nb) reinterpret_cast<float4 *>(lds)[(int)threadIdx.x + kMxThreads * i] = c.frag[i];
    if (threadIdx.x < 32) lds[nb * 1024 + threadIdx.x] = c.vec;
}

// NBT: the tile's sub-blocks when they are the usual four (0: read from the arguments; see mx_pass2p_kernel)
template <bool XM, bool NT, int NBT>
__global__ void __launch_bounds__(kMxThreads)
mx_pass1s_kernel(const float *__restrict__ src, MxPassArgs a) {
    const int NB = NBT ? NBT : a.NB;
    extern __shared__ __attribute__((aligned(16))) float mx_lds[];
    float *consts = mx_lds + kMxWaves * kMxStageFloats<XM>;       // H [NB][16][64], dH [32]
    const MxBlock blk = mx_block<XM>(a);
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    float *stage = mx_lds + w * kMxStageFloats<XM>;
    const MxStream<XM> st = mx_stream<XM>(a, blk, w, lane);
    const MxLane ln = mx_lane<XM>(a, blk, 32 * w + u);
    // (a pair stage: ONE contraction for both scans' tails -- its H has the causal scan's H in the rows 0 .. 15 and H21 in the rows
    // 16 .. 31, its border vector likewise: a pair's tails have at most 16 rows)
    const MxConsts cs = mx_consts_request(a.pair ? a.p_H : a.H, NB, a.pair ? a.p_dH : a.dH);
    // (pass 1 has registers to spare: four sub-blocks -- 16 KiB per wave -- are in flight)
    floatx4 pre[4][4];
#pragma unroll
    for (int d = 0; d < 4; d++)
        if (d < NB) mx_request<XM, NT>(src, st, d, pre[d]);
    mx_consts_store(consts, cs, NB);
    __syncthreads();
    floatx16 acc = mx_zero();
    float x0 = 0.0f;
    auto step = [&](int sb, floatx4 (&buf)[4]) __attribute__((always_inline)) {
        mx_stage_put<XM>(stage, st, sb, buf);
        if (sb + 4 < NB) mx_request<XM, NT>(src, st, sb + 4, buf);
        mx_wave_sync();
        float x[16];
        mx_stage_read<XM>(stage, u, h, x);
        if (sb == (a.causal ? 0 : NB - 1)) x0 = mx_stage_at<XM>(stage, u, a.causal ? 0 : 31);
        const float *Hf = consts + sb * 1024 + lane;
#pragma unroll
        for (int t = 0; t < 16; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Hf[t * 64], x[t], acc, 0, 0, 0);
        mx_wave_sync();
    };
    if constexpr (NBT != 0) {
#pragma unroll
        for (int sb = 0; sb < NBT; sb += 4) {
            step(sb, pre[0]);
            if (sb + 1 < NBT) step(sb + 1, pre[1]);
            if (sb + 2 < NBT) step(sb + 2, pre[2]);
            if (sb + 3 < NBT) step(sb + 3, pre[3]);
        }
    } else {
        for (int sb = 0; sb < NB; sb += 4) {
            step(sb, pre[0]);
            if (sb + 1 < NB) step(sb + 1, pre[1]);
            if (sb + 2 < NB) step(sb + 2, pre[2]);
            if (sb + 3 < NB) step(sb + 3, pre[3]);
        }
    }
    if (a.clamped && ln.valid && ln.border) {
        const float *dH = consts + NB * 1024;
#pragma unroll
        for (int t = 0; t < 16; t++) acc[t] = fmaf(dH[mx_row(t, h)], x0, acc[t]);
    }
    if (!ln.valid) return;
    if (a.pair) {
        floatx16 second = mx_zero();
#pragma unroll
        for (int t = 0; t < 8; t++) second[t] = acc[8 + t];        // rows 16 .. 31 -> the anticausal scan's rows 0 .. 15
        mx_store_tail(a.tails, ln.tidx, a.k, h, acc);               // (rows 0 .. KP - 1 <= 15 are stored)
        mx_store_tail(a.p_tails, ln.tidx, a.p_k, h, second);
    } else {
        mx_store_tail(a.tails, ln.tidx, a.k, h, acc);
    }
}

// NLQ = ceil(k / 8) and the direction are compile-time: the K steps of R that are not all zero -- the k most recent rows of the
// previous sub-block -- are then a fixed set of registers, requested from LDS ahead of the MFMAs that use them.
template <bool XM, int NLQ, bool CAUSAL, bool NT, int NBT>
__global__ void __launch_bounds__(kMxThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
mx_pass2s_kernel(const float *src, float *dst, MxPassArgs a) {      // (src == dst for every stage after the first and for in-place executes: no __restrict__)
    const int NB = NBT ? NBT : a.NB;       // (NBT: the usual eight sub-blocks of a tile of 256 as a constant; see mx_pass2p_kernel)
    extern __shared__ __attribute__((aligned(16))) float mx_lds[];
    // (one staging buffer per wave: x_b is in registers by the time y_b is written over it)
    // constants: dG [32], pad [32], G [16][64], R [16][64], next H [NB][16][64], next dH [32] -- G and R too: the 32 registers they
    // would take are what a second sub-block in flight takes
    float *consts = mx_lds + kMxWaves * kMxStageFloats<XM>;
    const float *Gl = consts + 64, *Rl = consts + 64 + 1024, *Hl = consts + 64 + 2048;
    const MxBlock blk = mx_block<XM>(a);
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    float *stage = mx_lds + w * kMxStageFloats<XM>;
    const MxStream<XM> st = mx_stream<XM>(a, blk, w, lane);
    const MxLane ln = mx_lane<XM>(a, blk, 32 * w + u);
    const float dg = threadIdx.x < 32 ? a.dG[threadIdx.x] : 0.0f;
    static_assert(kMxThreads == 256, "one 16-byte piece of G and of R per thread");
    const float4 gr0 = reinterpret_cast<const float4 *>(a.G)[threadIdx.x], gr1 = reinterpret_cast<const float4 *>(a.R)[threadIdx.x];
    const MxConsts cs = mx_consts_request(a.next ? a.next_H : a.G, a.next ? a.next_NB : 0, a.next ? a.next_dH : a.dG);
    // two sub-blocks in flight per wave
    floatx4 pre[2][4];
    mx_request<XM, NT>(src, st, CAUSAL ? 0 : NB - 1, pre[0]);
    if (NB > 1) mx_request<XM, NT>(src, st, CAUSAL ? 1 : NB - 2, pre[1]);
    constexpr int NL = 4 * NLQ, KP = 8 * NLQ, T_LO = CAUSAL ? 16 - NL : 0;
    floatx16 prev = mx_zero();
    if (ln.valid && (!ln.enters || (a.incoming != nullptr && !ln.border))) {
        const float *tp = ln.enters ? a.incoming + ln.line * KP : a.tails + ln.prev_tidx * KP;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r_lo = CAUSAL ? 28 - 8 * q - 4 * h : 8 * q + 4 * h;
            if (r_lo < KP) {
                const float4 v = *reinterpret_cast<const float4 *>(tp + r_lo);
                if (CAUSAL) { prev[4 * q] = v.w; prev[4 * q + 1] = v.z; prev[4 * q + 2] = v.y; prev[4 * q + 3] = v.x; }
                else          { prev[4 * q] = v.x; prev[4 * q + 1] = v.y; prev[4 * q + 2] = v.z; prev[4 * q + 3] = v.w; }
            }
        }
    }
    if (threadIdx.x < 32) consts[threadIdx.x] = dg;
    reinterpret_cast<float4 *>(consts + 64)[threadIdx.x] = gr0;
    reinterpret_cast<float4 *>(consts + 64 + 1024)[threadIdx.x] = gr1;
    if (a.next) mx_consts_store(consts + 64 + 2048, cs, a.next_NB);      // (next dH lands behind next H)
    __syncthreads();
    floatx16 nacc = mx_zero();
    float nx0 = 0.0f;
    auto step = [&](int bi, floatx4 (&buf)[4]) __attribute__((always_inline)) {
        const int sb = CAUSAL ? bi : NB - 1 - bi;
        mx_stage_put<XM>(stage, st, sb, buf);
        if (bi + 2 < NB) mx_request<XM, NT>(src, st, CAUSAL ? sb + 2 : sb - 2, buf);
        mx_wave_sync();
        float x[16];
        mx_stage_read<XM>(stage, u, h, x);
        floatx16 c = mx_zero();
        if (bi == 0 && a.clamped && ln.valid && ln.border) {
            const float x0 = mx_stage_at<XM>(stage, u, CAUSAL ? 0 : 31);
#pragma unroll
            for (int t = 0; t < 16; t++) c[t] = consts[mx_row(t, h)] * x0;
        }
        mx_wave_sync();
#pragma unroll
        for (int t = 0; t < 16; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Gl[t * 64 + lane], x[t], c, 0, 0, 0);
#pragma unroll
        for (int t = T_LO; t < T_LO + NL; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Rl[t * 64 + lane], prev[t], c, 0, 0, 0);
        prev = c;
        mx_stage_write<XM>(stage, u, h, c);
        if (a.next == 1) {
            // the tile-local tails of the NEXT scan of this dimension: its pass 1 without its read of the image (a finished
            // sub-block in accumulator layout IS the B operand)
            const float *Hn = Hl + sb * 1024 + lane;
#pragma unroll
            for (int t = 0; t < 16; t++) nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Hn[t * 64], c[t], nacc, 0, 0, 0);
        }
        mx_wave_sync();
        if (a.next == 1 && sb == (a.next_causal ? 0 : NB - 1)) nx0 = mx_stage_at<XM>(stage, u, a.next_causal ? 0 : 31);
        if constexpr (XM) {
            if (a.next == 2) {
                // The first y scan's tile-local tails (MxPassArgs::next == 2): the workgroup's 128 lines are one y tile, the 32
                // samples of this sub-block 32 of its columns.  A wave contracts ITS 32 lines (sub-block w of the y tile) with
                // its part of the y scan's H -- the staged y_b read down its columns is the B operand --, the four partial sums
                // meet in LDS and wave q adds up and stores rows 8 q .. 8 q + 7 of the 32 tails.
                floatx16 part = mx_zero();
                const float *Hy = Hl + w * 1024 + lane;
#pragma unroll
                for (int t = 0; t < 16; t++) part = __builtin_amdgcn_mfma_f32_32x32x2f32(Hy[t * 64], stage[mx_row(t, h) * kMxStagePitchX + u], part, 0, 0, 0);
                const int ytile = (int)(blk.first / kMxUnits), MY = (int)(a.lines / kMxUnits);
                const bool yenters = (a.next_causal ? ytile == 0 : ytile == MY - 1) && (a.next_causal ? a.slab_first != 0 : a.slab_last != 0);
                if (a.clamped && yenters && w == (a.next_causal ? 0 : kMxWaves - 1)) {
                    const float y0 = stage[(a.next_causal ? 0 : 31) * kMxStagePitchX + u];
#pragma unroll
                    for (int t = 0; t < 16; t++) part[t] = fmaf(Hl[a.next_NB * 1024 + mx_row(t, h)], y0, part[t]);
                }
                float *red = consts + 64 + 2048 + a.next_NB * 1024 + 32;         // [wave][q][lane] pieces of 16 bytes
                const int nq = (a.next_k + 7) >> 3;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < nq) reinterpret_cast<float4 *>(red)[(w * nq + q) * 64 + lane] = make_float4(part[4 * q], part[4 * q + 1], part[4 * q + 2], part[4 * q + 3]);
                __syncthreads();
                if (w < nq) {
                    float4 sum = reinterpret_cast<const float4 *>(red)[w * 64 + lane];
#pragma unroll
                    for (int ww = 1; ww < kMxWaves; ww++) {
                        const float4 v = reinterpret_cast<const float4 *>(red)[(ww * nq + w) * 64 + lane];
                        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    }
                    const int64_t column = (int64_t)blk.tile * a.T + 32 * sb + u;
                    *reinterpret_cast<float4 *>(a.next_tails + ((int64_t)ytile * a.N + column) * (8 * nq) + 8 * w + 4 * h) = sum;
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pos = 32 * sb + st.p0[i];
            if (pos >= st.lo[i] && pos < st.hi[i])
            {
                const floatx4 y = *reinterpret_cast<const floatx4 *>(stage + st.slot[i]);
                floatx4 *to = reinterpret_cast<floatx4 *>(dst + st.base[i] + sb * st.step);
                if constexpr (NT) __builtin_nontemporal_store(y, to);
                else *to = y;
            }
        }
        mx_wave_sync();
    };
    if constexpr (NBT != 0) {
#pragma unroll
        for (int bi = 0; bi < NBT; bi += 2) {
            step(bi, pre[0]);
            if (bi + 1 < NBT) step(bi + 1, pre[1]);
        }
    } else {
        for (int bi = 0; bi < NB; bi += 2) {
            step(bi, pre[0]);
            if (bi + 1 < NB) step(bi + 1, pre[1]);
        }
    }
    if (a.next == 1) {
        const bool nenters = (a.next_causal ? ln.tile == 0 : ln.tile == a.M - 1) && (a.next_causal ? a.slab_first != 0 : a.slab_last != 0);
        if (a.clamped && ln.valid && nenters) {
#pragma unroll
            for (int t = 0; t < 16; t++) nacc[t] = fmaf(Hl[a.next_NB * 1024 + mx_row(t, h)], nx0, nacc[t]);
        }
        if (ln.valid) mx_store_tail(a.next_tails, ln.tidx, a.next_k, h, nacc);
    }
}

// ---- the final pass of a PAIR stage: y = anticausal(causal(x)) on the tile, in one read and one write ---------------------
// Forward walk: w_b = G1 x_b + R1 w_(b-1) (w_(-1): the neighbouring tile's completed causal tail), the finished sub-blocks stay in
// registers (accumulator layout IS the B operand of what follows: NB x 16 registers, which is why a pair's tile has at most four
// sub-blocks); backward walk: y_b = G2 w_b + R2 y_(b+1) (y_(NB): the completed anticausal tail of the tile behind), stored as it
// appears.  Both scans have the same number of tail pieces (NLQ).  Clamped borders: the causal scan's term dG1 x_0 in the
// image's first tile, the anticausal scan's dG2 w_(T-1) in its last one (lib/recfilter.cpp:330-336 reads the partially
// updated buffer: the anticausal scan's border sample is the causal RESULT).
// NBT: the tile's sub-blocks when they are the usual four (0: read from the arguments) -- with the count a constant no request sits
// under a branch, and the compiler keeps count of what is in flight (behind a branch it waits for more than the step needs).
template <bool XM, int NLQ, bool NT, int NBT>
__global__ void __launch_bounds__(kMxThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
mx_pass2p_kernel(const float *src, float *dst, MxPassArgs a) {      // (src == dst for every stage after the first and for in-place executes: no __restrict__)
    const int NB = NBT ? NBT : a.NB;
    extern __shared__ __attribute__((aligned(16))) float mx_lds[];
    // constants: dG1 [32], dG2 [32], G1, R1, G2, R2 [16][64] each
    float *consts = mx_lds + kMxWaves * kMxStageFloats<XM>;
    const float *G1 = consts + 64, *R1 = G1 + 1024, *G2 = R1 + 1024, *R2 = G2 + 1024;
    const MxBlock blk = mx_block<XM>(a);
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    float *stage = mx_lds + w * kMxStageFloats<XM>;
    const MxStream<XM> st = mx_stream<XM>(a, blk, w, lane);
    const MxLane ln = mx_lane<XM>(a, blk, 32 * w + u);
    static_assert(kMxThreads == 256, "one 16-byte piece of every operator per thread");
    const float dg = threadIdx.x < 32 ? a.dG[threadIdx.x] : threadIdx.x < 64 ? a.p_dG[threadIdx.x - 32] : 0.0f;
    const float4 c0 = reinterpret_cast<const float4 *>(a.G)[threadIdx.x], c1 = reinterpret_cast<const float4 *>(a.R)[threadIdx.x];
    const float4 c2 = reinterpret_cast<const float4 *>(a.p_G)[threadIdx.x], c3 = reinterpret_cast<const float4 *>(a.p_R)[threadIdx.x];
    floatx4 pre[2][4];
    mx_request<XM, NT>(src, st, 0, pre[0]);
    if (NB > 1) mx_request<XM, NT>(src, st, 1, pre[1]);
    constexpr int NL = 4 * NLQ, KP = 8 * NLQ;
    // the carries: the causal one enters from the tile in front (row i of a sub-block that precedes the tile = tail 31 - i), the
    // anticausal one from the tile behind (row i = tail i)
    const bool first_tile = ln.tile == 0, last_tile = ln.tile == a.M - 1;
    floatx16 prev = mx_zero(), back = mx_zero();
    if (ln.valid && !first_tile) {
        const float *tp = a.tails + ln.prev_tidx * KP;          // (mx_lane: the causal scan's previous tile)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r_lo = 28 - 8 * q - 4 * h;
            if (r_lo < KP) {
                const float4 v = *reinterpret_cast<const float4 *>(tp + r_lo);
                prev[4 * q] = v.w; prev[4 * q + 1] = v.z; prev[4 * q + 2] = v.y; prev[4 * q + 3] = v.x;
            }
        }
    }
    if (ln.valid && !last_tile) {
        const float *tp = a.p_tails + (2 * ln.tidx - ln.prev_tidx) * KP;        // the tile behind: tidx + (tidx - prev_tidx)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r_lo = 8 * q + 4 * h;
            if (r_lo < KP) {
                const float4 v = *reinterpret_cast<const float4 *>(tp + r_lo);
                back[4 * q] = v.x; back[4 * q + 1] = v.y; back[4 * q + 2] = v.z; back[4 * q + 3] = v.w;
            }
        }
    }
    if (threadIdx.x < 64) consts[threadIdx.x] = dg;
    reinterpret_cast<float4 *>(consts + 64)[threadIdx.x] = c0;
    reinterpret_cast<float4 *>(consts + 64 + 1024)[threadIdx.x] = c1;
    reinterpret_cast<float4 *>(consts + 64 + 2048)[threadIdx.x] = c2;
    reinterpret_cast<float4 *>(consts + 64 + 3072)[threadIdx.x] = c3;
    __syncthreads();
    floatx16 wv[4];
    float w_last = 0.0f;
    auto forward = [&](int bi, floatx4 (&buf)[4], floatx16 &out) __attribute__((always_inline)) {
        mx_stage_put<XM>(stage, st, bi, buf);
        if (bi + 2 < NB) mx_request<XM, NT>(src, st, bi + 2, buf);
        mx_wave_sync();
        float x[16];
        mx_stage_read<XM>(stage, u, h, x);
        floatx16 c = mx_zero();
        if (bi == 0 && a.clamped && ln.valid && first_tile && a.slab_first) {
            const float x0 = mx_stage_at<XM>(stage, u, 0);
#pragma unroll
            for (int t = 0; t < 16; t++) c[t] = consts[mx_row(t, h)] * x0;
        }
        mx_wave_sync();
#pragma unroll
        for (int t = 0; t < 16; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(G1[t * 64 + lane], x[t], c, 0, 0, 0);
#pragma unroll
        for (int t = 16 - NL; t < 16; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(R1[t * 64 + lane], prev[t], c, 0, 0, 0);
        prev = c;
        out = c;
        if (bi == NB - 1) {                   // the tile's last causal output, for the anticausal scan's border term
            mx_stage_write<XM>(stage, u, h, c);
            mx_wave_sync();
            w_last = mx_stage_at<XM>(stage, u, 31);
            mx_wave_sync();
        }
    };
    auto backward = [&](int bi, const floatx16 &in) __attribute__((always_inline)) {
        floatx16 c = mx_zero();
        if (bi == NB - 1 && a.clamped && ln.valid && last_tile && a.slab_last) {
#pragma unroll
            for (int t = 0; t < 16; t++) c[t] = consts[32 + mx_row(t, h)] * w_last;
        }
#pragma unroll
        for (int t = 0; t < 16; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(G2[t * 64 + lane], in[t], c, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NL; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(R2[t * 64 + lane], back[t], c, 0, 0, 0);
        back = c;
        mx_stage_write<XM>(stage, u, h, c);
        mx_wave_sync();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pos = 32 * bi + st.p0[i];
            if (pos >= st.lo[i] && pos < st.hi[i]) {
                const floatx4 y = *reinterpret_cast<const floatx4 *>(stage + st.slot[i]);
                floatx4 *to = reinterpret_cast<floatx4 *>(dst + st.base[i] + bi * st.step);
                if constexpr (NT) __builtin_nontemporal_store(y, to);
                else *to = y;
            }
        }
        mx_wave_sync();
    };
    forward(0, pre[0], wv[0]);
    if (NB > 1) forward(1, pre[1], wv[1]);
    if (NB > 2) forward(2, pre[0], wv[2]);
    if (NB > 3) forward(3, pre[1], wv[3]);
    if (NB > 3) backward(3, wv[3]);
    if (NB > 2) backward(2, wv[2]);
    if (NB > 1) backward(1, wv[1]);
    backward(0, wv[0]);
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

// elements of the chain requested ahead of the step that consumes them: a step is 4 NLQ dependent MFMAs (0.1 us per NLQ), and
// the requests in flight have to cover the memory latency (2-3 us) -- eight steps for short tails, four for long ones
template <int NLQ> constexpr int kMxAhead = NLQ <= 2 ? 8 : 4;

template <int NLQ, bool CROSS>
__global__ void __launch_bounds__(kMxThreads)
mx_chain_kernel(MxChainArgs a) {
    constexpr int NL = 4 * NLQ, KP = 8 * NLQ, AHEAD = kMxAhead<NLQ>;
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, h = lane >> 5, u = lane & 31;
    const MxCol col = mx_col(a, ((int64_t)blockIdx.x * kMxWaves + w) * 32 + u);
    float Af[NL];
#pragma unroll
    for (int t = 0; t < NL; t++) Af[t] = a.A[t * 64 + lane];
    auto elem = [&](int j) { return a.seq + (col.off + (int64_t)j * a.s_j) * KP + 4 * h; };
    float4 ring[AHEAD][NLQ];
    // (requests without a branch: a load under a condition makes the compiler lose count of what is in flight, and it then waits
    // for EVERYTHING at the top of the loop -- the memory latency once per trip; an element that does not exist reads the
    // sequence's first one and is replaced by zeros where it is used)
    auto request = [&](int slot, int j) {
        const float *p = (col.valid && j < col.len) ? elem(j) : a.seq + 4 * h;
#pragma unroll
        for (int q = 0; q < NLQ; q++) ring[slot][q] = *reinterpret_cast<const float4 *>(p + 8 * q);
    };
    // CROSS (MxChainArgs::cross): a second ring with the elements of the cross term -- off the dependent path, the MFMA pipe has room
    float Wf[CROSS ? NL : 1];
    float4 ringx[CROSS ? AHEAD : 1][NLQ];
    auto requestx = [&](int slot, int j) {
        const float *p = (col.valid && j < col.len && j < a.cross_steps) ? a.cross + (col.off + (int64_t)j * a.s_j + a.cross_shift) * KP + 4 * h : a.seq + 4 * h;
#pragma unroll
        for (int q = 0; q < NLQ; q++) ringx[slot][q] = *reinterpret_cast<const float4 *>(p + 8 * q);
    };
    floatx16 border = mx_zero();
    if constexpr (CROSS) {
#pragma unroll
        for (int t = 0; t < NL; t++) Wf[t] = a.crossW[t * 64 + lane];
        if (a.crossD != nullptr) {
            floatx16 e0 = mx_zero();
            if (col.valid) {
#pragma unroll
                for (int q = 0; q < NLQ; q++) {
                    const float4 v = *reinterpret_cast<const float4 *>(a.cross + col.off * KP + 4 * h + 8 * q);
                    e0[4 * q] = v.x; e0[4 * q + 1] = v.y; e0[4 * q + 2] = v.z; e0[4 * q + 3] = v.w;
                }
            }
#pragma unroll
            for (int t = 0; t < NL; t++) border = __builtin_amdgcn_mfma_f32_32x32x2f32(a.crossD[t * 64 + lane], e0[t], border, 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < AHEAD; d++) requestx(d, d);
    }
#pragma unroll
    for (int d = 0; d < AHEAD; d++) request(d, d);
    floatx16 x = mx_zero();
    for (int j0 = 0; j0 < a.C; j0 += AHEAD) {
#pragma unroll
        for (int d = 0; d < AHEAD; d++) {
            const int j = j0 + d;
            floatx16 c = mx_zero();
            const bool live = col.valid && j < col.len;
#pragma unroll
            for (int q = 0; q < NLQ; q++) {
                c[4 * q] = live ? ring[d][q].x : 0.0f; c[4 * q + 1] = live ? ring[d][q].y : 0.0f;
                c[4 * q + 2] = live ? ring[d][q].z : 0.0f; c[4 * q + 3] = live ? ring[d][q].w : 0.0f;
            }
            request(d, j + AHEAD);                       // before this step's arithmetic and stores
            if constexpr (CROSS) {
                floatx16 xv = mx_zero();
                const bool crossed = live && j < a.cross_steps;
#pragma unroll
                for (int q = 0; q < NLQ; q++) {
                    xv[4 * q] = crossed ? ringx[d][q].x : 0.0f; xv[4 * q + 1] = crossed ? ringx[d][q].y : 0.0f;
                    xv[4 * q + 2] = crossed ? ringx[d][q].z : 0.0f; xv[4 * q + 3] = crossed ? ringx[d][q].w : 0.0f;
                }
                requestx(d, j + AHEAD);
                if (j == 0) {
#pragma unroll
                    for (int t = 0; t < 16; t++) c[t] += border[t];
                }
#pragma unroll
                for (int t = 0; t < NL; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Wf[t], xv[t], c, 0, 0, 0);
            }
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

dim3 mx_grid(const MxPassArgs &a) {
    if (a.mode == MX_X1) return dim3((unsigned)((a.units + kMxUnits - 1) / kMxUnits));
    if (a.mode == MX_XL) return dim3((unsigned)(((a.lines + kMxUnits - 1) / kMxUnits) * a.M));
    return dim3((unsigned)((a.inner + kMxUnits - 1) / kMxUnits), (unsigned)a.M, (unsigned)(a.lines / a.inner));
}

int mx_check(const MxPassArgs &a) {
    if (a.T != 32 * a.NB || a.NB < 1 || a.NB > kMxMaxNB || a.k < 1 || a.k > 32 || a.M < 1 || a.off < 0 || a.off >= a.T ||
        (int64_t)a.M * a.T < a.N + a.off || (int64_t)(a.M - 1) * a.T >= a.N + a.off || (a.mode != MX_Y && (a.N % 4 != 0 || a.off % 4 != 0))) {
        set_error("matrix path: bad tile geometry (T %d, NB %d, M %d, k %d)", a.T, a.NB, a.M, a.k);
        return RF_ERR_INVALID_ARG;
    }
    if (a.mode == MX_XL && ((a.lines + kMxUnits - 1) / kMxUnits) * a.M >= (1ll << 31)) { set_error("matrix path: extents out of range"); return RF_ERR_UNSUPPORTED; }
    const dim3 g = mx_grid(a);
    if (g.y > 65535u || g.z > 65535u || (a.mode == MX_Y && (a.inner % 4 != 0 || a.lines % a.inner != 0))) {
        set_error("matrix path: extents out of range");
        return RF_ERR_UNSUPPORTED;
    }
    return RF_OK;
}

}  // namespace

// whether every 128-byte run of a sub-block is one whole cache line (the non-temporal variants, mx_request)
static bool mx_whole_lines(const MxPassArgs &a, const void *src, const void *dst) {
    const int64_t pitch = a.mode == MX_Y ? a.inner : (a.mode == MX_XL || a.lines != 1) ? a.N : 32;
    return ((uintptr_t)src | (uintptr_t)dst) % 128 == 0 && pitch % 32 == 0 && a.off % 32 == 0;
}

// LDS of a pass: the waves' staging buffers and the constants; more than 64 KiB would need an opt-in per kernel and the passes
// stay well below it on purpose (three or four workgroups per CU)
int launch_mx_pass1(const float *src, const MxPassArgs &a, hipStream_t stream) {
    if (int rc = mx_check(a)) return rc;
    if (a.pair && (a.causal == 0 || a.ragged || a.NB > 4 || a.k > 16 || a.p_k > 16)) { set_error("matrix path: bad pair stage"); return RF_ERR_INVALID_ARG; }
    const size_t consts = ((size_t)a.NB * 1024 + 32) * sizeof(float);
    const size_t lds_x = kMxWaves * kMxStageFloats<true> * sizeof(float) + consts, lds_y = kMxWaves * kMxStageFloats<false> * sizeof(float) + consts;
#define RF_MX_P1S_NB(NT, NBT)                                                                                                       \
    if (a.mode == MX_Y) hipLaunchKernelGGL((mx_pass1s_kernel<false, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_y, stream, src, a);  \
    else hipLaunchKernelGGL((mx_pass1s_kernel<true, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_x, stream, src, a);
#define RF_MX_P1S(NT) if (a.NB == 4) { RF_MX_P1S_NB(NT, 4) } else if (a.NB == 8) { RF_MX_P1S_NB(NT, 8) } else { RF_MX_P1S_NB(NT, 0) }
    if (mx_whole_lines(a, src, src)) { RF_MX_P1S(true) } else { RF_MX_P1S(false) }
#undef RF_MX_P1S
#undef RF_MX_P1S_NB
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

int launch_mx_pass2(const float *src, float *dst, const MxPassArgs &a, hipStream_t stream) {
    if (int rc = mx_check(a)) return rc;
    if ((a.next == 2 && (a.mode != MX_XL || a.ragged || a.lines % kMxUnits != 0 || a.next_NB != kMxWaves)) || (a.next == 1 && a.next_NB != a.NB)) { set_error("matrix path: bad x -> y hand-over"); return RF_ERR_INVALID_ARG; }
    const size_t consts = (64 + 2048 + (a.next ? (size_t)a.next_NB * 1024 + 32 : 0) + (a.next == 2 ? (size_t)((a.next_k + 7) >> 3) * kMxWaves * 64 * 4 : 0)) * sizeof(float);
    const size_t lds_x = kMxWaves * kMxStageFloats<true> * sizeof(float) + consts, lds_y = kMxWaves * kMxStageFloats<false> * sizeof(float) + consts;
#define RF_MX_P2S_NB(NLQ, NT, NBT)                                                                                                     \
    if (a.mode == MX_Y) {                                                                                                              \
        if (a.causal) hipLaunchKernelGGL((mx_pass2s_kernel<false, NLQ, true, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_y, stream, src, dst, a);   \
        else hipLaunchKernelGGL((mx_pass2s_kernel<false, NLQ, false, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_y, stream, src, dst, a);           \
    } else {                                                                                                                           \
        if (a.causal) hipLaunchKernelGGL((mx_pass2s_kernel<true, NLQ, true, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_x, stream, src, dst, a);    \
        else hipLaunchKernelGGL((mx_pass2s_kernel<true, NLQ, false, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_x, stream, src, dst, a);            \
    }
#define RF_MX_P2S_NT(NLQ, NT) if (a.NB == 8) { RF_MX_P2S_NB(NLQ, NT, 8) } else { RF_MX_P2S_NB(NLQ, NT, 0) }
#define RF_MX_P2S(NLQ) if (whole) { RF_MX_P2S_NT(NLQ, true) } else { RF_MX_P2S_NT(NLQ, false) }
    const bool whole = mx_whole_lines(a, src, dst);
    switch ((a.k + 7) >> 3) {
        case 1: RF_MX_P2S(1) break;
        case 2: RF_MX_P2S(2) break;
        case 3: RF_MX_P2S(3) break;
        default: RF_MX_P2S(4) break;
    }
#undef RF_MX_P2S
#undef RF_MX_P2S_NT
#undef RF_MX_P2S_NB
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

int launch_mx_pass2_pair(const float *src, float *dst, const MxPassArgs &a, hipStream_t stream) {
    if (int rc = mx_check(a)) return rc;
    if (a.pair == 0 || a.causal == 0 || a.ragged || a.NB > 4 || ((a.k + 7) >> 3) != ((a.p_k + 7) >> 3) || a.next != 0 || a.incoming != nullptr) {
        set_error("matrix path: bad pair stage");
        return RF_ERR_INVALID_ARG;
    }
    const size_t consts = (64 + 4 * 1024) * sizeof(float);
    const size_t lds_x = kMxWaves * kMxStageFloats<true> * sizeof(float) + consts, lds_y = kMxWaves * kMxStageFloats<false> * sizeof(float) + consts;
    const bool whole = mx_whole_lines(a, src, dst);
#define RF_MX_P2P_NB(NLQ, NT, NBT)                                                                                                     \
    if (a.mode == MX_Y) hipLaunchKernelGGL((mx_pass2p_kernel<false, NLQ, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_y, stream, src, dst, a);  \
    else hipLaunchKernelGGL((mx_pass2p_kernel<true, NLQ, NT, NBT>), mx_grid(a), dim3(kMxThreads), lds_x, stream, src, dst, a);
#define RF_MX_P2P_NT(NLQ, NT) if (a.NB == 4) { RF_MX_P2P_NB(NLQ, NT, 4) } else { RF_MX_P2P_NB(NLQ, NT, 0) }
#define RF_MX_P2P(NLQ) if (whole) { RF_MX_P2P_NT(NLQ, true) } else { RF_MX_P2P_NT(NLQ, false) }
    switch ((a.k + 7) >> 3) {
        case 1: RF_MX_P2P(1) break;
        case 2: RF_MX_P2P(2) break;
        case 3: RF_MX_P2P(3) break;
        default: RF_MX_P2P(4) break;
    }
#undef RF_MX_P2P
#undef RF_MX_P2P_NT
#undef RF_MX_P2P_NB
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

int launch_mx_chain(const MxChainArgs &a, hipStream_t stream) {
    if (a.ncols <= 0 || a.C <= 0) return RF_OK;
    const int64_t blocks = (a.ncols + kMxUnits - 1) / kMxUnits;
    if (blocks >= (1ll << 31)) { set_error("matrix path: too many chain columns"); return RF_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)blocks), block(kMxThreads);
    if (a.cross != nullptr) {           // (a pair's tails have at most 16 rows)
        if (((a.k + 7) >> 3) > 2 || a.crossW == nullptr || a.exits != nullptr) { set_error("matrix path: bad cross term"); return RF_ERR_INVALID_ARG; }
        if (((a.k + 7) >> 3) == 1) hipLaunchKernelGGL((mx_chain_kernel<1, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((mx_chain_kernel<2, true>), grid, block, 0, stream, a);
        RF_HIP_CHECK(hipGetLastError());
        return RF_OK;
    }
    switch ((a.k + 7) >> 3) {
        case 1: hipLaunchKernelGGL((mx_chain_kernel<1, false>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((mx_chain_kernel<2, false>), grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((mx_chain_kernel<3, false>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((mx_chain_kernel<4, false>), grid, block, 0, stream, a); break;
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
