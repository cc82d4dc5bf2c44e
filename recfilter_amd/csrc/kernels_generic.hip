// kernels_generic.hip -- shape-agnostic gfx950 kernels.
//
//  * untiled_scan      one serial recurrence per line: the operator of
//                      RecFilter::add_filter (lib/recfilter.cpp:302-343) executed literally.
//                      This is what the reference runs for a filter that was never split()
//                      (gpu_auto_full_schedule, lib/recfilter.cpp:692-760).
//  * generic_pass1/pass2, carry_apply, gather_incoming
//                      the tiled algorithm of lib/split.cpp for ONE dimension at a time with a
//                      run-time tile width: intra-tile scans + tail extraction
//                      (create_intra_tile_term :503-665, extract_tails_from_each_scan :256-499),
//                      cross-tile carry recurrence with same-dimension chaining
//                      (create_complete_tail_term :743-867, create_tail_residual_term :912-1004)
//                      and the final correction pass (create_final_residual_term :1008-1130,
//                      add_residuals_to_final_result :1647-1780).
//
// These are the always-correct paths (any extent, any tile that divides it, any pixel type,
// order <= RF_MAX_ORDER).  The bandwidth-tuned path is kernels_fused.hip.
#include <type_traits>

#include "kernels.h"
#include "plan_clamp1d.h"

namespace rf {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int64_t line_base(const LineGeom &g, int64_t line) {
    return (line / g.inner) * g.n * g.inner + (line % g.inner);
}

// One in-tile scan step shared by every kernel in this file.  hist[j] is the output at
// direction position p-1-j (from this tile or from the carry), y0 the first output of a
// clamped border tile.
// KMAX: the order bucket the kernel is instantiated for -- kLowOrder (8) for the orders every tiled path takes, RF_MAX_ORDER
// (32) for the direct form of the reference's high-order sweeps (apps/audio/audio_filter_high_order.cpp:38-42); the
// register window of the recurrence is KMAX deep.
constexpr int kLowOrder = 8;
template <typename Acc, int KMAX>
__device__ __forceinline__ Acc scan_step(Acc x, int p, const DevScan<Acc> &sc, int k, bool clamp_first,
                                         Acc (&hist)[KMAX], Acc &y0) {
    Acc acc = sc.b * x;
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        if (j < k) {
            Acc g = hist[j];
            if (clamp_first && p <= j) g = (p == 0) ? x : y0;
            acc = acc + sc.a[j] * g;
        }
    }
#pragma unroll
    for (int j = KMAX - 1; j > 0; j--) hist[j] = hist[j - 1];
    hist[0] = acc;
    if (p == 0) y0 = acc;
    return acc;
}

// ---------------------------------------------------------------------------------------
template <typename P, int KMAX>
__global__ void __launch_bounds__(kBlock)
untiled_scan_kernel(const P *__restrict__ in, P *__restrict__ out, LineGeom g,
                    DevScan<typename PixelTraits<P>::Acc> sc, int clamped) {
    using Tr = PixelTraits<P>;
    using Acc = typename Tr::Acc;
    int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= g.lines) return;
    int64_t base = line_base(g, line);
    Acc hist[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) hist[j] = Acc(0);
    Acc y0 = Acc(0);
    for (int64_t r = 0; r < g.n; r++) {
        int64_t i = sc.causal ? r : g.n - 1 - r;
        int p = r < KMAX ? (int)r : KMAX;  // only p <= j matters
        Acc x = Tr::load(in[base + i * g.inner]);
        Acc y = scan_step<Acc, KMAX>(x, p, sc, sc.order, clamped != 0, hist, y0);
        out[base + i * g.inner] = Tr::store(y);
    }
}

// ---------------------------------------------------------------------------------------
// Stand-alone pointwise stage (rf_pointwise_desc) for the paths that do not fuse it:
//   dst = c0 * f + c1 * x + c2          (x may be null when c1 == 0)
template <typename P, typename X>
__global__ void __launch_bounds__(kBlock)
pointwise_kernel(const P *f, const X *x, P *dst, int64_t n, P c0, P c1, P c2) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        P v = c2;
        if (f) v = v + c0 * f[i];
        if (x) v = v + c1 * (P)x[i];
        dst[i] = v;
    }
}

// ---------------------------------------------------------------------------------------
// rf_box_difference: out = D_z^oz D_y^oy D_x^ox s with D(i) = (s(min(i+B, N-1)) - s(max(i-B-1, 0))) / (2B+1) and the
// nested clamps of apps/box/box_filter.h:128-139 kept as written.  Every application of D doubles the number of
// taps: per dimension up to four (position, sign) pairs, combined as a product over the dimensions.  A gather over a
// summed-area table: the taps of neighbouring outputs are neighbouring table entries, so the reads coalesce and hit
// L2; the kernel is bound by its one write per sample.
constexpr int kBoxRows = 16;      // rows one workgroup walks (amortises its setup; a row per workgroup is launch-bound;
                                  // taller strips measured no faster: the re-read rows miss L1/L2 either way)

template <typename P>
__global__ void __launch_bounds__(kBlock)
box_difference_kernel(const P *__restrict__ in, P *__restrict__ out, BoxDiffArgs a) {
    // grid = (x blocks, groups of kBoxRows rows, z): no index division in the kernel; the x taps are computed once per
    // thread, the y / z taps are wave-uniform (scalar), all in 32 bits (extents are checked on the host)
    const int c0 = (int)(blockIdx.x * kBlock + threadIdx.x), c2 = (int)blockIdx.z;
    const int n0 = (int)a.n[0], n1 = (int)a.n[1], n2 = (int)a.n[2];
    if (c0 >= n0) return;
    const int B = a.radius;
    P inv = P(1);
    for (int d = 0; d < RF_MAX_DIMS; d++)
        for (int o = 0; o < a.order[d]; o++) inv = inv / P(2 * B + 1);
    // tap k of a dimension: order 0 -> the sample itself; order 1 -> (up, dn); order 2 -> (up up, dn up, up dn, dn dn)
    // with signs (+), (+,-), (+,-,-,+)
    auto tap = [&](int c, int N, int order, int k) -> int {
        auto up = [&](int i) { return i + B < N - 1 ? i + B : N - 1; };
        auto dn = [&](int i) { return i - B - 1 > 0 ? i - B - 1 : 0; };
        if (order == 0) return c;
        if (order == 1) return k == 0 ? up(c) : dn(c);
        const int first = (k & 2) ? dn(c) : up(c);              // inner application
        return (k & 1) ? dn(first) : up(first);                  // outer application
    };
    const int nkx = 1 << a.order[0], nky = 1 << a.order[1], nkz = 1 << a.order[2];
    uint32_t px[4];
#pragma unroll
    for (int kx = 0; kx < 4; kx++) px[kx] = (uint32_t)tap(c0, n0, a.order[0], kx < nkx ? kx : 0);
    const int row_begin = (int)blockIdx.y * kBoxRows;
#pragma unroll 4
    for (int rr = 0; rr < kBoxRows; rr++) {
        const int c1 = row_begin + rr;
        if (c1 >= n1) break;
        P acc = P(0);
        for (int kz = 0; kz < nkz; kz++) {
            const int64_t p2 = tap(c2, n2, a.order[2], kz);
            for (int ky = 0; ky < nky; ky++) {
                const int64_t p1 = tap(c1, n1, a.order[1], ky);
                const P *row = in + (p2 * n1 + p1) * n0;         // wave-uniform
                const bool negyz = ((ky == 1 || ky == 2) ? 1 : 0) ^ ((kz == 1 || kz == 2) ? 1 : 0);
                P part = row[px[0]];
                if (nkx > 1) part = part - row[px[1]];
                if (nkx > 2) part = part - row[px[2]] + row[px[3]];
                acc = negyz ? acc - part : acc + part;
            }
        }
        __builtin_nontemporal_store(acc * inv, out + (((int64_t)c2 * n1 + c1) * n0 + c0));      // written once, streamed
    }
}

// The same operator for 2-D tables as a stream: a workgroup owns 256 columns, walks a strip of rows top to bottom and
// keeps the last few table rows in an LDS ring -- the raw row, every sample loaded ONCE (aligned, coalesced), plus a
// halo of order_x * (2B + 1) columns fetched by the first threads.  The x taps of a sample are ring entries a few
// columns to the left and right (what other threads loaded), the y taps are other ring rows; an output is a signed sum
// of 1..16 ring entries.  One barrier per block of rows.  The gather kernel above reads every table row once per tap:
// rows 2B+1 apart do not survive in L1/L2 between their uses at 16384^2 (PMC: 2.1 GB fetched per launch for the 1.07 GB
// table) and every x tap is a separate, misaligned load.  [1,1] 0.86 -> 0.61 ms, [0,2] 1.34 -> 0.70 ms at 16384^2.
// The rows of the next block are requested before the current block's outputs are computed.
// barrier that orders the workgroup's LDS traffic only: __syncthreads() also drains the outstanding global loads
// (s_waitcnt vmcnt(0)), which here are the next block's rows, requested early precisely to stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename P, int BATCH, bool SHARED>      // SHARED: x taps -> threads read columns other threads loaded
__global__ void __launch_bounds__(kBlock)
box_difference_stream_kernel(const P *__restrict__ in, P *__restrict__ out, BoxDiffArgs a, int strip_rows, int ring_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char box_lds[];
    P *ring = reinterpret_cast<P *>(box_lds);
    const int t = (int)threadIdx.x;
    const int n0 = (int)a.n[0], n1 = (int)a.n[1];
    const int x0 = (int)(blockIdx.x * kBlock), c2 = (int)blockIdx.z;
    const bool active = x0 + t < n0;
    const int c0 = active ? x0 + t : n0 - 1;                      // idle lanes shadow the last column (they take part in the barriers)
    const int B = a.radius;
    P inv = P(1);
    for (int d = 0; d < 2; d++)
        for (int o = 0; o < a.order[d]; o++) inv = inv / P(2 * B + 1);
    auto tap = [&](int c, int N, int order, int k) -> int {
        auto up = [&](int i) { return i + B < N - 1 ? i + B : N - 1; };
        auto dn = [&](int i) { return i - B - 1 > 0 ? i - B - 1 : 0; };
        if (order == 0) return c;
        if (order == 1) return k == 0 ? up(c) : dn(c);
        const int first = (k & 2) ? dn(c) : up(c);
        return (k & 1) ? dn(first) : up(first);
    };
    const int ordx = a.order[0], ord = a.order[1];
    const int nkx = 1 << ordx, nky = 1 << ord;
    // ring row = [left halo | the workgroup's 256 columns | right halo]; column c sits at c - x0 + halo_l
    const int halo_l = ordx * (B + 1), halo_r = ordx * B;
    const int pitch = SHARED ? kBlock + halo_l + halo_r : kBlock;
    int ix[4];                                                    // ring positions of this sample's x taps
#pragma unroll
    for (int kx = 0; kx < 4; kx++) ix[kx] = tap(c0, n0, ordx, kx < nkx ? kx : 0) - x0 + halo_l;
    // the halo column this thread fetches besides its own (clamped into the image; never referenced when outside)
    const bool has_halo = t < halo_l + halo_r;
    int hc = t < halo_l ? x0 - halo_l + t : x0 + kBlock + (t - halo_l);
    const int hslot = t < halo_l ? t : kBlock + t;                // = hc - x0 + halo_l
    hc = hc < 0 ? 0 : (hc > n0 - 1 ? n0 - 1 : hc);
    const P *plane = in + (int64_t)c2 * n1 * n0;
    P *oplane = out + (int64_t)c2 * n1 * n0;
    auto load_row = [&](int r, P &main, P &halo) {
        const P *row = plane + (int64_t)r * n0;                      // wave-uniform
        main = row[c0];
        halo = P(0);
        if constexpr (SHARED) {
            if (has_halo) halo = row[hc];
        }
    };
    auto store_row = [&](int r, P main, P halo) {
        P *dst = ring + (size_t)(r & ring_mask) * pitch;
        dst[halo_l + t] = main;
        if constexpr (SHARED) {
            if (has_halo) dst[hslot] = halo;
        }
    };
    auto h_of = [&](int r) -> P {                                    // x difference of table row r at this column
        const P *src = ring + (size_t)(r & ring_mask) * pitch;
        if constexpr (!SHARED) return src[t];
        P part = src[ix[0]];
        if (nkx > 1) part = part - src[ix[1]];
        if (nkx > 2) part = part - src[ix[2]] + src[ix[3]];
        return part;
    };
    const int reach_up = ord * B, reach_dn = ord * (B + 1);          // how far the nested y taps of a row can lie
    const int y0 = (int)blockIdx.y * strip_rows;
    const int y1 = y0 + strip_rows < n1 ? y0 + strip_rows : n1;
    int next = y0 - reach_dn > 0 ? y0 - reach_dn : 0;                 // first table row not yet in the ring
    auto need_hi_of = [&](int cb) {                                    // last table row the outputs [cb, cb + BATCH) read
        const int ce = cb + BATCH < y1 ? cb + BATCH : y1;
        return ce - 1 + reach_up < n1 - 1 ? ce - 1 + reach_up : n1 - 1;
    };
    // prime the ring with everything the first block of outputs needs
    {
        const int need_hi = need_hi_of(y0);
        while (next <= need_hi) {
            P v[BATCH], w[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; i++) load_row(next + i <= need_hi ? next + i : need_hi, v[i], w[i]);
#pragma unroll
            for (int i = 0; i < BATCH; i++)
                if (next + i <= need_hi) store_row(next + i, v[i], w[i]);
            next = next + BATCH <= need_hi ? next + BATCH : need_hi + 1;
        }
    }
    if (SHARED) lds_barrier();                 // without x taps a thread only reads its own column: no barriers
    for (int cb = y0; cb < y1; cb += BATCH) {
        const int ce = cb + BATCH < y1 ? cb + BATCH : y1;
        // request the rows of the NEXT block now (at most BATCH new ones), use them after this block's outputs
        const int next_hi = cb + BATCH < y1 ? need_hi_of(cb + BATCH) : next - 1;
        P v[BATCH], w[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; i++) load_row(next + i <= next_hi ? next + i : next - 1, v[i], w[i]);   // next >= 1 here; the filler row is cached
        for (int c1 = cb; c1 < ce; c1++) {
            P acc = P(0);
            for (int ky = 0; ky < nky; ky++) {
                const P h = h_of(tap(c1, n1, ord, ky));
                acc = (ky == 1 || ky == 2) ? acc - h : acc + h;
            }
            if (active) __builtin_nontemporal_store(acc * inv, oplane + ((int64_t)c1 * n0 + c0));
        }
        if (SHARED) lds_barrier();      // the new rows replace rows other threads may still be reading for this block
#pragma unroll
        for (int i = 0; i < BATCH; i++)
            if (next + i <= next_hi) store_row(next + i, v[i], w[i]);
        next = next_hi + 1 > next ? next_hi + 1 : next;
        if (SHARED) lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------
// Generic tiled path.  tails index: ((s*M + t)*k + r)*lines + line
template <typename Acc>
__device__ __forceinline__ int64_t tail_idx(const GenericDimArgs<Acc> &a, int s, int t, int r, int64_t line) {
    return (((int64_t)s * a.M + t) * a.k + r) * a.g.lines + line;
}

template <typename Acc>
__device__ __forceinline__ bool tile_is_first(const GenericDimArgs<Acc> &a, bool causal, int t) {
    return causal ? (t == 0) : (t == a.M - 1);
}
// the tile is the image's first tile in the scan direction (clamped prologue applies)
template <typename Acc>
__device__ __forceinline__ bool tile_is_border(const GenericDimArgs<Acc> &a, bool causal, int t) {
    return causal ? (t == 0 && a.first_is_border) : (t == a.M - 1 && a.last_is_border);
}
template <typename Acc>
__device__ __forceinline__ int tile_variant(const GenericDimArgs<Acc> &a, int t) {
    return ((t == 0 && a.first_is_border) ? 1 : 0) | ((t == a.M - 1 && a.last_is_border) ? 2 : 0);
}

// carry entering tile t for scan s: complete tail of the previous tile, or the slab's incoming carry
template <typename Acc, int KMAX>
__device__ __forceinline__ void load_carry(const GenericDimArgs<Acc> &a, int s, bool causal, int t, int64_t line,
                                           Acc (&c)[KMAX]) {
#pragma unroll
    for (int j = 0; j < KMAX; j++) c[j] = Acc(0);
    if (tile_is_first(a, causal, t)) {
        for (int j = 0; j < a.k; j++) c[j] = a.incoming[((int64_t)s * a.k + j) * a.g.lines + line];
    } else {
        int tp = causal ? t - 1 : t + 1;
        for (int j = 0; j < a.k; j++) c[j] = a.tails[tail_idx(a, s, tp, j, line)];
    }
}

template <typename P, bool kFinal, int KMAX>
__global__ void __launch_bounds__(kBlock)
generic_pass_kernel(const P *__restrict__ src, P *__restrict__ dst, GenericDimArgs<typename PixelTraits<P>::Acc> a) {
    using Tr = PixelTraits<P>;
    using Acc = typename Tr::Acc;
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= a.g.lines * a.M) return;
    int64_t line = idx % a.g.lines;
    int t = (int)(idx / a.g.lines);
    int64_t base = line_base(a.g, line) + (int64_t)t * a.T * a.g.inner;

    Acc v[kGenericMaxTile];
    for (int m = 0; m < a.T; m++) v[m] = Tr::load(src[base + m * a.g.inner]);

    for (int s = 0; s < a.n_scans; s++) {
        DevScan<Acc> sc = a.scans[s];
        bool causal = sc.causal != 0;
        bool clamp_first = a.clamped && tile_is_border(a, causal, t);
        Acc hist[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; j++) hist[j] = Acc(0);
        if (kFinal) load_carry<Acc, KMAX>(a, s, causal, t, line, hist);
        Acc y0 = Acc(0);
        for (int p = 0; p < a.T; p++) {
            int m = causal ? p : a.T - 1 - p;
            v[m] = scan_step<Acc, KMAX>(v[m], p < KMAX ? p : KMAX, sc, a.k, clamp_first, hist, y0);
        }
        if (!kFinal) {
            // tail r = value at direction position T-1-r, extracted right after the scan
            for (int r = 0; r < a.k; r++) {
                int p = a.T - 1 - r;
                int m = causal ? p : a.T - 1 - p;
                a.tails[tail_idx(a, s, t, r, line)] = v[m];
            }
        }
    }
    if (kFinal) {
        for (int m = 0; m < a.T; m++) dst[base + m * a.g.inner] = Tr::store(v[m]);
    }
}

// add the effect of the slab's incoming carry to every tile's complete tail
template <typename Acc, int KMAX>
__global__ void __launch_bounds__(kBlock)
generic_carry_apply_kernel(GenericDimArgs<Acc> a, int s) {
    int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int k = a.k;
    const bool causal = a.scans[s].causal != 0;
    const Acc *A = a.A + (int64_t)s * k * k;
    Acc x[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) x[j] = Acc(0);
    for (int j = 0; j < k; j++) x[j] = a.incoming[((int64_t)s * k + j) * a.g.lines + line];
    for (int i = 0; i < a.M; i++) {
        int t = causal ? i : a.M - 1 - i;
        Acc nx[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; j++) nx[j] = Acc(0);
        for (int r = 0; r < k; r++)
            for (int j = 0; j < k; j++) nx[r] = nx[r] + A[r * k + j] * x[j];
        for (int r = 0; r < k; r++) {
            a.tails[tail_idx(a, s, t, r, line)] = a.tails[tail_idx(a, s, t, r, line)] + nx[r];
            x[r] = nx[r];
        }
    }
}

// The cross-tile recurrence of orders above kLowOrder on this path (integer and f64 pixels, shapes the matrix path of
// kernels_matrix.hip does not take): one thread per line walks the tiles in scan direction -- the reference's own schedule
// (gpu_auto_inter_schedule, lib/recfilter.cpp:763-785) -- for the scans [s_begin, s_end) of the dimension:
//     tail_s[t] += sum_{q<s} W[v][q][s] * (carry entering tile t of scan q) + A_s * tail_s[t -/+ 1]
// with zero state entering the slab (what the slab's incoming carry adds is carry_apply's), exit carries to `send`.
// The blocked parallel scan of kernels_carry.hip keeps its k-vectors in registers and stops at kLowOrder.
template <typename Acc>
__global__ void __launch_bounds__(kBlock)
generic_carry_serial_kernel(GenericDimArgs<Acc> a, uint32_t causal_mask, int s_begin, int s_end, Acc *__restrict__ send) {
    const int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int k = a.k, M = a.M, n = a.n_scans;
    const int64_t L = a.g.lines;
    for (int s = s_begin; s < s_end; s++) {
        const bool causal = ((causal_mask >> s) & 1u) != 0;
        const Acc *Am = a.A + (int64_t)s * k * k;
        Acc x[RF_MAX_ORDER];
        for (int j = 0; j < k; j++) x[j] = Acc(0);
        for (int i = 0; i < M; i++) {
            const int tt = causal ? i : M - 1 - i;
            const int v = tile_variant(a, tt);
            Acc cur[RF_MAX_ORDER];
            for (int r = 0; r < k; r++) cur[r] = a.tails[tail_idx(a, s, tt, r, line)];
            for (int q = 0; q < s; q++) {
                const bool qc = ((causal_mask >> q) & 1u) != 0;
                const bool q_first = qc ? (tt == 0) : (tt == M - 1);
                const int tp = qc ? tt - 1 : tt + 1;
                const Acc *Wm = a.W + ((((int64_t)v * n + q) * n + s) * k) * k;
                for (int o = 0; o < k; o++) {
                    const Acc c = q_first ? a.incoming[((int64_t)q * k + o) * L + line] : a.tails[tail_idx(a, q, tp, o, line)];
                    for (int r = 0; r < k; r++) cur[r] = cur[r] + Wm[r * k + o] * c;
                }
            }
            for (int r = 0; r < k; r++) {
                Acc acc = cur[r];
                for (int j = 0; j < k; j++) acc = acc + Am[r * k + j] * x[j];
                cur[r] = acc;
            }
            for (int r = 0; r < k; r++) {
                x[r] = cur[r];
                a.tails[tail_idx(a, s, tt, r, line)] = cur[r];
            }
        }
        if (send != nullptr)
            for (int r = 0; r < k; r++) send[((int64_t)(s - s_begin) * k + r) * L + line] = x[r];
    }
}

// The same update with every tile independent: tail(t) += A^(i+1) * incoming, the powers tabulated on the host
// (GenericDimArgs::Apow).  Thread = line, blockIdx.y = a chunk of kApplyTiles tiles; the matrix of a tile is
// wave-uniform (scalar loads), the tails are read and written once, coalesced across lines.  The serial kernel above
// walks the tiles of a line in one thread -- a chain of dependent read-modify-writes that took 0.33 ms per scan on a
// cfg3 slab; this one is bound by the tails' traffic.
constexpr int kApplyTiles = 8;
template <typename Acc, int KMAX>
__global__ void __launch_bounds__(kBlock)
carry_apply_parallel_kernel(GenericDimArgs<Acc> a, int s) {
    const int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int k = a.k;
    const bool causal = a.scans[s].causal != 0;
    Acc x[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) x[j] = Acc(0);
    for (int j = 0; j < k; j++) x[j] = a.incoming[((int64_t)s * k + j) * a.g.lines + line];
    const int i0 = (int)blockIdx.y * kApplyTiles;
    // all loads of the chunk first, then the stores: interleaved, every store could alias the next load and the
    // read-modify-writes would run one after the other
    for (int r = 0; r < k; r++) {
        Acc cur[kApplyTiles];
#pragma unroll
        for (int u = 0; u < kApplyTiles; u++) {
            const int i = i0 + u;
            cur[u] = Acc(0);
            if (i < a.M) cur[u] = a.tails[tail_idx(a, s, causal ? i : a.M - 1 - i, r, line)];
        }
#pragma unroll
        for (int u = 0; u < kApplyTiles; u++) {
            const int i = i0 + u;
            if (i < a.M) {
                const Acc *Ap = a.Apow + ((int64_t)s * a.M + i) * k * k;          // wave-uniform
                Acc add = Acc(0);
                for (int j = 0; j < k; j++) add = add + Ap[r * k + j] * x[j];
                a.tails[tail_idx(a, s, causal ? i : a.M - 1 - i, r, line)] = cur[u] + add;
            }
        }
    }
}

template <typename Acc, int KMAX>
__global__ void __launch_bounds__(kBlock)
gather_incoming_kernel(GenericDimArgs<Acc> a, int s, const Acc *__restrict__ gathered, int64_t rank_stride,
                       int64_t plane_offset, int rank, int world, const Acc *__restrict__ AM) {
    int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int k = a.k;
    const bool causal = a.scans[s].causal != 0;
    Acc x[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) x[j] = Acc(0);
    // walk the slabs that precede this one in scan direction, nearest last
    int count = causal ? rank : world - 1 - rank;
    for (int i = 0; i < count; i++) {
        int h = causal ? i : world - 1 - i;
        Acc nx[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; j++) nx[j] = Acc(0);
        for (int r = 0; r < k; r++) {
            Acc acc = gathered[h * rank_stride + plane_offset + (int64_t)r * a.g.lines + line];
            for (int j = 0; j < k; j++) acc = acc + AM[(h * k + r) * k + j] * x[j];      // AM[h] = A^(tiles of slab h)
            nx[r] = acc;
        }
        for (int r = 0; r < k; r++) x[r] = nx[r];
    }
    for (int r = 0; r < k; r++) a.incoming[((int64_t)s * k + r) * a.g.lines + line] = x[r];
}

// ---- merged exchange: one all-gather for all scans of the sharded dimension ---------------------------------
// Every slab has completed its scans with zero entering carries and published the exit carry of each
// (gathered[h][plane][s][r][line]).  The true exit of slab h for scan s is
//     E_s[h] = E_s^local[h] + sum_{q <= s} X[h][q][s] * in_q[h]
// (X = the exit-tile rows of the cross-scan transfer Y, plan_generic.h), and in_s of the next slab in scan
// direction is E_s[h].  One thread per line walks the slabs for every scan in order; the carries entering EVERY
// slab are kept (the cross terms of later scans need them), n * world * k values in a private array.
constexpr int kMergeMaxState = 128;
template <typename Acc, int KMAX>
__global__ void __launch_bounds__(kBlock)
merged_gather_kernel(GenericDimArgs<Acc> a, const Acc *__restrict__ gathered, int64_t rank_stride, int64_t plane_offset,
                     int rank, int world, const Acc *__restrict__ X) {
    const int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int k = a.k, n = a.n_scans;
    const int64_t L = a.g.lines;
    Acc in[kMergeMaxState];
    for (int s = 0; s < n; s++) {
        const bool causal = a.scans[s].causal != 0;
        Acc prev[KMAX];
        for (int j = 0; j < k; j++) prev[j] = Acc(0);
        for (int i = 0; i < world; i++) {
            const int h = causal ? i : world - 1 - i;
            for (int j = 0; j < k; j++) in[(s * world + h) * k + j] = prev[j];
            if (i == world - 1) break;        // nobody follows the last slab
            Acc e[KMAX];
            for (int r = 0; r < k; r++) {
                Acc acc = gathered[h * rank_stride + plane_offset + ((int64_t)s * k + r) * L + line];
                for (int q = 0; q <= s; q++) {
                    const Acc *Xm = X + (((int64_t)h * n + q) * n + s) * k * k;
                    for (int j = 0; j < k; j++) acc = acc + Xm[r * k + j] * in[(q * world + h) * k + j];
                }
                e[r] = acc;
            }
            for (int r = 0; r < k; r++) prev[r] = e[r];
        }
        for (int r = 0; r < k; r++) a.incoming[((int64_t)s * k + r) * L + line] = in[(s * world + rank) * k + r];
    }
}

// The same walk with order, scan count and a bound on the slab count known at compile time (one node: world <= 8):
// every index is static, so the entering carries of all slabs stay in registers (the kernel above keeps them in a
// dynamically indexed private array, i.e. scratch memory: 25 us on a cfg3 slab), and all exits are requested before
// the first dependent multiply.  Same summation order as above.
constexpr int kMergeRegWorld = 8;
template <typename Acc, int K, int NS>
__global__ void __launch_bounds__(kBlock)
merged_gather_reg_kernel(GenericDimArgs<Acc> a, const Acc *__restrict__ gathered, int64_t rank_stride, int64_t plane_offset,
                         int rank, int world, const Acc *__restrict__ X) {
    constexpr int W = kMergeRegWorld;
    const int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int64_t L = a.g.lines;
    Acc ex[NS][W][K], in[NS][W][K];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int h = 0; h < W; h++)
#pragma unroll
            for (int r = 0; r < K; r++) {
                in[s][h][r] = Acc(0);
                ex[s][h][r] = h < world ? gathered[h * rank_stride + plane_offset + ((int64_t)s * K + r) * L + line] : Acc(0);
            }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const bool causal = a.scans[s].causal != 0;
        Acc prev[K];
#pragma unroll
        for (int j = 0; j < K; j++) prev[j] = Acc(0);
        // slab h of the walk (ascending for a causal scan, descending otherwise; h is a constant after unrolling): the
        // carry entering it is what the slabs before it in scan direction handed on
        auto walk = [&](auto ascending) {
#pragma unroll
            for (int i = 0; i < W; i++) {
                const int h = decltype(ascending)::value ? i : W - 1 - i;
                if (h < world) {
#pragma unroll
                    for (int j = 0; j < K; j++) in[s][h][j] = prev[j];
                    Acc e[K];
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        Acc acc = ex[s][h][r];
#pragma unroll
                        for (int q = 0; q <= s; q++) {
                            const Acc *Xm = X + (((int64_t)h * NS + q) * NS + s) * K * K;
#pragma unroll
                            for (int j = 0; j < K; j++) acc = acc + Xm[r * K + j] * in[q][h][j];
                        }
                        e[r] = acc;
                    }
                    // (the exit of the last slab in scan direction enters nobody: prev is not read again)
#pragma unroll
                    for (int r = 0; r < K; r++) prev[r] = e[r];
                }
            }
        };
        if (causal) walk(std::true_type{});
        else walk(std::false_type{});
#pragma unroll
        for (int r = 0; r < K; r++) {
            Acc mine = Acc(0);
#pragma unroll
            for (int h = 0; h < W; h++) mine = (h == rank) ? in[s][h][r] : mine;
            a.incoming[((int64_t)s * K + r) * L + line] = mine;
        }
    }
}

// tails_s[t] += sum_{q <= s} Y[q][s][t] * in_q for every scan and tile of the slab: one pass over the tails, parallel
// over lines and chunks of tiles (memory order).  K and the scan count are compile-time so the entering carries stay
// in registers.
template <typename Acc, int K, int NS>
__global__ void __launch_bounds__(kBlock)
merged_apply_kernel(GenericDimArgs<Acc> a, const Acc *__restrict__ Y) {
    const int64_t line = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (line >= a.g.lines) return;
    const int64_t L = a.g.lines;
    const int M = a.M;
    Acc in[NS][K];
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
        for (int j = 0; j < K; j++) in[q][j] = a.incoming[((int64_t)q * K + j) * L + line];
    const int t0 = (int)blockIdx.y * kApplyTiles;
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int r = 0; r < K; r++) {
            Acc cur[kApplyTiles];
#pragma unroll
            for (int u = 0; u < kApplyTiles; u++) {
                cur[u] = Acc(0);
                if (t0 + u < M) cur[u] = a.tails[tail_idx(a, s, t0 + u, r, line)];
            }
#pragma unroll
            for (int u = 0; u < kApplyTiles; u++) {
                const int t = t0 + u;
                if (t < M) {
                    Acc add = Acc(0);
#pragma unroll
                    for (int q = 0; q <= s; q++) {
                        const Acc *Ym = Y + ((((int64_t)q * NS + s) * M + t) * K + r) * K;      // wave-uniform
#pragma unroll
                        for (int j = 0; j < K; j++) add = add + Ym[j] * in[q][j];
                    }
                    a.tails[tail_idx(a, s, t, r, line)] = cur[u] + add;
                }
            }
        }
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

template <typename P>
int launch_untiled_scan(const P *in, P *out, LineGeom g, const DevScan<typename PixelTraits<P>::Acc> &sc,
                        bool clamped, hipStream_t stream) {
    if (g.lines <= 0 || g.n <= 0) return RF_OK;
    if (sc.order <= kLowOrder)
        hipLaunchKernelGGL((untiled_scan_kernel<P, kLowOrder>), dim3(grid_for(g.lines)), dim3(kBlock), 0, stream, in, out, g, sc,
                           clamped ? 1 : 0);
    else
        hipLaunchKernelGGL((untiled_scan_kernel<P, RF_MAX_ORDER>), dim3(grid_for(g.lines)), dim3(kBlock), 0, stream, in, out, g, sc,
                           clamped ? 1 : 0);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename P>
int launch_generic_pass1(const P *src, GenericDimArgs<typename PixelTraits<P>::Acc> a, hipStream_t stream) {
    if (a.k <= kLowOrder)
        hipLaunchKernelGGL((generic_pass_kernel<P, false, kLowOrder>), dim3(grid_for(a.g.lines * a.M)), dim3(kBlock), 0, stream, src,
                           (P *)nullptr, a);
    else
        hipLaunchKernelGGL((generic_pass_kernel<P, false, RF_MAX_ORDER>), dim3(grid_for(a.g.lines * a.M)), dim3(kBlock), 0, stream, src,
                           (P *)nullptr, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename P>
int launch_generic_pass2(const P *src, P *dst, GenericDimArgs<typename PixelTraits<P>::Acc> a, hipStream_t stream) {
    if (a.k <= kLowOrder)
        hipLaunchKernelGGL((generic_pass_kernel<P, true, kLowOrder>), dim3(grid_for(a.g.lines * a.M)), dim3(kBlock), 0, stream, src,
                           dst, a);
    else
        hipLaunchKernelGGL((generic_pass_kernel<P, true, RF_MAX_ORDER>), dim3(grid_for(a.g.lines * a.M)), dim3(kBlock), 0, stream, src,
                           dst, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename Acc>
int launch_generic_carry_apply(GenericDimArgs<Acc> a, int s, hipStream_t stream) {
    if (a.Apow != nullptr && a.M > 0) {
        dim3 grid(grid_for(a.g.lines), (unsigned)((a.M + kApplyTiles - 1) / kApplyTiles));
        if (a.k <= kLowOrder) hipLaunchKernelGGL((carry_apply_parallel_kernel<Acc, kLowOrder>), grid, dim3(kBlock), 0, stream, a, s);
        else hipLaunchKernelGGL((carry_apply_parallel_kernel<Acc, RF_MAX_ORDER>), grid, dim3(kBlock), 0, stream, a, s);
    } else {
        if (a.k <= kLowOrder) hipLaunchKernelGGL((generic_carry_apply_kernel<Acc, kLowOrder>), dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, s);
        else hipLaunchKernelGGL((generic_carry_apply_kernel<Acc, RF_MAX_ORDER>), dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, s);
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename Acc>
int launch_generic_carry_serial(GenericDimArgs<Acc> a, uint32_t causal_mask, int s_begin, int s_end, Acc *send, hipStream_t stream) {
    if (a.g.lines <= 0 || a.M <= 0 || s_end <= s_begin) return RF_OK;
    hipLaunchKernelGGL(generic_carry_serial_kernel<Acc>, dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, causal_mask, s_begin,
                       s_end, send);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename Acc>
int launch_gather_incoming(GenericDimArgs<Acc> a, int s, const Acc *gathered, int64_t rank_stride,
                           int64_t plane_offset, int rank, int world, const Acc *AM, hipStream_t stream) {
    if (a.k <= kLowOrder)
        hipLaunchKernelGGL((gather_incoming_kernel<Acc, kLowOrder>), dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, s, gathered,
                           rank_stride, plane_offset, rank, world, AM);
    else
        hipLaunchKernelGGL((gather_incoming_kernel<Acc, RF_MAX_ORDER>), dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, s, gathered,
                           rank_stride, plane_offset, rank, world, AM);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename Acc>
int launch_merged_gather(GenericDimArgs<Acc> a, const Acc *gathered, int64_t rank_stride, int64_t plane_offset, int rank,
                         int world, const Acc *X, hipStream_t stream) {
    if (world <= kMergeRegWorld) {
#define RF_CASE(KK, NN)                                                                                            \
    if (a.k == KK && a.n_scans == NN) {                                                                            \
        hipLaunchKernelGGL((merged_gather_reg_kernel<Acc, KK, NN>), dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, \
                           gathered, rank_stride, plane_offset, rank, world, X);                                   \
        RF_HIP_CHECK(hipGetLastError());                                                                           \
        return RF_OK;                                                                                              \
    }
        RF_CASE(1, 1) RF_CASE(1, 2) RF_CASE(1, 3) RF_CASE(1, 4)
        RF_CASE(2, 1) RF_CASE(2, 2) RF_CASE(2, 3) RF_CASE(2, 4)
        RF_CASE(3, 1) RF_CASE(3, 2) RF_CASE(3, 3) RF_CASE(3, 4)
#undef RF_CASE
    }
    if (a.n_scans * world * a.k > kMergeMaxState) { set_error("merged exchange: too many carries per line"); return RF_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL((merged_gather_kernel<Acc, kLowOrder>), dim3(grid_for(a.g.lines)), dim3(kBlock), 0, stream, a, gathered,
                       rank_stride, plane_offset, rank, world, X);      // (the merged exchange takes orders <= 3, plan_generic.h)
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename Acc>
int launch_merged_apply(GenericDimArgs<Acc> a, const Acc *Y, hipStream_t stream) {
    if (a.M <= 0) return RF_OK;
    dim3 grid(grid_for(a.g.lines), (unsigned)((a.M + kApplyTiles - 1) / kApplyTiles));
#define RF_CASE(KK, NN)                                                                                            \
    if (a.k == KK && a.n_scans == NN) {                                                                            \
        hipLaunchKernelGGL((merged_apply_kernel<Acc, KK, NN>), grid, dim3(kBlock), 0, stream, a, Y);               \
        RF_HIP_CHECK(hipGetLastError());                                                                           \
        return RF_OK;                                                                                              \
    }
    RF_CASE(1, 1) RF_CASE(1, 2) RF_CASE(1, 3) RF_CASE(1, 4)
    RF_CASE(2, 1) RF_CASE(2, 2) RF_CASE(2, 3) RF_CASE(2, 4)
    RF_CASE(3, 1) RF_CASE(3, 2) RF_CASE(3, 3) RF_CASE(3, 4)
#undef RF_CASE
    set_error("merged exchange: order %d with %d scans not instantiated", a.k, a.n_scans);
    return RF_ERR_UNSUPPORTED;
}

// ---- clamped 1-D signals on the fused kernels (plan_clamp1d.h): the two small launches around the zero-border plan ----
// dots[s] = sum_i w[s][i] * in[window of scan s][i], one workgroup per scan, double accumulation
template <typename P>
__global__ void __launch_bounds__(kBlock)
clamp1d_dots_kernel(const P *__restrict__ in, int64_t N, int L, const int32_t *__restrict__ side, const double *__restrict__ w,
                    double *__restrict__ dots) {
    const int s = blockIdx.x;
    const P *win = side[s] == 0 ? in : in + (N - L);
    double acc = 0.0;
    for (int i = threadIdx.x; i < L; i += kBlock) acc += w[(int64_t)s * L + i] * (double)win[i];
    __shared__ double red[kBlock];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int step = kBlock / 2; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) red[threadIdx.x] += red[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) dots[s] = red[0];
}

// beta_s = dots[s] + sum_{q<s} beta_q H[q][s] (every thread: n <= RF_MAX_SCANS), then out[window] += sum_s beta_s G_s
template <typename P>
__global__ void __launch_bounds__(kBlock)
clamp1d_fix_kernel(P *__restrict__ out, int64_t N, int L, int n, const int32_t *__restrict__ side, const double *__restrict__ H,
                   const double *__restrict__ G, const double *__restrict__ dots) {
    double beta[RF_MAX_SCANS];
    for (int s = 0; s < n; s++) {
        double b = dots[s];
        for (int q = 0; q < s; q++) b += beta[q] * H[q * n + s];
        beta[s] = b;
    }
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);       // [0, 2L): the start window, then the end window
    if (i >= 2 * L) return;
    const int which = i / L, k = i % L;
    double add = 0.0;
    for (int s = 0; s < n; s++)
        if (side[s] == which) add += beta[s] * G[(int64_t)s * L + k];
    P *p = which == 0 ? out + k : out + (N - L) + k;
    *p = (P)((double)*p + add);
}

template <typename P>
int launch_clamp1d_dots(const P *in, int64_t N, int L, int n, const int32_t *side, const double *w, double *dots, hipStream_t stream) {
    hipLaunchKernelGGL(clamp1d_dots_kernel<P>, dim3((unsigned)n), dim3(kBlock), 0, stream, in, N, L, side, w, dots);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
template <typename P>
int launch_clamp1d_fix(P *out, int64_t N, int L, int n, const int32_t *side, const double *H, const double *G, const double *dots,
                       hipStream_t stream) {
    hipLaunchKernelGGL(clamp1d_fix_kernel<P>, dim3((unsigned)((2 * L + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, out, N, L, n, side,
                       H, G, dots);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
template int launch_clamp1d_dots<float>(const float *, int64_t, int, int, const int32_t *, const double *, double *, hipStream_t);
template int launch_clamp1d_fix<float>(float *, int64_t, int, int, const int32_t *, const double *, const double *, const double *, hipStream_t);

template <typename P>
int launch_pointwise(const P *f, const P *x, P *dst, int64_t n, double c0, double c1, double c2, hipStream_t stream) {
    if (n <= 0) return RF_OK;
    const int64_t want = (n + kBlock - 1) / kBlock;
    const unsigned blocks = (unsigned)(want < 256 * 64 ? want : 256 * 64);
    hipLaunchKernelGGL((pointwise_kernel<P, P>), dim3(blocks), dim3(kBlock), 0, stream, f, c1 != 0.0 ? x : (const P *)nullptr, dst, n,
                       (P)c0, (P)c1, (P)c2);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
template <typename P, typename X>
int launch_pointwise_from(const P *f, const X *x, P *dst, int64_t n, double c0, double c1, double c2, hipStream_t stream) {
    if (n <= 0) return RF_OK;
    const int64_t want = (n + kBlock - 1) / kBlock;
    const unsigned blocks = (unsigned)(want < 256 * 64 ? want : 256 * 64);
    hipLaunchKernelGGL((pointwise_kernel<P, X>), dim3(blocks), dim3(kBlock), 0, stream, c0 != 0.0 ? f : (const P *)nullptr, x, dst, n,
                       (P)c0, (P)c1, (P)c2);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
template int launch_pointwise_from<float, uint8_t>(const float *, const uint8_t *, float *, int64_t, double, double, double, hipStream_t);
template int launch_pointwise<float>(const float *, const float *, float *, int64_t, double, double, double, hipStream_t);
template int launch_pointwise<double>(const double *, const double *, double *, int64_t, double, double, double, hipStream_t);

template <typename P>
int launch_box_difference(const P *in, P *out, const BoxDiffArgs &a, hipStream_t stream) {
    if (a.n[0] <= 0 || a.n[1] <= 0 || a.n[2] <= 0) return RF_OK;
    if ((a.n[1] + kBoxRows - 1) / kBoxRows > 65535 || a.n[2] > 65535 || a.n[0] >= (1ll << 31) || a.n[1] >= (1ll << 31)) {
        set_error("box_difference: extents too large");
        return RF_ERR_UNSUPPORTED;
    }
    // 2-D tables: the streaming kernel, when its ring of table rows fits 64 KiB of LDS and a thread has at most one
    // halo column to fetch
    if (a.order[2] == 0 && (a.order[0] > 0 || a.order[1] > 0) && RF_KNOB("RF_BOX_GATHER") == nullptr) {
        const int halo = a.order[0] * (2 * a.radius + 1);
        const int batch = a.order[1] <= 1 ? 16 : 8;      // rows in flight; the second-order window is twice as tall
        const int window = a.order[1] * (2 * a.radius + 1) + batch;
        int ring = 16;
        while (ring < window) ring *= 2;
        const size_t lds = (size_t)ring * (kBlock + halo) * sizeof(P);
        if (halo <= kBlock && lds <= 64 * 1024) {
            // strips as tall as possible (the reach_up + reach_dn rows around a strip are read again by its neighbours)
            // that still give the chip ~1024 workgroups
            const int64_t xb = (a.n[0] + kBlock - 1) / kBlock;
            int strip = 256;
            while (strip > 32 && xb * ((a.n[1] + strip - 1) / strip) * a.n[2] < 1024) strip /= 2;
            dim3 grid((unsigned)xb, (unsigned)((a.n[1] + strip - 1) / strip), (unsigned)a.n[2]);
            const bool shared = a.order[0] > 0;
            if (batch == 16 && shared) hipLaunchKernelGGL((box_difference_stream_kernel<P, 16, true>), grid, dim3(kBlock), lds, stream, in, out, a, strip, ring - 1);
            else if (batch == 16)      hipLaunchKernelGGL((box_difference_stream_kernel<P, 16, false>), grid, dim3(kBlock), lds, stream, in, out, a, strip, ring - 1);
            else if (shared)           hipLaunchKernelGGL((box_difference_stream_kernel<P, 8, true>), grid, dim3(kBlock), lds, stream, in, out, a, strip, ring - 1);
            else                       hipLaunchKernelGGL((box_difference_stream_kernel<P, 8, false>), grid, dim3(kBlock), lds, stream, in, out, a, strip, ring - 1);
            RF_HIP_CHECK(hipGetLastError());
            return RF_OK;
        }
    }
    dim3 grid((unsigned)((a.n[0] + kBlock - 1) / kBlock), (unsigned)((a.n[1] + kBoxRows - 1) / kBoxRows), (unsigned)a.n[2]);
    hipLaunchKernelGGL((box_difference_kernel<P>), grid, dim3(kBlock), 0, stream, in, out, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
// rf_tap_filter: out(p) = sum_t w_t * in[plane_t](clamp(p + off_t)).  One thread per four consecutive x samples; every
// tap of a row is a (shifted, clamped) run of the same row or of a row a few lines away -- L2 serves the re-reads.
template <typename P>
__global__ void __launch_bounds__(kBlock)
tap_filter_kernel(P *__restrict__ out, TapArgs a) {
    const int64_t nx = a.n[0], ny = a.n[1], nz = a.n[2];
    const int64_t xg = (nx + 3) / 4;
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= xg * ny * nz) return;
    const int64_t x0 = (idx % xg) * 4, y = (idx / xg) % ny, z = idx / (xg * ny);
    P acc[4] = {P(0), P(0), P(0), P(0)};
    for (int t = 0; t < a.n_taps; t++) {
        const P *src = reinterpret_cast<const P *>(a.in[a.plane[t]]);
        int64_t yy = y + a.off[t][1], zz = z + a.off[t][2];
        yy = yy < 0 ? 0 : (yy > ny - 1 ? ny - 1 : yy);
        zz = zz < 0 ? 0 : (zz > nz - 1 ? nz - 1 : zz);
        const P *row = src + (zz * ny + yy) * nx;
        const P w = (P)a.weight[t];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int64_t xx = x0 + i + a.off[t][0];
            xx = xx < 0 ? 0 : (xx > nx - 1 ? nx - 1 : xx);
            acc[i] = acc[i] + w * row[xx];
        }
    }
    P *dst = out + (z * ny + y) * nx + x0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (x0 + i < nx) dst[i] = acc[i];
}

template <typename P>
int launch_tap_filter(P *out, const TapArgs &a, hipStream_t stream) {
    const int64_t threads = ((a.n[0] + 3) / 4) * a.n[1] * a.n[2];
    const int64_t blocks = (threads + kBlock - 1) / kBlock;
    if (blocks >= (1ll << 31)) { set_error("tap_filter: extents too large"); return RF_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL((tap_filter_kernel<P>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, out, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
template int launch_tap_filter<float>(float *, const TapArgs &, hipStream_t);
template int launch_tap_filter<double>(double *, const TapArgs &, hipStream_t);

template int launch_box_difference<float>(const float *, float *, const BoxDiffArgs &, hipStream_t);
template int launch_box_difference<double>(const double *, double *, const BoxDiffArgs &, hipStream_t);

// explicit instantiations ----------------------------------------------------------------
#define RF_INSTANTIATE_PIXEL(P)                                                                                    \
    template int launch_untiled_scan<P>(const P *, P *, LineGeom, const DevScan<PixelTraits<P>::Acc> &, bool,      \
                                        hipStream_t);                                                              \
    template int launch_generic_pass1<P>(const P *, GenericDimArgs<PixelTraits<P>::Acc>, hipStream_t);             \
    template int launch_generic_pass2<P>(const P *, P *, GenericDimArgs<PixelTraits<P>::Acc>, hipStream_t);
RF_INSTANTIATE_PIXEL(float)
RF_INSTANTIATE_PIXEL(double)
RF_INSTANTIATE_PIXEL(int32_t)
RF_INSTANTIATE_PIXEL(int16_t)

#define RF_INSTANTIATE_ACC(Acc)                                                                                    \
    template int launch_generic_carry_apply<Acc>(GenericDimArgs<Acc>, int, hipStream_t);                           \
    template int launch_generic_carry_serial<Acc>(GenericDimArgs<Acc>, uint32_t, int, int, Acc *, hipStream_t);    \
    template int launch_gather_incoming<Acc>(GenericDimArgs<Acc>, int, const Acc *, int64_t, int64_t, int, int,    \
                                             const Acc *, hipStream_t);                                            \
    template int launch_merged_gather<Acc>(GenericDimArgs<Acc>, const Acc *, int64_t, int64_t, int, int, const Acc *, \
                                           hipStream_t);                                                           \
    template int launch_merged_apply<Acc>(GenericDimArgs<Acc>, const Acc *, hipStream_t);
RF_INSTANTIATE_ACC(float)
RF_INSTANTIATE_ACC(double)
RF_INSTANTIATE_ACC(uint32_t)

}  // namespace rf
