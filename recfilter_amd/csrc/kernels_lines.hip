// kernels_lines.hip -- the UNTILED operator (lib/recfilter.cpp:302-343: one recurrence over the whole line per scan)
// with the parallelism of the fused x phase inside the line.
//
// untiled_scan_kernel (kernels_generic.hip) is the literal form: one thread walks one line, n dependent steps per
// scan -- the shape of the reference's gpu_auto_full_schedule (lib/recfilter.cpp:690-730).  Here a workgroup owns 16
// lines; a line is walked in tiles of 256 samples, and a tile is scanned by the 16 lanes of a DPP row exactly like a
// tile row of the fused final pass (scan_rows16: 16-sample segment recurrences, Kogge-Stone over the lanes with the
// precomputed segment transfer powers, rank-k correction), the state leaving a tile handed to the next one in
// registers.  No tails, no carry stage, no second pass: ONE launch per filtered dimension for all its scans.
//
// What it is for: images whose five-launch tiled pipeline is launch-bound.  The reference's own benchmark sweep
// (scripts/profile_app.sh:6-19: widths 64 .. 4096) lives there: below ~1024^2 the tiled path costs 30-35 us per
// image, of which 23-26 us is the host enqueueing five dependent launches (tools/small_probe.py); two launches of
// this kernel finish a 512^2 image in a fraction of that.  It also is what RF_PATH_UNTILED runs whenever the shape
// admits it (f32 / i32 / i16 pixels, orders <= 3, <= 4 scans per dimension, extents that are multiples of 16).
// Larger images stay on the tiled paths: this one moves 8 bytes per sample and SCAN (every scan re-reads the line).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

constexpr int kLineTile = kFusedTX;       // samples per tile of the walk: 16 lanes x 16 samples
constexpr int kLinesPerWG = 16;           // lines per workgroup (one per DPP row of the 256 threads)

template <typename P, typename Acc>
__device__ __forceinline__ void load16(const P *p, Acc (&v)[1][kFusedSeg]) {
    if constexpr (sizeof(P) == sizeof(Acc)) {
        using A4 = typename Vec4<Acc>::type;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const A4 w = reinterpret_cast<const A4 *>(p)[j];
            v[0][4 * j + 0] = w.x; v[0][4 * j + 1] = w.y; v[0][4 * j + 2] = w.z; v[0][4 * j + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int m = 0; m < kFusedSeg; m++) v[0][m] = PixelTraits<P>::load(p[m]);
    }
}

template <typename P, typename Acc>
__device__ __forceinline__ void store16(P *p, const Acc (&v)[1][kFusedSeg]) {
    if constexpr (sizeof(P) == sizeof(Acc)) {
        using A4 = typename Vec4<Acc>::type;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            A4 w;
            w.x = v[0][4 * j + 0]; w.y = v[0][4 * j + 1]; w.z = v[0][4 * j + 2]; w.w = v[0][4 * j + 3];
            reinterpret_cast<A4 *>(p)[j] = w;
        }
    } else {
#pragma unroll
        for (int m = 0; m < kFusedSeg; m++) p[m] = PixelTraits<P>::store(v[0][m]);
    }
}

// one sample of the workgroup's 16 consecutive lines (64 contiguous bytes for 4-byte pixels) <-> a padded LDS row
template <typename P, typename Acc>
__device__ __forceinline__ void load_lines16(const P *p, Acc *row, int nl) {
    if constexpr (sizeof(P) == sizeof(Acc)) {
        using A4 = typename Vec4<Acc>::type;
        if (nl == kLinesPerWG) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const A4 w = reinterpret_cast<const A4 *>(p)[j];
                row[4 * j + 0] = w.x; row[4 * j + 1] = w.y; row[4 * j + 2] = w.z; row[4 * j + 3] = w.w;
            }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < kLinesPerWG; c++) row[c] = c < nl ? PixelTraits<P>::load(p[c]) : Acc(0);
}

template <typename P, typename Acc>
__device__ __forceinline__ void store_lines16(P *p, const Acc *row, int nl) {
    if constexpr (sizeof(P) == sizeof(Acc)) {
        using A4 = typename Vec4<Acc>::type;
        if (nl == kLinesPerWG) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                A4 w;
                w.x = row[4 * j + 0]; w.y = row[4 * j + 1]; w.z = row[4 * j + 2]; w.w = row[4 * j + 3];
                reinterpret_cast<A4 *>(p)[j] = w;
            }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < kLinesPerWG; c++)
        if (c < nl) p[c] = PixelTraits<P>::store(row[c]);
}

// STRIDED = false: lines are contiguous (scans along x), a lane reads its 16 samples straight from memory.
// STRIDED = true : lines run along y / z (sample stride `inner`); the workgroup's 16 lines are 16 consecutive x, so a
//                  block of 256 samples x 16 lines is moved through LDS: thread = sample, 64 contiguous bytes each.
template <typename P, int K, bool STRIDED>
__global__ void __launch_bounds__(256)
line_scans_kernel(const P *__restrict__ src, P *__restrict__ dst, LineScanArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    __shared__ Acc block[STRIDED ? kLineTile * (kLinesPerWG + 1) : 1];
    const int t = threadIdx.x;
    const int l = t & 15, slot = t >> 4;
    const int64_t line0 = (int64_t)blockIdx.x * kLinesPerWG;           // first line of the workgroup
    const int64_t n = a.n, inner = a.inner;
    const int MT = (int)((n + kLineTile - 1) / kLineTile);
    const int last_lane_partial = (int)((n - (int64_t)(MT - 1) * kLineTile - 1) / kFusedSeg);     // n % 16 == 0
    // element (line, i) = (line / inner) * n * inner + line % inner + i * inner
    const int64_t my_line = line0 + slot;
    const bool line_ok = my_line < a.lines;
    const int64_t wg_base = (line0 / inner) * n * inner + (line0 % inner);      // STRIDED: the 16 lines are consecutive from here
    const int64_t my_base = STRIDED ? 0 : my_line * n;

    for (int s = 0; s < a.n_scans; s++) {
        const FusedScan<Acc> &sc = a.scans[s];
        const bool causal = sc.causal != 0;
        const P *from = s == 0 ? src : dst;                 // the first scan reads the input, the others filter in place
        Acc state[K];
#pragma unroll
        for (int j = 0; j < K; j++) state[j] = Acc(0);
        for (int step = 0; step < MT; step++) {
            const int tt = causal ? step : MT - 1 - step;
            const int last_lane = (tt == MT - 1) ? last_lane_partial : 15;
            const bool lane_in = l <= last_lane;
            Acc v[1][kFusedSeg];
#pragma unroll
            for (int m = 0; m < kFusedSeg; m++) v[0][m] = Acc(0);
            if constexpr (!STRIDED) {
                if (line_ok && lane_in) load16<P, Acc>(from + my_base + (int64_t)tt * kLineTile + 16 * l, v);
            } else {
                __syncthreads();                             // the previous step's readers of `block` are done
                const int64_t gi = (int64_t)tt * kLineTile + t;         // this thread's sample of the 16 lines
                if (gi < n) {
                    const int nl = (int)((a.lines - line0) < kLinesPerWG ? (a.lines - line0) : kLinesPerWG);
                    load_lines16<P, Acc>(from + wg_base + gi * inner, block + t * (kLinesPerWG + 1), nl);
                }
                __syncthreads();
                if (lane_in) {
#pragma unroll
                    for (int m = 0; m < kFusedSeg; m++) v[0][m] = block[(16 * l + m) * (kLinesPerWG + 1) + slot];
                }
            }
            const bool first_lane = causal ? (l == 0) : (l == last_lane);
            const bool clamp_first = a.clamped && step == 0 && first_lane;
            Acc cx[1][K];
#pragma unroll
            for (int j = 0; j < K; j++) cx[0][j] = state[j];
            if (causal) scan_rows16<Acc, true, K, 1>(v, sc, first_lane, clamp_first, cx);
            else        scan_rows16<Acc, false, K, 1>(v, sc, first_lane, clamp_first, cx, l > last_lane, kFusedSeg);
            // the state leaving the tile: its last K outputs in scan direction, from the lane that holds them
#pragma unroll
            for (int j = 0; j < K; j++) {
                const Acc mine = causal ? v[0][kFusedSeg - 1 - j] : v[0][j];
                state[j] = __shfl(mine, causal ? 15 : 0, 16);
            }
            if constexpr (!STRIDED) {
                if (line_ok && lane_in) store16<P, Acc>(dst + my_base + (int64_t)tt * kLineTile + 16 * l, v);
            } else {
                if (lane_in) {
#pragma unroll
                    for (int m = 0; m < kFusedSeg; m++) block[(16 * l + m) * (kLinesPerWG + 1) + slot] = v[0][m];
                }
                __syncthreads();
                const int64_t gi = (int64_t)tt * kLineTile + t;
                if (gi < n) {
                    const int nl = (int)((a.lines - line0) < kLinesPerWG ? (a.lines - line0) : kLinesPerWG);
                    store_lines16<P, Acc>(dst + wg_base + gi * inner, block + t * (kLinesPerWG + 1), nl);
                }
            }
        }
        // the next scan reads what this one stored (other threads' samples in the STRIDED form)
        __threadfence_block();
        __syncthreads();
    }
}

// Short lines (at most MT tiles = 256 MT samples): the WHOLE line stays in registers (16 MT per lane) across all scans of
// the dimension -- one round of loads, the scans back to back (a tile is still one 16-lane segment scan, the tiles of a
// scan are walked in registers), one round of stores.  The walking kernel above pays a trip to memory per tile and scan,
// which is most of its time on the small images this path exists for (512^2: 15 us per dimension against ~5 us).
template <typename P, int K, int MT, bool STRIDED>
__global__ void __launch_bounds__(256)
line_scans_short_kernel(const P *__restrict__ src, P *__restrict__ dst, LineScanArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    extern __shared__ __attribute__((aligned(16))) unsigned char lines_lds[];
    Acc *block = reinterpret_cast<Acc *>(lines_lds);               // STRIDED: [MT * 256][17]
    constexpr int PITCH = kLinesPerWG + 1;
    const int t = threadIdx.x;
    const int l = t & 15, slot = t >> 4;
    const int64_t line0 = (int64_t)blockIdx.x * kLinesPerWG;
    const int64_t n = a.n, inner = a.inner;
    const int mt = (int)((n + kLineTile - 1) / kLineTile);          // tiles that exist (<= MT)
    const int last_lane_partial = (int)((n - (int64_t)(mt - 1) * kLineTile - 1) / kFusedSeg);
    const int64_t my_line = line0 + slot;
    const bool line_ok = my_line < a.lines;
    const int64_t wg_base = (line0 / inner) * n * inner + (line0 % inner);
    const int nl = (int)((a.lines - line0) < kLinesPerWG ? (a.lines - line0) : kLinesPerWG);

    Acc v[MT][1][kFusedSeg];
#pragma unroll
    for (int tt = 0; tt < MT; tt++)
#pragma unroll
        for (int m = 0; m < kFusedSeg; m++) v[tt][0][m] = Acc(0);
    auto lane_in = [&](int tt) { return tt < mt && l <= (tt == mt - 1 ? last_lane_partial : 15); };
    if constexpr (!STRIDED) {
#pragma unroll
        for (int tt = 0; tt < MT; tt++)
            if (line_ok && lane_in(tt)) load16<P, Acc>(src + my_line * n + (int64_t)tt * kLineTile + 16 * l, v[tt]);
    } else {
#pragma unroll
        for (int tt = 0; tt < MT; tt++) {
            const int64_t gi = (int64_t)tt * kLineTile + t;
            if (gi < n) load_lines16<P, Acc>(src + wg_base + gi * inner, block + (tt * kLineTile + t) * PITCH, nl);
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < MT; tt++)
            if (lane_in(tt)) {
#pragma unroll
                for (int m = 0; m < kFusedSeg; m++) v[tt][0][m] = block[(tt * kLineTile + 16 * l + m) * PITCH + slot];
            }
    }

#pragma unroll 1
    for (int s = 0; s < a.n_scans; s++) {
        const FusedScan<Acc> &sc = a.scans[s];
        Acc cx[1][K];
#pragma unroll
        for (int j = 0; j < K; j++) cx[0][j] = Acc(0);
        if (sc.causal != 0) {
#pragma unroll
            for (int tt = 0; tt < MT; tt++) {
                if (tt < mt) {
                    const bool first_lane = l == 0;
                    scan_rows16<Acc, true, K, 1>(v[tt], sc, first_lane, a.clamped && tt == 0 && first_lane, cx);
#pragma unroll
                    for (int j = 0; j < K; j++) cx[0][j] = __shfl(v[tt][0][kFusedSeg - 1 - j], 15, 16);
                }
            }
        } else {
#pragma unroll
            for (int tt = MT - 1; tt >= 0; tt--) {
                if (tt < mt) {
                    const int last_lane = tt == mt - 1 ? last_lane_partial : 15;
                    const bool first_lane = l == last_lane;
                    scan_rows16<Acc, false, K, 1>(v[tt], sc, first_lane, a.clamped && tt == mt - 1 && first_lane, cx, l > last_lane, kFusedSeg);
#pragma unroll
                    for (int j = 0; j < K; j++) cx[0][j] = __shfl(v[tt][0][j], 0, 16);
                }
            }
        }
    }

    if constexpr (!STRIDED) {
#pragma unroll
        for (int tt = 0; tt < MT; tt++)
            if (line_ok && lane_in(tt)) store16<P, Acc>(dst + my_line * n + (int64_t)tt * kLineTile + 16 * l, v[tt]);
    } else {
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < MT; tt++)
            if (lane_in(tt)) {
#pragma unroll
                for (int m = 0; m < kFusedSeg; m++) block[(tt * kLineTile + 16 * l + m) * PITCH + slot] = v[tt][0][m];
            }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < MT; tt++) {
            const int64_t gi = (int64_t)tt * kLineTile + t;
            if (gi < n) store_lines16<P, Acc>(dst + wg_base + gi * inner, block + (tt * kLineTile + t) * PITCH, nl);
        }
    }
}

}  // namespace

template <typename P>
int launch_line_scans(int K, bool strided, const P *src, P *dst, const LineScanArgs<typename PixelTraits<P>::Acc> &a,
                      hipStream_t stream) {
    if (a.lines <= 0 || a.n <= 0 || a.n_scans <= 0) return RF_OK;
    const int64_t groups = (a.lines + kLinesPerWG - 1) / kLinesPerWG;
    if (groups >= (1ll << 31)) { set_error("line scans: too many lines"); return RF_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)groups), blk(256);
    // lines of at most 1024 samples stay in registers across all scans
    const int mt = (int)((a.n + kLineTile - 1) / kLineTile);
#define RF_SHORT(KK, MTT)                                                                                             \
    if (K == KK && mt <= MTT) {                                                                                       \
        const size_t lds = strided ? (size_t)MTT * kLineTile * (kLinesPerWG + 1) * sizeof(typename PixelTraits<P>::Acc) : 0; \
        if (strided) {                                                                                                \
            if (lds > 64 * 1024) {                                                                                    \
                static std::atomic<bool> attr_set[64];                                                                \
                int dev = 0;                                                                                          \
                RF_HIP_CHECK(hipGetDevice(&dev));                                                                     \
                if (!attr_set[dev & 63].load(std::memory_order_acquire)) {                                            \
                    RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&line_scans_short_kernel<P, KK, MTT, true>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
                    attr_set[dev & 63].store(true, std::memory_order_release);                                        \
                }                                                                                                     \
            }                                                                                                         \
            hipLaunchKernelGGL((line_scans_short_kernel<P, KK, MTT, true>), grid, blk, lds, stream, src, dst, a);     \
        } else hipLaunchKernelGGL((line_scans_short_kernel<P, KK, MTT, false>), grid, blk, 0, stream, src, dst, a);   \
        RF_HIP_CHECK(hipGetLastError());                                                                              \
        return RF_OK;                                                                                                 \
    }
    static const bool no_short = RF_KNOB("RF_LINES_NO_SHORT") != nullptr;          // A/B runs against the walking kernel
    if (!no_short && (sizeof(typename PixelTraits<P>::Acc) == 4 || mt <= 2)) {       // (f64: 32 registers per tile and lane)
        RF_SHORT(1, 1) RF_SHORT(2, 1) RF_SHORT(3, 1) RF_SHORT(1, 2) RF_SHORT(2, 2) RF_SHORT(3, 2) RF_SHORT(1, 4) RF_SHORT(2, 4) RF_SHORT(3, 4)
        // lines of up to 2048 samples: 128 registers of samples per lane, one workgroup per CU (136 KiB of LDS when strided)
        if constexpr (sizeof(typename PixelTraits<P>::Acc) == 4) {
            static const bool no_long = RF_KNOB("RF_LINES_NO_SHORT8") != nullptr;
            if (!no_long) { RF_SHORT(1, 8) RF_SHORT(2, 8) RF_SHORT(3, 8) }
        }
    }
#undef RF_SHORT
#define RF_CASE(KK)                                                                                                   \
    if (K == KK) {                                                                                                    \
        if (strided) hipLaunchKernelGGL((line_scans_kernel<P, KK, true>), grid, blk, 0, stream, src, dst, a);         \
        else         hipLaunchKernelGGL((line_scans_kernel<P, KK, false>), grid, blk, 0, stream, src, dst, a);        \
        RF_HIP_CHECK(hipGetLastError());                                                                              \
        return RF_OK;                                                                                                 \
    }
    RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    set_error("line scans: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}

template int launch_line_scans<float>(int, bool, const float *, float *, const LineScanArgs<float> &, hipStream_t);
template int launch_line_scans<double>(int, bool, const double *, double *, const LineScanArgs<double> &, hipStream_t);
template int launch_line_scans<int32_t>(int, bool, const int32_t *, int32_t *, const LineScanArgs<uint32_t> &, hipStream_t);
template int launch_line_scans<int16_t>(int, bool, const int16_t *, int16_t *, const LineScanArgs<uint32_t> &, hipStream_t);

}  // namespace rf
