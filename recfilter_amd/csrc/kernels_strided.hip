// kernels_strided.hip -- tiled scans along a NON-contiguous dimension (z of a volume, or y when a
// filter has no x scans) with the column held in registers.
//
// Thread = one line position (lane = x, so every access of a wave is 256 contiguous bytes), tile =
// TZ consecutive samples along the filtered dimension, kept in TZ registers.  Every scan of the
// dimension is a serial recurrence up or down the registers -- no LDS, no transposition.  Pass 1
// stores only the k-sample tails of each scan (lib/split.cpp:256-499), pass 2 injects the completed
// carries (lib/split.cpp:1008-1130) and stores the tile; the carry stage in between is
// carry_block_kernel (kernels_carry.hip).  This is the y phase of the fused kernel fed straight from
// HBM; it is what the 3-D configs run for z after the fused x/y stage.
#include <cstdlib>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

// same recurrence as the fused kernel's y phase (kernels_fused.hip: scan_col)
template <typename Acc, bool CAUSAL, int K, int TZ>
__device__ __forceinline__ void scan_regs(Acc (&col)[TZ], const FusedScanY<Acc> &sc, bool clamp_first, const Acc (&carry)[K]) {
    Acc h[K];
#pragma unroll
    for (int j = 0; j < K; j++) h[j] = carry[j];
    Acc y0 = Acc(0);
#pragma unroll
    for (int p = 0; p < TZ; p++) {
        const int m = CAUSAL ? p : TZ - 1 - p;
        Acc x = col[m];
        Acc acc = sc.b * x;
#pragma unroll
        for (int j = K - 1; j >= 0; j--) {
            Acc g = h[j];
            if (p <= j) g = clamp_first ? (p == 0 ? x : y0) : g;
            acc = acc + sc.a[j] * g;
        }
#pragma unroll
        for (int j = K - 1; j > 0; j--) h[j] = h[j - 1];
        h[0] = acc;
        if (p == 0) y0 = acc;
        col[m] = acc;
    }
}

// UNI: a.inner is a multiple of 256, so the 256 lines of a workgroup sit in one run of the inner dimensions: every access
// is (a scalar base advanced per sample) + (one 32-bit lane offset) instead of an address of its own per sample -- that was
// one register per sample on top of the column (133 registers at 64 samples: three waves per SIMD).
// PAT: the directions of the scans when they are the usual ones -- 1: one causal scan, 2: causal then anticausal; 0: any
// (a run-time direction inside the loop over the scans costs a register copy per sample and scan, kernels_fused.hip).
template <typename P, int K, int TZ, bool FINAL, bool UNI, int PAT>
__global__ void __launch_bounds__(256)
strided_pass_kernel(const P *__restrict__ src, P *__restrict__ dst, StridedArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    const int64_t line = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= a.lines) return;
    const int t = blockIdx.y;
    const int64_t l0 = (int64_t)blockIdx.x * 256;                    // wave-uniform
    const int64_t ubase = (l0 / a.inner) * a.n * a.inner + (l0 % a.inner) + (int64_t)t * TZ * a.inner;
    const int64_t base = UNI ? ubase + threadIdx.x : (line / a.inner) * a.n * a.inner + (line % a.inner) + (int64_t)t * TZ * a.inner;
    const uint32_t lane = threadIdx.x;
    Acc col[TZ];
#pragma unroll
    for (int i = 0; i < TZ; i++) {
        if constexpr (UNI) col[i] = PixelTraits<P>::load(__builtin_nontemporal_load(src + (ubase + (int64_t)i * a.inner) + lane));
        else col[i] = PixelTraits<P>::load(__builtin_nontemporal_load(src + base + (int64_t)i * a.inner));
    }
    auto one_scan = [&](int s, auto causal_tag) __attribute__((always_inline)) {
        constexpr bool causal = decltype(causal_tag)::value;
        const FusedScanY<Acc> &sc = a.scans[s];
        const bool tile_first = causal ? (t == 0) : (t == a.M - 1);
        const bool border = causal ? (t == 0 && a.first_is_border) : (t == a.M - 1 && a.last_is_border);
        Acc carry[K];
#pragma unroll
        for (int r = 0; r < K; r++) carry[r] = Acc(0);
        if (FINAL) {
            if (tile_first) {
#pragma unroll
                for (int r = 0; r < K; r++) carry[r] = a.incoming[((int64_t)s * K + r) * a.lines + line];
            } else {
                const int tp = causal ? t - 1 : t + 1;
#pragma unroll
                for (int r = 0; r < K; r++) carry[r] = a.tails[(((int64_t)s * a.M + tp) * K + r) * a.lines + line];
            }
        }
        bool clamp_first = a.clamped && border;
        if constexpr (PAT == 0) {
            if (a.mod_form) {                       // zero-border form behind a border modification (scan_device.h)
                if (clamp_first) border_mod_col<Acc, causal, TZ>(col, sc);
                clamp_first = false;
            }
        }
        scan_regs<Acc, causal, K, TZ>(col, sc, clamp_first, carry);
        if (!FINAL) {
#pragma unroll
            for (int r = 0; r < K; r++)
                a.tails[(((int64_t)s * a.M + t) * K + r) * a.lines + line] = causal ? col[TZ - 1 - r] : col[r];
        }
    };
    if constexpr (PAT == 1) {
        one_scan(0, std::true_type{});
    } else if constexpr (PAT == 2) {
        one_scan(0, std::true_type{});
        one_scan(1, std::false_type{});
    } else {
#pragma unroll 1
        for (int s = 0; s < a.n_scans; s++) {
            if (a.scans[s].causal != 0) one_scan(s, std::true_type{});
            else one_scan(s, std::false_type{});
        }
    }
    if (FINAL) {
#pragma unroll
        for (int i = 0; i < TZ; i++) {
            if constexpr (UNI) __builtin_nontemporal_store(PixelTraits<P>::store(col[i]), dst + (ubase + (int64_t)i * a.inner) + lane);
            else __builtin_nontemporal_store(PixelTraits<P>::store(col[i]), dst + base + (int64_t)i * a.inner);
        }
    }
}

}  // namespace

template <typename P>
int launch_strided_pass(bool final_pass, int K, int TZ, const P *src, P *dst,
                        const StridedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    if (a.lines <= 0 || a.M <= 0) return RF_OK;
    if (a.M > 65535) { set_error("strided path: too many tiles"); return RF_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)((a.lines + 255) / 256), (unsigned)a.M);
    // the fast variants: whole runs of 256 lines per workgroup, the usual scan patterns (else the general one)
    const bool uni = a.inner % 256 == 0 && a.lines % 256 == 0;
    const int pat = a.mod_form ? 0 : (a.n_scans == 1 && a.scans[0].causal != 0) ? 1
                  : (a.n_scans == 2 && a.scans[0].causal != 0 && a.scans[1].causal == 0) ? 2 : 0;
    // Workgroups per CU: these kernels stream 64 or 128 rows per wave that lie a whole plane apart; beyond three waves
    // per SIMD more rows in flight make the memory system slower, not faster (2048^3, 64 samples per thread: 5.7 ms at
    // three waves per SIMD, 6.9 ms at six).  An unused LDS allocation bounds the residency.
    static const int wgs_per_cu = RF_KNOB("RF_STRIDED_WGS") ? atoi(RF_KNOB("RF_STRIDED_WGS")) : 3;
    const size_t pad_lds = wgs_per_cu >= 1 && wgs_per_cu <= 8 ? (size_t)(160 * 1024 / wgs_per_cu) & ~(size_t)1023 : 0;
    const size_t lds_bytes = pad_lds > 64 * 1024 ? 64 * 1024 : pad_lds;      // (more than 64 KiB would need an opt-in per kernel)
#define RF_LAUNCH(KK, TT, FF, UU, PP)                                                                                \
    { hipLaunchKernelGGL((strided_pass_kernel<P, KK, TT, FF, UU, PP>), grid, dim3(256), UU ? lds_bytes : 0, stream, src, dst, a); \
      RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
#define RF_CASE(KK, TT)                                                                                              \
    if (K == KK && TZ == TT) {                                                                                        \
        if (uni && pat == 2) { if (final_pass) RF_LAUNCH(KK, TT, true, true, 2) else RF_LAUNCH(KK, TT, false, true, 2) } \
        if (uni && pat == 1) { if (final_pass) RF_LAUNCH(KK, TT, true, true, 1) else RF_LAUNCH(KK, TT, false, true, 1) } \
        if (final_pass) RF_LAUNCH(KK, TT, true, false, 0) else RF_LAUNCH(KK, TT, false, false, 0)                     \
    }
    RF_CASE(1, 64) RF_CASE(2, 64) RF_CASE(3, 64)
    RF_CASE(1, 32) RF_CASE(2, 32) RF_CASE(3, 32)
    RF_CASE(1, 128) RF_CASE(2, 128) RF_CASE(3, 128)
#undef RF_CASE
#undef RF_LAUNCH
    set_error("strided path: unsupported order %d / tile %d", K, TZ);
    return RF_ERR_UNSUPPORTED;
}

// rf_stream_copy (include/recfilter_amd.h): the access shape of the fused final pass -- one workgroup of 256 threads per
// 256 x 128 tile, a tile row = 64 chunks of 16 bytes = one wave, non-temporal both ways, eight loads per thread in flight --
// without LDS and without arithmetic.
namespace {
typedef float CopyF4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t width) {
    const int cc = threadIdx.x & 63, rg = threadIdx.x >> 6;                 // chunk of the row, row of a group of four
    const int64_t base = ((int64_t)blockIdx.y * 128 + rg) * width + (int64_t)blockIdx.x * 256 + 4 * cc;
#pragma unroll
    for (int half = 0; half < 4; half++) {
        CopyF4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __builtin_nontemporal_load(reinterpret_cast<const CopyF4 *>(src + base + (int64_t)(32 * half + 4 * i) * width));
#pragma unroll
        for (int i = 0; i < 8; i++) __builtin_nontemporal_store(v[i], reinterpret_cast<CopyF4 *>(dst + base + (int64_t)(32 * half + 4 * i) * width));
    }
}
}  // namespace

int launch_stream_copy(const float *src, float *dst, int64_t width, int64_t rows, hipStream_t stream) {
    if (src == nullptr || dst == nullptr || src == dst || width <= 0 || rows <= 0 || width % 256 != 0 || rows % 128 != 0 ||
        rows / 128 > 65535 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 != 0) {
        set_error("stream copy: needs two distinct 16-byte aligned f32 images of whole 256 x 128 tiles");
        return RF_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)(width / 256), (unsigned)(rows / 128)), dim3(256), 0, stream, src, dst, width);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template int launch_strided_pass<float>(bool, int, int, const float *, float *, const StridedArgs<float> &, hipStream_t);
template int launch_strided_pass<int32_t>(bool, int, int, const int32_t *, int32_t *, const StridedArgs<uint32_t> &, hipStream_t);
template int launch_strided_pass<int16_t>(bool, int, int, const int16_t *, int16_t *, const StridedArgs<uint32_t> &, hipStream_t);

}  // namespace rf
