// kernels.h -- host-callable launchers of the gfx950 kernels (definitions in *.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pixel.h"
#include "rf_internal.h"

namespace rf {

// Device-side description of one scan, in the pixel's arithmetic type.
template <typename Acc>
struct DevScan {
    int32_t causal;
    int32_t order;
    Acc b;
    Acc a[RF_MAX_ORDER];
};

// Geometry of "lines along one dimension" of a dense x-fastest array:
//   element (line, i) lives at  (line / inner) * n * inner + (line % inner) + i * inner
struct LineGeom {
    int64_t n;       // extent of the filtered dimension
    int64_t inner;   // stride of the filtered dimension (product of lower extents)
    int64_t lines;   // number of lines = total / n
};

// ---- untiled path: one serial recurrence per line (kernels_generic.hip) -------------------
template <typename P>
int launch_untiled_scan(const P *in, P *out, LineGeom g, const DevScan<typename PixelTraits<P>::Acc> &sc,
                        bool clamped, hipStream_t stream);

// ---- stand-alone pointwise stage: dst = c0*f + c1*x + c2 (float pixel types; x ignored when c1 == 0) ----
template <typename P>
int launch_pointwise(const P *f, const P *x, P *dst, int64_t n, double c0, double c1, double c2, hipStream_t stream);
// same with the x operand of another element type (unsigned-byte input planes); f may be null when c0 == 0
template <typename P, typename X>
int launch_pointwise_from(const P *f, const X *x, P *dst, int64_t n, double c0, double c1, double c2, hipStream_t stream);

// ---- finite differences of summed-area tables (rf_box_difference) ----
struct BoxDiffArgs {
    int64_t n[RF_MAX_DIMS];       // extents, x first (1 for missing dimensions)
    int32_t order[RF_MAX_DIMS];   // 0..2 applications per dimension
    int32_t radius;
};
template <typename P>
int launch_box_difference(const P *in, P *out, const BoxDiffArgs &a, hipStream_t stream);

// ---- rf_stream_copy: the tile-shaped non-temporal copy bench.py measures its ceiling with (kernels_strided.hip) ----
int launch_stream_copy(const float *src, float *dst, int64_t width, int64_t rows, hipStream_t stream);

// ---- clamped tap combinations (rf_tap_filter) ----
struct TapArgs {
    int64_t n[RF_MAX_DIMS];                  // extents, x first (1 for missing dimensions)
    int32_t n_taps;
    int32_t plane[RF_MAX_TAPS];
    int32_t off[RF_MAX_TAPS][RF_MAX_DIMS];
    float weight[RF_MAX_TAPS];
    const void *in[RF_MAX_PLANES];
};
template <typename P>
int launch_tap_filter(P *out, const TapArgs &a, hipStream_t stream);

// ---- generic tiled path, any tile width T <= kGenericMaxTile dividing n -------------------
constexpr int kGenericMaxTile = 128;

template <typename Acc>
struct GenericDimArgs {
    LineGeom g;
    int32_t T;          // tile width
    int32_t M;          // tiles per line
    int32_t k;          // max order in the dimension
    int32_t n_scans;
    int32_t clamped;
    int32_t first_is_border;   // this slab holds the image's first tile along the dimension
    int32_t last_is_border;    // ... the last tile
    const DevScan<Acc> *scans; // n_scans entries (device)
    Acc *tails;                // [scan][tile][r][line]
    Acc *incoming;             // [scan][r][line]   carry entering the slab (zeros at the image border)
    const Acc *W;              // [variant 4][q][s][r][o]   (q < s used)
    const Acc *A;              // [s][r][j]
    const Acc *Apow;           // [s][i][r][j] = (A[s])^(i+1), i = 0..M-1: what the carry entering the slab adds to the
                               // tail of the i-th tile in scan direction (carry_apply); may be null (serial fallback)
    int32_t tile_major;        // tails laid out [tile][line / 256][scan][r][256] (the y tails of the fused path) -- only the
                               // blocked carry scan (kernels_carry.hip) reads this flag
    const Acc *tails_part2;    // the tile-local tails arrive in TWO parts (the tall patches of kernels_tails_walk.hip: a tile's
                               // column halves): tails + tails_part2, same layout; the blocked carry scan adds them up as it
                               // loads them and stores the completed tails to `tails`.  Null: one part.
};

template <typename P>
int launch_generic_pass1(const P *src, GenericDimArgs<typename PixelTraits<P>::Acc> a, hipStream_t stream);
// (the carry recurrence itself is the blocked scan of kernels_carry.hip, declared in kernels_fused.h)
template <typename Acc>
int launch_generic_carry_apply(GenericDimArgs<Acc> a, int s, hipStream_t stream);
// orders above 8: the carry recurrence of the scans [s_begin, s_end) as one thread per line (kernels_generic.hip)
constexpr int kCarryBlockMaxOrder = 8;   // what launch_carry_block (kernels_carry.hip) is instantiated for
template <typename Acc>
int launch_generic_carry_serial(GenericDimArgs<Acc> a, uint32_t causal_mask, int s_begin, int s_end, Acc *send, hipStream_t stream);
// incoming[s] = sum over the slabs before this one (in scan direction) of (A^M)^(distance-1) * their exit tails
template <typename Acc>
int launch_gather_incoming(GenericDimArgs<Acc> a, int s, const Acc *gathered, int64_t rank_stride,
                           int64_t plane_offset, int rank, int world, const Acc *AM /* k*k device */,
                           hipStream_t stream);
// merged exchange (all scans of the sharded dimension behind one all-gather): gathered[h][plane][s][r][line] holds
// every slab's zero-entering-carry exits; X[type][q][s] / Y[q][s][t] are the cross-scan transfers of plan_generic.h
template <typename Acc>
int launch_merged_gather(GenericDimArgs<Acc> a, const Acc *gathered, int64_t rank_stride, int64_t plane_offset, int rank,
                         int world, const Acc *X, hipStream_t stream);
template <typename Acc>
int launch_merged_apply(GenericDimArgs<Acc> a, const Acc *Y, hipStream_t stream);
template <typename P>
int launch_generic_pass2(const P *src, P *dst, GenericDimArgs<typename PixelTraits<P>::Acc> a, hipStream_t stream);

}  // namespace rf
