// kernels_overlap.h -- argument block and launchers of the fully overlapped N-D tiled path (kernels_overlap.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace rf {

constexpr int kOvMaxOrder = 8;       // feedback taps per scan on the overlapped path (its k-vectors live in registers)
constexpr int kOvMaxTile = 4096;      // samples per N-D tile (LDS: 16 KiB f32 / 32 KiB f64)

template <typename Acc>
struct OvDim {
    int64_t N;                   // extent (1 for a missing dimension)
    int64_t lines;               // total / N
    int32_t T, M;                // tile width (1 for a dimension without scans), tiles
    int32_t n, k;                // scans, order
    const DevScan<Acc> *scans;   // n entries (device)
    const Acc *G;                // [variant 4][q][pos T][o k]: tables.h prop[v][q][n-1]
    Acc *tails;                  // [s][tile][r][line] -- the layout the carry scan of kernels_carry.hip works on
};

template <typename Acc>
struct OvArgs {
    int32_t ndim;
    int32_t clamped;
    OvDim<Acc> d[3];
};

template <typename P>
int launch_overlap_pass1(const P *src, const OvArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream);
template <typename P>
int launch_overlap_pass2(const P *src, P *dst, const OvArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream);
// adds to the tails of dimension `dim` what the completed carries of the dimensions before it contribute
template <typename Acc>
int launch_overlap_residual(const OvArgs<Acc> &a, int dim, hipStream_t stream);

}  // namespace rf
