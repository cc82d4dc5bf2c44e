// kernels_overlap.hip -- the fully overlapped N-D tiling of lib/split.cpp: ALL tiled dimensions in one pass 1 and one
// pass 2, with the cross-dimension residuals between every pair of dimensions.
//
//   pass 1      one workgroup per N-D tile (Tx x Ty x Tz samples in LDS): every scan of every dimension with zero
//               entering carries, the k-sample tail of every scan kept             (split.cpp:503-665, 256-499)
//   carry(d)    the blocked carry scan of kernels_carry.hip, dimension by dimension (split.cpp:743-867, 912-1004)
//   residual(d) before carry(d), d >= 1: what the COMPLETED carries of every earlier dimension add to the tails of
//               dimension d -- add_prev_dimension_residual_to_tails, split.cpp:1215-1633, executed for every pair
//               of dimensions (x->y; x->z and y->z in 3-D, split.cpp:1814-1820).  Per tile the correction field
//                   r_0 = 0,   r_{e+1} = F_e(r_e) + sum_q G^e_q (x) c^e_q          (e = 0 .. d-1)
//               is rebuilt in LDS from the carries alone (F_e = the tile-local scans of dimension e, G^e_q = the
//               tile response to the carry entering scan q of dimension e, tables.h prop[v][q][n-1], c^e_q = that
//               completed carry), pushed through the tile-local scans of dimension d, and its tails are ADDED to the
//               stored tails of dimension d.  Everything is linear, so this equals extracting the tails of the true
//               field; no image byte is read.
//   pass 2      one workgroup per tile: every scan of every dimension entering with the completed carry of the
//               previous tile, one store                                             (split.cpp:1008-1130, 1647-1780)
//
// Two passes over the image whatever the number of dimensions (12 bytes per f32 sample + tails) -- against one pair
// of passes per dimension on RF_PATH_TILED_GENERIC.  The price is the tails: n_scans * k / T of the volume per
// dimension, so this path is for the small tiles the reference tiles with (its tests use 4, its apps 32); the
// bandwidth-tuned 256 x 64 tiles of RF_PATH_TILED_FUSED keep z as a second stage.
#include <type_traits>

#include "kernels.h"
#include "kernels_overlap.h"

namespace rf {

namespace {

constexpr int kOvThreads = 256;

template <typename Acc>
__device__ __forceinline__ Acc ov_scan_step(Acc x, int p, const DevScan<Acc> &sc, int k, bool clamp_first,
                                            Acc (&hist)[kOvMaxOrder], Acc &y0) {
    Acc acc = sc.b * x;
#pragma unroll
    for (int j = 0; j < kOvMaxOrder; j++) {
        if (j < k) {
            Acc g = hist[j];
            if (clamp_first && p <= j) g = (p == 0) ? x : y0;
            acc = acc + sc.a[j] * g;
        }
    }
#pragma unroll
    for (int j = kOvMaxOrder - 1; j > 0; j--) hist[j] = hist[j - 1];
    hist[0] = acc;
    if (p == 0) y0 = acc;
    return acc;
}

// geometry of one workgroup's tile
template <typename Acc>
struct OvTile {
    int t[3];        // tile coordinates
    int ls[3];       // LDS strides of the three dimensions
    int vol;
};

template <typename Acc>
__device__ __forceinline__ OvTile<Acc> ov_tile(const OvArgs<Acc> &a) {
    OvTile<Acc> tl;
    uint32_t b = blockIdx.x;
    tl.t[0] = (int)(b % (uint32_t)a.d[0].M); b /= (uint32_t)a.d[0].M;
    tl.t[1] = (int)(b % (uint32_t)a.d[1].M); b /= (uint32_t)a.d[1].M;
    tl.t[2] = (int)b;
    tl.ls[0] = 1; tl.ls[1] = a.d[0].T; tl.ls[2] = a.d[0].T * a.d[1].T;
    tl.vol = a.d[0].T * a.d[1].T * a.d[2].T;
    return tl;
}

// a tile-local line along dimension e, owned by thread i (< vol / T_e): LDS base and global line index
template <typename Acc>
__device__ __forceinline__ void ov_line(const OvArgs<Acc> &a, const OvTile<Acc> &tl, int e, int i, int &base, int64_t &line) {
    const int da = e == 0 ? 1 : 0, db = e == 2 ? 1 : 2;      // the two other dimensions, ascending
    const int ia = i % a.d[da].T, ib = i / a.d[da].T;
    base = ia * tl.ls[da] + ib * tl.ls[db];
    const int64_t ga = (int64_t)tl.t[da] * a.d[da].T + ia, gb = (int64_t)tl.t[db] * a.d[db].T + ib;
    line = ga + (int64_t)a.d[da].N * gb;
}

template <typename Acc>
__device__ __forceinline__ int ov_variant(const OvArgs<Acc> &a, const OvTile<Acc> &tl, int e) {
    return (tl.t[e] == 0 ? 1 : 0) | (tl.t[e] == a.d[e].M - 1 ? 2 : 0);
}

// global <-> LDS copy of the tile (x fastest in both)
template <typename P, typename Acc, bool LOAD>
__device__ __forceinline__ void ov_copy(const OvArgs<Acc> &a, const OvTile<Acc> &tl, const P *src, P *dst, Acc *tile) {
    using Tr = PixelTraits<P>;
    const int Tx = a.d[0].T, Ty = a.d[1].T;
    for (int i = threadIdx.x; i < tl.vol; i += kOvThreads) {
        const int x = i % Tx, y = (i / Tx) % Ty, z = i / (Tx * Ty);
        const int64_t g = ((int64_t)tl.t[0] * Tx + x) + a.d[0].N * (((int64_t)tl.t[1] * Ty + y) + a.d[1].N * ((int64_t)tl.t[2] * a.d[2].T + z));
        if (LOAD) tile[i] = Tr::load(src[g]);
        else dst[g] = Tr::store(tile[i]);
    }
}

// One scan of dimension e over every line of the LDS tile.
//   MODE 0: zero entering state, tails STORED (pass 1)      MODE 1: zero entering state, tails ADDED (residual)
//   MODE 2: entering state = completed carry of the previous tile, no tails (pass 2)
//   MODE 3: zero entering state, no tails (residual: F_e of the correction field)
template <typename Acc, int MODE>
__device__ __forceinline__ void ov_scan_dim(const OvArgs<Acc> &a, const OvTile<Acc> &tl, Acc *tile, int e, int s) {
    const OvDim<Acc> &d = a.d[e];
    const DevScan<Acc> sc = d.scans[s];
    const int T = d.T, k = d.k, stride = tl.ls[e];
    const bool causal = sc.causal != 0;
    const bool first = causal ? (tl.t[e] == 0) : (tl.t[e] == d.M - 1);
    const bool clamp_first = a.clamped && first;
    const int n_lines = tl.vol / T;
    for (int i = threadIdx.x; i < n_lines; i += kOvThreads) {
        int base;
        int64_t line;
        ov_line(a, tl, e, i, base, line);
        Acc hist[kOvMaxOrder];
#pragma unroll
        for (int j = 0; j < kOvMaxOrder; j++) hist[j] = Acc(0);
        if (MODE == 2 && !first) {
            const int tp = causal ? tl.t[e] - 1 : tl.t[e] + 1;
            for (int j = 0; j < k; j++) hist[j] = d.tails[(((int64_t)s * d.M + tp) * k + j) * d.lines + line];
        }
        Acc y0 = Acc(0);
        for (int p = 0; p < T; p++) {
            const int m = causal ? p : T - 1 - p;
            const Acc x = tile[base + m * stride];
            tile[base + m * stride] = ov_scan_step<Acc>(x, p < kOvMaxOrder ? p : kOvMaxOrder, sc, k, clamp_first, hist, y0);
        }
        if (MODE == 0 || MODE == 1) {
            for (int r = 0; r < k; r++) {             // hist[r] = output at direction position T-1-r = tail r
                Acc *tp = d.tails + (((int64_t)s * d.M + tl.t[e]) * k + r) * d.lines + line;
                if (MODE == 0) *tp = hist[r];
                else *tp = *tp + hist[r];
            }
        }
    }
}

template <typename P>
__global__ void __launch_bounds__(kOvThreads)
ov_pass1_kernel(const P *__restrict__ src, OvArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    __shared__ Acc tile[kOvMaxTile];
    const OvTile<Acc> tl = ov_tile(a);
    ov_copy<P, Acc, true>(a, tl, src, nullptr, tile);
    for (int e = 0; e < a.ndim; e++)
        for (int s = 0; s < a.d[e].n; s++) {
            __syncthreads();
            ov_scan_dim<Acc, 0>(a, tl, tile, e, s);
        }
}

template <typename P>
__global__ void __launch_bounds__(kOvThreads)
ov_pass2_kernel(const P *__restrict__ src, P *__restrict__ dst, OvArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    __shared__ Acc tile[kOvMaxTile];
    const OvTile<Acc> tl = ov_tile(a);
    ov_copy<P, Acc, true>(a, tl, src, nullptr, tile);
    for (int e = 0; e < a.ndim; e++)
        for (int s = 0; s < a.d[e].n; s++) {
            __syncthreads();
            ov_scan_dim<Acc, 2>(a, tl, tile, e, s);
        }
    __syncthreads();
    ov_copy<P, Acc, false>(a, tl, nullptr, dst, tile);
}

// residual of the dimensions before `dim` on the tails of `dim`
template <typename Acc>
__global__ void __launch_bounds__(kOvThreads)
ov_residual_kernel(OvArgs<Acc> a, int dim) {
    __shared__ Acc tile[kOvMaxTile];
    const OvTile<Acc> tl = ov_tile(a);
    for (int i = threadIdx.x; i < tl.vol; i += kOvThreads) tile[i] = Acc(0);
    bool nonzero = false;                               // r is still identically zero: nothing to push through F_e
    for (int e = 0; e < dim; e++) {
        const OvDim<Acc> &d = a.d[e];
        if (d.n == 0) continue;
        if (nonzero)
            for (int s = 0; s < d.n; s++) {
                __syncthreads();
                ov_scan_dim<Acc, 3>(a, tl, tile, e, s);
            }
        __syncthreads();
        // r += sum_q G^e_q (x) c^e_q: what the completed carries entering this tile add after all scans of dimension e
        const int v = ov_variant(a, tl, e);
        const int n_lines = tl.vol / d.T;
        for (int i = threadIdx.x; i < n_lines; i += kOvThreads) {
            int base;
            int64_t line;
            ov_line(a, tl, e, i, base, line);
            for (int q = 0; q < d.n; q++) {
                const bool causal = d.scans[q].causal != 0;
                const bool first = causal ? (tl.t[e] == 0) : (tl.t[e] == d.M - 1);
                if (first) continue;
                const int tp = causal ? tl.t[e] - 1 : tl.t[e] + 1;
                Acc c[kOvMaxOrder];
                for (int o = 0; o < d.k; o++) c[o] = d.tails[(((int64_t)q * d.M + tp) * d.k + o) * d.lines + line];
                const Acc *G = d.G + ((size_t)(v * d.n + q) * d.T) * d.k;
                for (int m = 0; m < d.T; m++) {
                    Acc acc = tile[base + m * tl.ls[e]];
                    for (int o = 0; o < d.k; o++) acc = acc + G[m * d.k + o] * c[o];
                    tile[base + m * tl.ls[e]] = acc;
                }
            }
        }
        nonzero = true;
    }
    if (!nonzero) return;
    for (int s = 0; s < a.d[dim].n; s++) {
        __syncthreads();
        ov_scan_dim<Acc, 1>(a, tl, tile, dim, s);
    }
}

}  // namespace

template <typename Acc>
static int ov_grid(const OvArgs<Acc> &a, unsigned *grid) {
    const int64_t tiles = (int64_t)a.d[0].M * a.d[1].M * a.d[2].M;
    if (tiles <= 0) { *grid = 0; return RF_OK; }
    if (tiles >= (1ll << 31)) { set_error("overlapped path: too many tiles"); return RF_ERR_UNSUPPORTED; }
    if ((int64_t)a.d[0].T * a.d[1].T * a.d[2].T > kOvMaxTile) { set_error("overlapped path: tile too large"); return RF_ERR_INVALID_ARG; }
    *grid = (unsigned)tiles;
    return RF_OK;
}

template <typename P>
int launch_overlap_pass1(const P *src, const OvArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    unsigned grid;
    int rc = ov_grid(a, &grid);
    if (rc != RF_OK || grid == 0) return rc;
    hipLaunchKernelGGL((ov_pass1_kernel<P>), dim3(grid), dim3(kOvThreads), 0, stream, src, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename P>
int launch_overlap_pass2(const P *src, P *dst, const OvArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    unsigned grid;
    int rc = ov_grid(a, &grid);
    if (rc != RF_OK || grid == 0) return rc;
    hipLaunchKernelGGL((ov_pass2_kernel<P>), dim3(grid), dim3(kOvThreads), 0, stream, src, dst, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename Acc>
int launch_overlap_residual(const OvArgs<Acc> &a, int dim, hipStream_t stream) {
    unsigned grid;
    int rc = ov_grid(a, &grid);
    if (rc != RF_OK || grid == 0) return rc;
    hipLaunchKernelGGL((ov_residual_kernel<Acc>), dim3(grid), dim3(kOvThreads), 0, stream, a, dim);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

#define RF_INST(P)                                                                                                  \
    template int launch_overlap_pass1<P>(const P *, const OvArgs<PixelTraits<P>::Acc> &, hipStream_t);               \
    template int launch_overlap_pass2<P>(const P *, P *, const OvArgs<PixelTraits<P>::Acc> &, hipStream_t);
RF_INST(float) RF_INST(double) RF_INST(int32_t) RF_INST(int16_t)
#undef RF_INST
template int launch_overlap_residual<float>(const OvArgs<float> &, int, hipStream_t);
template int launch_overlap_residual<double>(const OvArgs<double> &, int, hipStream_t);
template int launch_overlap_residual<uint32_t>(const OvArgs<uint32_t> &, int, hipStream_t);

}  // namespace rf
