// kernels_stream.hip -- pass 1 (tail extraction, kernels_tails.hip's contraction) as a STREAMING kernel:
// persistent workgroups that walk the tiles, fed by a ring of LDS slots that a loader wave fills with LDS-DMA.
//
// Why: fused_tails_kernel is latency-bound (profiles/r1/pmc_shader.json: its waves wait in 69 % of their cycles).
// Every tile byte travels HBM -> VGPR -> ds_write -> barrier, so the bytes a workgroup keeps in flight are capped
// by its staging registers, and every workgroup starts with an empty pipe.  Here
//   * a workgroup is 4 compute waves + 1 LOADER wave.  The loader issues `global_load_lds_dwordx4` (1 KiB per wave
//     instruction, straight into LDS, no VGPR hop, no ds_write) for slots up to kAhead quarters of a tile ahead of
//     the one being contracted, across tile boundaries: the bytes in flight per workgroup are constant
//     (kAhead x 16 KiB), whatever the compute waves are doing;
//   * the loader's `vmcnt` counts nothing but its own DMA pieces, so "slot g has landed" is an exact counted wait
//     (`s_waitcnt vmcnt(16 * slots still allowed in flight)`), and the compute waves never drain the ring when they
//     wait for their own stores;
//   * one `s_barrier` per slot orders both directions: slot g has landed (RAW) and slot g-1 has been read by every
//     compute wave, so the loader may refill it (WAR).
// The XOR chunk swizzle of the tile image is kept by permuting which 16-byte chunk of the row each LOADER LANE
// fetches (the LDS destination of an LDS-DMA is lane-linear, the global source is per lane).
//
// The arithmetic is that of fused_tails_kernel (same tables, same summation order inside a quarter); images made of
// whole 256 x 64 tiles with pixel-typed planes take this kernel, everything else (partial tiles, 32-row tiles,
// unsigned-byte planes, a fused prologue) stays on fused_tails_kernel.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

constexpr int kQRows = 16;                          // rows per ring slot: a quarter of a 64-row tile
constexpr int kSlotBytes = kQRows * kFusedTX * 4;   // 16 KiB
constexpr int kStreamTY = 64;
constexpr int kQPerTile = kStreamTY / kQRows;       // 4
constexpr int kStreamThreads = 320;                 // waves 0..3 compute, wave 4 loads

typedef float F2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_barrier() {
    // LDS traffic of this wave retired, then the workgroup barrier; a compiler barrier for memory as well.  (Not
    // __syncthreads(): its fence would add `vmcnt(0)` -- draining the loader's DMA ring, and making the compute
    // waves wait for their tail stores.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct TileCoord {
    int tx, ty;
    int64_t z;
};

// tile index (x fastest, then y, then plane) -> coordinates; the host checks that the tile count fits 31 bits
template <typename Acc>
__device__ __forceinline__ TileCoord tile_coord(const FusedArgs<Acc> &a, uint32_t i) {
    TileCoord c;
    const uint32_t per_plane = (uint32_t)a.MX * (uint32_t)a.MY;
    const uint32_t z = i / per_plane;
    const uint32_t rem = i - z * per_plane;
    c.z = z;
    c.ty = (int)(rem / (uint32_t)a.MX);
    c.tx = (int)(rem - (uint32_t)c.ty * (uint32_t)a.MX);
    return c;
}

// SLOTS ring slots, AHEAD (< SLOTS) of them in flight ahead of the slot being contracted
template <typename P, int K, int SLOTS, int AHEAD>
__global__ void __launch_bounds__(kStreamThreads)
stream_tails_kernel(const P *__restrict__ src, FusedArgs<typename PixelTraits<P>::Acc> a,
                    const typename PixelTraits<P>::Acc *__restrict__ Hx,     // [vx][s][r][256]
                    const typename PixelTraits<P>::Acc *__restrict__ Hy,     // [vy][j][r][64]
                    uint32_t n_tiles) {
    using Acc = typename PixelTraits<P>::Acc;
    using A4 = typename Vec4<Acc>::type;
    static_assert(sizeof(Acc) == 4, "the ring moves 4-byte samples");
    static_assert(AHEAD >= 1 && AHEAD < SLOTS && 16 * AHEAD <= 63, "ring geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ring: SLOTS x 16 KiB][hx: nx*K rows of 256][x-tail stage: 2 x nx*K x 64]
    const int nxk = a.nx * K, nyk = a.ny * K;
    Acc *hx_lds = reinterpret_cast<Acc *>(smem + SLOTS * kSlotBytes);
    Acc *stage = hx_lds + (size_t)(nxk > 0 ? nxk : 1) * kFusedTX;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t G = gridDim.x;
    const uint32_t first = blockIdx.x;
    const int n_my = first < n_tiles ? (int)((n_tiles - first + G - 1) / G) : 0;     // tiles this workgroup walks
    const int nq = n_my * kQPerTile;
    const uint32_t row_bytes = a.row_bytes;
    auto variant_x = [&](int tx) { return (tx == 0 ? 1 : 0) | (tx == a.MX - 1 ? 2 : 0); };

    if (wave == 4) {
        // ------------------------------------------------ loader wave ------------------------------------------------
        const int lane = t & 63;
        const uint32_t lane_off = (uint32_t)swz_chunk(lane) * 16u;       // LDS position `lane` holds chunk swz(lane)
        auto issue = [&](int g) {
            const TileCoord c = tile_coord(a, first + (uint32_t)(g / kQPerTile) * G);
            const P *plane = a.plane_batch ? reinterpret_cast<const P *>(a.in_planes[c.z]) : src + c.z * a.NX * a.NY;
            const char *base = reinterpret_cast<const char *>(plane + (int64_t)c.ty * kStreamTY * a.NX + (int64_t)c.tx * kFusedTX) +
                               (size_t)(g % kQPerTile) * kQRows * row_bytes;
            unsigned char *slot = smem + (g % SLOTS) * kSlotBytes;
#pragma unroll
            for (int r = 0; r < kQRows; r++)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(base + ((uint32_t)r * row_bytes + lane_off)),
                    (__attribute__((address_space(3))) void *)(slot + r * (kFusedTX * 4)), 16, 0, /*nt*/ 2);
        };
        for (int g = 0; g < AHEAD && g < nq; g++) issue(g);
        int cur_vx = -1;
        for (int g = 0; g < nq; g++) {
            // slot g has landed once at most min(AHEAD-1, nq-1-g) younger slots (16 pieces each) are outstanding
            const int younger = nq - 1 - g;
            if (younger >= AHEAD - 1) wait_vmcnt<16 * (AHEAD - 1)>();
            else if (AHEAD >= 3 && younger == 1) wait_vmcnt<16>();
            else wait_vmcnt<0>();
            lds_barrier();                                          // B_g
            if (g % kQPerTile == 0) {                               // the compute waves restage Hx behind a second barrier
                const int vx = variant_x(tile_coord(a, first + (uint32_t)(g / kQPerTile) * G).tx);
                if (vx != cur_vx) { cur_vx = vx; lds_barrier(); }
            }
            if (g + AHEAD < nq) issue(g + AHEAD);                   // refills the slot contracted before B_g
        }
        lds_barrier();                                              // final: x-tail stage of the last tile
        return;
    }

    // ------------------------------------------------ compute waves --------------------------------------------------
    const int l = t & 15, slot_row = t >> 4, sw = (l >> 2) & 3;       // x part: segment lane, row of the quarter
    const int e = (swz_chunk(t >> 2) << 2) | (t & 3);                 // y part: this column's place in a swizzled row
    const int64_t Lx = a.NYP * a.NZ, Ly = a.NXP * a.NZ;
    A4 *hx4 = reinterpret_cast<A4 *>(hx_lds);
    int cur_vx = -1;

    auto flush_xtails = [&](int n) {       // tile n's x tails: stage -> xt, 16 bytes per lane, 256 B per (s, r)
        if (nxk > 0 && t < nxk * 16) {
            const TileCoord c = tile_coord(a, first + (uint32_t)n * G);
            const int sr = t >> 4, j = t & 15;
            const A4 v = reinterpret_cast<const A4 *>(stage + (size_t)(n & 1) * nxk * kStreamTY + (size_t)sr * kStreamTY)[j];
            const int s = sr / K, r = sr % K;
            const int64_t line0 = (int64_t)c.ty * kStreamTY + a.NYP * c.z;
            *reinterpret_cast<A4 *>(a.xt + (((int64_t)s * a.MX + c.tx) * K + r) * Lx + line0 + 4 * j) = v;
        }
    };

    for (int n = 0; n < n_my; n++) {
        const TileCoord c = tile_coord(a, first + (uint32_t)n * G);
        const int vx = variant_x(c.tx);
        const int vy = ((c.ty == 0 && a.y_first_border) ? 1 : 0) | ((c.ty == a.MY - 1 && a.y_last_border) ? 2 : 0);
        Acc comb[kFusedMaxScans * K];
#pragma unroll
        for (int jr = 0; jr < kFusedMaxScans * K; jr++) comb[jr] = Acc(0);
        Acc *stage_n = stage + (size_t)(n & 1) * nxk * kStreamTY;

#pragma unroll 1
        for (int q = 0; q < kQPerTile; q++) {
            const int g = n * kQPerTile + q;
            lds_barrier();                                          // B_g: slot g landed, slot g-1 free
            if (q == 0) {
                if (vx != cur_vx) {
                    // impulse responses of the x tails for this border variant -> LDS, chunk-swizzled like the pixels
                    cur_vx = vx;
                    if (nxk > 0) {
                        const A4 *hsrc = reinterpret_cast<const A4 *>(Hx + (size_t)vx * nxk * kFusedTX);
                        for (int cidx = t; cidx < nxk * 64; cidx += 256) hx4[(cidx & ~63) | swz_chunk(cidx & 63)] = hsrc[cidx];
                    }
                    lds_barrier();
                }
                if (n > 0) flush_xtails(n - 1);                     // every wave's stage writes of tile n-1 are behind B_g
            }
            const Acc *tile = reinterpret_cast<const Acc *>(smem + (g % SLOTS) * kSlotBytes);
            const A4 *tile4 = reinterpret_cast<const A4 *>(tile);

            // ---- x tails of this quarter's 16 rows: dot products + reduction over the 16 lanes of a row ----
            if (nxk > 0) {
                Acc v[kFusedSeg];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const A4 w = tile4[slot_row * 64 + 4 * l + (j ^ sw)];
                    v[4 * j + 0] = w.x; v[4 * j + 1] = w.y; v[4 * j + 2] = w.z; v[4 * j + 3] = w.w;
                }
#pragma unroll 1
                for (int sr = 0; sr < nxk; sr++) {
                    Acc h[kFusedSeg];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const A4 w = hx4[sr * 64 + 4 * l + (j ^ sw)];
                        h[4 * j + 0] = w.x; h[4 * j + 1] = w.y; h[4 * j + 2] = w.z; h[4 * j + 3] = w.w;
                    }
                    Acc acc;
                    if constexpr (std::is_same<Acc, float>::value) {
                        F2s a0 = F2s{0.0f, 0.0f}, a1 = F2s{0.0f, 0.0f};
#pragma unroll
                        for (int m = 0; m < kFusedSeg; m += 4) {
                            a0 = F2s{h[m], h[m + 1]} * F2s{v[m], v[m + 1]} + a0;
                            a1 = F2s{h[m + 2], h[m + 3]} * F2s{v[m + 2], v[m + 3]} + a1;
                        }
                        const F2s s2 = a0 + a1;
                        acc = s2.x + s2.y;
                    } else {
                        acc = Acc(0);
#pragma unroll
                        for (int m = 0; m < kFusedSeg; m++) acc = acc + h[m] * v[m];
                    }
                    acc = acc + row_shift<true, 8>(acc);
                    acc = acc + row_shift<true, 4>(acc);
                    acc = acc + row_shift<true, 2>(acc);
                    acc = acc + row_shift<true, 1>(acc);
                    if (l == 15) stage_n[sr * kStreamTY + q * kQRows + slot_row] = acc;
                }
            }

            // ---- y: contract this quarter's rows with Hy (thread = column) ----
            if (nyk > 0) {
                Acc col[kQRows];
#pragma unroll
                for (int i = 0; i < kQRows; i++) col[i] = tile[i * kFusedTX + e];
                if constexpr (std::is_same<Acc, float>::value) {
#pragma unroll
                    for (int gp = 0; gp < (kFusedMaxScans * K + 1) / 2; gp++) {
                        if (2 * gp < nyk) {
                            const int j0 = 2 * gp, j1 = (2 * gp + 1 < nyk) ? 2 * gp + 1 : 2 * gp;
                            const Acc *h0 = Hy + (size_t)(vy * nyk + j0) * kStreamTY + q * kQRows;     // wave-uniform
                            const Acc *h1 = Hy + (size_t)(vy * nyk + j1) * kStreamTY + q * kQRows;
                            F2s c0 = F2s{0.0f, 0.0f}, c1 = F2s{0.0f, 0.0f};
#pragma unroll
                            for (int i = 0; i < kQRows; i += 2) {
                                const F2s cc = F2s{col[i], col[i + 1]};
                                c0 = F2s{h0[i], h0[i + 1]} * cc + c0;
                                c1 = F2s{h1[i], h1[i + 1]} * cc + c1;
                            }
                            comb[2 * gp] = comb[2 * gp] + (c0.x + c0.y);
                            if (2 * gp + 1 < kFusedMaxScans * K) comb[2 * gp + 1] = comb[2 * gp + 1] + (c1.x + c1.y);
                        }
                    }
                } else {
#pragma unroll
                    for (int jr = 0; jr < kFusedMaxScans * K; jr++) {
                        if (jr < nyk) {
                            const Acc *hy = Hy + (size_t)(vy * nyk + jr) * kStreamTY + q * kQRows;
#pragma unroll
                            for (int i = 0; i < kQRows; i++) comb[jr] = comb[jr] + hy[i] * col[i];
                        }
                    }
                }
            }
        }
        // combined rows -> yt; with x scans in the filter xscan_rows_kernel finishes them in place
        if (nyk > 0) {
            const int64_t line = (int64_t)c.tx * kFusedTX + t + a.NXP * c.z;
#pragma unroll
            for (int jr = 0; jr < kFusedMaxScans * K; jr++)
                if (jr < nyk) a.yt[(((int64_t)(jr / K) * a.MY + c.ty) * K + jr % K) * Ly + line] = comb[jr];
        }
    }
    lds_barrier();                                                  // final: the last tile's stage is complete
    if (n_my > 0) flush_xtails(n_my - 1);
}

int stream_grid_size(int64_t n_tiles, int MX) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    int64_t g = 2ll * cus;                       // two workgroups per CU (LDS: two rings)
    if (const char *env = getenv("RF_STREAM_GRID")) g = atoll(env);
    // a stride that is a multiple of the tiles per row keeps a walker in one tile column: one Hx variant for life
    if (g > MX && g % MX != 0) g -= g % MX;
    if (g > n_tiles) g = n_tiles;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

bool stream_tails_applicable(int K, int TY, bool src_u8, int pw_flags, int last_cols, int last_rows, int64_t n_tiles) {
    static const bool off = getenv("RF_NO_STREAM_TAILS") != nullptr;      // A/B runs against fused_tails_kernel
    if (off || src_u8 || (pw_flags & 1) || TY != kStreamTY || last_cols != kFusedTX || last_rows != TY) return false;
    if (K < 1 || K > 3) return false;
    static const int64_t min_tiles = getenv("RF_STREAM_MIN_TILES") ? atoll(getenv("RF_STREAM_MIN_TILES")) : 2048;
    return n_tiles >= min_tiles;      // below that a walker has too few tiles to amortise its pipeline fill
}

template <typename P>
int launch_stream_tails(int K, const P *src, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                        const typename PixelTraits<P>::Acc *Hx, const typename PixelTraits<P>::Acc *Hy, hipStream_t stream) {
    using Acc = typename PixelTraits<P>::Acc;
    const int64_t n_tiles = (int64_t)a.MX * a.MY * a.NZ;
    if (n_tiles <= 0) return RF_OK;
    if (n_tiles >= (1ll << 31)) { set_error("stream tails: too many tiles"); return RF_ERR_UNSUPPORTED; }
    const int grid = stream_grid_size(n_tiles, a.MX);
    if (grid <= 0) { set_error("stream tails: no device properties"); return RF_ERR_HIP; }
    constexpr int SLOTS = 4, AHEAD = 3;
    const int nxk = a.nx * K;
    const size_t lds = (size_t)SLOTS * kSlotBytes + (size_t)(nxk > 0 ? nxk : 1) * kFusedTX * sizeof(Acc) +
                       (size_t)2 * (nxk > 0 ? nxk : 1) * kStreamTY * sizeof(Acc);
#define RF_CASE(KK)                                                                                                      \
    if (K == KK) {                                                                                                       \
        auto kern = &stream_tails_kernel<P, KK, SLOTS, AHEAD>;                                                           \
        static std::atomic<bool> attr_set[64];                                                                           \
        int dev = 0;                                                                                                     \
        RF_HIP_CHECK(hipGetDevice(&dev));                                                                                \
        if (!attr_set[dev & 63].load(std::memory_order_acquire)) {                                                       \
            RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)(SLOTS * kSlotBytes + 3 * kFusedMaxScans * KK * kFusedTX * sizeof(Acc)))); \
            attr_set[dev & 63].store(true, std::memory_order_release);                                                   \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kStreamThreads), lds, stream, src, a, Hx, Hy, (uint32_t)n_tiles);      \
        RF_HIP_CHECK(hipGetLastError());                                                                                 \
        return RF_OK;                                                                                                    \
    }
    RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    set_error("stream tails: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}

template int launch_stream_tails<float>(int, const float *, const FusedArgs<float> &, const float *, const float *, hipStream_t);
template int launch_stream_tails<int32_t>(int, const int32_t *, const FusedArgs<uint32_t> &, const uint32_t *, const uint32_t *,
                                          hipStream_t);

}  // namespace rf
