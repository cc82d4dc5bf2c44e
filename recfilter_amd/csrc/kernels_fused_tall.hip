// kernels_fused_tall.hip -- pass 2 on 256 x 128 tiles.
//
// The bytes the fused path moves besides the image are tails: k samples per tile border, scan and line.  Their volume is
// proportional to 1/256 + 1/TY, and every kernel between the two passes (the x residual of the y tails, the y carry scan)
// reads and writes the y tails once -- at TY = 64 that is 0.45 GB against the image's 1 GB on cfg3.  Doubling the tile height
// halves the y tails and with them those kernels.  A 256 x 128 tile is 128 KiB: it does not fit the LDS twice per CU, but
// the y phase never needed the tile in LDS -- it holds its column in registers.  So the tile goes through the SAME 64 KiB of
// LDS as two halves of 64 rows (load, x phase, columns out), the thread keeps its 128-sample column in registers, runs all
// y scans on it and stores.  Everything else is fused_pass2_kernel (kernels_fused.hip): same carries, same tables, same
// arithmetic per sample.
#include <atomic>
#include <type_traits>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

constexpr int kTallTY = 128;
constexpr int kHalfRows = 64;

// YPAT: the directions of the y scans when they are the usual ones -- 1: one causal scan, 2: causal then anticausal; 0: any
// (a run-time direction inside the loop over the scans makes every sample of the column a phi of two register
// assignments: a hundred and more register copies per scan and, on a 128-sample column, spills).
// EARLY (whole tiles, one of the fixed patterns, no epilogue): the rows are stored from inside the last scan.
// XFIX: the x scans have the same pattern as the y scans (YPAT), fixed at compile time as well.
template <typename P, int K, bool EDGE, typename PI, int YPAT, bool EARLY, bool XFIX>
__global__ void __launch_bounds__(kFusedThreads, 2)
fused_pass2_tall_kernel(const PI *__restrict__ src, P *__restrict__ dst, FusedArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    using A4 = typename Vec4<Acc>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Acc *tile = reinterpret_cast<Acc *>(lds_raw);
    A4 *tile4 = reinterpret_cast<A4 *>(lds_raw);
    constexpr int TY = kTallTY, TL = kHalfRows, NR = TL / 16;

    const int t = threadIdx.x;
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (a.xcd_contig) {                                        // (uniform; FusedArgs::xcd_contig)
        const unsigned gxn = gridDim.x, total = gxn * gridDim.y;
        unsigned b = (unsigned)by * gxn + (unsigned)bx;
        b = (b & 7u) * (total >> 3) + (b >> 3);
        bx = (int)(b % gxn); by = (int)(b / gxn);
    }
    const int tx = bx + a.tx0, ty = by + a.ty0;
    const int64_t z = blockIdx.z;
    if (a.plane_batch) {
        src = reinterpret_cast<const PI *>(a.in_planes[z]);
        dst = reinterpret_cast<P *>(a.out_planes[z]);
    }
    const int64_t tile_off = (a.plane_batch ? 0 : z * a.NX * a.NY) + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX;
    const int l = t & 15, slot = t >> 4, sw = (l >> 2) & 3;    // x phase: segment lane, row slot
    const int cc = t & 63, rg = t >> 6;                        // load: 16-byte chunk, row group
    const int e = (swz_chunk(t >> 2) << 2) | (t & 3);          // y phase: swizzled column offset
    const int64_t Lx = a.NYP * a.NZ, Ly = a.NXP * a.NZ;
    const int64_t line = (int64_t)tx * kFusedTX + t + a.NXP * z;       // y phase: this thread's column
    const int last_lane = (EDGE && tx == a.MX - 1) ? a.last_lane : 15;
    const int last_cols = (EDGE && tx == a.MX - 1) ? a.last_cols : kFusedTX;
    const int entry_valid = last_cols - 16 * last_lane;
    const int rows_here = (EDGE && ty == a.MY - 1) ? a.last_rows : TY;

    const char *spb = reinterpret_cast<const char *>(src + tile_off);
    const uint32_t in_row_bytes = a.row_bytes / (uint32_t)sizeof(P) * (uint32_t)sizeof(PI);
    const uint32_t off0 = (uint32_t)rg * in_row_bytes + (uint32_t)cc * (uint32_t)(4 * sizeof(PI));
    const bool chunk_in = 4 * cc < last_cols;
    // ... and when the width is not a multiple of 4 the last of them is partial (tile-uniform flag; scan_device.h)
    const bool odd_cols = EDGE && (last_cols & 3) != 0;
    const int cols_valid = last_cols - 4 * cc;
    const A4 zero4 = A4{Acc(0), Acc(0), Acc(0), Acc(0)};

    Acc col[TY];              // this thread's column, all 128 rows
    A4 tmp[2][TL / 4];       // both halves are requested up front
    auto request_pixels = [&](int h) {
        // wave w streams rows w, w+4, ... of the half
#pragma unroll
        for (int i = 0; i < TL / 4; i++) {
            const int row = TL * h + rg + 4 * i;
            const bool in = chunk_in && (!EDGE || row < rows_here);
            if (odd_cols) tmp[h][i] = in ? load_chunk_cols<PI, Acc>(spb + (off0 + (uint32_t)(TL * h + 4 * i) * in_row_bytes), cols_valid) : zero4;
            else tmp[h][i] = in ? load_chunk<PI, Acc>(spb + (off0 + (uint32_t)(TL * h + 4 * i) * in_row_bytes)) : zero4;
        }
    };

    // Order of the requests: the first half's pixels, then every carry, then the second half's pixels.  (Carries first -- round
    // 1's order -- delays the first pixel request by ~90 load instructions per tile; the x phase of the first half waits for
    // "everything but the second half" either way.)
    request_pixels(0);
    // ---- y carries: requested first, used last ----
    Acc CY[kFusedMaxScans][K];
#pragma unroll
    for (int j = 0; j < kFusedMaxScans; j++) {
#pragma unroll
        for (int r = 0; r < K; r++) CY[j][r] = Acc(0);
        if (j < a.ny) {
            const bool causal = a.ys[j].causal != 0;
            const bool tile_first = causal ? (ty == 0) : (ty == a.MY - 1);
            if (tile_first) {
#pragma unroll
                for (int r = 0; r < K; r++) CY[j][r] = a.y_incoming[((int64_t)j * K + r) * Ly + line];
            } else {
                const int tp = causal ? ty - 1 : ty + 1;
#pragma unroll
                for (int r = 0; r < K; r++) CY[j][r] = a.yt[a.yt_index(j, tp, r, K, line)];
            }
        }
    }
    // row shard: tails completed with zero entering carries + Y * (true entering carries), as in fused_pass2_kernel.  The
    // entering carries are requested here, with the other carries; the correction itself waits until the pixels have been
    // requested (apply_entering_carries below) -- computed here it made every tile wait for its carry loads before it asked
    // for a single pixel.
    Acc yin[kFusedMaxScans][K];
    if (a.y_apply != nullptr) {
#pragma unroll
        for (int q = 0; q < kFusedMaxScans; q++)
#pragma unroll
            for (int o = 0; o < K; o++) yin[q][o] = q < a.ny ? a.y_incoming[((int64_t)q * K + o) * Ly + line] : Acc(0);
    }
    auto apply_entering_carries = [&]() {
        if (a.y_apply == nullptr) return;
#pragma unroll
        for (int j = 0; j < kFusedMaxScans; j++) {
            if (j < a.ny) {
                const bool causal = a.ys[j].causal != 0;
                const bool tile_first = causal ? (ty == 0) : (ty == a.MY - 1);
                if (!tile_first) {
                    const int tp = causal ? ty - 1 : ty + 1;
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        Acc add = Acc(0);
#pragma unroll
                        for (int q = 0; q <= j; q++) {
                            const Acc *Ym = a.y_apply + ((((int64_t)q * a.ny + j) * a.MY + tp) * K + r) * K;
#pragma unroll
                            for (int o = 0; o < K; o++) add = add + Ym[o] * yin[q][o];
                        }
                        CY[j][r] = CY[j][r] + add;
                    }
                }
            }
        }
    };

    // The x carries entering the rows of both halves: requested before the pixels, parked in LDS behind the tile once they
    // have arrived ([half][s][n][j][row slot]; read by every lane of a row, used by its entry lane) -- a 128-sample column,
    // a prefetched half tile and the x phase leave no registers for them.
    Acc *cx_lds = tile + TL * kFusedTX;
    Acc CX[2][kFusedMaxScans][NR][K];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int64_t line0 = (int64_t)ty * TY + TL * h + slot + a.NYP * z;
#pragma unroll
        for (int s = 0; s < kFusedMaxScans; s++) {
#pragma unroll
            for (int n = 0; n < NR; n++)
#pragma unroll
                for (int j = 0; j < K; j++) CX[h][s][n][j] = Acc(0);
            if (s < a.nx) {
                const bool causal = a.xs[s].causal != 0;
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == last_lane);
                if (first_lane) {
                    const int tp = causal ? tx - 1 : tx + 1;
                    const Acc *cp = tile_first ? a.x_incoming + (int64_t)s * K * Lx : a.xt + ((int64_t)s * a.MX + tp) * K * Lx;
#pragma unroll
                    for (int n = 0; n < NR; n++)
#pragma unroll
                        for (int j = 0; j < K; j++) CX[h][s][n][j] = cp[j * Lx + line0 + 16 * n];
                }
            }
        }
    }
    auto park_carries = [&]() {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s = 0; s < kFusedMaxScans; s++) {
                if (s < a.nx) {
                    const bool causal = a.xs[s].causal != 0;
                    const bool first_lane = causal ? (l == 0) : (l == last_lane);
                    if (first_lane) {
#pragma unroll
                        for (int n = 0; n < NR; n++)
#pragma unroll
                            for (int j = 0; j < K; j++) cx_lds[(((h * kFusedMaxScans + s) * NR + n) * K + j) * 16 + slot] = CX[h][s][n][j];
                    }
                }
            }
    };
    request_pixels(1);
    park_carries();
    apply_entering_carries();
    // ... and wait in LDS meanwhile, [j][r][column] behind the x carries (registers are the scarce resource here)
    Acc *cy_lds = tile + kHalfRows * kFusedTX + 2 * kFusedMaxScans * (kHalfRows / 16) * K * 16;
#pragma unroll
    for (int j = 0; j < kFusedMaxScans; j++)
        if (j < a.ny) {
#pragma unroll
            for (int r = 0; r < K; r++) cy_lds[(j * K + r) * kFusedTX + t] = CY[j][r];
        }

#pragma unroll
    for (int h = 0; h < TY / TL; h++) {
        if constexpr (!PixelTraits<P>::is_integer) {
            if (a.pw_flags & 1) {
#pragma unroll
                for (int i = 0; i < TL / 4; i++) {
                    const bool in = chunk_in && TL * h + rg + 4 * i < rows_here;      // samples beyond the image stay zero
                    const Acc s = in ? a.pre_s : Acc(0), b = in ? a.pre_b : Acc(0);
                    tmp[h][i].x = s * tmp[h][i].x + b; tmp[h][i].y = s * tmp[h][i].y + b;
                    tmp[h][i].z = s * tmp[h][i].z + b; tmp[h][i].w = s * tmp[h][i].w + b;
                    if (odd_cols) clear_dead_cols<A4, Acc>(tmp[h][i], cols_valid);
                }
            }
        }
        if (h > 0) __syncthreads();                             // the previous half's columns are out of LDS
#pragma unroll
        for (int i = 0; i < TL / 4; i++) tile4[(rg + 4 * i) * 64 + swz_chunk(cc)] = tmp[h][i];
        __syncthreads();
        // ---- x phase of this half ----
        if (a.nx > 0) {
            Acc v[NR][kFusedSeg];
#pragma unroll
            for (int n = 0; n < NR; n++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    A4 q = tile4[(slot + 16 * n) * 64 + 4 * l + (j ^ sw)];
                    v[n][4 * j + 0] = q.x; v[n][4 * j + 1] = q.y; v[n][4 * j + 2] = q.z; v[n][4 * j + 3] = q.w;
                }
            }
            auto x_scan = [&](int s, auto causal_tag) __attribute__((always_inline)) {
                constexpr bool causal = decltype(causal_tag)::value;
                const FusedScan<Acc> &sc = a.xs[s];
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == last_lane);
                const bool clamp_first = a.clamped && tile_first && first_lane;
                Acc cx[NR][K];
#pragma unroll
                for (int n = 0; n < NR; n++)
#pragma unroll
                    for (int j = 0; j < K; j++) cx[n][j] = cx_lds[(((h * kFusedMaxScans + s) * NR + n) * K + j) * 16 + slot];
                bool cf = clamp_first;
                if constexpr (!(XFIX && YPAT >= 1)) {              // (the general-pattern variants: what a mod-form plan launches)
                    if (a.mod_form) {
                        if (a.clamped && tile_first) border_mod_rows16<Acc, causal, NR>(v, sc, clamp_first);
                        cf = false;
                    }
                }
                if constexpr (causal) scan_rows16<Acc, true, K, NR>(v, sc, first_lane, cf, cx);
                else scan_rows16<Acc, false, K, NR>(v, sc, first_lane, cf, cx, l > last_lane, EDGE ? entry_valid : kFusedSeg);
            };
            // (the x scans' directions are compile-time too where they follow the y pattern: every row of the tile would
            // otherwise be a phi of two register assignments per scan, as the column is in the y phase)
            if constexpr (XFIX && YPAT == 1) {
                x_scan(0, std::true_type{});
            } else if constexpr (XFIX && YPAT == 2) {
                x_scan(0, std::true_type{});
                x_scan(1, std::false_type{});
            } else {
#pragma unroll 1
                for (int s = 0; s < a.nx; s++) {
                    if (a.xs[s].causal != 0) x_scan(s, std::true_type{});
                    else x_scan(s, std::false_type{});
                }
            }
#pragma unroll
            for (int n = 0; n < NR; n++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    A4 q;
                    q.x = v[n][4 * j + 0]; q.y = v[n][4 * j + 1]; q.z = v[n][4 * j + 2]; q.w = v[n][4 * j + 3];
                    tile4[(slot + 16 * n) * 64 + 4 * l + (j ^ sw)] = q;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TL; i++) col[TL * h + i] = tile[i * kFusedTX + e];
    }

    // ---- y phase: thread = column, 128 rows ----
    auto y_scan = [&](int j, auto causal_tag) __attribute__((always_inline)) {
        constexpr bool causal = decltype(causal_tag)::value;
        const FusedScanY<Acc> &sc = a.ys[j];
        const bool border = causal ? (ty == 0 && a.y_first_border) : (ty == a.MY - 1 && a.y_last_border);
        const bool clamp_first = a.clamped && border;
        Acc c[K];
#pragma unroll
        for (int r = 0; r < K; r++) c[r] = cy_lds[(j * K + r) * kFusedTX + t];      // (written by this thread)
        bool cf = clamp_first;
        if constexpr (YPAT == 0) {
            if (a.mod_form) {
                if (clamp_first) border_mod_col<Acc, causal, TY>(col, sc);
                cf = false;
            }
        }
        if constexpr (causal) scan_col<Acc, true, K, TY>(col, sc, cf, c);
        else {
            if (rows_here == TY) scan_col<Acc, false, K, TY>(col, sc, cf, c);
            else scan_col_partial_up<Acc, K, TY>(col, sc, cf, rows_here);
        }
    };
    // whole tiles without an epilogue: the rows are stored from inside the last scan, each as soon as it is final
    char *dpb_early = reinterpret_cast<char *>(dst + tile_off);
    static_assert(!EARLY || (!EDGE && YPAT > 0), "early stores: whole tiles, fixed scan pattern");
    auto row_out = [&](int m, Acc v) __attribute__((always_inline)) {
        __builtin_nontemporal_store(PixelTraits<P>::store(v),
                                    reinterpret_cast<P *>(dpb_early + ((uint32_t)t * (uint32_t)sizeof(P) + (uint32_t)m * a.row_bytes)));
    };
    if constexpr (YPAT == 1 || YPAT == 2) {
        if (YPAT == 2) y_scan(0, std::true_type{});
        if constexpr (EARLY) {
            constexpr int jl = YPAT - 1;
            const bool clamp_first = a.clamped && (YPAT == 1 ? (ty == 0 && a.y_first_border) : (ty == a.MY - 1 && a.y_last_border));
            Acc c[K];
#pragma unroll
            for (int r = 0; r < K; r++) c[r] = cy_lds[(jl * K + r) * kFusedTX + t];
            scan_col<Acc, YPAT == 1, K, TY>(col, a.ys[jl], clamp_first, c, row_out);
            return;
        } else {
            if (YPAT == 1) y_scan(0, std::true_type{});
            else y_scan(1, std::false_type{});
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < a.ny; j++) {
            if (a.ys[j].causal != 0) y_scan(j, std::true_type{});
            else y_scan(j, std::false_type{});
        }
    }
    if constexpr (!PixelTraits<P>::is_integer) {
        if (a.pw_flags & 2) {
            if (a.post_i != Acc(0)) {
                // the epilogue's input operand comes back through L2 / the Infinity Cache (no registers to spare here)
                const PI *xp = src + tile_off;
                const uint32_t nxu = (uint32_t)a.NX;
                const Acc c1 = a.post_i * ((a.pw_flags & 1) ? a.pre_s : Acc(1));
                const Acc c2 = a.post_b + a.post_i * ((a.pw_flags & 1) ? a.pre_b : Acc(0));
                if (t < last_cols) {
#pragma unroll
                    for (int i = 0; i < TY; i++)
                        if (i < rows_here) col[i] = a.post_f * col[i] + (c1 * (Acc)xp[(uint32_t)t + (uint32_t)i * nxu] + c2);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TY; i++) col[i] = a.post_f * col[i] + a.post_b;
            }
        }
    }
    {
        char *dpb = reinterpret_cast<char *>(dst + tile_off);
        const uint32_t row_bytes = a.row_bytes;
        if (t < last_cols) {
#pragma unroll
            for (int i = 0; i < TY; i++)
                if (!EDGE || i < rows_here)
                    __builtin_nontemporal_store(PixelTraits<P>::store(col[i]),
                                                reinterpret_cast<P *>(dpb + ((uint32_t)t * (uint32_t)sizeof(P) + (uint32_t)i * row_bytes)));
        }
    }
}

template <typename P, int K, bool EDGE, typename PI, int YPAT, bool EARLY, bool XFIX = false>
int launch_tall_pat(const PI *src, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    using Acc = typename PixelTraits<P>::Acc;
    // the half tile + the x carries of both halves ([2][4 scans][4 rows][K][16 slots])
    // ... and the y carries of the columns ([ny * K][256])
    const size_t lds = ((size_t)kHalfRows * kFusedTX + 2 * kFusedMaxScans * (kHalfRows / 16) * K * 16 + (size_t)a.ny * K * kFusedTX) * sizeof(Acc);
    const size_t lds_max = ((size_t)kHalfRows * kFusedTX + 2 * kFusedMaxScans * (kHalfRows / 16) * K * 16 + (size_t)kFusedMaxScans * K * kFusedTX) * sizeof(Acc);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    std::atomic<bool> &done = attr_set[dev & 63];
    if (!done.load(std::memory_order_acquire)) {
        RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_pass2_tall_kernel<P, K, EDGE, PI, YPAT, EARLY, XFIX>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        done.store(true, std::memory_order_release);
    }
    dim3 grid((unsigned)(a.gx > 0 ? a.gx : a.MX), (unsigned)(a.gy > 0 ? a.gy : a.MY), (unsigned)a.NZ);
    FusedArgs<Acc> aa = a;
    aa.xcd_contig = (a.row_bytes % 128u != 0 && (grid.x * grid.y) % 8u == 0 && grid.x * grid.y >= 2048u) ? 1 : 0;
    hipLaunchKernelGGL((fused_pass2_tall_kernel<P, K, EDGE, PI, YPAT, EARLY, XFIX>), grid, dim3(kFusedThreads), lds, stream, src, dst, aa);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename P, int K, bool EDGE, typename PI>
int launch_tall_impl(const PI *src, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    // (a plan in mod form -- FusedArgs::mod_form -- takes the general-pattern code, which applies its border modifications)
    const int pat = a.mod_form ? 0 : (a.ny == 1 && a.ys[0].causal != 0) ? 1 : (a.ny == 2 && a.ys[0].causal != 0 && a.ys[1].causal == 0) ? 2 : 0;
    bool early = !EDGE && pat > 0;
    if constexpr (!PixelTraits<P>::is_integer) early = early && (a.pw_flags & 2) == 0;
    const int xpat = a.mod_form ? 0 : (a.nx == 1 && a.xs[0].causal != 0) ? 1 : (a.nx == 2 && a.xs[0].causal != 0 && a.xs[1].causal == 0) ? 2 : 0;
    if constexpr (!EDGE) {
        if (early && pat == 1 && xpat == 1) return launch_tall_pat<P, K, EDGE, PI, 1, true, true>(src, dst, a, stream);
        if (early && pat == 2 && xpat == 2) return launch_tall_pat<P, K, EDGE, PI, 2, true, true>(src, dst, a, stream);
        if (early && pat == 1) return launch_tall_pat<P, K, EDGE, PI, 1, true>(src, dst, a, stream);
        if (early && pat == 2) return launch_tall_pat<P, K, EDGE, PI, 2, true>(src, dst, a, stream);
    }
    if (pat == 1) return launch_tall_pat<P, K, EDGE, PI, 1, false>(src, dst, a, stream);
    if (pat == 2) return launch_tall_pat<P, K, EDGE, PI, 2, false>(src, dst, a, stream);
    return launch_tall_pat<P, K, EDGE, PI, 0, false>(src, dst, a, stream);
}

}  // namespace

template <typename P>
int launch_fused_pass2_tall(int K, const void *src, bool src_u8, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                            hipStream_t stream) {
    if (a.MX <= 0 || a.MY <= 0 || a.NZ <= 0) return RF_OK;
    if (a.NZ > 65535 || a.MY > 65535) { set_error("fused path: grid too large"); return RF_ERR_UNSUPPORTED; }
    const bool edge = a.last_cols != kFusedTX || a.last_rows != kTallTY;
    // one launch of `aa`'s tiles on the lean (edge = false) or the EDGE variant
    auto one = [&](const FusedArgs<typename PixelTraits<P>::Acc> &aa, bool e) -> int {
#define RF_CASE(KK)                                                                                                     \
    if (K == KK) {                                                                                                      \
        if constexpr (std::is_same<P, float>::value) {                                                                  \
            if (src_u8) return e ? launch_tall_impl<P, KK, true, uint8_t>((const uint8_t *)src, dst, aa, stream)         \
                                 : launch_tall_impl<P, KK, false, uint8_t>((const uint8_t *)src, dst, aa, stream);       \
        }                                                                                                               \
        return e ? launch_tall_impl<P, KK, true, P>((const P *)src, dst, aa, stream)                                    \
                 : launch_tall_impl<P, KK, false, P>((const P *)src, dst, aa, stream);                                  \
    }
        RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
        set_error("fused path: unsupported order %d", K);
        return (int)RF_ERR_UNSUPPORTED;
    };
    // Partial tiles.  The EDGE variant costs the final pass almost twice its time per tile (no rows leave from inside the last
    // scan, a bound and a select per access, 24-144 bytes of scratch: 16384 x 16256, 0.662 ms against 0.366 ms for 16384^2),
    // and only the last tile column / row needs it: the whole tiles run on the lean kernel, the two strips as launches of
    // their own on the EDGE variant (FusedArgs::tx0 ..): 0.386 + 0.042 ms.  (One launch with a per-workgroup branch between
    // the two bodies was slower than the three launches: 0.531 ms.)
    static const bool edge_everywhere = RF_KNOB("RF_TALL_EDGE_EVERYWHERE") != nullptr;      // A/B runs
    if (edge && !edge_everywhere) {
        const int MXf = a.MX - (a.last_cols != kFusedTX ? 1 : 0), MYf = a.MY - (a.last_rows != kTallTY ? 1 : 0);
        FusedArgs<typename PixelTraits<P>::Acc> part = a;
        part.gx = MXf; part.gy = MYf;
        int rc = (MXf > 0 && MYf > 0) ? one(part, false) : (int)RF_OK;
        if (rc == RF_OK && MXf < a.MX) { part = a; part.tx0 = a.MX - 1; part.gx = 1; part.gy = a.MY; rc = one(part, true); }
        if (rc == RF_OK && MYf < a.MY && MXf > 0) { part = a; part.ty0 = a.MY - 1; part.gy = 1; part.gx = MXf; rc = one(part, true); }
        return rc;
    }
    return one(a, edge);
}

template int launch_fused_pass2_tall<float>(int, const void *, bool, float *, const FusedArgs<float> &, hipStream_t);
template int launch_fused_pass2_tall<int32_t>(int, const void *, bool, int32_t *, const FusedArgs<uint32_t> &, hipStream_t);
template int launch_fused_pass2_tall<int16_t>(int, const void *, bool, int16_t *, const FusedArgs<uint32_t> &, hipStream_t);

}  // namespace rf
