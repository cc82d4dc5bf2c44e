// kernels_fused.hip -- the bandwidth-tuned path: LDS-staged fused x/y tiles for gfx950.
//
// One workgroup owns a 256 x TY tile of one image plane, stages it in LDS once and runs EVERY
// x scan and EVERY y scan of the filter on it before anything goes back to HBM -- the
// "overlapped" evaluation of lib/split.cpp (add_residuals_to_final_result :1647-1780) with
// MI355X-shaped tiles instead of the reference's 32 x 32 CUDA tiles:
//
//   load     256 threads, one 1 KiB tile row per wave instruction (16 B / lane, coalesced),
//            written to LDS with an XOR swizzle of the 16-byte chunk index
//   x phase  thread = (row, 16-sample segment); a tile row is exactly one 16-lane DPP row.
//            segment-local recurrence in registers -> Kogge-Stone scan of the k-vector segment
//            states across the 16 lanes with row_shr/row_shl DPP and precomputed powers of the
//            segment transfer matrix -> rank-k correction of the 16 samples.
//            LDS reads/writes are ds_read/write_b128, conflict-free thanks to the swizzle.
//   y phase  thread = column; the TY samples of the column sit in registers, every y scan is
//            a serial recurrence up or down the registers (no LDS traffic between scans)
//   store    the register column goes straight to HBM, 256 B per wave instruction, after the fused pointwise
//            epilogue if there is one
//
// This file holds pass 2, the final correction pass (lib/split.cpp:1008-1130, 1647-1780).  Pass 1 (tail extraction as
// a contraction) and the completion of the y tails are in kernels_tails.hip, the carry recurrences in
// kernels_carry.hip; plan_fused.cpp strings them together.
#include <atomic>
#include <type_traits>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

// ---- pass 2, the final correction pass: one workgroup per 256 x TY tile -----------------------------
// The completed carries the tile needs (first lane of each row for the x scans, every column for the y
// scans) are requested BEFORE the pixels, so their latency hides behind the 64 KiB pixel load instead of
// stalling every scan.  (Pass 1 is the contraction of kernels_tails.hip; the scan-everything pass 1 and a
// persistent, register-prefetching variant of this kernel were measured slower in round 1 and removed.)
// EDGE: the image has partial tiles (width not a multiple of 256 or height not a multiple of TY); without it the
// masks below are compile-time constants and the kernel stays lean.
// YPAT: the directions of the y scans when they are the usual ones -- 1: one causal scan, 2: causal then anticausal; 0: any.
// With a run-time direction inside the loop over the scans every sample of the column is a phi of two register
// assignments: ~TY register copies per scan (and spills on the 128-row tiles of kernels_fused_tall.hip).
// LIN: a folded 1-D signal whose end falls inside the image (FusedArgs::lin_limit) -- its masked loads and stores are a
// variant of the plain kernel (no epilogue operand, whole tiles, no y scans), so that no other kernel carries them
// MOD: the plan's scans are in zero-border form behind border modifications (FusedArgs::mod_form) -- an instance of its own of
// the general-pattern code (inside the common EDGE instance the eight modification factors per scan exhausted its scalar
// registers: 173 / 203 / 236 vector registers became 233 / 265 / 297 at orders 1 / 2 / 3, for every image with partial tiles)
template <typename P, int K, int TY, bool EPI, bool EDGE, typename PI, int YPAT = 0, bool XFIX = false, bool LIN = false, bool MOD = false>
__global__ void __launch_bounds__(kFusedThreads, (EPI || PixelTraits<P>::is_integer) ? 2 : 1)
fused_pass2_kernel(const PI *__restrict__ src, P *__restrict__ dst, FusedArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    using A4 = typename Vec4<Acc>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Acc *tile = reinterpret_cast<Acc *>(lds_raw);
    A4 *tile4 = reinterpret_cast<A4 *>(lds_raw);
    constexpr int NR = TY / 16;

    const int t = threadIdx.x;
    const int tx = (int)blockIdx.x + a.tx0, ty = (int)blockIdx.y + a.ty0;
    const int64_t z = blockIdx.z;
    // batched Tuple planes: plane z is its own buffer (wave-uniform pointer pick from the kernel arguments)
    if (a.plane_batch) {
        src = reinterpret_cast<const PI *>(a.in_planes[z]);
        dst = reinterpret_cast<P *>(a.out_planes[z]);
    }
    const int64_t tile_off = (a.plane_batch ? 0 : z * a.NX * a.NY) + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX;
    const int l = t & 15, slot = t >> 4, sw = (l >> 2) & 3;    // x phase: segment lane, row slot
    const int64_t Lx = a.NYP * a.NZ, Ly = a.NXP * a.NZ;
    const int64_t line0 = (int64_t)ty * TY + slot + a.NYP * z;         // x phase: row n -> line0 + 16 n
    const int64_t line = (int64_t)tx * kFusedTX + t + a.NXP * z;       // y phase: this thread's column
    // a row's last tile may be partial: its missing samples are zeros on load, skipped on store, and an anticausal
    // x scan enters at the last existing segment
    const int last_lane = (EDGE && tx == a.MX - 1) ? a.last_lane : 15;
    const int last_cols = (EDGE && tx == a.MX - 1) ? a.last_cols : kFusedTX;       // columns of this tile that exist
    const int entry_valid = last_cols - 16 * last_lane;                             // ... in the last existing segment
    // ... and so may the last tile row: rows_here of its TY rows exist, an anticausal y scan enters at the last of them
    const int rows_here = (EDGE && ty == a.MY - 1) ? a.last_rows : TY;

    // ---- carries (pass 2) ----
    Acc CX[kFusedMaxScans][NR][K];
    Acc CY[kFusedMaxScans][K];
#pragma unroll
    for (int s = 0; s < kFusedMaxScans; s++) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            CY[s][j] = Acc(0);
#pragma unroll
            for (int n = 0; n < NR; n++) CX[s][n][j] = Acc(0);
        }
    }
    {
#pragma unroll
        for (int s = 0; s < kFusedMaxScans; s++) {
            if (s < a.nx) {
                const bool causal = a.xs[s].causal != 0;
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == last_lane);
                if (first_lane) {
                    // previous tile's completed tail, or (first tile of the row) the state entering the row
                    const int tp = causal ? tx - 1 : tx + 1;
                    const Acc *cp = tile_first ? a.x_incoming + (int64_t)s * K * Lx
                                               : a.xt + ((int64_t)s * a.MX + tp) * K * Lx;
#pragma unroll
                    for (int n = 0; n < NR; n++)
#pragma unroll
                        for (int j = 0; j < K; j++) CX[s][n][j] = cp[j * Lx + line0 + 16 * n];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kFusedMaxScans; j++) {
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
        if (a.y_apply != nullptr) {
            // row shard: tails completed with zero entering carries + Y * (true entering carries), in the summation
            // order of merged_apply_kernel (kernels_generic.hip).  Y is uniform per workgroup: scalar loads.
            Acc yin[kFusedMaxScans][K];
#pragma unroll
            for (int q = 0; q < kFusedMaxScans; q++)
#pragma unroll
                for (int o = 0; o < K; o++)
                    yin[q][o] = q < a.ny ? a.y_incoming[((int64_t)q * K + o) * Ly + line] : Acc(0);
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
        }
    }

    // ---- load: wave w streams rows w, w+4, ...; one 1 KiB row per instruction ----
    {
        const int cc = t & 63, rg = t >> 6;
        // byte offsets inside the tile, kept in 32 bits: scalar base + 32-bit vector offset addressing
        const char *spb = reinterpret_cast<const char *>(src + tile_off);
        const uint32_t in_row_bytes = a.row_bytes / (uint32_t)sizeof(P) * (uint32_t)sizeof(PI);
        const uint32_t off0 = (uint32_t)rg * in_row_bytes + (uint32_t)cc * (uint32_t)(4 * sizeof(PI));
        auto ld = [&](int row) { return load_chunk<PI, Acc>(spb + (off0 + (uint32_t)row * in_row_bytes)); };
        auto ld_head = [&](int row, int n) { return load_chunk_head<PI, Acc>(spb + (off0 + (uint32_t)row * in_row_bytes), n); };
        A4 tmp[TY / 4];
        const bool chunk_in = 4 * cc < last_cols;            // this thread's 16-byte chunk exists in the image
        // ... and when the width is not a multiple of 4 the last of them is partial (tile-uniform flag; scan_device.h)
        const bool odd_cols = (last_cols & 3) != 0;
        const int cols_valid = last_cols - 4 * cc;
        auto ld_cols = [&](int row) { return load_chunk_cols<PI, Acc>(spb + (off0 + (uint32_t)row * in_row_bytes), cols_valid); };
        const A4 zero4 = A4{Acc(0), Acc(0), Acc(0), Acc(0)};
        // a folded 1-D signal that ends inside or before this tile (FusedArgs::lin_limit): zeros from its end on
        const int64_t lin0 = ((int64_t)ty * TY) * a.NX + (int64_t)tx * kFusedTX;
        if (LIN && a.lin_limit > 0 && lin0 + (int64_t)(TY - 1) * a.NX + kFusedTX > a.lin_limit) {       // (tile-uniform)
#pragma unroll
            for (int i = 0; i < TY / 4; i++) {
                const int64_t idx = lin0 + (int64_t)(rg + 4 * i) * a.NX + 4 * cc;
                // (the one chunk the end falls into is loaded sample by sample: nothing behind the end is read)
                tmp[i] = idx + 3 < a.lin_limit ? ld(4 * i) : idx < a.lin_limit ? ld_head(4 * i, (int)(a.lin_limit - idx)) : zero4;
            }
        } else if (odd_cols) {
#pragma unroll
            for (int i = 0; i < TY / 4; i++) tmp[i] = rg + 4 * i < rows_here ? ld_cols(4 * i) : zero4;
        } else if (rows_here == TY) {
#pragma unroll
            for (int i = 0; i < TY / 4; i++) tmp[i] = chunk_in ? ld(4 * i) : zero4;
        } else {
#pragma unroll
            for (int i = 0; i < TY / 4; i++)
                tmp[i] = (chunk_in && rg + 4 * i < rows_here) ? ld(4 * i) : zero4;
        }
        if constexpr (!PixelTraits<P>::is_integer) {
            if (a.pw_flags & 1) {
#pragma unroll
                for (int i = 0; i < TY / 4; i++) {
                    // samples beyond the image stay zero: they do not exist
                    const bool in = chunk_in && rg + 4 * i < rows_here;
                    const Acc s = in ? a.pre_s : Acc(0), b = in ? a.pre_b : Acc(0);
                    tmp[i].x = s * tmp[i].x + b; tmp[i].y = s * tmp[i].y + b;
                    tmp[i].z = s * tmp[i].z + b; tmp[i].w = s * tmp[i].w + b;
                    if (odd_cols) clear_dead_cols<A4, Acc>(tmp[i], cols_valid);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TY / 4; i++) tile4[(rg + 4 * i) * 64 + swz_chunk(cc)] = tmp[i];
    }

    // fused epilogue with an input operand: the thread's input column is taken out of LDS before the x phase
    // overwrites it and rides in registers to the final store (no second read of the image)
    Acc orig[EPI ? TY : 1];
    if constexpr (EPI) {
        __syncthreads();
        const int e = (swz_chunk(t >> 2) << 2) | (t & 3);
#pragma unroll
        for (int i = 0; i < TY; i++) orig[i] = tile[i * kFusedTX + e];
    }

    // ---- x phase: thread = (row slot, 16-sample segment); TY/16 rows per thread, interleaved ----
    if (a.nx > 0) {
        __syncthreads();
        Acc v[NR][kFusedSeg];
#pragma unroll
        for (int n = 0; n < NR; n++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                A4 q = tile4[(slot + 16 * n) * 64 + 4 * l + (j ^ sw)];
                v[n][4 * j + 0] = q.x; v[n][4 * j + 1] = q.y; v[n][4 * j + 2] = q.z; v[n][4 * j + 3] = q.w;
            }
        }
        if constexpr (XFIX) {
            // the x scans follow the y pattern (YPAT): directions and carries picked at compile time -- no register copies
            // at the merge of two scan directions, no select chain over the carries
            static_assert(YPAT >= 1, "fixed x pattern only with a fixed y pattern");
            constexpr int XP = YPAT > 2 ? YPAT - 2 : YPAT;
            {
                const bool first_lane = l == 0;
                scan_rows16<Acc, true, K, NR>(v, a.xs[0], first_lane, a.clamped && tx == 0 && first_lane, CX[0]);
            }
            if constexpr (XP == 2) {
                const bool first_lane = l == last_lane;
                scan_rows16<Acc, false, K, NR>(v, a.xs[1], first_lane, a.clamped && tx == a.MX - 1 && first_lane, CX[1], l > last_lane,
                                               EDGE ? entry_valid : kFusedSeg);
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < a.nx; s++) {
                const FusedScan<Acc> &sc = a.xs[s];
                const bool causal = sc.causal != 0;
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == last_lane);
                const bool clamp_first = a.clamped && tile_first && first_lane;
                Acc cx[NR][K];     // CX[s] with a run-time s: a select chain, not an indexed (scratch) array
#pragma unroll
                for (int n = 0; n < NR; n++)
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        cx[n][j] = CX[0][n][j];
#pragma unroll
                        for (int q = 1; q < kFusedMaxScans; q++) cx[n][j] = (s == q) ? CX[q][n][j] : cx[n][j];
                    }
                bool cf = clamp_first;
                if constexpr (MOD) {                               // zero-border form behind a border modification
                    if (a.clamped && tile_first) {
                        if (causal) border_mod_rows16<Acc, true, NR>(v, sc, clamp_first);
                        else        border_mod_rows16<Acc, false, NR>(v, sc, clamp_first);
                    }
                    cf = false;
                }
                if (causal) scan_rows16<Acc, true, K, NR>(v, sc, first_lane, cf, cx);
                else        scan_rows16<Acc, false, K, NR>(v, sc, first_lane, cf, cx, l > last_lane, EDGE ? entry_valid : kFusedSeg);
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
    }
    __syncthreads();

    // ---- y phase: thread = column ----
    {
        const int e = (swz_chunk(t >> 2) << 2) | (t & 3);
        Acc col[TY];
#pragma unroll
        for (int i = 0; i < TY; i++) col[i] = tile[i * kFusedTX + e];
        if constexpr (YPAT >= 1) {
            // YPAT 3 / 4: pattern 1 / 2 without an epilogue -- the rows are stored from inside the last scan, each as soon
            // as it is final, instead of TY stores in one burst behind the recurrence
            static_assert(!EDGE || YPAT <= 2, "rows leave from inside the last scan on whole tiles only");
            constexpr int PAT = YPAT > 2 ? YPAT - 2 : YPAT;
            constexpr bool EARLY = YPAT > 2;
            char *dpb_early = reinterpret_cast<char *>(dst + tile_off);
            // (with the pointwise epilogue applied on the way out: EPI has this thread's input column in registers)
            auto row_out = [&](int m, Acc v) __attribute__((always_inline)) {
                if constexpr (!PixelTraits<P>::is_integer) {
                    if constexpr (EPI) v = a.post_f * v + (a.post_i * orig[m] + a.post_b);
                    else if (a.pw_flags & 2) v = a.post_f * v + a.post_b;
                }
                __builtin_nontemporal_store(PixelTraits<P>::store(v),
                                            reinterpret_cast<P *>(dpb_early + ((uint32_t)t * (uint32_t)sizeof(P) + (uint32_t)m * a.row_bytes)));
            };
            {
                const bool clamp_first = a.clamped && ty == 0 && a.y_first_border;
                if constexpr (EARLY && PAT == 1) { scan_col<Acc, true, K, TY>(col, a.ys[0], clamp_first, CY[0], row_out); return; }
                else scan_col<Acc, true, K, TY>(col, a.ys[0], clamp_first, CY[0]);
            }
            if constexpr (PAT == 2) {
                const bool clamp_first = a.clamped && ty == a.MY - 1 && a.y_last_border;
                if constexpr (EARLY) { scan_col<Acc, false, K, TY>(col, a.ys[1], clamp_first, CY[1], row_out); return; }
                else if (EDGE && rows_here != TY) scan_col_partial_up<Acc, K, TY>(col, a.ys[1], clamp_first, rows_here);     // partial last tile row
                else scan_col<Acc, false, K, TY>(col, a.ys[1], clamp_first, CY[1]);
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < a.ny; j++) {
                const FusedScanY<Acc> &sc = a.ys[j];
                const bool causal = sc.causal != 0;
                const bool border = causal ? (ty == 0 && a.y_first_border) : (ty == a.MY - 1 && a.y_last_border);
                const bool clamp_first = a.clamped && border;
                Acc c[K];
#pragma unroll
                for (int r = 0; r < K; r++) {
                    c[r] = CY[0][r];
#pragma unroll
                    for (int q = 1; q < kFusedMaxScans; q++) c[r] = (j == q) ? CY[q][r] : c[r];
                }
                bool cf = clamp_first;
                if constexpr (MOD) {
                    if (clamp_first) {
                        if (causal) border_mod_col<Acc, true, TY>(col, sc);
                        else        border_mod_col<Acc, false, TY>(col, sc);
                    }
                    cf = false;
                }
                if (causal) scan_col<Acc, true, K, TY>(col, sc, cf, c);
                else if (rows_here == TY) scan_col<Acc, false, K, TY>(col, sc, cf, c);
                else scan_col_partial_up<Acc, K, TY>(col, sc, cf, rows_here);     // partial last tile row
            }
        }
        if constexpr (!PixelTraits<P>::is_integer) {
            // fused epilogue (compute_at of a pointwise consumer, lib/recfilter.cpp:473-573); x' was applied at the load
            if constexpr (EPI) {
#pragma unroll
                for (int i = 0; i < TY; i++) col[i] = a.post_f * col[i] + (a.post_i * orig[i] + a.post_b);
            } else if (a.pw_flags & 2) {
                if (a.post_i != Acc(0)) {
                    // order 3 has no registers to spare at two workgroups per CU: its input column comes back through
                    // L2 / the Infinity Cache instead, 64 coalesced dword loads per thread
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
            P *dp = dst + tile_off;
            // byte offsets kept in 32 bits (a tile spans TY rows: far below 4 GiB): scalar base + one 32-bit add per
            // row instead of a 64-bit address computation per store; the output is written once and not read again
            // by this filter: non-temporal stores
            char *dpb = reinterpret_cast<char *>(dp);
            const uint32_t row_bytes = a.row_bytes;
            // rows of this column that exist: all of the tile's, or (folded 1-D signal) those before the signal's end
            int my_rows = rows_here;
            if (LIN && a.lin_limit > 0) {
                const int64_t first = ((int64_t)ty * TY) * a.NX + (int64_t)tx * kFusedTX + t;
                const int64_t left = a.lin_limit > first ? (a.lin_limit - first + a.NX - 1) / a.NX : 0;
                my_rows = left < (int64_t)rows_here ? (int)left : rows_here;
            }
            if (t < last_cols) {
                if (my_rows == TY) {
#pragma unroll
                    for (int i = 0; i < TY; i++)
                        __builtin_nontemporal_store(PixelTraits<P>::store(col[i]), reinterpret_cast<P *>(dpb + ((uint32_t)t * (uint32_t)sizeof(P) + (uint32_t)i * row_bytes)));
                } else {
#pragma unroll
                    for (int i = 0; i < TY; i++)
                        if (i < my_rows)
                            __builtin_nontemporal_store(PixelTraits<P>::store(col[i]), reinterpret_cast<P *>(dpb + ((uint32_t)t * (uint32_t)sizeof(P) + (uint32_t)i * row_bytes)));
                }
            }
        }
    }
}

template <typename P, int K, int TY, bool EPI, bool EDGE, typename PI, int YPAT = 0, bool XFIX = false, bool LIN = false, bool MOD = false>
int launch_fused_pass2_impl(const PI *src, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    using Acc = typename PixelTraits<P>::Acc;
    const size_t lds = (size_t)TY * kFusedTX * sizeof(Acc);
    // the 64 KiB of dynamic LDS has to be opted into once per device
    // (concurrent plan creators/executors on different host threads may race to set it: the flag is atomic and the
    // attribute call itself is idempotent)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    std::atomic<bool> &done = attr_set[dev & 63];
    if (!done.load(std::memory_order_acquire)) {
        RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_pass2_kernel<P, K, TY, EPI, EDGE, PI, YPAT, XFIX, LIN, MOD>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done.store(true, std::memory_order_release);
    }
    dim3 grid((unsigned)(a.gx > 0 ? a.gx : a.MX), (unsigned)(a.gy > 0 ? a.gy : a.MY), (unsigned)a.NZ);
    hipLaunchKernelGGL((fused_pass2_kernel<P, K, TY, EPI, EDGE, PI, YPAT, XFIX, LIN, MOD>), grid, dim3(kFusedThreads), lds, stream, src, dst, a);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

}  // namespace

template <typename P, typename PI>
static int launch_fused_pass2_typed(int K, int TY, const PI *src, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                                    hipStream_t stream) {
    // the epilogue variant that keeps the input column in registers exists for float pixels only
    bool epi = false;
    const bool edge = a.last_cols != kFusedTX || a.last_rows != TY;
    // (three launches like the 128-row final pass's -- whole tiles lean, edge strips on the EDGE variant -- gain nothing here:
    // the 64-row EDGE variant costs 20-35 % per tile, what the two extra strips cost: 8192 x 8188, 0.134 / 0.135 ms)
    if constexpr (!PixelTraits<P>::is_integer) epi = (a.pw_flags & 2) && a.post_i != typename PixelTraits<P>::Acc(0) && K <= 2;
    if (a.lin_limit > 0 && (a.ny != 0 || a.NZ != 1 || a.plane_batch || epi || edge)) {
        set_error("fused pass 2: a signal that ends inside the image needs a 1-D plan of whole tiles without an input-operand epilogue");
        return RF_ERR_INVALID_ARG;
    }
    // (a plan in mod form -- FusedArgs::mod_form -- takes the general-pattern code, which applies its border modifications)
    const int ypat = a.mod_form ? 0 : (a.ny == 1 && a.ys[0].causal != 0) ? 1 : (a.ny == 2 && a.ys[0].causal != 0 && a.ys[1].causal == 0) ? 2 : 0;
    const int xpat = a.mod_form ? 0 : (a.nx == 1 && a.xs[0].causal != 0) ? 1 : (a.nx == 2 && a.xs[0].causal != 0 && a.xs[1].causal == 0) ? 2 : 0;
    bool early = true;        // rows can leave from inside the last scan (an affine epilogue is applied on the way out; one with
                              // an input operand only where the column is in registers, i.e. the EPI variants)
    if constexpr (!PixelTraits<P>::is_integer) early = (a.pw_flags & 2) == 0 || a.post_i == typename PixelTraits<P>::Acc(0);
#define RF_CASE(KK, TT)                                                                                         \
    if (K == KK && TY == TT) {                                                                                  \
        if constexpr (std::is_same<P, float>::value && std::is_same<P, PI>::value) {                            \
            if (a.mod_form) {                                                                                   \
                if (epi) { set_error("fused pass 2: clamped sections cannot take an epilogue with an input operand"); return RF_ERR_UNSUPPORTED; } \
                return edge ? launch_fused_pass2_impl<P, KK, TT, false, true, PI, 0, false, false, true>(src, dst, a, stream)   \
                            : launch_fused_pass2_impl<P, KK, TT, false, false, PI, 0, false, false, true>(src, dst, a, stream); \
            }                                                                                                   \
        }                                                                                                       \
        if constexpr (std::is_same<P, PI>::value) {                                                             \
            if (a.lin_limit > 0) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 0, false, true>(src, dst, a, stream); \
        }                                                                                                       \
        if constexpr (!PixelTraits<P>::is_integer) {                                                            \
            if (epi && edge) return launch_fused_pass2_impl<P, KK, TT, true, true, PI>(src, dst, a, stream);    \
            if constexpr (TT == 64 && std::is_same<P, PI>::value) {                                             \
                if (epi && ypat == 1) return launch_fused_pass2_impl<P, KK, TT, true, false, PI, 3>(src, dst, a, stream); \
                if (epi && ypat == 2) return launch_fused_pass2_impl<P, KK, TT, true, false, PI, 4>(src, dst, a, stream); \
            }                                                                                                   \
            if (epi) return launch_fused_pass2_impl<P, KK, TT, true, false, PI>(src, dst, a, stream);           \
        }                                                                                                       \
        if constexpr (std::is_same<P, float>::value && std::is_same<P, PI>::value) {   /* partial tiles, the usual y scans */ \
            if (edge && ypat == 1) return launch_fused_pass2_impl<P, KK, TT, false, true, PI, 1>(src, dst, a, stream); \
            if (edge && ypat == 2) return launch_fused_pass2_impl<P, KK, TT, false, true, PI, 2>(src, dst, a, stream); \
        }                                                                                                       \
        if (edge) return launch_fused_pass2_impl<P, KK, TT, false, true, PI>(src, dst, a, stream);              \
        if constexpr ((TT == 64 || std::is_same<P, float>::value) && std::is_same<P, PI>::value) {   /* the usual y scans, directions fixed at compile time */ \
            if (ypat == 1 && early && xpat == 1) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 3, true>(src, dst, a, stream); \
            if (ypat == 2 && early && xpat == 2) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 4, true>(src, dst, a, stream); \
            if (ypat == 1 && early) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 3>(src, dst, a, stream); \
            if (ypat == 2 && early) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 4>(src, dst, a, stream); \
            if (ypat == 1) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 1>(src, dst, a, stream); \
            if (ypat == 2) return launch_fused_pass2_impl<P, KK, TT, false, false, PI, 2>(src, dst, a, stream); \
        }                                                                                                       \
        return launch_fused_pass2_impl<P, KK, TT, false, false, PI>(src, dst, a, stream);                       \
    }
    RF_CASE(1, 64) RF_CASE(2, 64) RF_CASE(3, 64)
    RF_CASE(1, 32) RF_CASE(2, 32) RF_CASE(3, 32)
#undef RF_CASE
    set_error("fused path: unsupported order %d / tile height %d", K, TY);
    return RF_ERR_UNSUPPORTED;
}

template <typename P>
int launch_fused_pass2(int K, int TY, const void *src, bool src_u8, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                       hipStream_t stream) {
    if (a.MX <= 0 || a.MY <= 0 || a.NZ <= 0) return RF_OK;
    if (a.NZ > 65535 || a.MY > 65535) { set_error("fused path: grid too large"); return RF_ERR_UNSUPPORTED; }
    if constexpr (std::is_same<P, float>::value) {
        if (src_u8) return launch_fused_pass2_typed<P, uint8_t>(K, TY, (const uint8_t *)src, dst, a, stream);
    }
    return launch_fused_pass2_typed<P, P>(K, TY, (const P *)src, dst, a, stream);
}

template int launch_fused_pass2<float>(int, int, const void *, bool, float *, const FusedArgs<float> &, hipStream_t);
template int launch_fused_pass2<int32_t>(int, int, const void *, bool, int32_t *, const FusedArgs<uint32_t> &, hipStream_t);
template int launch_fused_pass2<int16_t>(int, int, const void *, bool, int16_t *, const FusedArgs<uint32_t> &, hipStream_t);
template int launch_fused_pass2<double>(int, int, const void *, bool, double *, const FusedArgs<double> &, hipStream_t);

}  // namespace rf
