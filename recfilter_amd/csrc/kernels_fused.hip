// kernels_fused.hip -- the bandwidth-tuned path: LDS-staged fused x/y tiles for gfx950.
//
// One workgroup owns a 256 x TY tile of one image plane, stages it in LDS once and runs EVERY
// x scan and EVERY y scan of the filter on it before anything goes back to HBM -- the
// "overlapped" evaluation of lib/split.cpp (create_intra_tile_term :503-665 for pass 1,
// add_residuals_to_final_result :1647-1780 for pass 2) with MI355X-shaped tiles instead of
// the reference's 32 x 32 CUDA tiles:
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
//   pass 1   writes only the k-sample tails of every scan (lib/split.cpp:256-499)
//   pass 2   injects the completed carries (lib/split.cpp:1008-1130) and stores the tile,
//            256 B per wave instruction
//
// Between the passes: the x carry recurrence (generic_carry_scan_kernel over the x tails), the
// cross-dimension residual of lib/split.cpp:1215-1633 split into tau_kernel (run the tile-local
// y scans on the completed x-carry strips, keep their tails) and carry_block_kernel (kernels_carry.hip:
// y carry recurrence with the residual sum_o G[x][o] * tau[o] folded into the tail it starts from).
#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

// ---- the fused pass kernel: one workgroup per tile ------------------------------------------------
// FINAL = false: intra-tile scans + tail extraction (the scan-everything pass 1 of the first round, kept as
//                an option; the default pass 1 is the contraction of kernels_tails.hip)
// FINAL = true : final correction pass.  The completed carries the tile needs (first lane of each row for
//                the x scans, every column for the y scans) are requested BEFORE the pixels, so their
//                latency hides behind the 64 KiB pixel load instead of stalling every scan.
template <typename P, int K, int TY, bool FINAL>
__global__ void __launch_bounds__(kFusedThreads)
fused_pass_kernel(const P *__restrict__ src, P *__restrict__ dst, FusedArgs<typename PixelTraits<P>::Acc> a) {
    using Acc = typename PixelTraits<P>::Acc;
    using A4 = typename Vec4<Acc>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Acc *tile = reinterpret_cast<Acc *>(lds_raw);
    A4 *tile4 = reinterpret_cast<A4 *>(lds_raw);
    constexpr int NR = TY / 16;

    const int t = threadIdx.x;
    const int tx = blockIdx.x, ty = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int64_t tile_off = z * a.NX * a.NY + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX;
    const int l = t & 15, slot = t >> 4, sw = (l >> 2) & 3;    // x phase: segment lane, row slot
    const int64_t Lx = a.NY * a.NZ, Ly = a.NX * a.NZ;
    const int64_t line0 = (int64_t)ty * TY + slot + a.NY * z;          // x phase: row n -> line0 + 16 n
    const int64_t line = (int64_t)tx * kFusedTX + t + a.NX * z;        // y phase: this thread's column

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
    if (FINAL) {
#pragma unroll
        for (int s = 0; s < kFusedMaxScans; s++) {
            if (s < a.nx) {
                const bool causal = a.xs[s].causal != 0;
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == 15);
                if (first_lane && !tile_first) {
                    const int tp = causal ? tx - 1 : tx + 1;
#pragma unroll
                    for (int n = 0; n < NR; n++)
#pragma unroll
                        for (int j = 0; j < K; j++)
                            CX[s][n][j] = a.xt[(((int64_t)s * a.MX + tp) * K + j) * Lx + line0 + 16 * n];
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
                    for (int r = 0; r < K; r++) CY[j][r] = a.yt[(((int64_t)j * a.MY + tp) * K + r) * Ly + line];
                }
            }
        }
    }

    // ---- load: wave w streams rows w, w+4, ...; one 1 KiB row per instruction ----
    {
        const int cc = t & 63, rg = t >> 6;
        const A4 *sp = reinterpret_cast<const A4 *>(src + tile_off);
        const uint32_t rs4 = (uint32_t)(a.NX / 4);
        const uint32_t off0 = (uint32_t)rg * rs4 + (uint32_t)cc;
        A4 tmp[TY / 4];
#pragma unroll
        for (int i = 0; i < TY / 4; i++) tmp[i] = sp[off0 + (uint32_t)(4 * i) * rs4];
#pragma unroll
        for (int i = 0; i < TY / 4; i++) tile4[(rg + 4 * i) * 64 + swz_chunk(cc)] = tmp[i];
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
#pragma unroll 1
        for (int s = 0; s < a.nx; s++) {
            const FusedScan<Acc> &sc = a.xs[s];
            const bool causal = sc.causal != 0;
            const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
            const bool first_lane = causal ? (l == 0) : (l == 15);
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
            if (causal) scan_rows16<Acc, true, K, NR>(v, sc, first_lane, clamp_first, cx);
            else        scan_rows16<Acc, false, K, NR>(v, sc, first_lane, clamp_first, cx);
            if (!FINAL) {
                // tail r = sample at direction position 255-r: the last lane's last K samples
                const bool last_lane = causal ? (l == 15) : (l == 0);
                if (last_lane) {
#pragma unroll
                    for (int n = 0; n < NR; n++)
#pragma unroll
                        for (int r = 0; r < K; r++)
                            a.xt[(((int64_t)s * a.MX + tx) * K + r) * Lx + line0 + 16 * n] =
                                causal ? v[n][kFusedSeg - 1 - r] : v[n][r];
                }
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
    if (!FINAL && a.ny == 0) return;   // pass 1 of an x-only filter has nothing left to extract
    __syncthreads();

    // ---- y phase: thread = column ----
    {
        const int e = (swz_chunk(t >> 2) << 2) | (t & 3);
        Acc col[TY];
#pragma unroll
        for (int i = 0; i < TY; i++) col[i] = tile[i * kFusedTX + e];
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
            if (causal) scan_col<Acc, true, K, TY>(col, sc, clamp_first, c);
            else        scan_col<Acc, false, K, TY>(col, sc, clamp_first, c);
            if (!FINAL) {
#pragma unroll
                for (int r = 0; r < K; r++)
                    a.yt[(((int64_t)j * a.MY + ty) * K + r) * Ly + line] = causal ? col[TY - 1 - r] : col[r];
            }
        }
        if (FINAL) {
            P *dp = dst + tile_off;
            const uint32_t nxu = (uint32_t)a.NX;
#pragma unroll
            for (int i = 0; i < TY; i++) dp[(uint32_t)t + (uint32_t)i * nxu] = PixelTraits<P>::store(col[i]);
        }
    }
}

// ---- the fused pass kernel, persistent and software pipelined ---------------------------------
// One workgroup loops over tiles id = blockIdx.x, +gridDim.x, ... (tx fastest).  LDS allows only
// two workgroups per CU (64 KiB each), so HBM latency is covered by the schedule, not by
// occupancy.  Per tile, in program order (vector-memory operations retire in issue order):
//     commit   pixels prefetched during the previous tile -> LDS              (waits: pixels)
//     x phase  every x scan on 16-sample segments                             (waits: x carries)
//     prefetch pixels of the next tile -> registers, in flight during the y phase
//     y phase  every y scan on register columns
//     carries  of the NEXT tile are requested here, i.e. BEFORE this tile's stores, so the wait
//              on them at the next x phase never has to drain the 64 stores behind them
//     stores   (pass 2) / tails (pass 1), fire and forget
template <typename P, int K, int TY, bool FINAL>
__global__ void __launch_bounds__(kFusedThreads, 2)
fused_pass_persistent_kernel(const P *__restrict__ src, P *__restrict__ dst,
                             FusedArgs<typename PixelTraits<P>::Acc> a, int64_t n_tiles) {
    using Acc = typename PixelTraits<P>::Acc;
    using A4 = typename Vec4<Acc>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Acc *tile = reinterpret_cast<Acc *>(lds_raw);
    A4 *tile4 = reinterpret_cast<A4 *>(lds_raw);
    constexpr int NL = TY / 4, NR = TY / 16;

    const int t = threadIdx.x;
    const int cc = t & 63, rg = t >> 6;                        // load phase: 16-byte chunk, row group
    const int l = t & 15, slot = t >> 4, sw = (l >> 2) & 3;    // x phase: segment lane, row slot
    const int e = (swz_chunk(t >> 2) << 2) | (t & 3);          // y phase: swizzled column offset
    const uint32_t rs4 = (uint32_t)(a.NX / 4);                 // row stride in 16-byte chunks
    const uint32_t ld_off0 = (uint32_t)rg * rs4 + (uint32_t)cc;
    const int64_t Lx = a.NY * a.NZ, Ly = a.NX * a.NZ;
    const int tiles_per_plane = a.MX * a.MY;                   // 32-bit tile arithmetic (n_tiles < 2^31)
    const int stride = (int)gridDim.x;

    A4 pre[NL];                            // pixels of the tile about to be committed
    Acc CX[kFusedMaxScans][NR][K];         // its x carries (first lane of each row only)
    Acc CY[kFusedMaxScans][K];             // its y carries (this thread's column)

    auto fetch_pixels = [&](int tid, uint32_t off) {
        const int z = tid / tiles_per_plane, rem = tid % tiles_per_plane;
        const int ty = rem / a.MX, tx = rem % a.MX;
        const A4 *sp = reinterpret_cast<const A4 *>(src + (int64_t)z * a.NX * a.NY + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX);
#pragma unroll
        for (int i = 0; i < NL; i++) pre[i] = sp[off + (uint32_t)(4 * i) * rs4];
    };
    auto fetch_carries = [&](int tid) {
        const int z = tid / tiles_per_plane, rem = tid % tiles_per_plane;
        const int ty = rem / a.MX, tx = rem % a.MX;
        const int64_t line0 = (int64_t)ty * TY + slot + a.NY * z;
        const int64_t line = (int64_t)tx * kFusedTX + t + a.NX * z;
#pragma unroll
        for (int s = 0; s < kFusedMaxScans; s++) {
#pragma unroll
            for (int n = 0; n < NR; n++)
#pragma unroll
                for (int j = 0; j < K; j++) CX[s][n][j] = Acc(0);
            if (s < a.nx) {
                const bool causal = a.xs[s].causal != 0;
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == 15);
                if (first_lane && !tile_first) {
                    const int tp = causal ? tx - 1 : tx + 1;
#pragma unroll
                    for (int n = 0; n < NR; n++)
#pragma unroll
                        for (int j = 0; j < K; j++)
                            CX[s][n][j] = a.xt[(((int64_t)s * a.MX + tp) * K + j) * Lx + line0 + 16 * n];
                }
            }
        }
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
                    for (int r = 0; r < K; r++) CY[j][r] = a.yt[(((int64_t)j * a.MY + tp) * K + r) * Ly + line];
                }
            }
        }
    };

#pragma unroll
    for (int s = 0; s < kFusedMaxScans; s++) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            CY[s][j] = Acc(0);
#pragma unroll
            for (int n = 0; n < NR; n++) CX[s][n][j] = Acc(0);       // pass 1: every tile scans with zero carries
        }
    }
    const int n_tiles_i = (int)n_tiles;
    int id = (int)blockIdx.x;
    if (id < n_tiles_i) {
        fetch_pixels(id, ld_off0);
        if (FINAL) fetch_carries(id);
    }
    for (; id < n_tiles_i; id += stride) {
        // 32-bit per-thread offsets from a wave-uniform tile base (SGPR base + VGPR offset addressing).
        // The empty asm makes them opaque per iteration: otherwise the compiler hoists all 16 load and
        // 64 store row offsets out of the tile loop and pays ~80 VGPRs for it.
        uint32_t ld_off = ld_off0, st_off = (uint32_t)t;
        asm volatile("" : "+v"(ld_off), "+v"(st_off));
        const int z = id / tiles_per_plane, rem = id % tiles_per_plane;
        const int ty = rem / a.MX, tx = rem % a.MX;
        const int64_t tile_off = (int64_t)z * a.NX * a.NY + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX;
        const int64_t line0 = (int64_t)ty * TY + slot + a.NY * z;          // x phase: row n -> line0 + 16 n
        const int64_t line = (int64_t)tx * kFusedTX + t + a.NX * z;        // y phase: this thread's column
        const int nid = id + stride;

        // ---- commit the prefetched pixels to LDS ----
#pragma unroll
        for (int i = 0; i < NL; i++) tile4[(rg + 4 * i) * 64 + swz_chunk(cc)] = pre[i];
        __syncthreads();

        // ---- x phase ----
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
#pragma unroll 1
            for (int s = 0; s < a.nx; s++) {
                const FusedScan<Acc> &sc = a.xs[s];
                const bool causal = sc.causal != 0;
                const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
                const bool first_lane = causal ? (l == 0) : (l == 15);
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
                if (causal) scan_rows16<Acc, true, K, NR>(v, sc, first_lane, clamp_first, cx);
                else        scan_rows16<Acc, false, K, NR>(v, sc, first_lane, clamp_first, cx);
                if (!FINAL) {
                    const bool last_lane = causal ? (l == 15) : (l == 0);
                    if (last_lane) {
#pragma unroll
                        for (int n = 0; n < NR; n++)
#pragma unroll
                            for (int r = 0; r < K; r++)
                                a.xt[(((int64_t)s * a.MX + tx) * K + r) * Lx + line0 + 16 * n] =
                                    causal ? v[n][kFusedSeg - 1 - r] : v[n][r];
                    }
                }
            }
            if (FINAL || a.ny > 0) {
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
        }

        // ---- prefetch the next tile's pixels: 64 KiB in flight while this tile's y phase runs ----
        if (nid < n_tiles_i) fetch_pixels(nid, ld_off);

        // ---- y phase: thread = column, the column lives in registers ----
        if (FINAL || a.ny > 0) {
            Acc col[TY];
#pragma unroll
            for (int i = 0; i < TY; i++) col[i] = tile[i * kFusedTX + e];
            __syncthreads();      // LDS is free for the next tile's commit
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
                if (causal) scan_col<Acc, true, K, TY>(col, sc, clamp_first, c);
                else        scan_col<Acc, false, K, TY>(col, sc, clamp_first, c);
                if (!FINAL) {
#pragma unroll
                    for (int r = 0; r < K; r++)
                        a.yt[(((int64_t)j * a.MY + ty) * K + r) * Ly + line] = causal ? col[TY - 1 - r] : col[r];
                }
            }
            if (FINAL) {
                if (nid < n_tiles_i) fetch_carries(nid);     // before the stores, see the header comment
                P *dp = dst + tile_off;
                const uint32_t nxu = (uint32_t)a.NX;
#pragma unroll
                for (int i = 0; i < TY; i++) dp[st_off + (uint32_t)i * nxu] = PixelTraits<P>::store(col[i]);
            }
        } else {
            __syncthreads();      // x-only pass 1: nothing to read back, just release LDS
        }
    }
}

// ---- tau: tile-local y scans of the completed x-carry strips, tails kept ----------------------
// tau[((tile*ny + j)*K + r)*nx*K + q*K + o], tile = (z*MY + ty)*MX + tx
template <typename Acc, int K, int TY>
__global__ void __launch_bounds__(256)
tau_kernel(FusedArgs<Acc> a, Acc *__restrict__ tau) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_tiles = (int64_t)a.MX * a.MY * a.NZ;
    const int strips = a.nx * K;
    if (idx >= n_tiles * strips) return;
    // tx fastest so that neighbouring threads work on neighbouring tiles of one tile row
    const int tx = (int)(idx % a.MX);
    int64_t rest = idx / a.MX;
    const int so = (int)(rest % strips);
    rest /= strips;
    const int ty = (int)(rest % a.MY);
    const int64_t z = rest / a.MY;
    const int q = so / K, o = so % K;
    const int64_t tile = (z * a.MY + ty) * a.MX + tx;
    Acc *out = tau + tile * (int64_t)a.ny * K * a.nx * K + q * K + o;      // + (j*K + r) * nx*K
    const int ostride = a.nx * K;

    const bool qc = a.xs[q].causal != 0;
    const bool q_first = qc ? (tx == 0) : (tx == a.MX - 1);
    if (q_first) {   // no carry enters this tile for scan q
        for (int e = 0; e < a.ny * K; e++) out[e * ostride] = Acc(0);
        return;
    }
    const int tp = qc ? tx - 1 : tx + 1;
    const int64_t Lx = a.NY * a.NZ;
    const Acc *strip = a.xt + (((int64_t)q * a.MX + tp) * K + o) * Lx + (int64_t)ty * TY + a.NY * z;
    Acc col[TY];
#pragma unroll
    for (int i = 0; i < TY; i++) col[i] = strip[i];
    Acc zero[K];
#pragma unroll
    for (int r = 0; r < K; r++) zero[r] = Acc(0);
#pragma unroll 1
    for (int j = 0; j < a.ny; j++) {
        const FusedScanY<Acc> &sc = a.ys[j];
        const bool causal = sc.causal != 0;
        const bool border = causal ? (ty == 0 && a.y_first_border) : (ty == a.MY - 1 && a.y_last_border);
        const bool clamp_first = a.clamped && border;
        if (causal) scan_col<Acc, true, K, TY>(col, sc, clamp_first, zero);
        else        scan_col<Acc, false, K, TY>(col, sc, clamp_first, zero);
#pragma unroll
        for (int r = 0; r < K; r++) out[(j * K + r) * ostride] = causal ? col[TY - 1 - r] : col[r];
    }
}

template <typename P, int K, int TY, bool FINAL>
int launch_fused_pass_one(const P *src, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    using Acc = typename PixelTraits<P>::Acc;
    const size_t lds = (size_t)TY * kFusedTX * sizeof(Acc);
    static int persistent = -1, resident = 0;
    if (persistent < 0) {
        const char *env = getenv("RF_FUSED_PERSIST");
        // measured on MI355X (profiles/r1): one workgroup per tile beats the persistent loop (0.30/0.42 ms vs
        // 0.31/0.45 ms for pass 1/2 of cfg3), so the persistent variant is opt-in
        persistent = (env && atoi(env) != 0) ? 1 : 0;
        RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_pass_kernel<P, K, TY, FINAL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_pass_persistent_kernel<P, K, TY, FINAL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0, cus = 0, per_cu = 0;
        RF_HIP_CHECK(hipGetDevice(&dev));
        RF_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        RF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, reinterpret_cast<const void *>(&fused_pass_persistent_kernel<P, K, TY, FINAL>), kFusedThreads, lds));
        if (per_cu < 1) per_cu = 1;
        if (const char *w = getenv("RF_FUSED_WGS_PER_CU")) per_cu = atoi(w) > 0 ? atoi(w) : per_cu;
        resident = cus * per_cu;
    }
    const int64_t n_tiles = (int64_t)a.MX * a.MY * a.NZ;
    if (persistent) {
        const unsigned grid = (unsigned)(n_tiles < resident ? n_tiles : resident);
        hipLaunchKernelGGL((fused_pass_persistent_kernel<P, K, TY, FINAL>), dim3(grid), dim3(kFusedThreads), lds, stream,
                           src, dst, a, n_tiles);
    } else {
        dim3 grid((unsigned)a.MX, (unsigned)a.MY, (unsigned)a.NZ);
        hipLaunchKernelGGL((fused_pass_kernel<P, K, TY, FINAL>), grid, dim3(kFusedThreads), lds, stream, src, dst, a);
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

template <typename P, int K, int TY>
int launch_fused_pass_impl(bool final_pass, const P *src, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                           hipStream_t stream) {
    return final_pass ? launch_fused_pass_one<P, K, TY, true>(src, dst, a, stream)
                      : launch_fused_pass_one<P, K, TY, false>(src, dst, a, stream);
}

}  // namespace

template <typename P>
int launch_fused_pass(bool final_pass, int K, int TY, const P *src, P *dst,
                      const FusedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream) {
    if (a.MX <= 0 || a.MY <= 0 || a.NZ <= 0) return RF_OK;
    if (a.NZ > 65535 || a.MY > 65535) { set_error("fused path: grid too large"); return RF_ERR_UNSUPPORTED; }
#define RF_CASE(KK, TT) if (K == KK && TY == TT) return launch_fused_pass_impl<P, KK, TT>(final_pass, src, dst, a, stream);
    RF_CASE(1, 64) RF_CASE(2, 64) RF_CASE(3, 64)
    RF_CASE(1, 32) RF_CASE(2, 32) RF_CASE(3, 32)
#undef RF_CASE
    set_error("fused path: unsupported order %d / tile height %d", K, TY);
    return RF_ERR_UNSUPPORTED;
}

template <typename Acc>
int launch_tau(int K, int TY, const FusedArgs<Acc> &a, Acc *tau, hipStream_t stream) {
    const int64_t n = (int64_t)a.MX * a.MY * a.NZ * a.nx * K;
    if (n <= 0 || a.ny == 0) return RF_OK;
    const unsigned grid = (unsigned)((n + 255) / 256);
#define RF_CASE(KK, TT) if (K == KK && TY == TT) { hipLaunchKernelGGL((tau_kernel<Acc, KK, TT>), dim3(grid), dim3(256), 0, stream, a, tau); RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
    RF_CASE(1, 64) RF_CASE(2, 64) RF_CASE(3, 64)
    RF_CASE(1, 32) RF_CASE(2, 32) RF_CASE(3, 32)
#undef RF_CASE
    set_error("tau: unsupported order %d / tile height %d", K, TY);
    return RF_ERR_UNSUPPORTED;
}

template int launch_fused_pass<float>(bool, int, int, const float *, float *, const FusedArgs<float> &, hipStream_t);
template int launch_fused_pass<int32_t>(bool, int, int, const int32_t *, int32_t *, const FusedArgs<uint32_t> &, hipStream_t);
template int launch_tau<float>(int, int, const FusedArgs<float> &, float *, hipStream_t);
template int launch_tau<uint32_t>(int, int, const FusedArgs<uint32_t> &, uint32_t *, hipStream_t);

}  // namespace rf
