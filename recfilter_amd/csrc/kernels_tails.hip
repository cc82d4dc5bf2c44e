// kernels_tails.hip -- pass 1 of the fused path as a contraction: per-tile tail extraction without
// running the scans over the tile.
//
// What pass 1 has to produce (extract_tails_from_each_scan, lib/split.cpp:256-499) is, per tile, the
// k-sample tail of every scan with all carries entering the tile set to zero.  Those tails are LINEAR
// in the tile, so they can be read off with precomputed impulse responses instead of running every
// recurrence over every sample (what the reference's <F>_Intra stage and the first version of this
// pass did, ~25 VALU instructions per sample with long dependent chains):
//
//   x tails   xt[s][r](row)  = sum_x Hx[s][r][x] * tile[row][x]            Hx = E_s F_s ... F_0 (k x 256)
//   y tails   yt[j][r](col)  = F_x( sum_i Hy[j][r][i] * tile[i][.] )(col)  Hy = E_j Y_j ... Y_0 (k x TY)
//
// For the y tails the TY rows of the tile are first contracted with Hy into k "combined rows" per y
// scan (fused_tails_kernel, no recurrence at all), and only those few rows go through the tile-local
// x scans F_x (xscan_rows_kernel, 1/16 of the samples).  Hx/Hy come from the same scan_tile routine as
// every other table (border variants included), so the carry stage and pass 2 see exactly the tails
// they saw before.
//
// fused_tails_kernel stages HALF tiles (256 x 32, 32 KiB of LDS) so that four to five workgroups fit a
// CU; the second half's pixels are already in flight while the first half is contracted.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {


typedef float F2 __attribute__((ext_vector_type(2)));     // operands of the packed f32 instructions
typedef float F4 __attribute__((ext_vector_type(4)));     // accumulator of v_mfma_f32_4x4x1_16b_f32

// MODE: what the loads of a tile have to mask.  0: nothing but whole chunks beyond the image (every 2-D / 3-D image whose
// width is a multiple of 4); 1: the end of a folded 1-D signal (FusedArgs::lin_limit); 2: a width that is not a multiple
// of 4 (a row's last chunk is partial).  Variants of their own, so that the kernel every other image runs stays small: with
// the sample-by-sample loads of modes 1 and 2 in the common kernel it grew from 115 / 121 to 127 / 132 registers at order
// 2 / 3 -- above 128 a SIMD holds three waves instead of four -- and pass 1 of an order-3 filter on 3 x 16384^2 took 0.89
// instead of 0.69 ms (order 2, one plane: 0.237 against 0.205 ms; same box).
//
// YM (f32 pixels): the y part -- 2 * ny * K multiply-adds per sample on the vector ALU otherwise -- runs on the matrix cores,
// v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4 x 4 outer products per issue, exact f32 fma chains; lane map:
// profiles/r2/mfma_4x4x1_16b_lane_map.txt): block b = four adjacent columns, A[b][i] = this step's pixel of column 4b + i
// (the lane's own column: the operand is the register the VALU form multiplies), B[b][j] = Hy[tail 4g + j][row] (the step's
// 32-row slice of Hy staged in LDS next to the tile, 16 bytes per lane and four rows), accumulator D[b][i][j] = combined row
// 4g + j at column 4b + i, alive across the tile's steps: lane 4b + j ends up with four adjacent columns of ONE tail and
// stores them as 16 bytes.  At order 3 with two y scans the vector ALU loses 96 of its ~330 instructions per step and wave.
template <typename P, int K, int TY, typename PI, int MODE = 0, bool YM = false>
__global__ void __launch_bounds__(kFusedThreads)
fused_tails_kernel(const PI *__restrict__ src, FusedArgs<typename PixelTraits<P>::Acc> a,
                   const typename PixelTraits<P>::Acc *__restrict__ Hx,     // [vx][s][r][256]
                   const typename PixelTraits<P>::Acc *__restrict__ Hy) {   // [vy][j][r][TY]
    using Acc = typename PixelTraits<P>::Acc;
    using A4 = typename Vec4<Acc>::type;
    constexpr int kTailRows = sizeof(Acc) == 8 ? 16 : 32;          // rows staged per step: 32 KiB of LDS
    __shared__ __attribute__((aligned(16))) Acc tile[kTailRows * kFusedTX];
    // Hx of this tile's variant: nx * K rows of 256, dynamic so that a filter with two x scans of order 3 (6 KiB instead
    // of 12) still fits four workgroups per CU
    extern __shared__ __attribute__((aligned(16))) unsigned char hx_raw[];
    Acc *hx_lds = reinterpret_cast<Acc *>(hx_raw);
    A4 *tile4 = reinterpret_cast<A4 *>(tile);
    A4 *hx4 = reinterpret_cast<A4 *>(hx_lds);
    static_assert(!YM || (std::is_same<Acc, float>::value && kTailRows == 32), "the matrix-core y part is f32");
    constexpr int kHyPitch4 = 9;                  // YM: 32 rows + 4 floats of padding per tail, in 16-byte units (the four
                                                  // tails of a group are read side by side: no two on the same banks)
    constexpr int NGY = (kFusedMaxScans * K + 3) / 4;     // YM: groups of four y tails
    constexpr int NH = TY / kTailRows;            // steps per tile (2 for TY = 64)
    constexpr int NL = kTailRows / 4;             // float4 loads per thread per step
    constexpr int NR = kTailRows / 16;            // rows per thread in the x part

    const int t = threadIdx.x;
    const int tx = blockIdx.x, ty = blockIdx.y;
    const int64_t z = blockIdx.z;
    if (a.plane_batch) src = reinterpret_cast<const PI *>(a.in_planes[z]);     // batched Tuple planes: own buffers
    const int64_t tile_off = (a.plane_batch ? 0 : z * a.NX * a.NY) + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX;
    const int vx = (tx == 0 ? 1 : 0) | (tx == a.MX - 1 ? 2 : 0);
    const int vy = ((ty == 0 && a.y_first_border) ? 1 : 0) | ((ty == a.MY - 1 && a.y_last_border) ? 2 : 0);
    const int nxk = a.nx * K, nyk = a.ny * K;

    const int cc = t & 63, rg = t >> 6;                        // load: 16-byte chunk, row group
    const int l = t & 15, slot = t >> 4, sw = (l >> 2) & 3;    // x part: segment lane, row slot
    const int e = (swz_chunk(t >> 2) << 2) | (t & 3);          // y part: swizzled column offset
    // byte offsets inside the tile, kept in 32 bits: scalar base + 32-bit vector offset addressing
    const char *spb = reinterpret_cast<const char *>(src + tile_off);
    const uint32_t in_row_bytes = a.row_bytes / (uint32_t)sizeof(P) * (uint32_t)sizeof(PI);
    const uint32_t off0 = (uint32_t)rg * in_row_bytes + (uint32_t)cc * (uint32_t)(4 * sizeof(PI));
    auto ld = [&](int row) { return load_chunk<PI, Acc>(spb + (off0 + (uint32_t)row * in_row_bytes)); };
    auto ld_head = [&](int row, int n) { return load_chunk_head<PI, Acc>(spb + (off0 + (uint32_t)row * in_row_bytes), n); };
    const int64_t Lx = a.NYP * a.NZ;
    // the row's last tile may be partial: 16-byte chunks beyond the image are taken as zeros
    const bool chunk_in = (tx != a.MX - 1) || (4 * cc < a.last_cols);
    // ... and when the width is not a multiple of 4 the last of them is partial (tile-uniform flag; scan_device.h)
    const bool odd_cols = MODE == 2 && tx == a.MX - 1;
    const int cols_valid = a.last_cols - 4 * cc;
    auto ld_cols = [&](int row) { return load_chunk_cols<PI, Acc>(spb + (off0 + (uint32_t)row * in_row_bytes), cols_valid); };
    const A4 zero4 = A4{Acc(0), Acc(0), Acc(0), Acc(0)};

    // rows of this half that exist (the last tile row may be partial): the fast path when all of them do
    const int rows_here = (ty == a.MY - 1) ? a.last_rows : TY;
    // a folded 1-D signal that ends inside or before this tile (FusedArgs::lin_limit): samples from the end on are zeros
    const int64_t lin0 = ((int64_t)ty * TY) * a.NX + (int64_t)tx * kFusedTX;         // linear index of the tile's first sample
    const bool lin_cut = MODE == 1 && a.lin_limit > 0 && lin0 + (int64_t)(TY - 1) * a.NX + kFusedTX > a.lin_limit;      // (tile-uniform)
    // Whole tiles -- all but the last tile of a row, the last tile row and the tile a folded signal ends in -- run a body of
    // their own (WHOLE), instantiated apart from the one that masks: with the load variants joined in ONE body the compiler
    // merged their registers through copies behind the loads (`global_load v[22:25]; s_waitcnt vmcnt(0); v_mov v26, v25`), i.e.
    // every step WAITED for the pixels it had just requested for the next step -- the memory latency was exposed once per
    // step instead of hidden behind the step's arithmetic (ISA of round 3's kernel; pass 1 at order 3 paid its whole VALU
    // time on top of the loads for it).
    const bool whole_tile = !lin_cut && (tx != a.MX - 1 || a.last_cols == kFusedTX) && rows_here == TY && !odd_cols;
    auto body = [&](auto whole_tag) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
        Acc comb[YM ? 1 : kFusedMaxScans * K];
#pragma unroll
        for (int jr = 0; jr < (YM ? 1 : kFusedMaxScans * K); jr++) comb[jr] = Acc(0);
        // YM: the step's slice of Hy behind Hx in the dynamic LDS, [tail (padded to groups of four)][kHyPitch4 * 4]; threads
        // (tail, chunk of four rows) fetch it with the step's pixels
        A4 *hy4 = hx4 + (a.nx > 0 ? a.nx : 1) * K * (kFusedTX / 4);
        const int nyk4 = (nyk + 3) & ~3;
        const int hy_jr = t >> 3, hy_m = t & 7;
        A4 hy_pre = zero4;
        F4 yacc[YM ? NGY : 1];
#pragma unroll
        for (int g = 0; g < (YM ? NGY : 1); g++) yacc[g] = F4{0.0f, 0.0f, 0.0f, 0.0f};

        A4 pre[NL];
        auto load_half = [&](int half) {
            if constexpr (YM) {
                if (hy_jr < nyk) hy_pre = *reinterpret_cast<const A4 *>(Hy + (size_t)(vy * nyk + hy_jr) * TY + kTailRows * half + 4 * hy_m);
            }
            const int r0 = kTailRows * half + rg;
            if (!WHOLE && lin_cut) {
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    const int64_t idx = lin0 + (int64_t)(r0 + 4 * i) * a.NX + 4 * cc;
                    // (the one chunk the end falls into is loaded sample by sample: nothing behind the end is read)
                    pre[i] = idx + 3 < a.lin_limit ? ld(kTailRows * half + 4 * i)
                             : idx < a.lin_limit   ? ld_head(kTailRows * half + 4 * i, (int)(a.lin_limit - idx)) : zero4;
                }
                return;
            }
            // (whole tiles -- all but the last of a row and the last tile row -- load without a condition: a per-lane select
            // between a load and zeros costs a branch per load and a wait for the load before the zeros may be written)
            if constexpr (WHOLE) {
#pragma unroll
                for (int i = 0; i < NL; i++) pre[i] = ld(kTailRows * half + 4 * i);
            } else if (odd_cols) {
#pragma unroll
                for (int i = 0; i < NL; i++) pre[i] = r0 + 4 * i < rows_here ? ld_cols(kTailRows * half + 4 * i) : zero4;
            } else if (rows_here == TY) {
#pragma unroll
                for (int i = 0; i < NL; i++) pre[i] = chunk_in ? ld(kTailRows * half + 4 * i) : zero4;
            } else {
#pragma unroll
                for (int i = 0; i < NL; i++)
                    pre[i] = (chunk_in && r0 + 4 * i < rows_here) ? ld(kTailRows * half + 4 * i) : zero4;
            }
        };
        load_half(0);
        // impulse responses of the x tails for this tile's border variant -> LDS (read back per segment below);
        // 16-byte chunk c of a row is stored at chunk c ^ ((c >> 4) & 3), the same swizzle as the pixels
        if (nxk > 0) {
            const A4 *hsrc = reinterpret_cast<const A4 *>(Hx + (size_t)vx * nxk * kFusedTX);
            for (int c = t; c < nxk * 64; c += kFusedThreads) hx4[(c & ~63) | swz_chunk(c & 63)] = hsrc[c];
        }

#pragma unroll
        for (int half = 0; half < NH; half++) {
            if (half > 0) __syncthreads();                          // previous step's readers are done
            if constexpr (!PixelTraits<P>::is_integer) {
                if (a.pw_flags & 1) {                                   // fused prologue x' = pre_s * in + pre_b
#pragma unroll
                    for (int i = 0; i < NL; i++) {
                        // samples beyond the image stay zero: they do not exist
                        const bool in = WHOLE || (chunk_in && kTailRows * half + rg + 4 * i < rows_here);
                        const Acc s = in ? a.pre_s : Acc(0), b = in ? a.pre_b : Acc(0);
                        pre[i].x = s * pre[i].x + b; pre[i].y = s * pre[i].y + b;
                        pre[i].z = s * pre[i].z + b; pre[i].w = s * pre[i].w + b;
                        if (!WHOLE && odd_cols) clear_dead_cols<A4, Acc>(pre[i], cols_valid);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NL; i++) tile4[(rg + 4 * i) * 64 + swz_chunk(cc)] = pre[i];
            if constexpr (YM) {
                if (hy_jr < nyk4) hy4[hy_jr * kHyPitch4 + hy_m] = hy_pre;       // (tails beyond nyk: zeros)
            }
            __syncthreads();
            if (half + 1 < NH) load_half(half + 1);                 // next half in flight during this one's math

            // ---- x tails of this half's rows: dot products + reduction over the 16 lanes of a row ----
            if (nxk > 0) {
                Acc v[NR][kFusedSeg];
#pragma unroll
                for (int n = 0; n < NR; n++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        A4 q = tile4[(slot + 16 * n) * 64 + 4 * l + (j ^ sw)];
                        v[n][4 * j + 0] = q.x; v[n][4 * j + 1] = q.y; v[n][4 * j + 2] = q.z; v[n][4 * j + 3] = q.w;
                    }
                }
                const int64_t line0 = (int64_t)ty * TY + kTailRows * half + slot + a.NYP * z;
#pragma unroll 1
                for (int sr = 0; sr < nxk; sr++) {
                    Acc h[kFusedSeg];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        A4 q = hx4[sr * 64 + 4 * l + (j ^ sw)];
                        h[4 * j + 0] = q.x; h[4 * j + 1] = q.y; h[4 * j + 2] = q.z; h[4 * j + 3] = q.w;
                    }
                    Acc acc[NR];
                    if constexpr (std::is_same<Acc, float>::value) {
                        // two samples per instruction (v_pk_fma_f32): neighbours along the row sit in adjacent registers
                        // (two accumulators per row: consecutive packed FMAs are independent, a dependent pair costs a wait state)
                        F2 acc2[NR][2];
#pragma unroll
                        for (int n = 0; n < NR; n++) acc2[n][0] = acc2[n][1] = F2{0.0f, 0.0f};
#pragma unroll
                        for (int m = 0; m < kFusedSeg; m += 2) {
                            const F2 hh = F2{h[m], h[m + 1]};
#pragma unroll
                            for (int n = 0; n < NR; n++)
                                acc2[n][(m >> 1) & 1] = hh * F2{v[n][m], v[n][m + 1]} + acc2[n][(m >> 1) & 1];
                        }
#pragma unroll
                        for (int n = 0; n < NR; n++) {
                            const F2 s2 = acc2[n][0] + acc2[n][1];
                            acc[n] = s2.x + s2.y;
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < NR; n++) {
                            acc[n] = Acc(0);
#pragma unroll
                            for (int m = 0; m < kFusedSeg; m++) acc[n] = acc[n] + h[m] * v[n][m];
                        }
                    }
#pragma unroll
                    for (int n = 0; n < NR; n++) {          // sum over the row's 16 lanes; total lands in lane 15
                        acc[n] = acc[n] + row_shift<true, 8>(acc[n]);
                        acc[n] = acc[n] + row_shift<true, 4>(acc[n]);
                        acc[n] = acc[n] + row_shift<true, 2>(acc[n]);
                        acc[n] = acc[n] + row_shift<true, 1>(acc[n]);
                    }
                    if (l == 15) {
                        const int s = sr / K, r = sr % K;
#pragma unroll
                        for (int n = 0; n < NR; n++)
                            a.xt[(((int64_t)s * a.MX + tx) * K + r) * Lx + line0 + 16 * n] = acc[n];
                    }
                }
            }

            // ---- y: contract this half's rows with Hy (thread = column) ----
            if (nyk > 0) {
                Acc col[kTailRows];
#pragma unroll
                for (int i = 0; i < kTailRows; i++) col[i] = tile[i * kFusedTX + e];
                if constexpr (YM) {
#pragma unroll
                    for (int m = 0; m < kTailRows / 4; m++) {
#pragma unroll
                        for (int g = 0; g < NGY; g++) {
                            if (4 * g < nyk) {                                   // (uniform)
                                const A4 h = hy4[(4 * g + (t & 3)) * kHyPitch4 + m];
                                yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[4 * m + 0], h.x, yacc[g], 0, 0, 0);
                                yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[4 * m + 1], h.y, yacc[g], 0, 0, 0);
                                yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[4 * m + 2], h.z, yacc[g], 0, 0, 0);
                                yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[4 * m + 3], h.w, yacc[g], 0, 0, 0);
                            }
                        }
                    }
                } else if constexpr (std::is_same<Acc, float>::value) {
                    // two rows per instruction (v_pk_fma_f32), two tails at a time so that consecutive packed FMAs are
                    // independent (a dependent pair costs a wait state); an odd last tail is paired with itself
#pragma unroll
                    for (int g = 0; g < (kFusedMaxScans * K + 1) / 2; g++) {
                        if (2 * g < nyk) {
                            const int j0 = 2 * g, j1 = (2 * g + 1 < nyk) ? 2 * g + 1 : 2 * g;
                            const Acc *h0 = Hy + (size_t)(vy * nyk + j0) * TY + kTailRows * half;     // wave-uniform
                            const Acc *h1 = Hy + (size_t)(vy * nyk + j1) * TY + kTailRows * half;
                            F2 c0 = F2{0.0f, 0.0f}, c1 = F2{0.0f, 0.0f};
#pragma unroll
                            for (int i = 0; i < kTailRows; i += 2) {
                                const F2 cc = F2{col[i], col[i + 1]};
                                c0 = F2{h0[i], h0[i + 1]} * cc + c0;
                                c1 = F2{h1[i], h1[i + 1]} * cc + c1;
                            }
                            comb[2 * g] = comb[2 * g] + (c0.x + c0.y);
                            if (2 * g + 1 < kFusedMaxScans * K) comb[2 * g + 1] = comb[2 * g + 1] + (c1.x + c1.y);
                        }
                    }
                } else {
#pragma unroll
                    for (int jr = 0; jr < kFusedMaxScans * K; jr++) {
                        if (jr < nyk) {
                            const Acc *hy = Hy + (size_t)(vy * nyk + jr) * TY + kTailRows * half;     // wave-uniform
#pragma unroll
                            for (int i = 0; i < kTailRows; i++) comb[jr] = comb[jr] + hy[i] * col[i];
                        }
                    }
                }
            }
        }
        // combined rows -> yt; with x scans in the filter xscan_rows_kernel finishes them in place
        if constexpr (YM) {
            if (nyk > 0) {
                // lane 4b + j, register i: combined row 4g + j at the tile's column (t & ~3) + i (thread t reads column t: `e` is
                // where the swizzle put it)
                const int64_t line = (int64_t)tx * kFusedTX + (t & ~3) + a.NXP * z;
#pragma unroll
                for (int g = 0; g < NGY; g++) {
                    const int jr = 4 * g + (t & 3);
                    if (jr < nyk) *reinterpret_cast<F4 *>(a.yt + a.yt_index(jr / K, ty, jr % K, K, line)) = yacc[g];
                }
            }
        } else if (nyk > 0) {
            const int64_t line = (int64_t)tx * kFusedTX + t + a.NXP * z;
#pragma unroll
            for (int jr = 0; jr < kFusedMaxScans * K; jr++)
                if (jr < nyk) a.yt[a.yt_index(jr / K, ty, jr % K, K, line)] = comb[jr];
        }
    };
    if (whole_tile) body(std::true_type{});
    else body(std::false_type{});
}

// tile-local x scans (all of them, zero carries) of the combined rows, in place in yt, plus the
// cross-dimension residual.  A workgroup takes 16 row tiles (runs of 256 consecutive yt samples), moves
// them through LDS with fully coalesced 16-byte accesses and scans each in one 16-lane DPP row.
//
// Residual (lib/split.cpp:1215-1633): what the completed x carries entering a tile add to the row after
// all x scans, sum_{q,o} G_q[x][o] * tau[q][o], tau = the same tail functional applied to the carry strip.  It is folded in here because this kernel
// touches every tail sample anyway; the y carry scan then needs no knowledge of x.  The G table of the
// interior tile variant is staged in LDS next to the rows (its load overlaps the rows' load); the few
// border tiles read their variant from memory.
//
// XC (images of at most kXcMaxTiles tiles per row): the kernel also COMPLETES the x tails it reads, i.e. it is the carry
// scan along x (carry_block_kernel on xt) as well.  On such images that launch is a dozen microseconds of launch and
// latency around 1-2 MiB of tails; here every workgroup takes the x tails of its tile row (nx * MX * K strips of TY
// rows, 16 KiB) into LDS, runs the recurrence over each row's tiles (blocked over the four waves) -- the same sums as
// carry_block_kernel: own tail + W * (completed carries of the earlier scans entering the tile) + A * (completed tail of
// the previous tile) -- and the first workgroup of the tile row stores the completed tails for the final pass, into
// a second array (xt_done): the other workgroups of the tile row read the incomplete ones whenever they get to run.  The
// workgroups of a tile row repeat that recurrence (redundant, but it needs no hand-off between them).
constexpr int kXcMaxTiles = 32;   // XC = tiles per wave (four waves): 4 up to 16 tiles per row, 8 up to 32 (order 1)


template <typename Acc, int K, bool EDGE, bool TALL, int XC = 0, bool MOD = false>
__global__ void __launch_bounds__(256, XC ? (sizeof(Acc) == 8 ? 2 : 3) : (sizeof(Acc) == 8 ? 3 : 6))
xscan_rows_kernel(FusedArgs<Acc> a, int gj, int TY, const Acc *__restrict__ Hy, const Acc *__restrict__ G,
                  const Acc *__restrict__ Wx = nullptr, const Acc *__restrict__ Ax = nullptr, Acc *__restrict__ xt_done = nullptr) {
    using A4 = typename Vec4<Acc>::type;
    __shared__ __attribute__((aligned(16))) Acc rows[16 * kFusedTX];
    extern __shared__ __attribute__((aligned(16))) unsigned char g_raw[];                  // G[variant 0][q][o][x]: nx * K
    Acc *g_lds = reinterpret_cast<Acc *>(g_raw);                                           // rows of 256 (dynamic: more
                                                                                           // workgroups per CU with few scans)
    Acc *xc = g_lds + a.nx * K * kFusedTX;                                                 // XC: [q][tx][o][TY] behind it
    A4 *rows4 = reinterpret_cast<A4 *>(rows);
    const int t = threadIdx.x;
    const int cc = t & 63, rg = t >> 6;
    const int l = t & 15, row = t >> 4, sw = (l >> 2) & 3;
    const int nxk = a.nx * K;
    // The workgroup's 16 row tiles: gj combined rows (j, r) x 16/gj consecutive x tiles of one tile row ty -- the rows of
    // one x tile need the same carry strips (tau below), so they sit in the same wave and fetch them once.
    // block -> (ty, z, group of combined rows, group of x tiles), x groups fastest
    const int txp = 16 / gj;
    const int n_xg = (a.MX + txp - 1) / txp, n_jg = a.ny * K / gj;
    int b = blockIdx.x;
    const int xg = b % n_xg; b /= n_xg;
    const int jg = b % n_jg; b /= n_jg;
    const int64_t z = b % a.NZ;
    const int ty = (int)(b / a.NZ);
    auto tile_of = [&](int r, int &jr, int &tx_out) { jr = jg * gj + r % gj; tx_out = xg * txp + r / gj; };
    auto row_tile_index = [&](int jr, int tx_i) {      // index of a run of 256 y-tail samples (FusedArgs::yt_index / 256)
        return a.yt_index(jr / K, ty, jr % K, K, ((int64_t)z * a.MX + tx_i) * kFusedTX) >> 8;
    };
    A4 *yt4 = reinterpret_cast<A4 *>(a.yt);
    A4 tmp[4];
    if (a.yt_parts <= 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int jr_i, tx_i;
            tile_of(rg + 4 * i, jr_i, tx_i);
            tmp[i] = (tx_i < a.MX) ? yt4[row_tile_index(jr_i, tx_i) * 64 + cc] : A4{Acc(0), Acc(0), Acc(0), Acc(0)};
        }
    } else {
        // the combined rows come in parts (FusedArgs::yt_parts): add them up
        const A4 *p4 = reinterpret_cast<const A4 *>(a.ytp);
        const int64_t stride4 = a.yt_part_stride >> 2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int jr_i, tx_i;
            tile_of(rg + 4 * i, jr_i, tx_i);
            A4 v = A4{Acc(0), Acc(0), Acc(0), Acc(0)};
            if (tx_i < a.MX) {
                const int64_t e = row_tile_index(jr_i, tx_i) * 64 + cc;
                v = p4[e];
#pragma unroll 1
                for (int p = 1; p < a.yt_parts; p++) v = v + p4[p * stride4 + e];
            }
            tmp[i] = v;
        }
    }
    if constexpr (XC > 0) {
        const int64_t Lx = a.NYP * a.NZ;
        const int64_t ybase = (int64_t)ty * TY + a.NYP * z;
        const int MX = a.MX;
        const int n_strips = a.nx * MX * K, cps = TY / 4;                 // 16-byte chunks per strip
        A4 *xc4 = reinterpret_cast<A4 *>(xc);
        for (int e = t; e < n_strips * cps; e += 256)
            xc4[e] = *reinterpret_cast<const A4 *>(a.xt + (int64_t)(e / cps) * Lx + ybase + 4 * (e % cps));
        __syncthreads();
        {
            // Blocked like carry_block_kernel: wave c owns the tiles [c*C, (c+1)*C) of every row in scan order (C = ceil(MX/4)
            // <= XC), runs the recurrence inside its chunk from a zero state, the chunks' exit states are combined through LDS
            // with A^C, and the entering state is propagated through the chunk.  No branch inside the loops over a chunk's
            // tiles (a uniform branch per tile would serialise the LDS and table latencies): tiles beyond MX are computed
            // on clamped indices and never stored.
            const int y = t & 63, ch = __builtin_amdgcn_readfirstlane(t >> 6);
            const bool y_ok = y < TY;
            const int yc = y_ok ? y : TY - 1;
            const int C = (MX + 3) / 4;
            Acc *exits = reinterpret_cast<Acc *>(rows);             // [4][64][K]; the rows buffer is not in use yet
#pragma unroll 1
            for (int s = 0; s < a.nx; s++) {
                const bool causal = a.xs[s].causal != 0;
                Acc cur[XC][K];                       // the row's tails of scan s, this wave's chunk, in scan order
#pragma unroll
                for (int ii = 0; ii < XC; ii++) {
                    const int i = ch * C + ii;
                    const int ic = (ii < C && i < MX) ? i : MX - 1;
                    const int tt = causal ? ic : MX - 1 - ic;
#pragma unroll
                    for (int r = 0; r < K; r++) cur[ii][r] = xc[((s * MX + tt) * K + r) * TY + yc];
                }
#pragma unroll 1
                for (int q = 0; q < s; q++) {               // chaining on the completed carries of the earlier scans
                    const bool qc = a.xs[q].causal != 0;
                    Acc Wv[4][K * K], entering[K];          // the four border variants of W[q -> s] (uniform)
#pragma unroll
                    for (int v = 0; v < 4; v++)
#pragma unroll
                        for (int e = 0; e < K * K; e++) Wv[v][e] = Wx[(((v * a.nx + q) * a.nx + s) * K) * K + e];
#pragma unroll
                    for (int o = 0; o < K; o++) entering[o] = a.x_incoming[(int64_t)(q * K + o) * Lx + ybase + yc];
#pragma unroll
                    for (int ii = 0; ii < XC; ii++) {
                        const int i = ch * C + ii;
                        const int ic = (ii < C && i < MX) ? i : MX - 1;
                        const int tt = causal ? ic : MX - 1 - ic;
                        const bool t_first = tt == 0, t_last = tt == MX - 1;
                        const bool q_first = qc ? t_first : t_last;
                        const int tp = q_first ? tt : (qc ? tt - 1 : tt + 1);
                        Acc c[K];
#pragma unroll
                        for (int o = 0; o < K; o++) {
                            const Acc from_tile = xc[((q * MX + tp) * K + o) * TY + yc];
                            c[o] = q_first ? entering[o] : from_tile;
                        }
#pragma unroll
                        for (int r = 0; r < K; r++)
#pragma unroll
                            for (int o = 0; o < K; o++) {
                                const int e = r * K + o;
                                const Acc w = t_first ? (t_last ? Wv[3][e] : Wv[1][e]) : (t_last ? Wv[2][e] : Wv[0][e]);
                                cur[ii][r] = cur[ii][r] + w * c[o];
                            }
                    }
                }
                Acc A[K][K], AC[K][K];                      // A and A^C (uniform)
#pragma unroll
                for (int r = 0; r < K; r++)
#pragma unroll
                    for (int o = 0; o < K; o++) { A[r][o] = Ax[(s * K + r) * K + o]; AC[r][o] = A[r][o]; }
                for (int p = 1; p < C; p++) {
                    Acc nxt[K][K];
#pragma unroll
                    for (int r = 0; r < K; r++)
#pragma unroll
                        for (int o = 0; o < K; o++) {
                            Acc acc = Acc(0);
#pragma unroll
                            for (int m = 0; m < K; m++) acc = acc + A[r][m] * AC[m][o];
                            nxt[r][o] = acc;
                        }
#pragma unroll
                    for (int r = 0; r < K; r++)
#pragma unroll
                        for (int o = 0; o < K; o++) AC[r][o] = nxt[r][o];
                }
                // chunk-local recurrence from a zero state (tiles beyond the chunk's end must not move the state)
                Acc state[K];
#pragma unroll
                for (int r = 0; r < K; r++) state[r] = Acc(0);
#pragma unroll
                for (int ii = 0; ii < XC; ii++) {
                    const bool live = ii < C && ch * C + ii < MX;
#pragma unroll
                    for (int r = 0; r < K; r++)
#pragma unroll
                        for (int o = 0; o < K; o++) cur[ii][r] = cur[ii][r] + A[r][o] * state[o];
#pragma unroll
                    for (int r = 0; r < K; r++) state[r] = live ? cur[ii][r] : state[r];
                }
                // (a partial last chunk: its exit state is never read -- no chunk follows it)
#pragma unroll
                for (int r = 0; r < K; r++) exits[(ch * 64 + y) * K + r] = state[r];
                __syncthreads();
                Acc inc[K];
#pragma unroll
                for (int r = 0; r < K; r++) inc[r] = Acc(0);
                for (int c = 0; c < ch; c++) {
                    Acc nx[K];
#pragma unroll
                    for (int r = 0; r < K; r++) nx[r] = exits[(c * 64 + y) * K + r];
#pragma unroll
                    for (int r = 0; r < K; r++)
#pragma unroll
                        for (int o = 0; o < K; o++) nx[r] = nx[r] + AC[r][o] * inc[o];
#pragma unroll
                    for (int r = 0; r < K; r++) inc[r] = nx[r];
                }
#pragma unroll
                for (int ii = 0; ii < XC; ii++) {
                    Acc yv[K];
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        yv[r] = Acc(0);
#pragma unroll
                        for (int o = 0; o < K; o++) yv[r] = yv[r] + A[r][o] * inc[o];
                    }
                    const int i = ch * C + ii;
                    const int tt = causal ? i : MX - 1 - i;
                    const bool live = ii < C && i < MX;
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        inc[r] = yv[r];
                        if (live && y_ok) xc[((s * MX + tt) * K + r) * TY + y] = cur[ii][r] + yv[r];
                    }
                }
                __syncthreads();      // scan s + 1 chains on these; the exit states may be overwritten
            }
        }
        __syncthreads();
        if (xg == 0 && jg == 0) {      // one workgroup of the tile row publishes the completed tails (the final pass reads them)
            for (int e = t; e < n_strips * cps; e += 256)
                *reinterpret_cast<A4 *>(xt_done + (int64_t)(e / cps) * Lx + ybase + 4 * (e % cps)) = xc4[e];
        }
    }
    int jr, tx;
    tile_of(row, jr, tx);
    const bool row_ok = tx < a.MX;
    const int vx = (tx == 0 ? 1 : 0) | (tx == a.MX - 1 ? 2 : 0);
    Acc tv[kFusedMaxScans * K];
#pragma unroll
    for (int qo = 0; qo < kFusedMaxScans * K; qo++) tv[qo] = Acc(0);
    // tau[q][o] = tail r of the tile-local y scans (through scan j) of the strip c_q[o](y): the completed x
    // carry entering this tile, as a function of the row.  Like every tail it is a contraction with Hy; the
    // 16 lanes of the row split the TY-term sum and all-reduce it with DPP.
    const bool residual = (G != nullptr);
    if (residual) {
        if (row_ok) {
            const int r = jr % K, j = jr / K;
            const int vy = ((ty == 0 && a.y_first_border) ? 1 : 0) | ((ty == a.MY - 1 && a.y_last_border) ? 2 : 0);
            const int64_t Lx = a.NYP * a.NZ;
            // the row's 16 lanes split the TY-term sums: four consecutive rows per lane and 64-row block (TY <= 128)
#pragma unroll
            for (int blk = 0; blk < (TALL ? 128 : 64); blk += 64) {
                const bool lane_in = blk + 4 * l < TY;
                A4 hy = A4{Acc(0), Acc(0), Acc(0), Acc(0)};
                if (lane_in) hy = *reinterpret_cast<const A4 *>(Hy + ((size_t)(vy * a.ny + j) * K + r) * TY + blk + 4 * l);
                const int64_t y0 = (int64_t)ty * TY + a.NYP * z + blk + 4 * l;
#pragma unroll
                for (int q = 0; q < kFusedMaxScans; q++) {
                    if (q < a.nx) {
                        const bool qc = a.xs[q].causal != 0;
                        const bool q_first = qc ? (tx == 0) : (tx == a.MX - 1);
                        const int tp = qc ? tx - 1 : tx + 1;
#pragma unroll
                        for (int o = 0; o < K; o++) {
                            A4 c = A4{Acc(0), Acc(0), Acc(0), Acc(0)};
                            if (!q_first && lane_in) {
                                if constexpr (XC > 0) c = *reinterpret_cast<const A4 *>(xc + ((q * a.MX + tp) * K + o) * TY + blk + 4 * l);
                                else c = *reinterpret_cast<const A4 *>(a.xt + (((int64_t)q * a.MX + tp) * K + o) * Lx + y0);
                            }
                            tv[q * K + o] = tv[q * K + o] + (hy.x * c.x + hy.y * c.y + hy.z * c.z + hy.w * c.w);
                        }
                    }
                }
            }
        }
        // all-reduce over the 16 lanes of the row: xor 1, xor 2, half mirror, mirror
#pragma unroll
        for (int qo = 0; qo < kFusedMaxScans * K; qo++) {
            if (qo < nxk) {
                tv[qo] = tv[qo] + dpp_move<0xB1>(tv[qo]);
                tv[qo] = tv[qo] + dpp_move<0x4E>(tv[qo]);
                tv[qo] = tv[qo] + dpp_move<0x141>(tv[qo]);
                tv[qo] = tv[qo] + dpp_move<0x140>(tv[qo]);
            }
        }
        // G[variant 0][q][o][256] -> LDS, each 256-sample row chunk-swizzled like the pixel rows
        const A4 *gs = reinterpret_cast<const A4 *>(G);
        A4 *gl4 = reinterpret_cast<A4 *>(g_lds);
        for (int c = t; c < nxk * 64; c += 256) gl4[(c & ~63) | swz_chunk(c & 63)] = gs[c];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) rows4[(rg + 4 * i) * 64 + swz_chunk(cc)] = tmp[i];
    __syncthreads();
    {
        Acc v[1][kFusedSeg];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            A4 q = rows4[row * 64 + 4 * l + (j ^ sw)];
            v[0][4 * j + 0] = q.x; v[0][4 * j + 1] = q.y; v[0][4 * j + 2] = q.z; v[0][4 * j + 3] = q.w;
        }
        Acc zero[1][K];
#pragma unroll
        for (int j = 0; j < K; j++) zero[0][j] = Acc(0);
#pragma unroll 1
        for (int s = 0; s < a.nx; s++) {
            const FusedScan<Acc> &sc = a.xs[s];
            const bool causal = sc.causal != 0;
            const bool tile_first = causal ? (tx == 0) : (tx == a.MX - 1);
            const int last_lane = (EDGE && tx == a.MX - 1) ? a.last_lane : 15;       // a row's last tile may be partial
            const bool first_lane = causal ? (l == 0) : (l == last_lane);
            const bool clamp_first = a.clamped && tile_first && first_lane;
            bool cf = clamp_first;
            if constexpr (MOD) {        // FusedArgs::mod_form: zero-border scans behind border modifications (scan_device.h)
                if (causal) border_mod_rows16<Acc, true, 1>(v, sc, clamp_first);
                else        border_mod_rows16<Acc, false, 1>(v, sc, clamp_first);
                cf = false;
            }
            if (causal) scan_rows16<Acc, true, K, 1>(v, sc, first_lane, cf, zero);
            else        scan_rows16<Acc, false, K, 1>(v, sc, first_lane, cf, zero, l > last_lane,
                                                          (EDGE && tx == a.MX - 1) ? a.last_cols - 16 * a.last_lane : kFusedSeg);
        }
        if (residual) {
            // v += sum G_q[.][o] * tau[q][o].  Interior tiles (variant 0) read G from the LDS copy, the few border tiles their
            // variant from memory -- two loops, each with its own address space.  (As ONE select between the two pointers every
            // chunk was a generic (flat) load with two selects on its address; flat loads count against the LDS and the memory
            // counter.)
            typedef __attribute__((address_space(3))) const A4 *LdsA4;
            typedef __attribute__((address_space(1))) const A4 *GlobA4;
            auto add_residual = [&](auto from_lds) __attribute__((always_inline)) {
                constexpr bool LDS = decltype(from_lds)::value;
#pragma unroll
                for (int q = 0; q < kFusedMaxScans; q++) {
                    if (q < a.nx) {
#pragma unroll
                        for (int o = 0; o < K; o++) {
                            // this lane's 16 columns of G_q[.][o]
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                A4 w;
                                if constexpr (LDS) w = ((LdsA4)reinterpret_cast<const A4 *>(g_lds))[(q * K + o) * 64 + 4 * l + (c ^ sw)];
                                else w = ((GlobA4)reinterpret_cast<const A4 *>(G + (((size_t)vx * a.nx + q) * K + o) * kFusedTX))[4 * l + c];
                                const Acc tq = tv[q * K + o];
                                v[0][4 * c + 0] = v[0][4 * c + 0] + w.x * tq; v[0][4 * c + 1] = v[0][4 * c + 1] + w.y * tq;
                                v[0][4 * c + 2] = v[0][4 * c + 2] + w.z * tq; v[0][4 * c + 3] = v[0][4 * c + 3] + w.w * tq;
                            }
                        }
                    }
                }
            };
            if (vx == 0) add_residual(std::true_type{});
            else add_residual(std::false_type{});
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            A4 q;
            q.x = v[0][4 * j + 0]; q.y = v[0][4 * j + 1]; q.z = v[0][4 * j + 2]; q.w = v[0][4 * j + 3];
            rows4[row * 64 + 4 * l + (j ^ sw)] = q;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = rg + 4 * i;
        int jr_i, tx_i;
        tile_of(r, jr_i, tx_i);
        if (tx_i < a.MX) yt4[row_tile_index(jr_i, tx_i) * 64 + cc] = rows4[r * 64 + swz_chunk(cc)];
    }
}

}  // namespace

template <typename P>
int launch_fused_tails(int K, int TY, const void *src, bool src_u8, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                       const typename PixelTraits<P>::Acc *Hx, const typename PixelTraits<P>::Acc *Hy,
                       hipStream_t stream) {
    if (a.MX <= 0 || a.MY <= 0 || a.NZ <= 0) return RF_OK;
    if (a.NZ > 65535 || a.MY > 65535) { set_error("fused path: grid too large"); return RF_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)a.MX, (unsigned)a.MY, (unsigned)a.NZ);
    static const size_t pad_bytes = RF_KNOB("RF_TAILS_PAD_LDS") ? (size_t)atoi(RF_KNOB("RF_TAILS_PAD_LDS")) : 0;     // A/B: bounds the residency
    const size_t hx_bytes = (size_t)(a.nx > 0 ? a.nx : 1) * K * kFusedTX * sizeof(typename PixelTraits<P>::Acc) + pad_bytes;
    // the y part on the matrix cores (f32 pixels, whole-chunk loads): order 3 by default -- its vector ALU is the loaded pipe
    const size_t hy_bytes = (size_t)((a.ny * K + 3) & ~3) * 9 * 16;
    static const char *ym_knob = RF_KNOB("RF_TAILS_YMFMA");                  // A/B: 0 = never, 1 = every order
    const bool ymfma = a.ny > 0 && (ym_knob ? atoi(ym_knob) != 0 : K == 3);
#define RF_CASE(KK, TT)                                                                                             \
    if (K == KK && TY == TT) {                                                                                       \
        if constexpr (std::is_same<P, float>::value) {                                                               \
            if (src_u8) {                                                                                            \
                hipLaunchKernelGGL((fused_tails_kernel<P, KK, TT, uint8_t>), grid, dim3(kFusedThreads), hx_bytes, stream, \
                                   (const uint8_t *)src, a, Hx, Hy);                                                 \
                RF_HIP_CHECK(hipGetLastError());                                                                     \
                return RF_OK;                                                                                        \
            }                                                                                                        \
        }                                                                                                            \
        if (a.lin_limit > 0) {                  /* folded 1-D signal whose end is masked */                             \
            hipLaunchKernelGGL((fused_tails_kernel<P, KK, TT, P, 1>), grid, dim3(kFusedThreads), hx_bytes, stream,       \
                               (const P *)src, a, Hx, Hy);                                                           \
            RF_HIP_CHECK(hipGetLastError());                                                                         \
            return RF_OK;                                                                                            \
        }                                                                                                            \
        if constexpr (sizeof(P) >= 4) {                                                                              \
            if ((a.last_cols & 3) != 0) {       /* width not a multiple of 4: the variant with partial-chunk loads */    \
                hipLaunchKernelGGL((fused_tails_kernel<P, KK, TT, P, 2>), grid, dim3(kFusedThreads), hx_bytes, stream,   \
                                   (const P *)src, a, Hx, Hy);                                                       \
                RF_HIP_CHECK(hipGetLastError());                                                                     \
                return RF_OK;                                                                                        \
            }                                                                                                        \
        }                                                                                                            \
        if constexpr (std::is_same<P, float>::value && TT >= 32) {                                                   \
            if (ymfma) {        /* y tails on the matrix cores; the step's Hy slice behind Hx in the dynamic LDS */         \
                hipLaunchKernelGGL((fused_tails_kernel<P, KK, TT, P, 0, true>), grid, dim3(kFusedThreads), hx_bytes + hy_bytes, \
                                   stream, (const P *)src, a, Hx, Hy);                                               \
                RF_HIP_CHECK(hipGetLastError());                                                                     \
                return RF_OK;                                                                                        \
            }                                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL((fused_tails_kernel<P, KK, TT, P>), grid, dim3(kFusedThreads), hx_bytes, stream, (const P *)src, \
                           a, Hx, Hy);                                                                               \
        RF_HIP_CHECK(hipGetLastError());                                                                             \
        return RF_OK;                                                                                                \
    }
    RF_CASE(1, 64) RF_CASE(2, 64) RF_CASE(3, 64)
    RF_CASE(1, 32) RF_CASE(2, 32) RF_CASE(3, 32)
    RF_CASE(1, 128) RF_CASE(2, 128) RF_CASE(3, 128)
#undef RF_CASE
    set_error("fused tails: unsupported order %d / tile height %d", K, TY);
    return RF_ERR_UNSUPPORTED;
}

bool xscan_completes_x_tails(int K, int TY, int MX, int nx, int ny, size_t acc_bytes, int64_t tile_rows) {
    static const bool off = RF_KNOB("RF_NO_MERGED_CARRY_X") != nullptr;      // A/B runs
    // Measured (tools/ab_mcx.sh, tools/mid_probe.py; summed-area table, bicubic prefilter x 3 planes, order-2 and order-3
    // Gaussians at 1280^2 ... 4096^2): orders 1 and 2 gain 2-5 us of 22-78 us at every size; order 3 gains 10 us of 90 on
    // one plane of 2112^2 ... 4096^2 and loses 3 us of 186 on three planes of 4096^2 -- that launch is 1152 workgroups
    // of 47 KiB of LDS, a round and a half -- so order 3 takes this path while the launch fits one round.
    static const bool all = RF_KNOB("RF_MERGED_CARRY_X_ALL") != nullptr;     // A/B runs: order 3 whatever the launch size
    // (beyond 16 tiles per row only order 1 still gains: summed-area table 8192^2 160.5 -> 158.2 us, bicubic x 3 planes 6144^2
    //  272 -> 266 us; order 2 loses 3 us at 6144^2 and 8 us at 8192^2, where every workgroup repeats a 32-tile recurrence)
    if (off || nx <= 0 || ny <= 0 || MX > (K == 1 ? kXcMaxTiles : 16) || TY > 64 || TY % 4 != 0) return false;
    if (K >= 3 && !all) {
        int gj = 1;
        while (gj < 16 && (ny * K) % (2 * gj) == 0) gj *= 2;
        const int txp = 16 / gj;
        const int64_t blocks = tile_rows * (ny * K / gj) * ((MX + txp - 1) / txp);      // (launch_xscan_rows)
        if (blocks > 768) return false;
    }
    // rows + G + the tile row's x tails within the 64 KiB of LDS a kernel gets without asking for more
    const size_t lds = ((size_t)16 * kFusedTX + (size_t)nx * K * kFusedTX + (size_t)nx * MX * K * TY) * acc_bytes;
    if (lds > 64 * 1024) return false;
    return true;
}

// one launch of xscan_rows_kernel: the MOD instance (float accumulators only) for plans in mod form
#define RF_XSCAN_LAUNCH(PLAIN, MODDED, ...)                                                                          \
    do {                                                                                                            \
        bool launched_ = false;                                                                                     \
        if constexpr (std::is_same<Acc, float>::value) {                                                            \
            if (a.mod_form) { hipLaunchKernelGGL((xscan_rows_kernel<RF_UNPAREN MODDED>), __VA_ARGS__); launched_ = true; } \
        }                                                                                                           \
        if (!launched_) hipLaunchKernelGGL((xscan_rows_kernel<RF_UNPAREN PLAIN>), __VA_ARGS__);                     \
    } while (0)
#define RF_UNPAREN(...) __VA_ARGS__

template <typename Acc>
int launch_xscan_rows(int K, int TY, const FusedArgs<Acc> &a, const Acc *Hy, const Acc *G, hipStream_t stream,
                      const Acc *Wx, const Acc *Ax, Acc *xt_done) {
    // yt is [j][ty][r][x + NX*z]: every run of 256 consecutive samples is one combined row of one x tile
    const int n_jr = a.ny * K;
    if (n_jr <= 0 || a.MY <= 0 || a.MX <= 0 || a.NZ <= 0 || a.nx == 0) return RF_OK;
    // combined rows per workgroup: the largest power of two (<= 16) dividing their number; the rest of the 16 row
    // tiles are consecutive x tiles
    int gj = 1;
    while (gj < 16 && n_jr % (2 * gj) == 0) gj *= 2;
    const int txp = 16 / gj;
    const int64_t blocks = (int64_t)a.MY * a.NZ * (n_jr / gj) * ((a.MX + txp - 1) / txp);
    if (blocks >= (1ll << 31)) { set_error("xscan rows: grid too large"); return RF_ERR_UNSUPPORTED; }
    const unsigned grid = (unsigned)blocks;
    const bool edge = a.last_cols != kFusedTX;       // images of whole tiles keep the lean kernel
    const size_t g_bytes = (size_t)a.nx * K * kFusedTX * sizeof(Acc);
    if (Wx != nullptr) {      // the kernel completes the x tails too (XC)
        if (G == nullptr || Ax == nullptr || xt_done == nullptr || TY > 64 || a.MX > kXcMaxTiles) { set_error("xscan rows: merged carry scan misconfigured"); return RF_ERR_INVALID_ARG; }
        const size_t xc_bytes = g_bytes + (size_t)a.nx * a.MX * K * TY * sizeof(Acc);
#define RF_CASE(KK, CH)                                                                                                    \
        if (K == KK && (a.MX + 3) / 4 <= CH) {                                                                             \
            if (edge) RF_XSCAN_LAUNCH((Acc, KK, true, false, CH), (Acc, KK, true, false, CH, true), dim3(grid), dim3(256), xc_bytes, stream, a, gj, TY, Hy, G, Wx, Ax, xt_done);  \
            else      RF_XSCAN_LAUNCH((Acc, KK, false, false, CH), (Acc, KK, false, false, CH, true), dim3(grid), dim3(256), xc_bytes, stream, a, gj, TY, Hy, G, Wx, Ax, xt_done); \
            RF_HIP_CHECK(hipGetLastError());                                                                               \
            return RF_OK;                                                                                                  \
        }
        RF_CASE(1, 4) RF_CASE(2, 4) RF_CASE(3, 4) RF_CASE(1, 8)
#undef RF_CASE
        set_error("xscan rows: no merged carry scan for order %d at %d tiles per row", K, (int)a.MX);
        return RF_ERR_UNSUPPORTED;
    }
#define RF_CASE(KK)                                                                                                        \
    if (K == KK) {                                                                                                         \
        if (TY > 64) {      /* 128-row tiles: two 64-row blocks of the carry strips per lane */                            \
            if (edge) RF_XSCAN_LAUNCH((Acc, KK, true, true), (Acc, KK, true, true, 0, true), dim3(grid), dim3(256), g_bytes, stream, a, gj, TY, Hy, G);  \
            else      RF_XSCAN_LAUNCH((Acc, KK, false, true), (Acc, KK, false, true, 0, true), dim3(grid), dim3(256), g_bytes, stream, a, gj, TY, Hy, G); \
        } else {                                                                                                           \
            if (edge) RF_XSCAN_LAUNCH((Acc, KK, true, false), (Acc, KK, true, false, 0, true), dim3(grid), dim3(256), g_bytes, stream, a, gj, TY, Hy, G);  \
            else      RF_XSCAN_LAUNCH((Acc, KK, false, false), (Acc, KK, false, false, 0, true), dim3(grid), dim3(256), g_bytes, stream, a, gj, TY, Hy, G); \
        }                                                                                                                  \
        RF_HIP_CHECK(hipGetLastError());                                                                                   \
        return RF_OK;                                                                                                      \
    }
    RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    set_error("xscan rows: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}

template int launch_fused_tails<float>(int, int, const void *, bool, const FusedArgs<float> &, const float *, const float *, hipStream_t);
template int launch_fused_tails<int32_t>(int, int, const void *, bool, const FusedArgs<uint32_t> &, const uint32_t *,
                                         const uint32_t *, hipStream_t);
template int launch_fused_tails<int16_t>(int, int, const void *, bool, const FusedArgs<uint32_t> &, const uint32_t *,
                                         const uint32_t *, hipStream_t);
template int launch_fused_tails<double>(int, int, const void *, bool, const FusedArgs<double> &, const double *, const double *, hipStream_t);
template int launch_xscan_rows<float>(int, int, const FusedArgs<float> &, const float *, const float *, hipStream_t, const float *, const float *, float *);
template int launch_xscan_rows<uint32_t>(int, int, const FusedArgs<uint32_t> &, const uint32_t *, const uint32_t *, hipStream_t, const uint32_t *, const uint32_t *, uint32_t *);
template int launch_xscan_rows<double>(int, int, const FusedArgs<double> &, const double *, const double *, hipStream_t, const double *, const double *, double *);

}  // namespace rf
