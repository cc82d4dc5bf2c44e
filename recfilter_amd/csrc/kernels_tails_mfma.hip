// kernels_tails_mfma.hip -- pass 1 of the fused path (kernels_tails.hip's contraction, one tile per workgroup, register-
// staged steps of 32 rows) with the x-tail contraction on the matrix cores.
//
// Why (round 4, profiles/r4/pass1_order3.txt): fused_tails_kernel at order 3 with two x and two y scans streams at
// 4.7 TB/s where the order-1 instance of the same kernel reaches 6.1.  Per 32-row step and workgroup it moves 258 KiB through
// LDS -- 98 KiB of that the Hx table, re-read by every row slot for every step -- and issues ~330 VALU instructions per wave,
// 96 packed FMAs + 48 DPP adds of them the x tails: the LDS pipe is busy in half of the kernel's cycles, the VALU in a third,
// and with two barriers per step the phases of a workgroup do not overlap.  Here
//   * x tails: v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4 x 4 outer products per issue, exact f32 fma chains), the
//     form kernels_stream.hip uses: wave w owns columns 64w .. 64w+63 of all 32 rows; block b = (column phase cg, row group
//     rg), A[b][i] = pixel (row 4 rg + i, column 4 (16w + 4m + cg) + e), B[b][j] = Hx[tail 4g + j][that column] -- SIXTEEN
//     REGISTERS per group of four tails, loaded once per tile, so Hx never enters LDS -- one ds_read_b128 feeds four MFMAs;
//     the four column phases meet through two cross-lane adds, the four waves' partial sums through a 3 KiB LDS stage that
//     the next step's first barrier publishes (no barrier of its own);
//   * y tails: as in fused_tails_kernel -- thread = column, Hy through scalar loads, packed FMAs -- or (YM) on the matrix
//     cores too, with the step's slice of Hy in LDS.
// LDS per workgroup: 32 KiB tile + 0.5 KiB per x tail (+ 1-2 KiB Hy slice): four workgroups per CU as before; LDS traffic
// per step 96-110 KiB instead of 258.  The tile image is XOR-swizzled per ROW (kernels_stream.hip, slot_pos): the 16 rows an
// A operand gathers at one column sit on 16 different bank groups.
//
// Takes f32 images of whole tiles (width % 256 == 0, height % TY == 0) without a fused prologue; everything else stays on
// fused_tails_kernel.  Same tables, same tails (summation order differs: results equal to rounding).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "kernels_fused.h"
#include "scan_device.h"

namespace rf {

namespace {

typedef float F2 __attribute__((ext_vector_type(2)));
typedef float F4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 32;                 // rows staged per step
constexpr int kHyPitch4 = 9;              // YM: 32 rows + 4 floats of padding per tail, in 16-byte units

// v[l] + v[l ^ 16] + v[l ^ 32] + v[l ^ 48] in every lane: two gfx950 lane swaps (v_permlane16_swap / v_permlane32_swap exchange
// odd rows / the upper half of one register with even rows / the lower half of another; with both registers a copy of v the
// two results are "this half" and "the other half") instead of two ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ float sum_lanes_xor_16_32(float v) {
    typedef unsigned U2 __attribute__((ext_vector_type(2)));
    U2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ int row_swz(int row, int c) { return c ^ (row & 15) ^ ((c >> 4) & 3); }

// NX, NY: scans along x / y (compile time: they size the Hx register fragments and the y accumulators -- at four workgroups
// per CU the kernel has 128 registers and uses nearly all of them); YM: y tails on the matrix cores
template <int K, int TY, int NX, int NY, bool YM>
__global__ void __launch_bounds__(kFusedThreads, 4)
mfma_tails_kernel(const float *__restrict__ src, FusedArgs<float> a,
                  const float *__restrict__ Hx,     // [vx][s][r][256]
                  const float *__restrict__ Hy) {   // [vy][j][r][TY]
    __shared__ __attribute__((aligned(16))) float tile[kRows * kFusedTX];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_raw[];      // [stage: 4 waves x nxk x 32][YM: Hy slice]
    F4 *tile4 = reinterpret_cast<F4 *>(tile);
    float *stage = reinterpret_cast<float *>(dyn_raw);
    constexpr int NH = TY / kRows;
    constexpr int NL = kRows / 4;
    constexpr int NGX = NX > 0 ? (NX * K + 3) / 4 : 1;
    constexpr int NGY = NY > 0 ? (NY * K + 3) / 4 : 1;
    constexpr int NYK = NY > 0 ? NY * K : 1;

    const int t = threadIdx.x;
    const int tx = blockIdx.x, ty = blockIdx.y;
    const int64_t z = blockIdx.z;
    if (a.plane_batch) src = reinterpret_cast<const float *>(a.in_planes[z]);
    const int64_t tile_off = (a.plane_batch ? 0 : z * a.NX * a.NY) + (int64_t)ty * TY * a.NX + (int64_t)tx * kFusedTX;
    const int vx = (tx == 0 ? 1 : 0) | (tx == a.MX - 1 ? 2 : 0);
    const int vy = ((ty == 0 && a.y_first_border) ? 1 : 0) | ((ty == a.MY - 1 && a.y_last_border) ? 2 : 0);
    constexpr int nxk = NX * K, nyk = NY * K;
    const int64_t Lx = a.NYP * a.NZ;
    F4 *stage4 = reinterpret_cast<F4 *>(stage);
    F4 *hy4 = stage4 + 4 * (nxk > 0 ? nxk : 1) * (kRows / 4);

    const int cc = t & 63, rg = t >> 6;                              // load: 16-byte chunk, row group
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row16 = lane & 15, cg = lane >> 4, j4 = lane & 3;      // x part: A row, column phase; B tail within its group
    const char *spb = reinterpret_cast<const char *>(src + tile_off);
    const uint32_t off0 = (uint32_t)rg * a.row_bytes + (uint32_t)cc * 16u;
    auto ld = [&](int row) { return __builtin_nontemporal_load(reinterpret_cast<const F4 *>(spb + (off0 + (uint32_t)row * a.row_bytes))); };

    F4 pre[NL];
    F4 hy_pre = F4{0.f, 0.f, 0.f, 0.f};
    const int hy_jr = t >> 3, hy_m = t & 7;
    const int nyk4 = (nyk + 3) & ~3;
    auto load_step = [&](int h) {
        if constexpr (YM) {
            if (hy_jr < nyk) hy_pre = *reinterpret_cast<const F4 *>(Hy + (size_t)(vy * nyk + hy_jr) * TY + kRows * h + 4 * hy_m);
        }
#pragma unroll
        for (int i = 0; i < NL; i++) pre[i] = ld(kRows * h + 4 * i);
    };
    // Hx fragments of this tile's border variant: [g][4m + e] = Hx[tail 4g + j4][column 4 (16w + 4m + cg) + e].  Requested
    // BEFORE the first step's pixels: the wait for those pixels then covers them (loads return in order), and the step loop --
    // where only the next step's pixels are ever pending -- needs no wait of its own for them.  (Requested behind the pixels,
    // the loop's first use of a fragment was a `s_waitcnt vmcnt(0)`: every step waited for the pixels it had just requested.)
    float Bx[NGX][16];
    if constexpr (NX > 0) {
#pragma unroll
        for (int g = 0; g < NGX; g++) {
            // (lanes of a tail that does not exist fetch tail 0: what they contribute lands in accumulator columns nobody
            //  stores -- no select behind the load, which would make the kernel wait for the table before it requests a pixel)
            const int sr = 4 * g + j4 < nxk ? 4 * g + j4 : 0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const F4 h = *reinterpret_cast<const F4 *>(Hx + ((size_t)vx * nxk + sr) * kFusedTX + 4 * (16 * w + 4 * m + cg));
                Bx[g][4 * m + 0] = h.x; Bx[g][4 * m + 1] = h.y; Bx[g][4 * m + 2] = h.z; Bx[g][4 * m + 3] = h.w;
            }
        }
    }
    load_step(0);

    // the x tails of step h: the four waves' partial sums -> xt, 16 bytes (four rows) per thread
    auto flush_xtails = [&](int h) {
        if (t < nxk * (kRows / 4)) {
            const int sr = t >> 3, q = t & 7;
            F4 v = stage4[(0 * nxk + sr) * (kRows / 4) + q];
#pragma unroll
            for (int ww = 1; ww < 4; ww++) v = v + stage4[(ww * nxk + sr) * (kRows / 4) + q];
            const int s = sr / K, r = sr % K;
            const int64_t line0 = (int64_t)ty * TY + kRows * h + a.NYP * z;
            *reinterpret_cast<F4 *>(a.xt + (((int64_t)s * a.MX + tx) * K + r) * Lx + line0 + 4 * q) = v;
        }
    };

    float comb[YM ? 1 : NYK];
#pragma unroll
    for (int jr = 0; jr < (YM ? 1 : NYK); jr++) comb[jr] = 0.0f;
    F4 yacc[YM ? NGY : 1];
#pragma unroll
    for (int g = 0; g < (YM ? NGY : 1); g++) yacc[g] = F4{0.f, 0.f, 0.f, 0.f};

    // (the step loop is NOT unrolled: a quarter of the code for a 128-row tile -- four workgroups per CU run different
    // phases of it out of one instruction cache -- and no cross-step scheduling that would cost registers)
#pragma unroll 1
    for (int h = 0; h < NH; h++) {
        if (h > 0) {
            __syncthreads();                                         // step h-1: readers done, its x-tail stage complete
            if (nxk > 0) flush_xtails(h - 1);
        }
#pragma unroll
        for (int i = 0; i < NL; i++) tile4[(rg + 4 * i) * 64 + row_swz(rg + 4 * i, cc)] = pre[i];
        if constexpr (YM) {
            if (hy_jr < nyk4) hy4[hy_jr * kHyPitch4 + hy_m] = hy_pre;
        }
        __syncthreads();
        if (h + 1 < NH) load_step(h + 1);                            // next step in flight during this one's math

        // ---- x tails: 2 sets of 16 rows x this wave's 64 columns on the matrix cores ----
        // (one accumulator per group of four tails: the groups' chains interleave; a single group alternates between two)
        if (nxk > 0) {
            constexpr int NA = NGX == 1 ? 2 : 1;
#pragma unroll
            for (int n = 0; n < kRows / 16; n++) {
                F4 av[4];
#pragma unroll
                for (int m = 0; m < 4; m++) av[m] = tile4[(16 * n + row16) * 64 + row_swz(row16, 16 * w + 4 * m + cg)];
                F4 acc[NGX][NA];
#pragma unroll
                for (int g = 0; g < NGX; g++)
#pragma unroll
                    for (int u = 0; u < NA; u++) acc[g][u] = F4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 4; m++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
#pragma unroll
                        for (int g = 0; g < NGX; g++)
                            acc[g][e % NA] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m][e], Bx[g][4 * m + e], acc[g][e % NA], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int g = 0; g < NGX; g++) {
                    F4 d = acc[g][0];
                    if constexpr (NA == 2) d = d + acc[g][1];
                    // the four column phases: lanes l, l ^ 16, l ^ 32, l ^ 48
#pragma unroll
                    for (int i = 0; i < 4; i++) d[i] = sum_lanes_xor_16_32(d[i]);
                    // lane 4 rg' + j (cg == 0), register i: row 16 n + 4 rg' + i of tail 4 g + j
                    const int sr = 4 * g + j4;
                    if (cg == 0 && sr < nxk) stage4[((w * nxk + sr) * kRows + 16 * n + (row16 & 12)) >> 2] = d;
                }
            }
        }

        // ---- y tails: contract this step's rows with Hy (thread = column t), eight rows at a time ----
        if (nyk > 0) {
            const int cbase = (t >> 2) ^ ((t >> 6) & 3);              // chunk t/4 under the row-independent part of the swizzle
            if constexpr (YM) {
#pragma unroll
                for (int m = 0; m < kRows / 4; m++) {
                    float col[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) col[i] = tile[(4 * m + i) * kFusedTX + ((cbase ^ ((4 * m + i) & 15)) << 2) + (t & 3)];
#pragma unroll
                    for (int g = 0; g < NGY; g++) {
                        if (4 * g < nyk) {                           // (uniform)
                            const F4 hh = hy4[(4 * g + (t & 3)) * kHyPitch4 + m];
                            yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[0], hh.x, yacc[g], 0, 0, 0);
                            yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[1], hh.y, yacc[g], 0, 0, 0);
                            yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[2], hh.z, yacc[g], 0, 0, 0);
                            yacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(col[3], hh.w, yacc[g], 0, 0, 0);
                        }
                    }
                }
            } else {
                // two rows per instruction (v_pk_fma_f32), two tails at a time (consecutive packed FMAs independent); the
                // chunk loop is NOT unrolled: eight column registers live instead of thirty-two
#pragma unroll 1
                for (int c8 = 0; c8 < kRows / 8; c8++) {
                    float col[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) col[i] = tile[(8 * c8 + i) * kFusedTX + ((cbase ^ ((8 * c8 + i) & 15)) << 2) + (t & 3)];
#pragma unroll
                    for (int g = 0; g < (NYK + 1) / 2; g++) {
                        if (2 * g < nyk) {
                            const int j0 = 2 * g, j1 = (2 * g + 1 < nyk) ? 2 * g + 1 : 2 * g;
                            const float *h0 = Hy + (size_t)(vy * nyk + j0) * TY + kRows * h + 8 * c8;     // wave-uniform: scalar loads
                            const float *h1 = Hy + (size_t)(vy * nyk + j1) * TY + kRows * h + 8 * c8;
                            F2 c0 = F2{0.0f, 0.0f}, c1 = F2{0.0f, 0.0f};
#pragma unroll
                            for (int i = 0; i < 8; i += 2) {
                                const F2 c2 = F2{col[i], col[i + 1]};
                                c0 = F2{h0[i], h0[i + 1]} * c2 + c0;
                                c1 = F2{h1[i], h1[i + 1]} * c2 + c1;
                            }
                            comb[2 * g] = comb[2 * g] + (c0.x + c0.y);
                            if (2 * g + 1 < NYK) comb[2 * g + 1] = comb[2 * g + 1] + (c1.x + c1.y);
                        }
                    }
                }
            }
        }
    }
    if (nxk > 0) {
        __syncthreads();                                             // the last step's stage is complete
        flush_xtails(NH - 1);
    }
    // combined rows -> yt; with x scans in the filter xscan_rows_kernel finishes them in place
    if constexpr (YM) {
        if (nyk > 0) {
            // lane 4b + j, register i: combined row 4g + j at the tile's column (t & ~3) + i
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
        for (int jr = 0; jr < NYK; jr++)
            if (jr < nyk) a.yt[a.yt_index(jr / K, ty, jr % K, K, line)] = comb[jr];
    }
}

}  // namespace

// When pass 1 takes this kernel: f32 images of whole tiles, no prologue, pixel-typed planes (the launch is the same grid as
// fused_tails_kernel's: one workgroup per tile).  mode: 0 automatic, +1 wherever the shape allows, -1 never.
bool mfma_tails_applicable(int K, int TY, bool src_u8, int pw_flags, int last_cols, int last_rows, int64_t lin_limit, int nx, int ny,
                           int mode) {
    static const char *knob = RF_KNOB("RF_TAILS_XMFMA");              // A/B: 0 = never, 1 = wherever the shape allows
    if (knob) mode = atoi(knob) != 0 ? 1 : -1;
    if (mode < 0) return false;
    if (src_u8 || (pw_flags & 1) || last_cols != kFusedTX || last_rows != TY || lin_limit != 0) return false;
    // (at most two scans per dimension -- the causal / anticausal pair: the instances that exist)
    if (K < 1 || K > 3 || (TY != 32 && TY != 64 && TY != 128) || nx + ny == 0 || nx > 2 || ny > 2) return false;
    if (K == 3 && TY == 64 && nx == 2) return false;      // (the two instances that do not fit 128 registers)
    if (mode > 0) return true;
    // Measured (profiles/r4/ab_pass1_mfma.txt, same box, alternating): order 3 x 3 planes of 16384^2 630-673 -> 589-596 us,
    // order 2 on 16384^2 197 -> 188 us; order 1 loses 1-2 % (two of a group's four tails exist: half of every MFMA is idle)
    return K >= 2;
}

int launch_mfma_tails(int K, int TY, const float *src, const FusedArgs<float> &a, const float *Hx, const float *Hy, hipStream_t stream) {
    if (a.MX <= 0 || a.MY <= 0 || a.NZ <= 0) return RF_OK;
    if (a.NZ > 65535 || a.MY > 65535) { set_error("fused path: grid too large"); return RF_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)a.MX, (unsigned)a.MY, (unsigned)a.NZ);
    const int nxk = a.nx * K, nyk = a.ny * K;
    // (YM = false: with the y tails on the matrix cores too -- the template's other form -- every instance needs more than
    //  the 128 registers four workgroups per CU leave it, and spill traffic shares the pixel loads' counter)
    const size_t stage_bytes = (size_t)4 * (nxk > 0 ? nxk : 1) * kRows * sizeof(float);
    (void)nyk;
#define RF_CASE(KK, TT, XX, YY)                                                                                           \
    if (K == KK && TY == TT && a.nx == XX && a.ny == YY) {                                                                \
        hipLaunchKernelGGL((mfma_tails_kernel<KK, TT, XX, YY, false>), grid, dim3(kFusedThreads), stage_bytes, stream, src, a, Hx, Hy);            \
        RF_HIP_CHECK(hipGetLastError());                                                                                  \
        return RF_OK;                                                                                                     \
    }
#define RF_CASES(KK, TT) RF_CASE(KK, TT, 2, 2) RF_CASE(KK, TT, 1, 1) RF_CASE(KK, TT, 2, 0) RF_CASE(KK, TT, 0, 2) RF_CASE(KK, TT, 1, 0) \
    RF_CASE(KK, TT, 0, 1) RF_CASE(KK, TT, 2, 1) RF_CASE(KK, TT, 1, 2)
    RF_CASES(1, 32) RF_CASES(1, 64) RF_CASES(1, 128)
    RF_CASES(2, 32) RF_CASES(2, 64) RF_CASES(2, 128)
    RF_CASES(3, 32) RF_CASES(3, 64) RF_CASES(3, 128)
#undef RF_CASES
#undef RF_CASE
    set_error("mfma tails: unsupported order %d / tile height %d / %d + %d scans", K, TY, a.nx, a.ny);
    return RF_ERR_UNSUPPORTED;
}

}  // namespace rf
