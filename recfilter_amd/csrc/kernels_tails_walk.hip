// kernels_tails_walk.hip -- pass 1 of a 3-D filter in ONE read of the volume: the x tails, the y tails' combined rows AND
// the z tails.
//
// Why: a 3-D filter runs as the fused x/y stage per plane followed by the strided z stage (plan_fused.cpp): four reads and
// two writes of the volume, 24 B per sample.  The z operators commute with the x/y filter (plan_strided.h, "early form"), so
// the z tails may be taken from the RAW volume, which pass 1 of the x/y stage reads anyway.  A z tail is a sum over the TZ
// planes of a z tile, sum_z Hz[tail][z] * v(x, y, z): whoever forms it keeps n_z * k partial sums per (x, y) sample while it
// walks the planes.  Here a workgroup of 1024 threads -- ONE per CU, sixteen waves -- owns a patch of 256 x 32 samples and
// walks the TZ planes of one z tile: eight samples and 32 accumulator registers per thread, a quarter of the CU's register
// file for the patch.  Per plane (32 KiB) the patch is staged in LDS and three contractions run on the matrix cores
// (v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer products per issue, exact f32 fma chains; 24 per wave and plane):
//   x tails   wave = 16 columns x 32 rows; block = (four rows, column half), A = pixel, B = Hx[tail j][column]; the two column
//             halves meet through one v_permlane32_swap, the sixteen waves' partial sums through an LDS stage
//   y tails   thread = (column, eight rows); block = four adjacent columns, B = Hy[tail j][row]; lane 4 b + j ends up with
//             tail j of the block's four columns; the four row quarters meet through an LDS stage
//   z tails   A = the lane's own sample, straight from the load registers, B = Hz[tail j][plane]; accumulators alive across
//             the whole walk
// (with the products on the vector ALU the kernel took as long as the two passes it replaces: NOTES.md, round 4).
// A y tile of TY = 32 * parts rows is `parts` patches, i.e. workgroups: each stores its PART of the combined rows (ytp[part]);
// xscan_rows_kernel, the next reader of the rows, adds the parts up as it loads them (FusedArgs::yt_parts).
// Patch and stages are double-buffered by the plane's parity: one workgroup barrier per plane.  Two planes of a thread's loads
// in flight (four: no faster).  20 B per sample instead of 24 (+ 1 B of parts).
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

constexpr int kRows = 32;                 // rows of a patch (TALL patches: kTallRows x kTallCols)
constexpr int kTallRows = 64, kTallCols = 128;
constexpr int kWalkThreads = 1024;

// v[l] + v[l ^ 32] in every lane (v_permlane32_swap: kernels_tails_mfma.hip)
__device__ __forceinline__ float sum_lanes_xor_32(float v) {
    typedef unsigned U2 __attribute__((ext_vector_type(2)));
    const U2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The patch in LDS: rows padded to 65 chunks of 16 bytes.  The sixteen rows an A operand gathers at one column then sit on
// sixteen different bank groups, and every access of the step is one address register plus an immediate (with the XOR swizzle of
// kernels_tails_mfma.hip the eight rows of the y part alone took eight address registers; this kernel has 128 in all).
constexpr int kPitch4 = kFusedTX / 4 + 1;
constexpr int kTallPitch4 = kTallCols / 4 + 1;

// K: order of the x/y stage; NX, NY: scans along x / y.  EDGE: the patch is not whole -- the last tile of a row may have fewer than
// 256 columns (a multiple of four), the last tile row fewer than TY rows: what does not exist loads as zeros (the tables of the
// last tiles are built for their extent, as for the staged pass 1) and is never stored.  Whole patches run a body of their own:
// joined in one, the masks sit behind every load of every patch (kernels_tails.hip, WHOLE).
// TALL (round 6): the patch is 128 columns x 64 rows instead of 256 x 32 -- the same 8192 samples, eight per thread.  A y tile of
// 128 rows is then TWO patches instead of four: half the combined-row parts, the kernel's dearest stores (and half of what
// xscan_rows reads); the x tails of a tile now come in two parts of their own (its column halves: xt and WalkArgs::xt2, a
// quarter of the bytes the y parts save), which the carry scan along x adds up as it loads them (CarryGeom::part2).  The bare
// read of that shape is 5 % faster as well (tools/microbench/walk_read.hip: 1.19 against 1.25 ms per 8 GiB; with the stores
// 1.75 against 1.91).  Orders <= 2 along x / y (four x tails: one flushing wave), 128-row y tiles.
// U4 (round 6): widths that are not multiples of four.  Rows are then only element-aligned: a thread's chunk is four 4-byte
// loads instead of one 16-byte load (same registers, four times the load instructions), the last chunk of a row may exist in
// part (masked per element, like everything that does not exist: zeros when the plane is staged), and the z tails -- the one
// unpadded destination -- are stored sample by sample.  Everything behind the staging works on the padded patch as before.
template <int K, int NX, int NY, bool EDGE, bool TALL, bool U4>
__device__ __forceinline__ void walk_tails_body(const float *__restrict__ src, const FusedArgs<float> &a, const WalkArgs &wa,
                                                const float *__restrict__ Hx,     // [vx][s][r][256]
                                                const float *__restrict__ Hy,     // [vy][j][r][TY]
                                                const float *__restrict__ HzT) {  // [vz][z][4]
    // [patch x 2][x stage: 16 waves x nxk x 32, x 2][y stage: 4 x nyk x 256, x 2][Hz: 128 x 4][B operands]: the patch and the two
    // stages are double-buffered by the plane's parity, so that a step needs ONE workgroup barrier (below)
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_raw[];
    constexpr int nxk = NX * K, nyk = NY * K;
    constexpr int NGX = (nxk + 3) / 4, NGY = (nyk + 3) / 4;
    constexpr int PH = TALL ? kTallRows : kRows, PW = TALL ? kTallCols : kFusedTX;         // the patch
    constexpr int CPR = PW / 4, P4 = TALL ? kTallPitch4 : kPitch4;                          // chunks of a patch row, its pitch in LDS
    constexpr int NQ = PH / 8;                                                              // row groups of the y part (eight rows each)
    constexpr int kTile4 = PH * P4, kXs4 = 16 * nxk * 8, kYs4 = NQ * nyk * CPR;
    static_assert(!TALL || nxk <= 4, "tall patches: one wave flushes the x tails");
    F4 *tile4 = reinterpret_cast<F4 *>(dyn_raw);
    F4 *stage4 = tile4 + 2 * kTile4;
    F4 *ystage4 = stage4 + 2 * kXs4;
    float *hz_lds = reinterpret_cast<float *>(ystage4 + 2 * kYs4);
    // the B operands of the x and y parts (fixed for the whole walk): 32 bytes per lane, kept in LDS -- as registers they are the
    // sixteen that decide between two and four planes of loads in flight
    F4 *hx_b = reinterpret_cast<F4 *>(hz_lds + 4 * 128);             // [g][wave][half][j][2]
    F4 *hy_b = hx_b + NGX * 16 * 2 * 4 * 2;                          // [g][yq][j][2]

    const int t = threadIdx.x;
    const int tx = TALL ? (int)blockIdx.x >> 1 : (int)blockIdx.x, xh = TALL ? (int)blockIdx.x & 1 : 0, tz = blockIdx.z;      // (tile column, half of it)
    const int ty = blockIdx.y >> wa.parts_log2, h = blockIdx.y & ((1 << wa.parts_log2) - 1);     // (tile row, patch of it)
    const int TZ = wa.TZ;
    const int vx = (tx == 0 ? 1 : 0) | (tx == a.MX - 1 ? 2 : 0);
    const int vy = (ty == 0 ? 1 : 0) | (ty == a.MY - 1 ? 2 : 0);
    const int vz = ((tz == 0 && wa.z_first_border) ? 1 : 0) | ((tz == wa.MZ - 1 && wa.z_last_border) ? 2 : 0);
    const int64_t Lx = a.NYP * a.NZ;

    const int cc = t & (CPR - 1), rg = t / CPR;                      // load: 16-byte chunk, row (and row + PH / 2)
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j4 = lane & 3;                                         // all three contractions: the tail this lane's B operand and results belong to
    const int64_t plane = a.NX * a.NY;
    // (a wave-uniform base advanced per plane + one 32-bit lane offset: no address registers per load)
    const char *spb = reinterpret_cast<const char *>(src + ((int64_t)tz * TZ) * plane + ((int64_t)ty * wa.TY + PH * h) * a.NX + (int64_t)tx * kFusedTX + PW * xh);
    const uint32_t off0 = (uint32_t)rg * a.row_bytes + (uint32_t)cc * 16u, off1 = off0 + (uint32_t)(PH / 2) * a.row_bytes;
    const int64_t plane_bytes = plane * (int64_t)sizeof(float);
    // the rows rg and rg + 16 and the chunk cc of this thread: do they exist?
    const int cols_here = (tx == a.MX - 1 ? a.last_cols : kFusedTX) - PW * xh;                // columns of the patch that exist (may be <= 0)
    const int rows_left = (ty == a.MY - 1 ? a.last_rows : wa.TY) - PH * h;                    // rows of the patch that exist (may be <= 0)
    const bool ok0 = !EDGE || (4 * cc < cols_here && rg < rows_left), ok1 = !EDGE || (4 * cc < cols_here && rg + PH / 2 < rows_left);
    bool okE[4];                                                     // U4: which samples of this thread's chunk exist
#pragma unroll
    for (int e = 0; e < 4; e++) okE[e] = !EDGE || !U4 || 4 * cc + e < cols_here;
    auto ld = [&](const char *pb, uint32_t off, bool ok) {
        // (a chunk that does not exist reads the volume's first chunk; `put` replaces it by zeros -- not here: a select behind
        //  the load would be waited for on the spot)
        if constexpr (U4) {
            F4 r;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const char *p = pb + off + 4 * e;
                if constexpr (EDGE) p = (ok && okE[e]) ? p : reinterpret_cast<const char *>(src);
                r[e] = __builtin_nontemporal_load(reinterpret_cast<const float *>(p));
            }
            return r;
        } else {
            if constexpr (EDGE) return __builtin_nontemporal_load(reinterpret_cast<const F4 *>(ok ? pb + off : reinterpret_cast<const char *>(src)));
            else return __builtin_nontemporal_load(reinterpret_cast<const F4 *>(pb + off));
        }
    };

    // ---- operands that do not change from plane to plane, requested before the first pixels ----
    // x: wave w owns columns 16 w .. 16 w + 15 of all 32 rows (TALL: columns 16 (w & 7) .. of rows 32 (w >> 3) ..);
    // block b = (rows 4 (b & 7) .., column half b >> 3)
    const int xcg = TALL ? (w & 7) : w, xrh = TALL ? (w >> 3) : 0;
    const int xrow = 32 * xrh + 4 * ((lane >> 2) & 7) + (lane & 3), xhf = lane >> 5;
    // (filled by the lanes that would hold them: lane (b, j) of wave w writes what lanes (.., j) of its half read back)
#pragma unroll
    for (int g = 0; g < NGX; g++) {
        const int sr = 4 * g + j4 < nxk ? 4 * g + j4 : 0;           // (a tail that does not exist: its products land in columns nobody stores)
        if ((lane & 28) == 0) {
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
                hx_b[(((g * 16 + w) * 2 + xhf) * 4 + j4) * 2 + mi] =
                    *reinterpret_cast<const F4 *>(Hx + ((size_t)vx * nxk + sr) * kFusedTX + PW * xh + 16 * xcg + 8 * xhf + 4 * mi);
        }
    }
    // y: thread = (column yc, rows 8 yq ..); block = four adjacent columns
    const int yc = t & (PW - 1), yq = __builtin_amdgcn_readfirstlane(t / PW);
    if (t < NGY * NQ * 4) {
        const int g = t / (NQ * 4), q = (t >> 2) & (NQ - 1), j = t & 3;
        const int jr = 4 * g + j < nyk ? 4 * g + j : 0;
        const float *hr = Hy + ((size_t)vy * nyk + jr) * wa.TY + PH * h + 8 * q;
        hy_b[((g * NQ + q) * 4 + j) * 2 + 0] = *reinterpret_cast<const F4 *>(hr);
        hy_b[((g * NQ + q) * 4 + j) * 2 + 1] = *reinterpret_cast<const F4 *>(hr + 4);
    }
    // z: the tile's impulse responses -> LDS (one 4-byte read per lane and plane)
    if (t < TZ) reinterpret_cast<F4 *>(hz_lds)[t] = *reinterpret_cast<const F4 *>(HzT + ((size_t)vz * TZ + t) * 4);

    constexpr int D = 2;                                             // planes of a thread's loads in flight (four: no faster)
    F4 pre[D][2];
#pragma unroll
    for (int d = 0; d < D; d++) {
        // (plane by plane, as the loop requests them: requested row by row instead -- the compiler's order without the fence --
        //  the first plane is complete only when three of the first four loads are, and the loop's wait for `its` plane became
        //  `s_waitcnt vmcnt(1)`: half of the next plane as well, every step)
        pre[d][0] = ld(spb + d * plane_bytes, off0, ok0);
        pre[d][1] = ld(spb + d * plane_bytes, off1, ok1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // z accumulators: [row rg / rg + PH / 2][element e of the lane's chunk]; lane 4 b + j, register i: tail j of column 16 b + 4 i + e
    F4 zacc[2][4];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int e = 0; e < 4; e++) zacc[k][e] = F4{0.f, 0.f, 0.f, 0.f};
    const float *hzp = hz_lds + j4;

    // Plane z of the tile complete in the two stages: x tails -> xt (wave 8), this patch's part of the combined rows -> ytp
    // (waves 0 .. nyk - 1).  Destinations: a wave-uniform base that advances by a fixed number of elements per plane + one
    // 32-bit offset per flushing lane (the tails of a volume that fits the memory stay below 4 GiB; launch_walk_tails checks).
    const int64_t ystride = (int64_t)a.MX * a.ny * K * kFusedTX, xstride = a.NYP;          // (y tails tile-major)
    const int64_t zg0 = (int64_t)tz * TZ;
    float *const ybase = wa.ytp + (int64_t)h * wa.part_stride + a.yt_index(0, ty, 0, K, (int64_t)tx * kFusedTX + a.NXP * zg0);
    float *const xbase = (xh ? wa.xt2 : a.xt) + ((int64_t)tx * K) * Lx + (int64_t)ty * wa.TY + PH * h + a.NYP * zg0;
    // (the x tails of a plane are nxk * 8 chunks of four rows: one flushing wave per 32 of them -- wave 8, and wave 9 for the
    //  six tails of an order-3 pair; TALL: nxk * 16 chunks, one per lane of wave 8)
    constexpr int kXChunks = TALL ? nxk * 16 : nxk * (kRows / 4), kXWaves = TALL ? 1 : (kXChunks + 31) / 32;
    static_assert(nyk <= 8 && kXWaves <= 2, "flushing waves: y parts on waves 0 .. nyk - 1, x tails on waves 8 and 9");
    const int xu = TALL ? (t & 63) : 32 * (w - 8) + (t & 31);                               // this lane's chunk of the x tails (waves 8, 9)
    const bool yflush = TALL ? t < nyk * CPR : w < nyk;                                     // (TALL: nyk rows of 32 chunks -- waves 0 and 1, or half of wave 0)
    uint32_t foff = 0;
    if (yflush) foff = TALL ? (uint32_t)((t / CPR) * kFusedTX + PW * xh + 4 * (t & (CPR - 1))) : (uint32_t)(t * 4);      // [jr][256]: jr = t / CPR, chunk t % CPR
    else if (w >= 8 && w < 8 + kXWaves) {
        const int u = xu < kXChunks ? xu : 0, sr = TALL ? u >> 4 : u >> 3, q = TALL ? u & 15 : u & 7;
        foff = (uint32_t)((((int64_t)(sr / K) * a.MX * K + sr % K) * Lx + 4 * q));
    }
    auto flush = [&](int zl, int par) {                             // zl: the plane, counted inside the z tile; par: its parity
        if (yflush) {
            const F4 *ys = ystage4 + par * kYs4;
            const int jr = t / CPR, c4 = t & (CPR - 1);
            F4 v = ys[(0 * nyk + jr) * CPR + c4];
#pragma unroll
            for (int q = 1; q < NQ; q++) v = v + ys[(q * nyk + jr) * CPR + c4];
            *reinterpret_cast<F4 *>(ybase + (int64_t)zl * ystride + foff) = v;
        } else if (TALL && w == 8) {
            // lane u = (tail sr, four-row group q of the 64 rows): the eight waves of its row half hold the partial sums
            const F4 *xs = stage4 + par * kXs4;
            const int u = xu < kXChunks ? xu : 0, sr = u >> 4, q = u & 15;
            const int first = ((8 * (q >> 3)) * nxk + sr) * 8 + (q & 7);
            F4 v = xs[first];
#pragma unroll
            for (int p = 1; p < 8; p++) v = v + xs[first + p * nxk * 8];
            if (xu < kXChunks) *reinterpret_cast<F4 *>(xbase + (int64_t)zl * xstride + foff) = v;
        } else if (!TALL && w >= 8 && w < 8 + kXWaves) {
            // lanes u and u + 32 each add up eight of the sixteen waves' partial sums, the halves meet across the wave
            const F4 *xs = stage4 + par * kXs4;
            const int u = xu < kXChunks ? xu : 0, half8 = (t >> 5) & 1;
            F4 v = xs[(8 * half8) * kXChunks + u];
#pragma unroll
            for (int p = 1; p < 8; p++) v = v + xs[(8 * half8 + p) * kXChunks + u];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = sum_lanes_xor_32(v[i]);
            if ((t & 63) < 32 && xu < kXChunks) *reinterpret_cast<F4 *>(xbase + (int64_t)zl * xstride + foff) = v;
        }
    };

    // plane zn (in the registers p0, p1) -> patch buffer `par` + its z products; `between`: what the caller wants issued behind
    // the wait for the plane's pixels and in front of the next request (the flush: the flushing waves' store is then never the
    // youngest operation their next wait has to count); LOAD: request plane zn + 2 into the same registers
    auto put = [&](int zn, int par, F4 &p0, F4 &p1, auto load_tag, auto between) {
        constexpr bool LOAD = decltype(load_tag)::value;
        F4 v0 = p0, v1 = p1;
        if (a.pw_flags & 1) {                                      // fused prologue x' = pre_s * in + pre_b
            v0 = v0 * a.pre_s + a.pre_b;
            v1 = v1 * a.pre_s + a.pre_b;
        }
        if constexpr (EDGE) {                                      // samples beyond the image stay zero: they do not exist
            if constexpr (U4) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (!(ok0 && okE[e])) v0[e] = 0.f;
                    if (!(ok1 && okE[e])) v1[e] = 0.f;
                }
            } else {
                if (!ok0) v0 = F4{0.f, 0.f, 0.f, 0.f};
                if (!ok1) v1 = F4{0.f, 0.f, 0.f, 0.f};
            }
        }
        F4 *tl = tile4 + par * kTile4;
        tl[rg * P4 + cc] = v0;
        tl[(rg + PH / 2) * P4 + cc] = v1;
        between();
        // z tails: block = four adjacent chunks of the row; A = the lane's own sample, B = Hz[tail j4][zn]
        const float hz = hzp[4 * zn];                              // (zero for a tail that does not exist: the table is padded)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            zacc[0][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(v0[e], hz, zacc[0][e], 0, 0, 0);
            zacc[1][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(v1[e], hz, zacc[1][e], 0, 0, 0);
        }
        if constexpr (LOAD) {
            const char *lp = spb + (int64_t)(zn + 2) * plane_bytes;       // (wave-uniform: scalar base + lane offset)
            p0 = ld(lp, off0, ok0);
            p1 = ld(lp, off1, ok1);
        }
    };

    // the x and y tails of the plane in patch buffer `par` -> the stages of that parity
    auto contract = [&](int par) {
        const F4 *tl = tile4 + par * kTile4;
        const float *tlf = reinterpret_cast<const float *>(tl);
        // ---- x tails: 32 rows x this wave's 16 columns ----
        {
            F4 *xs = stage4 + par * kXs4;
            F4 av[2];
#pragma unroll
            for (int mi = 0; mi < 2; mi++) av[mi] = tl[xrow * P4 + 4 * xcg + 2 * xhf + mi];
#pragma unroll
            for (int g = 0; g < NGX; g++) {
                const F4 b0 = hx_b[(((g * 16 + w) * 2 + xhf) * 4 + j4) * 2], b1 = hx_b[(((g * 16 + w) * 2 + xhf) * 4 + j4) * 2 + 1];
                F4 acc0 = F4{0.f, 0.f, 0.f, 0.f}, acc1 = F4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0][e], b0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[1][e], b1[e], acc1, 0, 0, 0);
                }
                F4 dsum = acc0 + acc1;
#pragma unroll
                for (int i = 0; i < 4; i++) dsum[i] = sum_lanes_xor_32(dsum[i]);       // the two column halves
                // lane 4 rg' + j (lanes 0..31), register i: row 4 rg' + i of tail 4 g + j, this wave's sixteen columns
                const int sr = 4 * g + j4;
                if (xhf == 0 && sr < nxk) xs[(w * nxk + sr) * 8 + (lane >> 2)] = dsum;
            }
        }
        // ---- y tails: eight rows of column yc; lane 4 b + j ends up with tail j of the block's four columns ----
        {
            F4 *ys = ystage4 + par * kYs4;
            float col[8];
#pragma unroll
            for (int i = 0; i < 8; i++) col[i] = tlf[(8 * yq + i) * (P4 * 4) + yc];
#pragma unroll
            for (int g = 0; g < NGY; g++) {
                const F4 b0 = hy_b[((g * NQ + yq) * 4 + j4) * 2], b1 = hy_b[((g * NQ + yq) * 4 + j4) * 2 + 1];
                F4 acc0 = F4{0.f, 0.f, 0.f, 0.f}, acc1 = F4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(col[i], b0[i], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(col[4 + i], b1[i], acc1, 0, 0, 0);
                }
                const int jr = 4 * g + j4;
                if (jr < nyk) ys[(yq * nyk + jr) * CPR + (yc >> 2)] = acc0 + acc1;
            }
        }
    };

    // One step = one plane z (parity par): contract it out of its patch buffer into the stages of its parity; flush plane z - 1
    // out of the other stages; put plane z + 1 into the other patch buffer.  Everything a step writes was last read in the
    // step before: one barrier per step.
    auto step = [&](int z, int par, F4 &p0, F4 &p1, auto write_tag, auto load_tag) {
        contract(par);
        if constexpr (decltype(write_tag)::value) put(z + 1, par ^ 1, p0, p1, load_tag, [&] { if (z > 0) flush(z - 1, par ^ 1); });
        else if (z > 0) flush(z - 1, par ^ 1);
        __syncthreads();
    };
    const std::true_type yes{};
    const std::false_type no{};

    __syncthreads();                                                 // Hz and the B operands are in LDS
    put(0, 0, pre[0][0], pre[0][1], yes, [] {});
    __syncthreads();                                                 // plane 0
    // planes 0 .. TZ - 5 request a further plane; TZ is a multiple of 32
#pragma unroll 1
    for (int z0 = 0; z0 < TZ - 4; z0 += 2) {
        step(z0, 0, pre[1][0], pre[1][1], yes, yes);
        step(z0 + 1, 1, pre[0][0], pre[0][1], yes, yes);
    }
    step(TZ - 4, 0, pre[1][0], pre[1][1], yes, yes);                 // puts plane TZ - 3, requests plane TZ - 1
    step(TZ - 3, 1, pre[0][0], pre[0][1], yes, no);                  // puts plane TZ - 2
    step(TZ - 2, 0, pre[1][0], pre[1][1], yes, no);                  // puts plane TZ - 1
    step(TZ - 1, 1, pre[0][0], pre[0][1], no, no);
    flush(TZ - 1, 1);

    // z tails of the patch: [s][tz][r][line], line = y * NX + x (StridedArgs::tails); lane 4 b + j stores tail j of the block's
    // sixteen columns
    if (j4 < wa.nzk) {
        const int64_t half = (PH / 2) * a.NX;
        const int cblk = (lane >> 2) & (CPR / 4 - 1);                        // the block's sixteen columns inside the patch row
        const int64_t line = ((int64_t)ty * wa.TY + PH * h + rg) * a.NX + (int64_t)tx * kFusedTX + PW * xh + 16 * cblk;
        float *q = wa.zt + ((((int64_t)(j4 / wa.KZ)) * wa.MZ + tz) * wa.KZ + j4 % wa.KZ) * plane + line;
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // (the z tails are [..][NX * NY], unpadded: nothing of a row or column that does not exist is stored)
                if constexpr (U4) {              // element-aligned rows: sample by sample
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (!EDGE || (rg + (PH / 2) * k < rows_left && 16 * cblk + 4 * i + e < cols_here)) q[k * half + 4 * i + e] = zacc[k][e][i];
                    continue;
                }
                if (EDGE && !(rg + (PH / 2) * k < rows_left && 16 * cblk + 4 * i < cols_here)) continue;
                *reinterpret_cast<F4 *>(q + k * half + 4 * i) = F4{zacc[k][0][i], zacc[k][1][i], zacc[k][2][i], zacc[k][3][i]};
            }
    }
}

template <int K, int NX, int NY, bool TALL = false, bool U4 = false>
__global__ void __launch_bounds__(kWalkThreads)
walk_tails_kernel(const float *__restrict__ src, FusedArgs<float> a, WalkArgs wa, const float *__restrict__ Hx, const float *__restrict__ Hy,
                  const float *__restrict__ HzT) {
    constexpr int PH = TALL ? kTallRows : kRows, PW = TALL ? kTallCols : kFusedTX;
    const int ty = blockIdx.y >> wa.parts_log2, h = blockIdx.y & ((1 << wa.parts_log2) - 1);
    const int tx = TALL ? (int)blockIdx.x >> 1 : (int)blockIdx.x, xh = TALL ? (int)blockIdx.x & 1 : 0;
    const bool whole = (tx != a.MX - 1 || a.last_cols - PW * xh >= PW) && (ty != a.MY - 1 || a.last_rows - PH * h >= PH);
    if (whole) walk_tails_body<K, NX, NY, false, TALL, U4>(src, a, wa, Hx, Hy, HzT);
    else walk_tails_body<K, NX, NY, true, TALL, U4>(src, a, wa, Hx, Hy, HzT);
}

}  // namespace

// When pass 1 of a 3-D plan walks: f32 volumes whose depth is whole z tiles (z slabs too) and whose width is a multiple of 4 (the last
// tile of a row and the last tile row may be partial), x, y and z scans all present, orders <= 3 along x / y and <= 2 along z,
// at most two scans per dimension.
bool walk_tails_applicable(int K, int TY, int nx, int ny, int nz, int KZ, int TZ, int last_cols, int last_rows) {
    if (K < 1 || K > 3 || KZ < 1 || KZ > 2) return false;          // (order 3 along x / y; the four z accumulators per sample stay)
    if (nx < 1 || nx > 2 || ny < 1 || ny > 2 || nz < 1 || nz > 2) return false;
    if (TY != 32 && TY != 64 && TY != 128) return false;
    if (TZ != 32 && TZ != 64 && TZ != 128) return false;
    if (last_cols < 1 || last_cols > kFusedTX || last_rows < 1 || last_rows > TY) return false;      // (any width since round 6: U4)
    return true;
}

int launch_walk_tails(int K, const float *src, const FusedArgs<float> &a, const WalkArgs &wa, const float *Hx, const float *Hy,
                      hipStream_t stream) {
    if (a.MX <= 0 || a.MY <= 0 || wa.MZ <= 0) return RF_OK;
    const bool tall = wa.tall != 0;
    const int parts = wa.TY / (tall ? kTallRows : kRows);
    const int nxk = a.nx * K, nyk = a.ny * K;
    if (tall && (wa.TY != 2 * kTallRows || nxk > 4 || K > 2 || wa.xt2 == nullptr)) { set_error("walk tails: tall patches need 128-row tiles, orders <= 2, at most four x tails and a second x part"); return RF_ERR_INVALID_ARG; }
    if ((int64_t)a.MY * parts > 65535 || wa.MZ > 65535 || (int64_t)a.MX * 2 > 65535) { set_error("walk tails: grid too large"); return RF_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)(tall ? 2 * a.MX : a.MX), (unsigned)(a.MY * parts), (unsigned)wa.MZ);
    // [patch x 2][x stage x 2][y stage x 2][Hz][Hx operands][Hy operands] (walk_tails_body)
    const size_t lds = ((size_t)2 * (tall ? kTallRows * kTallPitch4 : kRows * kPitch4) * 4 + (size_t)2 * 16 * nxk * kRows + (size_t)2 * 4 * nyk * kFusedTX + (size_t)4 * 128 +
                        (size_t)((nxk + 3) / 4) * 1024 + (size_t)((nyk + 3) / 4) * (tall ? 256 : 128)) * sizeof(float);
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    const bool u4 = a.NX % 4 != 0;                   // rows that are only element-aligned
#define RF_CASE_TU(KK, XX, YY, TT, UU)                                                                                     \
    if (K == KK && a.nx == XX && a.ny == YY && tall == TT && u4 == UU) {                                                   \
        auto kern = walk_tails_kernel<KK, XX, YY, TT, UU>;                                                                 \
        static std::atomic<bool> opted[64];                                                                                \
        std::atomic<bool> &done = opted[dev & 63];                                                                         \
        if (!done.load(std::memory_order_acquire)) {        /* more than 64 KiB of dynamic LDS: opt in, once per kernel and device */ \
            RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            done.store(true, std::memory_order_release);                                                                   \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(kWalkThreads), lds, stream, src, a, wa, Hx, Hy, wa.HzT);                        \
        RF_HIP_CHECK(hipGetLastError());                                                                                   \
        return RF_OK;                                                                                                      \
    }
#define RF_CASE_T(KK, XX, YY, TT) RF_CASE_TU(KK, XX, YY, TT, false) RF_CASE_TU(KK, XX, YY, TT, true)
#define RF_CASE(KK, XX, YY) RF_CASE_T(KK, XX, YY, false)
    RF_CASE(2, 2, 2) RF_CASE(2, 1, 1) RF_CASE(2, 2, 1) RF_CASE(2, 1, 2) RF_CASE(1, 2, 2) RF_CASE(1, 1, 1) RF_CASE(1, 2, 1) RF_CASE(1, 1, 2)
    RF_CASE(3, 2, 2) RF_CASE(3, 1, 1) RF_CASE(3, 2, 1) RF_CASE(3, 1, 2)
    RF_CASE_T(2, 2, 2, true) RF_CASE_T(2, 1, 1, true) RF_CASE_T(2, 2, 1, true) RF_CASE_T(2, 1, 2, true)
    RF_CASE_T(1, 2, 2, true) RF_CASE_T(1, 1, 1, true) RF_CASE_T(1, 2, 1, true) RF_CASE_T(1, 1, 2, true)
#undef RF_CASE
#undef RF_CASE_T
#undef RF_CASE_TU
    set_error("walk tails: unsupported order %d / %d scans / %d z tails", K, a.nx, wa.nzk);
    return RF_ERR_UNSUPPORTED;
}

}  // namespace rf
