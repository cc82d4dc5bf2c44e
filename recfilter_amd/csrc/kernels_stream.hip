// kernels_stream.hip -- pass 1 (tail extraction, kernels_tails.hip's contraction) as a STREAMING kernel:
// persistent workgroups that walk the tiles, fed by a ring of LDS slots that a loader wave fills with LDS-DMA.
//
// Why: fused_tails_kernel is latency-bound (profiles/r1/pmc_shader.json: its waves wait in 69 % of their cycles).
// Every tile byte travels HBM -> VGPR -> ds_write -> barrier, so the bytes a workgroup keeps in flight are capped
// by its staging registers, and every workgroup starts with an empty pipe.  Here
//   * a workgroup is 8 compute waves (4 for the x tails, 4 for the y tails of the same slot) + 1 LOADER wave.  The loader issues `global_load_lds_dwordx4` (1 KiB per wave
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
constexpr int kHyPitch = kStreamTY + 4;             // floats per row of the Hy table in LDS (padded: bank spread)
constexpr int kStreamThreads = 576;                 // waves 0..3: x tails, waves 4..7: y tails, wave 8: loader


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

typedef float F4s __attribute__((ext_vector_type(4)));

// Where chunk c (16 bytes) of row i sits inside a 1 KiB slot row: the low four bits of the chunk index are XORed with
// the row (the 16 rows a matrix-core A operand gathers at one column then sit on 16 different bank groups) and with
// bits 4..5 of the chunk (segments 256 B apart otherwise share banks).  Both reads of the kernel are conflict-free
// under it: x role, lane = (row i, chunk 4m + kk); y role, lane = (row ks + 4 kk, 16 consecutive columns).
__device__ __forceinline__ int slot_pos(int i, int c) { return c ^ (i & 15) ^ ((c >> 4) & 3); }

// SLOTS ring slots, AHEAD (< SLOTS) of them in flight ahead of the slot being contracted.
//
// The contractions run on the matrix cores, which leaves the vector ALU almost idle.  With the dot products on the
// VALU (16 + 16 packed FMAs per thread and slot plus DPP reductions) the compute waves alone needed 0.16-0.18 ms of
// the loader's 0.19 ms; v_mfma_f32_16x16x4_f32 was no better (only nx*K = 4 of its 16 output columns are useful,
// the matrix pipe itself became the bound).  The instruction that fits is v_mfma_f32_4x4x1_16b_f32: sixteen
// independent 4 x 4 outer products per issue (8 cycles), D[b][i][j] += A[b][i] * B[b][j] with block b = lane / 4,
// i.e. a rank-4 "tails" dimension at full efficiency (exact f32: a chain of fma's).
//   x role (waves 0..3, wave w owns columns 64w..64w+63 of every row):
//       block = (column phase cg = lane/16, row group rg): A[b][i] = pixel (row 4rg + i, column 4(16w + 4m + cg) + e),
//       B[b][j] = Hx[sr = j][that column] (16 registers per group of four tails, loaded when the walker's tile column
//       changes: never, while the stride is a multiple of the tiles per row).  One ds_read_b128 feeds four MFMAs.  The
//       four column phases are added with two cross-lane steps, the four waves' partial sums meet in the LDS stage.
//   y role (waves 4..7, wave w owns the same columns):
//       block = group of four columns, one MFMA per slot row: A[b][i] = pixel (row, column 64w + 4b + i) = lane-linear
//       ds_read_b32, B[b][j] = Hy[jr = j][row] (read from an LDS copy of the table, 16 bytes per 4 rows); the
//       accumulator D[b][i][j] = combined row jr at column 64w + 4b + i lives across the tile's four slots and needs
//       no reduction at all.
// NGX / NGY: groups of four tails along x / y (compile time: they size the register arrays); K is a run-time value here
// Register budget: two 9-wave workgroups per CU need a SIMD with three free slots for the second one; at five waves per
// SIMD (96 registers) it often finds none and the kernel runs with one workgroup per CU (a build of this kernel at 95
// registers took 0.27 ms against 0.205 at 68).  Six waves per SIMD (80 registers) keep both resident.
template <int NGX, int NGY, int SLOTS, int AHEAD>
__global__ void __launch_bounds__(kStreamThreads, 6)
stream_tails_kernel(const float *__restrict__ src, FusedArgs<float> a,
                    const float *__restrict__ Hx,     // [vx][s][r][256]
                    const float *__restrict__ Hy,     // [vy][j][r][64]
                    uint32_t n_tiles, int K) {
    static_assert(AHEAD >= 1 && AHEAD < SLOTS && 16 * AHEAD <= 63, "ring geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ring: SLOTS x 16 KiB][x-tail stage: 2 tiles x 4 waves x nx*K x 64][Hy table: 4 variants x ny*K x 64]
    const int nxk = a.nx * K, nyk = a.ny * K;
    float *stage = reinterpret_cast<float *>(smem + SLOTS * kSlotBytes);
    const int stage_tile = 4 * (nxk > 0 ? nxk : 1) * kStreamTY;       // floats per tile in the stage
    float *hy_lds = stage + 2 * stage_tile;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const uint32_t G = gridDim.x;
    const uint32_t first = blockIdx.x;
    const int n_my = first < n_tiles ? (int)((n_tiles - first + G - 1) / G) : 0;     // tiles this workgroup walks
    const int nq = n_my * kQPerTile;
    const uint32_t row_bytes = a.row_bytes;
    auto variant_x = [&](int tx) { return (tx == 0 ? 1 : 0) | (tx == a.MX - 1 ? 2 : 0); };

    if (wave == 8) {
        // ------------------------------------------------ loader wave ------------------------------------------------
        auto issue = [&](int g) {
            const TileCoord c = tile_coord(a, first + (uint32_t)(g / kQPerTile) * G);
            const float *plane = a.plane_batch ? reinterpret_cast<const float *>(a.in_planes[c.z]) : src + c.z * a.NX * a.NY;
            const char *base = reinterpret_cast<const char *>(plane + (int64_t)c.ty * kStreamTY * a.NX + (int64_t)c.tx * kFusedTX) +
                               (size_t)(g % kQPerTile) * kQRows * row_bytes;
            unsigned char *slot = smem + (g % SLOTS) * kSlotBytes;
#pragma unroll
            for (int r = 0; r < kQRows; r++) {
                // LDS position `lane` of row r receives the chunk that slot_pos maps there (an involution per row)
                const uint32_t lane_off = (uint32_t)slot_pos(r, lane) * 16u;
                // each sample is read once by this pass: non-temporal (leaves L2 / the memory-side cache to the tails)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(base + ((uint32_t)r * row_bytes + lane_off)),
                    (__attribute__((address_space(3))) void *)(slot + r * (kFusedTX * 4)), 16, 0, /*nt*/ 2);
            }
        };
        for (int g = 0; g < AHEAD && g < nq; g++) issue(g);
        for (int g = 0; g < nq; g++) {
            // slot g has landed once at most min(AHEAD-1, nq-1-g) younger slots (16 pieces each) are outstanding
            const int younger = nq - 1 - g;
            if (younger >= AHEAD - 1) wait_vmcnt<16 * (AHEAD - 1)>();
            else if (AHEAD >= 3 && younger == 1) wait_vmcnt<16>();
            else wait_vmcnt<0>();
            lds_barrier();                                          // B_g
            if (g + AHEAD < nq) issue(g + AHEAD);                   // refills the slot contracted before B_g
        }
        lds_barrier();                                              // final: x-tail stage of the last tile
        return;
    }

    // Hy table -> LDS (all four border variants; ordered before its first use by B_0).  Rows are padded by four floats:
    // the four tails of a group are read side by side (lanes 4b + j, 16 bytes each), 256-byte rows would put them on
    // the same banks (4-way conflict on every read: profiles/r2/pmc_shader.json, 34 % of this kernel's LDS cycles).
    for (int i = tid; i < 4 * nyk * kStreamTY; i += 512) hy_lds[(i / kStreamTY) * kHyPitch + i % kStreamTY] = Hy[i];

    const int64_t Lx = a.NYP * a.NZ;
    const int w = wave & 3;                                          // column group 64w .. 64w+63

    if (wave < 4) {
        // ------------------------------------------------ x role ----------------------------------------------------
        const int t = tid;                                           // 0..255
        const int row = lane & 15, cg = lane >> 4, j4 = lane & 3;    // A: row of the slot, column phase; B: tail in group
        float Bx[NGX][16];                                           // Hx[sr = 4 gx + j4][4 (16w + 4m + cg) + e] at [gx][4m + e]
        auto flush_xtails = [&](int n) {       // tile n's x tails: sum of the four waves' partials -> xt, 16 B per lane
            if (nxk > 0 && t < nxk * 16) {
                const TileCoord c = tile_coord(a, first + (uint32_t)n * G);
                const int sr = t >> 4, j = t & 15;
                const F4s *sp = reinterpret_cast<const F4s *>(stage + (size_t)(n & 1) * stage_tile + (size_t)sr * kStreamTY) + j;
                F4s v = sp[0];
#pragma unroll
                for (int ww = 1; ww < 4; ww++) v = v + sp[(size_t)ww * nxk * (kStreamTY / 4)];
                const int s = sr / K, r = sr % K;
                const int64_t line0 = (int64_t)c.ty * kStreamTY + a.NYP * c.z;
                *reinterpret_cast<F4s *>(a.xt + (((int64_t)s * a.MX + c.tx) * K + r) * Lx + line0 + 4 * j) = v;
            }
        };
        // byte offsets of this lane's four A reads inside a slot: row `row`, chunk 16w + 4m + cg
        uint32_t a_off[4];
#pragma unroll
        for (int m = 0; m < 4; m++) a_off[m] = (uint32_t)row * 1024u + (uint32_t)slot_pos(row, 16 * w + 4 * m + cg) * 16u;
        // The stride G is a multiple of the tiles per row (stream_grid_size), so this walker never leaves its tile
        // column: one border variant, its Hx fragments loaded once.  (Loaded inside the tile loop the compiler joins the
        // branch with `s_waitcnt vmcnt(0)`, which also waits for the previous tile's x-tail STORE: a store latency per
        // tile on the critical path of every wave of the workgroup.)
        {
            const int vx = variant_x((int)(first % (uint32_t)a.MX));
#pragma unroll
            for (int gx = 0; gx < NGX; gx++)
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int sr = 4 * gx + j4;
                        Bx[gx][4 * m + e] = (n_my > 0 && sr < nxk) ? Hx[((size_t)vx * nxk + sr) * kFusedTX + 4 * (16 * w + 4 * m + cg) + e] : 0.0f;
                    }
        }
        for (int n = 0; n < n_my; n++) {
            float *stage_w = stage + (size_t)(n & 1) * stage_tile + (size_t)w * nxk * kStreamTY;
#pragma unroll 1
            for (int q = 0; q < kQPerTile; q++) {
                const int g = n * kQPerTile + q;
                lds_barrier();                                          // B_g: slot g landed, slot g-1 free
                if (q == 0 && n > 0) flush_xtails(n - 1);               // every wave's stage writes of tile n-1 are behind B_g
                if (nxk == 0) continue;
                const unsigned char *slot = smem + (g % SLOTS) * kSlotBytes;
                F4s av[4];
#pragma unroll
                for (int m = 0; m < 4; m++) av[m] = *reinterpret_cast<const F4s *>(slot + a_off[m]);
#pragma unroll
                for (int gx = 0; gx < NGX; gx++) {
                    {
                        // four accumulators: consecutive MFMAs do not wait for each other
                        F4s acc[4];
#pragma unroll
                        for (int m = 0; m < 4; m++) acc[m] = F4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m = 0; m < 4; m++) {
                            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m].x, Bx[gx][4 * m + 0], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m].y, Bx[gx][4 * m + 1], acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m].z, Bx[gx][4 * m + 2], acc[2], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m].w, Bx[gx][4 * m + 3], acc[3], 0, 0, 0);
                        }
                        F4s d = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                        // add the four column phases: lanes l, l^16, l^32, l^48
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            d[i] = d[i] + __shfl_xor(d[i], 16);
                            d[i] = d[i] + __shfl_xor(d[i], 32);
                        }
                        // lane 4 rg + j (cg == 0), register i: row 4 rg + i of tail sr = 4 gx + j
                        const int sr = 4 * gx + j4;
                        if (cg == 0 && sr < nxk)
                            *reinterpret_cast<F4s *>(stage_w + sr * kStreamTY + q * kQRows + (row & 12)) = d;
                    }
                }
            }
        }
        lds_barrier();                                                  // final: the last tile's stage is complete
        if (n_my > 0) flush_xtails(n_my - 1);
        return;
    }

    // ------------------------------------------------ y role ----------------------------------------------------
    // this lane's pixel of slot row r: column 64w + lane -> chunk 16w + lane/4, element lane%4
    const int j4 = lane & 3;
    const uint32_t px_base = (uint32_t)((16 * w + (lane >> 2)) ^ (w & 3)) * 16u + (uint32_t)(lane & 3) * 4u;
    for (int n = 0; n < n_my; n++) {
        const TileCoord c = tile_coord(a, first + (uint32_t)n * G);
        const int vy = ((c.ty == 0 && a.y_first_border) ? 1 : 0) | ((c.ty == a.MY - 1 && a.y_last_border) ? 2 : 0);
        F4s acc[NGY];
#pragma unroll
        for (int gy = 0; gy < NGY; gy++) acc[gy] = F4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < kQPerTile; q++) {
            const int g = n * kQPerTile + q;
            lds_barrier();                                          // B_g
            if (nyk == 0) continue;
            const unsigned char *slot = smem + (g % SLOTS) * kSlotBytes;
            float px[kQRows];
#pragma unroll
            for (int r = 0; r < kQRows; r++)
                px[r] = *reinterpret_cast<const float *>(slot + ((px_base ^ ((uint32_t)r << 4)) + (uint32_t)r * 1024u));
#pragma unroll
            for (int gy = 0; gy < NGY; gy++) {
                {
                    const int jr = 4 * gy + j4;
                    const float *hp = hy_lds + (size_t)(vy * nyk + (jr < nyk ? jr : 0)) * kHyPitch + q * kQRows;
                    const float keep = jr < nyk ? 1.0f : 0.0f;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const F4s h = *reinterpret_cast<const F4s *>(hp + 4 * m) * keep;
                        acc[gy] = __builtin_amdgcn_mfma_f32_4x4x1f32(px[4 * m + 0], h.x, acc[gy], 0, 0, 0);
                        acc[gy] = __builtin_amdgcn_mfma_f32_4x4x1f32(px[4 * m + 1], h.y, acc[gy], 0, 0, 0);
                        acc[gy] = __builtin_amdgcn_mfma_f32_4x4x1f32(px[4 * m + 2], h.z, acc[gy], 0, 0, 0);
                        acc[gy] = __builtin_amdgcn_mfma_f32_4x4x1f32(px[4 * m + 3], h.w, acc[gy], 0, 0, 0);
                    }
                }
            }
        }
        // lane 4b + j, register i: combined row jr = 4 gy + j at column 64w + 4b + i -> yt, 16 bytes per lane (with x
        // scans in the filter xscan_rows_kernel finishes them in place)
#pragma unroll
        for (int gy = 0; gy < NGY; gy++) {
            const int jr = 4 * gy + j4;
            if (jr < nyk)
                *reinterpret_cast<F4s *>(a.yt + a.yt_index(jr / K, c.ty, jr % K, K,
                                                          a.NXP * c.z + (int64_t)c.tx * kFusedTX + 64 * w + (lane & ~3))) = acc[gy];
        }
    }
    lds_barrier();                                                  // final (the x waves flush the last stage behind it)
}

// Workgroups to launch: two per CU (LDS: two rings), rounded down to a multiple of the tiles per row so that a walker
// stays in one tile column (one Hx border variant for life); 0 when the image is too wide for that.
int stream_grid_size(int64_t n_tiles, int MX) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    int64_t g = 2ll * cus;
    if (const char *env = RF_KNOB("RF_STREAM_GRID")) g = atoll(env);
    if (g >= n_tiles) return (int)n_tiles;        // one tile per walker
    if (g < MX) return 0;
    g -= g % MX;
    return (int)g;
}

}  // namespace

// When pass 1 takes the streaming kernel.  Measured (profiles/r2/pass1_stream_vs_staged.txt): the LDS-DMA ring streams
// at 5.2-5.8 TB/s whatever the size, the register-staged fused_tails_kernel at 5.0 TB/s on a single 16384^2 plane
// (launch ramp and tail of its 16384 short workgroups) but at 5.9-6.0 TB/s once a launch is three times as long
// (3 planes, volumes).  So: single planes with at most four tails per dimension stream, everything else is staged.
bool stream_tails_applicable(int K, int TY, bool src_u8, int pw_flags, int last_cols, int last_rows, int64_t n_tiles, int MX,
                             int64_t NZ, int nxk, int nyk, int mode) {
    // mode: rf_filter_desc.flags -- RF_PLAN_STREAM_PASS1 (+1: wherever the shape rules allow, whatever the size; the tests
    // cover every shape class that way), RF_PLAN_STAGED_PASS1 (-1: never), 0: single planes of at least 2048 tiles
    static const bool off = RF_KNOB("RF_NO_STREAM_TAILS") != nullptr;      // A/B runs against fused_tails_kernel
    if (mode < 0) return false;
    const bool force = mode > 0 || RF_KNOB("RF_STREAM_FORCE") != nullptr;
    if (off || src_u8 || (pw_flags & 1) || TY != kStreamTY || last_cols != kFusedTX || last_rows != TY) return false;
    if (K < 1 || K > 3) return false;
    if (!force && (NZ != 1 || nxk > 4 || nyk > 4)) return false;
    // Round 4 (profiles/r4/ab_pass1_three_way.txt, 64-row tiles, same box): the register-staged kernels have caught up -- order
    // 2 on 16384^2: this kernel 194 us, mfma_tails_kernel 192, fused_tails_kernel 202; order 1 on 8192^2: 48.9 / 48.5 / 47.9 -- so
    // the automatic choice no longer takes this kernel; RF_PLAN_STREAM_PASS1 (and the A/B knob) still does.
    if (!force) return false;
    const char *mt = RF_KNOB("RF_STREAM_MIN_TILES");              // (read per call: the tests lower it to cover small shapes)
    const int64_t min_tiles = mode > 0 ? 1 : mt ? atoll(mt) : 2048;
    if (n_tiles < min_tiles) return false;      // below that a walker has too few tiles to amortise its pipeline fill
    return stream_grid_size(n_tiles, MX) > 0;
}

int launch_stream_tails(int K, const float *src, const FusedArgs<float> &a, const float *Hx, const float *Hy, hipStream_t stream) {
    const int64_t n_tiles = (int64_t)a.MX * a.MY * a.NZ;
    if (n_tiles <= 0) return RF_OK;
    if (n_tiles >= (1ll << 31)) { set_error("stream tails: too many tiles"); return RF_ERR_UNSUPPORTED; }
    const int grid = stream_grid_size(n_tiles, a.MX);
    if (grid <= 0) { set_error("stream tails: no device properties"); return RF_ERR_HIP; }
    static const int ring = RF_KNOB("RF_STREAM_RING") ? atoi(RF_KNOB("RF_STREAM_RING")) : 4;           // ring slots (tuning)
    const int nxk = a.nx * K;
    const size_t lds_rest = (size_t)2 * 4 * (nxk > 0 ? nxk : 1) * kStreamTY * sizeof(float) +
                            (size_t)4 * (a.ny * K > 0 ? a.ny * K : 1) * kHyPitch * sizeof(float);
    const int ngx = nxk > 0 ? (nxk + 3) / 4 : 1, ngy = a.ny * K > 0 ? (a.ny * K + 3) / 4 : 1;
#define RF_CASE(GX, GY, SLOTS, AHEAD)                                                                                    \
    if (ngx == GX && ngy == GY && ring == SLOTS) {                                                                       \
        auto kern = &stream_tails_kernel<GX, GY, SLOTS, AHEAD>;                                                          \
        static std::atomic<bool> attr_set[64];                                                                           \
        int dev = 0;                                                                                                     \
        RF_HIP_CHECK(hipGetDevice(&dev));                                                                                \
        if (!attr_set[dev & 63].load(std::memory_order_acquire)) {                                                       \
            RF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)(SLOTS * kSlotBytes + (8 * 4 * GX + 4 * 4 * GY) * kHyPitch * sizeof(float)))); \
            attr_set[dev & 63].store(true, std::memory_order_release);                                                   \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kStreamThreads), (size_t)SLOTS * kSlotBytes + lds_rest, stream, src, a, Hx, \
                           Hy, (uint32_t)n_tiles, K);                                                                    \
        RF_HIP_CHECK(hipGetLastError());                                                                                 \
        return RF_OK;                                                                                                    \
    }
    RF_CASE(1, 1, 4, 3) RF_CASE(1, 2, 4, 3) RF_CASE(1, 3, 4, 3) RF_CASE(2, 1, 4, 3) RF_CASE(2, 2, 4, 3) RF_CASE(2, 3, 4, 3)
    RF_CASE(3, 1, 4, 3) RF_CASE(3, 2, 4, 3) RF_CASE(3, 3, 4, 3)
    RF_CASE(1, 1, 3, 2)       // RF_STREAM_RING=3: the shallower ring (A/B runs on cfg3's shape)
#undef RF_CASE
    set_error("stream tails: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}

}  // namespace rf
