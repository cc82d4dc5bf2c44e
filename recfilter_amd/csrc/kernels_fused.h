// kernels_fused.h -- argument blocks and launchers of the fused x/y tile kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"
#include "pixel.h"
#include "rf_internal.h"

namespace rf {

constexpr int kFusedTX = 256;       // tile width: 16 lanes x 16 samples = one DPP row per tile row
constexpr int kFusedSeg = 16;       // samples per lane in the x phase
constexpr int kFusedThreads = 256;  // 4 waves
constexpr int kFusedMaxK = 3;       // max feedback order on the fused path
constexpr int kFusedMaxMod = 8;     // samples a border modification touches: the orders the fused path rewrites into sections (4..8; RF_MAX_ORDER is 32 since round 5)
constexpr int kFusedMaxScans = 4;   // max scans per dimension on the fused path

// One scan as the fused kernels read it (device memory, uniform -> scalar loads).
template <typename Acc>
struct FusedScan {
    int32_t causal;
    Acc b;
    Acc a[kFusedMaxK];
    // x phase only; "direction coordinates": position p counts from where the scan enters
    Acc R[kFusedMaxK][kFusedSeg];           // R[j][m]: effect on the segment's sample m (MEMORY order, so that
                                            // neighbours pair up for packed FMAs) of component j of the entering state
    Acc P[4][kFusedMaxK][kFusedMaxK];       // segment exit-state transfer over 1, 2, 4, 8 segments
    // border modification of a scan in zero-border form (Scan::mod_n, rf_internal.h); read only when FusedArgs::mod_form
    int32_t mod_n;
    Acc mod_g[kFusedMaxMod];
};

// y scans only need their coefficients
template <typename Acc>
struct FusedScanY {
    int32_t causal;
    Acc b;
    Acc a[kFusedMaxK];
    int32_t mod_n;                          // as FusedScan::mod_n
    Acc mod_g[kFusedMaxMod];
};

// Passed to the kernels BY VALUE: the scan tables then live in the kernarg segment (constant address
// space), so a run-time scan index still compiles to scalar loads.  Behind a pointer the persistent
// kernel's own stores make them "possibly clobbered" and the compiler falls back to per-lane vector
// loads that cost ~50 VGPRs.
constexpr int kFusedMaxPlanes = 16;     // == RF_MAX_PLANES

template <typename Acc>
struct FusedArgs {
    int64_t NX, NY, NZ;      // extents (NZ = batch of planes along z, 1 for 2-D)
    int64_t NXP;             // MX * 256: pitch of the y tails / y carries (the last tile of a row may be partial)
    int32_t MX, MY;          // tiles along x / y
    int32_t last_lane;       // x phase: lane holding the last existing segment of a row's last tile (15 when full)
    int32_t last_cols;       // columns that exist in a row's last tile (256 when full; any number for 4- and 8-byte pixels)
    uint32_t row_bytes;      // NX * sizeof(pixel): image pitch in bytes, for 32-bit offset arithmetic inside a tile
    int64_t NYP;             // MY * TY: pitch of the x tails / x carries (the last tile row may be partial)
    int32_t last_rows;       // rows that exist in the last tile row (TY when full)
    int32_t nx, ny;          // scans along x / y
    int32_t clamped;
    int32_t y_first_border;  // the slab holds the image's first / last tile row
    int32_t y_last_border;
    FusedScan<Acc> xs[kFusedMaxScans];
    FusedScanY<Acc> ys[kFusedMaxScans];
    Acc *xt;                 // x tails   [s][tx][r][y + NY*z]
    Acc *yt;                 // y tails   [j][ty][r][x + NX*z]
    const Acc *y_incoming;   // carry entering the slab along y, [j][r][x + NX*z]
    const Acc *y_apply;      // row shards with the merged exchange: Y[q][j][ty][r][o] (cross_scan_transfer, plan_generic.h).
                             // The slab's y tails were completed with ZERO entering carries; pass 2 adds what the true
                             // ones (y_incoming) contribute, Y * in, as it loads a carry, so the tails are not rewritten
                             // by a separate launch.  Null otherwise.
    const Acc *x_incoming;   // carry entering each row along x, [s][r][y + NY*z]: zeros for an image; for a long
                             // 1-D signal folded into rows it is the state the previous row hands over
    // pointwise stages fused into the passes (rf_pointwise_desc; float pixels only): bit 0 = x' = pre_s*in + pre_b
    // on every pixel load, bit 1 = out = post_f*F + post_i*x' + post_b on the final store of pass 2
    int32_t pw_flags;
    Acc pre_s, pre_b, post_f, post_i, post_b;
    // Tuple planes of a 2-D filter batched into one launch: plane z of the "volume" is its own buffer (the kernels'
    // src/dst arguments are unused); the tails treat the planes exactly like the z planes of a 3-D image
    int32_t plane_batch;
    // The final pass of an image with partial tiles runs as up to three launches: the whole tiles on the lean kernel, the
    // last tile column and the last tile row on the EDGE variant (kernels_fused.hip, launch_fused_pass2).  tx0 / ty0: the
    // tile this launch's block (0, 0) stands for; gx / gy: its grid (0 = MX / MY).  Read by the final-pass kernels only.
    int32_t tx0, ty0, gx, gy;
    // Final pass on 128-row tiles: workgroup b of the launch takes tile (b mod 8) * (tiles / 8) + b / 8 instead of tile b, i.e.
    // every XCD (workgroups are dealt to the eight XCDs round robin) walks a contiguous eighth of the tiles.  Set by the
    // launcher for images whose row pitch is not a multiple of 128 bytes: their rows start anywhere inside a 128-byte line,
    // so the line at every tile boundary is written in two parts by two workgroups -- in the same L2 when they are
    // neighbours inside an XCD (tools/microbench/hbm_read_patterns pitch 16380: tile-shaped copy 0.438 -> 0.413 ms; an
    // aligned pitch copies in 0.324 ms and prefers the plain order, 0.337 ms with this one).
    int32_t xcd_contig;
    // The plan's scans are in zero-border form behind border modifications (clamped filters whose high-order scans were split
    // into sections, plan.cpp): `clamped` then only says WHERE the modifications apply; the kernels that run recurrences take
    // their general-pattern code and call border_mod_* (scan_device.h) instead of the native clamped prologue.
    int32_t mod_form;
    const void *in_planes[kFusedMaxPlanes];
    void *out_planes[kFusedMaxPlanes];
    // Layout of the y tails.  0: [j][ty][r][column] -- a tile's rows are 1-KiB pieces a whole image row apart; 1: tile-major,
    // [ty][z*MX + tx][j][r][256] -- every tile owns one contiguous block of ny*K KiB (what pass 1 writes and pass 2 reads in
    // one piece; pass 1 is sensitive to how its tail stores reach memory).  Row shards keep layout 0.
    int32_t yt_tile_major;
    // The combined rows of pass 1 arrive in `yt_parts` parts (kernels_tails_walk.hip: a y tile is several 32-row patches, each
    // the work of its own workgroup): part p of element e is ytp[p * yt_part_stride + e] in the layout of yt, and
    // xscan_rows_kernel adds the parts up as it loads a row (it stores to yt).  0 / 1: the rows are in yt.
    int32_t yt_parts;
    int64_t yt_part_stride;
    const Acc *ytp;
    // A long 1-D signal folded into rows whose length is not a whole number of rows (plan_fused.cpp, "chained rows"): the
    // image the kernels see is the signal followed by zeros, but only its first lin_limit samples exist in the caller's
    // buffers -- loads beyond them yield zeros, stores beyond them are dropped.  0: every sample of the image exists.
    int64_t lin_limit;
    // element index of y tail (j, ty, r) of column `line` (= x + NXP * z)
    __host__ __device__ int64_t yt_index(int j, int ty, int r, int K, int64_t line) const {
        if (yt_tile_major)
            return ((((int64_t)ty * (NXP * NZ / kFusedTX) + (line >> 8)) * ny + j) * K + r) * kFusedTX + (line & 255);
        return (((int64_t)j * MY + ty) * K + r) * (NXP * NZ) + line;
    }
};

// Register-column scans along a strided dimension (kernels_strided.hip), by value like FusedArgs.
template <typename Acc>
struct StridedArgs {
    int64_t n, inner, lines;     // extent and stride of the filtered dimension, number of lines
    int32_t M;                   // tiles of TZ samples
    int32_t n_scans;
    int32_t clamped, first_is_border, last_is_border;
    int32_t mod_form;            // as FusedArgs::mod_form
    FusedScanY<Acc> scans[kFusedMaxScans];
    Acc *tails;                  // [s][t][r][line]
    const Acc *incoming;         // [s][r][line]
};

// Untiled scans with the x phase's parallelism inside the line (kernels_lines.hip): all scans of ONE dimension in one launch.
template <typename Acc>
struct LineScanArgs {
    int64_t n, inner, lines;     // extent and stride of the filtered dimension, number of lines
    int32_t n_scans, clamped;
    FusedScan<Acc> scans[kFusedMaxScans];      // with the 16-sample segment tables R / P
};
template <typename P>
int launch_line_scans(int K, bool strided, const P *src, P *dst, const LineScanArgs<typename PixelTraits<P>::Acc> &a,
                      hipStream_t stream);

template <typename P>
int launch_strided_pass(bool final_pass, int K, int TZ, const P *src, P *dst,
                        const StridedArgs<typename PixelTraits<P>::Acc> &a, hipStream_t stream);

// Blocked parallel carry scan over the tails of one dimension (kernels_carry.hip); scans
// [s_begin, s_end) of the dimension in one launch.  AC[s] = A[s]^C, C = carry_chunk_length(M, lines).
// chained-rows plans (long 1-D signals): a carry launch that first finishes the previous scan (kernels_carry.hip, PRE)
template <typename Acc>
struct ChainPre {
    const Acc *exit_states;    // [K][lines] of scan s_begin - 1 (NOT the buffer this launch publishes its own exits to)
    Acc *incoming_prev;        // [K][lines]: the states entering the rows for scan s_begin - 1 (workgroup 0 stores them)
    const Acc *AM, *AMS;       // A^MX and (A^MX)^S of scan s_begin - 1
    const Acc *Apow;           // [i][K][K] = A^(i+1) of scan s_begin - 1
    int32_t S, causal_prev;
};

template <typename Acc>
int launch_carry_block(int K, const GenericDimArgs<Acc> &a, uint32_t causal_mask, int s_begin, int s_end, Acc *send,
                       const Acc *AC, int C, hipStream_t stream, const ChainPre<Acc> *pre = nullptr);
int carry_chunk_length(int64_t M, int64_t lines, int K = 1);   // chunk length depends on the order only above 3
// Chains the rows of a 1-D signal folded into NY rows (kernels_carry.hip): from the rows' local exit states
// exit[r][y] it forms the state entering every row, incoming[r][y] (entry of the next row = AM * entry + exit,
// AM = A^MX), with 64 lanes each owning S consecutive rows; AMS = AM^S.
template <typename Acc>
int launch_row_chain(int K, const Acc *exit_states, Acc *incoming, int NY, bool causal, const Acc *AM, const Acc *AMS,
                     int S, hipStream_t stream);
int carry_chunk_count(int64_t M, int64_t lines, int C, int K = 1);
// row chain + propagation through the rows' tails in one launch (when the rows' entering states fit the LDS)
bool chain_apply_applies(int K, int64_t NY, size_t acc_bytes);
template <typename Acc>
int launch_chain_apply(int K, const GenericDimArgs<Acc> &a, int s, const Acc *exit_states, Acc *incoming, bool causal,
                       const Acc *AM, const Acc *AMS, int S, hipStream_t stream);

// pass 2: the final correction pass (kernels_fused.hip)
// (src_u8: the input plane holds unsigned bytes, rf_pointwise_desc.in_dtype == RF_IN_U8; float pixels only)
template <typename P>
int launch_fused_pass2(int K, int TY, const void *src, bool src_u8, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                       hipStream_t stream);
// the same pass on 256 x 128 tiles: two 64-row halves through the LDS, the 128-sample column in registers
// (kernels_fused_tall.hip); halves the y tails and every kernel that walks them
template <typename P>
int launch_fused_pass2_tall(int K, const void *src, bool src_u8, P *dst, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                            hipStream_t stream);
// pass 1 as a contraction with precomputed impulse responses (kernels_tails.hip)
template <typename P>
int launch_fused_tails(int K, int TY, const void *src, bool src_u8, const FusedArgs<typename PixelTraits<P>::Acc> &a,
                       const typename PixelTraits<P>::Acc *Hx, const typename PixelTraits<P>::Acc *Hy,
                       hipStream_t stream);
// the same pass as a streaming kernel (kernels_stream.hip): persistent workgroups, LDS-DMA ring, loader wave
bool stream_tails_applicable(int K, int TY, bool src_u8, int pw_flags, int last_cols, int last_rows, int64_t n_tiles, int MX,
                             int64_t NZ, int nxk, int nyk, int mode /* +1 RF_PLAN_STREAM_PASS1, -1 RF_PLAN_STAGED_PASS1 */);
int launch_stream_tails(int K, const float *src, const FusedArgs<float> &a, const float *Hx, const float *Hy, hipStream_t stream);
// pass 1 with the x-tail contraction on the matrix cores, one tile per workgroup (kernels_tails_mfma.hip)
bool mfma_tails_applicable(int K, int TY, bool src_u8, int pw_flags, int last_cols, int last_rows, int64_t lin_limit, int nx, int ny,
                           int mode);
int launch_mfma_tails(int K, int TY, const float *src, const FusedArgs<float> &a, const float *Hx, const float *Hy, hipStream_t stream);
// pass 1 of a 3-D plan in one read of the volume: x tails, the parts of the y tails' combined rows and the z tails
// (kernels_tails_walk.hip)
struct WalkArgs {
    float *ytp;              // parts of the combined rows, [part][layout of FusedArgs::yt]; == yt when the y tile is one patch
    int64_t part_stride;     // elements between two parts
    float *zt;               // z tails, [s][tz][r][y * NX + x] (StridedArgs::tails)
    const float *HzT;        // impulse responses of the z tails, [variant][z][4]
    int32_t TY, TZ, MZ;      // rows of a y tile (32 * parts), planes of a z tile, z tiles
    int32_t parts_log2;      // log2(TY / 32)
    int32_t z_first_border, z_last_border;      // the slab holds the volume's first / last z tile
    int32_t nzk, KZ;         // z tails per sample (= scans along z * KZ), order of the z scans
    // tall patches (128 columns x 64 rows, kernels_tails_walk.hip TALL): TY / 64 parts of the combined rows, and the x tails of
    // a tile in two parts -- its left half in FusedArgs::xt, its right half in xt2 (same layout)
    int32_t tall;
    float *xt2;
};
bool walk_tails_applicable(int K, int TY, int nx, int ny, int nz, int KZ, int TZ, int last_cols, int last_rows);
int launch_walk_tails(int K, const float *src, const FusedArgs<float> &a, const WalkArgs &wa, const float *Hx, const float *Hy,
                      hipStream_t stream);
// tile-local x scans of the combined rows + cross-dimension residual, in place in yt (G == nullptr: no residual)
template <typename Acc>
int launch_xscan_rows(int K, int TY, const FusedArgs<Acc> &a, const Acc *Hy, const Acc *G, hipStream_t stream,
                      const Acc *Wx = nullptr, const Acc *Ax = nullptr, Acc *xt_done = nullptr);
// Wx / Ax / xt_done given: the launch also completes the x tails (the carry scan along x), into xt_done -- for images this
// predicate accepts (few tiles per row: the separate carry launch is all launch and latency there)
bool xscan_completes_x_tails(int K, int TY, int MX, int nx, int ny, size_t acc_bytes, int64_t tile_rows /* MY * NZ */);
}  // namespace rf
