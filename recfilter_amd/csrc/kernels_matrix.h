// kernels_matrix.h -- argument blocks and launchers of the matrix path (kernels_matrix.hip): scans of any order up to
// RF_MAX_ORDER = 32 in their direct form, every stage of the tiled algorithm a small dense f32 GEMM on the matrix cores.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rf_internal.h"

namespace rf {

constexpr int kMxSB = 32;                  // sub-block: samples one 32x32x2 MFMA chain covers
constexpr int kMxWaves = 4;                // waves per workgroup of the pass kernels (the final passes copy one 16-byte piece of
                                           // every 32 x 32 operator per thread; the x -> y hand-over has one wave per sub-block of a y tile)
constexpr int kMxUnits = 32 * kMxWaves;    // units (line, tile) per workgroup: one per lane column
constexpr int kMxMaxNB = 8;                // sub-blocks per tile: T <= 256 (the H fragments of a tile sit in LDS: 4 KiB per sub-block)
constexpr int kMxChunk = 16;               // tiles per chunk of the carry chain (levels of the blocked scan)
constexpr int kMxTopMax = 24;              // a sequence this short is chained in one go
constexpr int kMxTopWide = 512;            // ... when the lines alone fill the chip (plan_matrix.cpp)

// How the units of a scanned dimension lie in memory:
//   MX_X1  scan along x, few lines (1-D signals): unit U = line * M + tile, a workgroup takes 128 consecutive units =
//          128 * T consecutive samples; lane = tile
//   MX_XL  scan along x, >= 32 lines: a workgroup takes 128 consecutive lines of one tile; lane = line
//   MX_Y   scan along y or z: a workgroup takes 128 consecutive columns of one tile; lane = column
enum MxMode { MX_X1 = 0, MX_XL = 1, MX_Y = 2 };

struct MxPassArgs {
    int32_t mode;
    int32_t T, NB, M;          // tile width, sub-blocks per tile, tiles per line
    int32_t k;                 // feedback order (rows of a tail)
    int32_t causal;
    int32_t clamped;           // the tile where the scan enters the image takes the border correction
    int64_t N, inner, lines;   // extent and stride of the scanned dimension, number of lines
    int32_t ragged;            // the tiles do not divide the extent (off != 0 or M T != N): elements are checked one by one
    int64_t off;               // tile t covers the samples [t T - off, (t + 1) T - off): the tiles need not divide N, the padding
                               // (zeros, never stored) lies where the scan LEAVES the image -- 0 for a causal scan, M T - N otherwise
    int64_t units;             // lines * M
    const float *G, *R;        // A-operand fragments [16][64] of the sub-block operators (see kernels_matrix.hip)
    const float *dG;           // [32]: what a clamped border adds to the first sub-block per unit first sample
    const float *H;            // A-operand fragments [NB][16][64] of the tail extraction
    const float *dH;           // [32]: ... to the tile-local tail
    float *tails;              // [unit][KP], KP = 8 ceil(k / 8); unit = line * M + tile (MX_X1) or tile * lines + line
    // A slab of a sharded image (the outermost dimension, MX_Y): whether it holds the image's first / last tile along the
    // dimension (else the scan enters the slab from a neighbouring one) and the carry that enters it, [line][KP] (zeros before the
    // exchange has formed it; null for an unsharded plan)
    int32_t slab_first, slab_last;
    const float *incoming;
    // The stage that follows (pass 2 only): the final pass holds the finished block in LDS, which is exactly what the NEXT
    // scan's pass 1 would read from HBM -- so it contracts it with the next scan's H and stores that scan's tile-local tails
    // (lib/reorder.cpp:100-176 chains stages through memory; VERDICT r4 "next" 3).  next: 0 none; 1 the next scan runs along
    // the same dimension (same units); 2 this scan runs along x with lane = line and the next one along y of a 2-D image whose
    // 128 x 128 blocks coincide (lane = column of the same block).
    int32_t next;
    int32_t next_k, next_causal, next_NB;      // (next_NB: sub-blocks of the next scan's tile = of its H)
    const float *next_H, *next_dH;
    float *next_tails;
    // A PAIR stage: a causal scan and the anticausal scan that follows it along the same dimension, in ONE final pass (the tile's
    // causal result stays in registers: 8 instead of 16 bytes per sample and pair).  This block describes the causal scan; p_*
    // the anticausal one.  Pass 1 forms both scans' tile-local tails from the tile: H x and H21 x, H21 = (tail extraction of the
    // anticausal scan) . (the causal scan's tile operator); between the two carry chains the planner adds W21 . (the completed
    // causal carry entering the tile) to the anticausal tails (create_tail_residual_term, lib/split.cpp:912-1004).
    int32_t pair;              // 0: a stage of one scan
    int32_t p_k;
    const float *p_G, *p_R, *p_dG;     // pass 2: the anticausal scan's operators
    const float *p_H, *p_dH;           // pass 1: fragments [NB][16][64] of the STACKED tail extraction -- rows 0 .. 15 the causal scan's H,
                                       // rows 16 .. 31 H21 (a pair's tails have at most 16 rows) -- and its border vector [32] likewise
    float *p_tails;                    // the anticausal scan's tails (pass 1: tile-local; pass 2: completed), [unit][KP]
};

// One level of the carry chain / of its propagation.  An ELEMENT is a k-vector stored as KP = 8 ceil(k / 8) floats (rows
// fastest, zero padded).  Column c of the launch (a lane) is split as c_hi = c / cdiv, c_lo = c % cdiv; its element of
// step j is element number base + c_hi * s_hi + c_lo * s_lo + j * s_j of seq.
struct MxChainArgs {
    float *seq;
    float *exits;              // next level: element c_hi * e_hi + c_lo * e_lo; null on the top level
    const float *A;            // A-operand fragments [16][64] of the level's transfer matrix
    const float *P;            // propagation: fragments [C][16][64] of its powers 1..C
    int32_t k, C;
    int32_t chunk_is_lo;       // the chunk index is c_lo (MX_X1) or c_hi
    int32_t enter_fixed;       // propagation of a slab's entering carry: every element takes exits[c_lo * e_lo] (no chunk before it)
    int64_t ncols, cdiv;
    int64_t base, s_hi, s_lo, s_j;
    int64_t e_hi, e_lo;
    int64_t Mtot;              // elements per line on this level (the last chunk may be short)
    // The anticausal chain of a PAIR stage when it runs over the whole line in one go: the cross term is added on the way -- step j
    // takes crossW . cross[element of step j + cross_shift] (the completed causal carry entering the tile) for j < cross_steps, and
    // step 0 crossD . cross[its own element] (the anticausal scan's clamped border) -- instead of a launch of its own before the chain.
    const float *cross;        // null: a plain chain
    const float *crossW, *crossD;      // fragments [16][64]; crossD null without a clamped border
    int64_t cross_shift;
    int32_t cross_steps;
};

int launch_mx_pass1(const float *src, const MxPassArgs &a, hipStream_t stream);
int launch_mx_pass2(const float *src, float *dst, const MxPassArgs &a, hipStream_t stream);
int launch_mx_pass2_pair(const float *src, float *dst, const MxPassArgs &a, hipStream_t stream);
int launch_mx_chain(const MxChainArgs &a, hipStream_t stream);
int launch_mx_apply(const MxChainArgs &a, hipStream_t stream);

}  // namespace rf
