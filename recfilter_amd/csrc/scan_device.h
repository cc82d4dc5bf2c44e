// scan_device.h -- device-side building blocks shared by the fused kernels: DPP row shifts, the LDS
// chunk swizzle, the 16-lane segment scan of the x phase and the register-column scan of the y phase.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_fused.h"

namespace rf {
namespace {

// ---- DPP helpers ------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_move(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {          // two 32-bit moves (f64 pixels on the line kernels)
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
// shift towards higher lanes (causal) = row_shr, towards lower lanes (anticausal) = row_shl;
// lanes without a source inside their 16-lane row receive 0
template <bool CAUSAL, int D, typename Acc>
__device__ __forceinline__ Acc row_shift(Acc v) {
    return dpp_move<(CAUSAL ? 0x110 : 0x100) + D>(v);
}

__device__ __forceinline__ int swz_chunk(int c) { return c ^ ((c >> 4) & 3); }

// ---- x phase: one scan over NR 256-sample tile rows, each held as 16 samples per lane -----------
// The NR rows are independent recurrences; every step is written row-innermost so the compiler
// interleaves them (ILP = NR) and the dependent-FMA latency of one row hides behind the others.
template <typename Acc, bool CAUSAL, int K, int NR>
__device__ __forceinline__ void scan_rows16(Acc (&v)[NR][kFusedSeg], const FusedScan<Acc> &sc, bool first_lane,
                                            bool clamp_first, const Acc (&carry)[NR][K], bool dead_lane = false,
                                            int entry_valid = kFusedSeg) {
    Acc h[NR][K];
    Acc y0[NR];
#pragma unroll
    for (int n = 0; n < NR; n++) {
#pragma unroll
        for (int j = 0; j < K; j++) h[n][j] = first_lane ? carry[n][j] : Acc(0);
        y0[n] = Acc(0);
    }
    // feed-forward first, two samples per instruction for f32 (v_pk_mul_f32); the clamped prologue still needs the
    // unscaled first sample
    constexpr int m_first = CAUSAL ? 0 : kFusedSeg - 1;
    Acc x_first[NR];
#pragma unroll
    for (int n = 0; n < NR; n++) {
        x_first[n] = v[n][m_first];
        if (!CAUSAL && entry_valid < kFusedSeg) {                    // partial entry segment: its last existing sample
            // (entry_valid is wave-uniform: a scalar switch over statically indexed registers, each case behind an opaque
            // asm -- left alone the compiler folds the cases into v[n][entry_valid - 1], a run-time index that sends the
            // whole register array to scratch)
            Acc pick = v[n][0];
#define RF_PICK(c) case c + 1: pick = v[n][c]; asm volatile("" : "+v"(pick)); break;
            switch (entry_valid) {
                RF_PICK(1) RF_PICK(2) RF_PICK(3) RF_PICK(4) RF_PICK(5) RF_PICK(6) RF_PICK(7) RF_PICK(8)
                RF_PICK(9) RF_PICK(10) RF_PICK(11) RF_PICK(12) RF_PICK(13) RF_PICK(14)
                default: break;
            }
#undef RF_PICK
            x_first[n] = first_lane ? pick : x_first[n];
        }
    }
    if constexpr (std::is_same<Acc, float>::value) {
        typedef float F2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int m = 0; m < kFusedSeg; m += 2)
#pragma unroll
            for (int n = 0; n < NR; n++) {
                const F2 r = F2{sc.b, sc.b} * F2{v[n][m], v[n][m + 1]};
                v[n][m] = r.x;
                v[n][m + 1] = r.y;
            }
    } else {
#pragma unroll
        for (int m = 0; m < kFusedSeg; m++)
#pragma unroll
            for (int n = 0; n < NR; n++) v[n][m] = sc.b * v[n][m];
    }
    // An anticausal scan of a row's partial last tile enters inside the entry lane's segment when the width is not a
    // multiple of 16: only its first entry_valid (1..15; wave-uniform) samples exist.  The dead ones are cleared --
    // an earlier causal scan ran on into them -- and the clamped prologue is positioned per lane; with a zero state
    // the dead positions then stay zero until the scan reaches the image.  Only those tiles take this path.
    if (!CAUSAL && entry_valid < kFusedSeg) {
        const int off = first_lane ? kFusedSeg - entry_valid : 0;           // direction positions before the image
#pragma unroll
        for (int p = 0; p < kFusedSeg; p++) {
            const int m = kFusedSeg - 1 - p;
            const int pr = p - off;
#pragma unroll
            for (int n = 0; n < NR; n++) {
                Acc acc = pr < 0 ? Acc(0) : v[n][m];
#pragma unroll
                for (int j = K - 1; j >= 0; j--) {
                    Acc g = h[n][j];
                    g = (clamp_first && pr >= 0 && pr <= j) ? (pr == 0 ? x_first[n] : y0[n]) : g;
                    acc = acc + sc.a[j] * g;
                }
#pragma unroll
                for (int j = K - 1; j > 0; j--) h[n][j] = h[n][j - 1];
                h[n][0] = acc;
                // every position before the border reads the scan's first output from here on (lib/recfilter.cpp:330-336):
                // as the state, so that a prologue longer than the entry segment (fewer than K samples exist in it, e.g. a
                // width of 16 m + 1) carries over into the next lane
                if (clamp_first && pr == 0) {
#pragma unroll
                    for (int j = 1; j < K; j++) h[n][j] = acc;
                }
                y0[n] = (pr == 0) ? acc : y0[n];
                v[n][m] = acc;
            }
        }
    } else
    // 1. segment-local recurrence (exact for the first lane, which owns the tile's carry)
#pragma unroll
    for (int p = 0; p < kFusedSeg; p++) {
        const int m = CAUSAL ? p : kFusedSeg - 1 - p;
#pragma unroll
        for (int n = 0; n < NR; n++) {
            const Acc x = x_first[n];       // only read at p == 0
            Acc acc = v[n][m];
            // oldest tap first: the newest output h[0] enters last, one dependent FMA per sample
#pragma unroll
            for (int j = K - 1; j >= 0; j--) {
                Acc g = h[n][j];
                if (p <= j) g = clamp_first ? (p == 0 ? x : y0[n]) : g;
                acc = acc + sc.a[j] * g;
            }
#pragma unroll
            for (int j = K - 1; j > 0; j--) h[n][j] = h[n][j - 1];
            h[n][0] = acc;
            if (p == 0) y0[n] = acc;
            v[n][m] = acc;
        }
    }
    // A lane beyond the last existing segment of a partial tile (dead_lane) holds no image samples.  Whatever an
    // earlier causal scan left there must not flow back into the row: its exit state is dropped.
    if (!CAUSAL) {
#pragma unroll
        for (int n = 0; n < NR; n++)
#pragma unroll
            for (int j = 0; j < K; j++) h[n][j] = dead_lane ? Acc(0) : h[n][j];
    }
    // 2. Kogge-Stone over the 16 lanes of each row: S_l <- sum_{j<=l} P^(l-j) S_j
#define RF_KS_STEP(D, IDX)                                                                        \
    _Pragma("unroll") for (int n = 0; n < NR; n++) {                                              \
        Acc Sh[K];                                                                                \
        _Pragma("unroll") for (int j = 0; j < K; j++) Sh[j] = row_shift<CAUSAL, D>(h[n][j]);      \
        _Pragma("unroll") for (int r = 0; r < K; r++)                                             \
            _Pragma("unroll") for (int j = 0; j < K; j++) h[n][r] = h[n][r] + sc.P[IDX][r][j] * Sh[j]; \
    }
    RF_KS_STEP(1, 0)
    RF_KS_STEP(2, 1)
    RF_KS_STEP(4, 2)
    RF_KS_STEP(8, 3)
#undef RF_KS_STEP
    // 3. state entering this lane's segment, then the rank-K correction of its 16 samples
    Acc C[NR][K];
#pragma unroll
    for (int n = 0; n < NR; n++)
#pragma unroll
        for (int j = 0; j < K; j++) C[n][j] = row_shift<CAUSAL, 1>(h[n][j]);
    if constexpr (std::is_same<Acc, float>::value) {
        // two neighbouring samples per instruction (v_pk_fma_f32); j outermost so that consecutive packed FMAs are
        // independent (a dependent pair costs a wait state)
        typedef float F2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < K; j++)
#pragma unroll
            for (int m = 0; m < kFusedSeg; m += 2)
#pragma unroll
                for (int n = 0; n < NR; n++) {
                    const F2 r = F2{sc.R[j][m], sc.R[j][m + 1]} * F2{C[n][j], C[n][j]} + F2{v[n][m], v[n][m + 1]};
                    v[n][m] = r.x;
                    v[n][m + 1] = r.y;
                }
    } else {
#pragma unroll
        for (int m = 0; m < kFusedSeg; m++)
#pragma unroll
            for (int n = 0; n < NR; n++)
#pragma unroll
                for (int j = 0; j < K; j++) v[n][m] = v[n][m] + sc.R[j][m] * C[n][j];
    }
}

// ---- y phase: one scan up or down a register column -----------------------------------------
// `done(m, value)` is called for every row as soon as it is final (the last scan of a pass stores from there: the
// rows leave while the recurrence still runs, instead of TY stores in one burst behind it)
struct NoRowSink { __device__ __forceinline__ void operator()(int, float) const {} __device__ __forceinline__ void operator()(int, double) const {}
                   __device__ __forceinline__ void operator()(int, uint32_t) const {} };
template <typename Acc, bool CAUSAL, int K, int TY, typename SC, typename Sink = NoRowSink>
__device__ __forceinline__ void scan_col(Acc (&col)[TY], const SC &sc, bool clamp_first,
                                         const Acc (&carry)[K], const Sink &done = Sink()) {
    Acc h[K];
#pragma unroll
    for (int j = 0; j < K; j++) h[j] = carry[j];
    Acc y0 = Acc(0);
    // feed-forward first, two rows per instruction for f32 (v_pk_mul_f32); the clamped prologue still needs the
    // unscaled first sample
    const Acc x_first = col[CAUSAL ? 0 : TY - 1];
    if constexpr (std::is_same<Acc, float>::value) {
        typedef float F2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < TY; i += 2) {
            const F2 r = F2{sc.b, sc.b} * F2{col[i], col[i + 1]};
            col[i] = r.x;
            col[i + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < TY; i++) col[i] = sc.b * col[i];
    }
#pragma unroll
    for (int p = 0; p < TY; p++) {
        const int m = CAUSAL ? p : TY - 1 - p;
        const Acc x = x_first;             // only read at p == 0
        Acc acc = col[m];
        // oldest tap first: the newest output h[0] enters last, one dependent FMA per sample
#pragma unroll
        for (int j = K - 1; j >= 0; j--) {
            Acc g = h[j];
            if (p <= j) g = clamp_first ? (p == 0 ? x : y0) : g;
            acc = acc + sc.a[j] * g;
        }
#pragma unroll
        for (int j = K - 1; j > 0; j--) h[j] = h[j - 1];
        h[0] = acc;
        if (p == 0) y0 = acc;
        col[m] = acc;
        done(m, acc);
    }
}

// Anticausal scan up a column whose tile row is partial: only rows [0, rows) exist, the scan enters at row rows-1
// (run-time, wave-uniform) instead of TY-1.  The non-existing rows are cleared first -- an earlier causal scan ran
// on into them -- so with a zero state they stay zero until the scan reaches the image; the clamped prologue is
// positioned with run-time (scalar) compares.  Only the last tile row of an image pays for this.
template <typename Acc, int K, int TY, typename SC>
__device__ __forceinline__ void scan_col_partial_up(Acc (&col)[TY], const SC &sc, bool clamp_first, int rows) {
    Acc h[K];
#pragma unroll
    for (int j = 0; j < K; j++) h[j] = Acc(0);
    Acc y0 = Acc(0);
    const int off = TY - rows;                   // direction positions before the image
#pragma unroll
    for (int p = 0; p < TY; p++) {
        const int m = TY - 1 - p;
        const int pr = p - off;                  // position counted from the entry row (negative: not in the image)
        const Acc x = pr < 0 ? Acc(0) : col[m];
        Acc acc = sc.b * x;
#pragma unroll
        for (int j = K - 1; j >= 0; j--) {
            Acc g = h[j];
            g = (clamp_first && pr <= j) ? (pr == 0 ? x : y0) : g;
            acc = acc + sc.a[j] * g;
        }
#pragma unroll
        for (int j = K - 1; j > 0; j--) h[j] = h[j - 1];
        h[0] = acc;
        y0 = (pr == 0) ? acc : y0;
        col[m] = acc;
    }
}

// ---- border modification of scans in zero-border form (FusedArgs::mod_form; plan.cpp, "clamped sections") ----------------
// A clamped scan is the zero-border scan of an input whose first k samples in scan direction are x_r + g_r * x_0
// (tables.h, scan_tile).  These run on the tile where the scan enters a clamped image, in front of a scan called with
// clamp_first = false.  The plan guarantees whole entry segments (width % 16 == 0) and whole tile rows.
template <typename Acc, bool CAUSAL, int NR, typename SC>
__device__ __forceinline__ void border_mod_rows16(Acc (&v)[NR][kFusedSeg], const SC &sc, bool entry_lane) {
#pragma unroll
    for (int n = 0; n < NR; n++) {
        const Acc x0 = v[n][CAUSAL ? 0 : kFusedSeg - 1];
#pragma unroll
        for (int r = 0; r < kFusedMaxMod; r++) {
            const Acc g = (entry_lane && r < sc.mod_n) ? sc.mod_g[r] : Acc(0);
            const int m = CAUSAL ? r : kFusedSeg - 1 - r;
            v[n][m] = v[n][m] + g * x0;
        }
    }
}

// (whole tile rows only: a plan in mod form picks a tile height that divides the image's, plan_fused.cpp)
template <typename Acc, bool CAUSAL, int TY, typename SC>
__device__ __forceinline__ void border_mod_col(Acc (&col)[TY], const SC &sc) {
    const Acc x0 = col[CAUSAL ? 0 : TY - 1];
#pragma unroll
    for (int r = 0; r < kFusedMaxMod && r < TY; r++) {
        const Acc g = r < sc.mod_n ? sc.mod_g[r] : Acc(0);
        const int m = CAUSAL ? r : TY - 1 - r;
        col[m] = col[m] + g * x0;
    }
}

template <typename Acc>
struct Vec4 {
    typedef Acc type __attribute__((ext_vector_type(4)));
};

// Four neighbouring input samples as arithmetic values.  PI = the pixel type: one 16-byte load; PI = uint8_t
// (rf_pointwise_desc.in_dtype == RF_IN_U8): one 4-byte load and four byte-to-float conversions.
// The image is streamed: each pass reads a sample once, so the loads are non-temporal (the `nt` cache hint) and leave
// L2 and the memory-side cache to the tails and the tables.  Together with non-temporal stores of the output this
// took cfg3 from 0.68 to 0.64 ms (pass 2: 5.76 -> 6.3 TB/s).
template <typename PI, typename Acc>
__device__ __forceinline__ typename Vec4<Acc>::type load_chunk(const char *p) {
    using A4 = typename Vec4<Acc>::type;
    if constexpr (sizeof(PI) == sizeof(Acc)) {
        return __builtin_nontemporal_load(reinterpret_cast<const A4 *>(p));
    } else if constexpr (sizeof(PI) == 2) {
        // int16 pixels (tests/test_type_invariance.cpp): one 8-byte load, sign-extended into the 32-bit ring
        typedef short S4 __attribute__((ext_vector_type(4)));
        const S4 w = __builtin_nontemporal_load(reinterpret_cast<const S4 *>(p));
        return A4{(Acc)(int32_t)w.x, (Acc)(int32_t)w.y, (Acc)(int32_t)w.z, (Acc)(int32_t)w.w};
    } else {
        static_assert(sizeof(PI) == 1, "input planes are the pixel type or unsigned bytes");
        const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
        return A4{(Acc)(w & 255u), (Acc)((w >> 8) & 255u), (Acc)((w >> 16) & 255u), (Acc)(w >> 24)};
    }
}

// The chunk of a folded 1-D signal that straddles the signal's end (FusedArgs::lin_limit): only its first `n_valid` (1..3)
// samples exist in the caller's buffer, so they are loaded one by one -- nothing behind the end is read -- and the rest
// of the chunk is zeros.
template <typename PI, typename Acc>
__device__ __forceinline__ typename Vec4<Acc>::type load_chunk_head(const char *p, int n_valid) {
    using A4 = typename Vec4<Acc>::type;
    const PI *q = reinterpret_cast<const PI *>(p);
    A4 v = A4{Acc(0), Acc(0), Acc(0), Acc(0)};
    v.x = (Acc)q[0];
    if (n_valid > 1) v.y = (Acc)q[1];
    if (n_valid > 2) v.z = (Acc)q[2];
    return v;
}

// A chunk of a row's LAST tile when the image width is not a multiple of 4: `valid` of its four samples exist
// (<= 0: none, zeros; >= 4: all).  The samples behind a row's end belong to the next row (or lie behind the plane),
// so a partial chunk is loaded sample by sample.  Rows of such an image are only element-aligned: the 16-byte loads of the
// full chunks are dword-aligned accesses, which global memory instructions take.
template <typename PI, typename Acc>
__device__ __forceinline__ typename Vec4<Acc>::type load_chunk_cols(const char *p, int valid) {
    using A4 = typename Vec4<Acc>::type;
    if (valid >= 4) return load_chunk<PI, Acc>(p);
    if (valid <= 0) return A4{Acc(0), Acc(0), Acc(0), Acc(0)};
    return load_chunk_head<PI, Acc>(p, valid);
}
// ... and after a prologue with a bias its samples that do not exist are zeros again
template <typename A4, typename Acc>
__device__ __forceinline__ void clear_dead_cols(A4 &v, int valid) {
    if (valid < 4) v.w = Acc(0);
    if (valid < 3) v.z = Acc(0);
    if (valid < 2) v.y = Acc(0);
    if (valid < 1) v.x = Acc(0);
}

}  // namespace
}  // namespace rf
