// tables.h -- the tiling algebra as host-side tables.
//
// This replaces the symbolic Func-graph rewriting of the reference's lib/split.cpp and the
// float matrices of lib/coefficients.cpp with a handful of small dense tables per filtered
// dimension.  Every table is produced by *running the scan recurrence on short vectors*
// (unit carries, unit inputs), never by a closed-form matrix expression, so the tables are
// consistent with what the kernels do by construction -- including the clamped-border
// prologue, where the reference's matrix_B(clamp) (lib/coefficients.cpp:38-39) differs from
// its own add_filter semantics (lib/recfilter.cpp:330-336) by ~1e-8.
//
// Notation (one filtered dimension, tile width T, scans s = 0..n-1 in application order,
// k = max feedback order in the dimension, shorter scans zero-padded as lib/split.cpp:575-578):
//   memory position m in [0,T); "direction position" p = m (causal) or T-1-m (anticausal)
//   carry c[j], j<k  = value at direction position -1-j (the j-th element before the tile)
//   tail  t[r], r<k  = value at direction position T-1-r   (what the next tile receives)
//
//   prop[v][q][s]  (T x k, s >= q, memory rows)   effect on the tile, after scans q..s, of the
//                   carry entering scan q:   F_s ... F_{q+1} R_q       (lib/split.cpp:152-203)
//   A[s]           (k x k)  tail rows of prop[.][s][s]: carry -> next carry (lib/split.cpp:770,832)
//   W[v][q][s]     (k x k)  tail rows (in scan s's direction) of prop[v][q][s], q < s
//                                                                      (lib/split.cpp:912-1004)
//   v = border variant of the tile: bit0 = tile is first along the dimension, bit1 = last;
//   it selects the clamped prologue for the scans whose first tile it is (lib/split.cpp:634-647).
//
// The scalar type S is double for floating pixel types and uint64_t (wrap-around ring
// arithmetic) for integer pixel types, so integer filters stay bit-exact modulo 2^32/2^16
// however large the true table entries grow.
#pragma once

#include <cstdint>
#include <vector>

#include "rf_internal.h"

namespace rf {

template <typename S>
struct ScanS {
    bool causal;
    S b;
    S a[RF_MAX_ORDER];
    int mod_n = -1;              // Scan::mod_n / mod_g (rf_internal.h): >= 0 -- zero-border recurrence behind a border modification
    S mod_g[RF_MAX_ORDER] = {};
};

// In-place scan of one tile in memory order -- the operator of lib/recfilter.cpp:321-343
// restricted to a tile (lib/split.cpp:628-654).
//   carry      != nullptr : history before the tile (zero-border form), carry[j] = y[-1-j]
//   clamp_first            : the tile is the first one in the scan's direction of a clamped image
//   T_valid < T             : only the first T_valid samples of the tile exist (the image's last, partial tile);
//                             an anticausal scan then enters at sample T_valid-1, the rest is never touched
template <typename S>
inline void scan_tile(S *v, int T, int k, const ScanS<S> &sc, bool clamp_first, const S *carry, int T_valid = -1) {
    if (T_valid < 0 || T_valid > T) T_valid = T;
    if (clamp_first && sc.mod_n >= 0) {
        // A clamped scan IS the zero-border scan of a modified input: y_r = b x_r + sum_{j<r} a_j y_{r-1-j} + (sum_{j>=r} a_j) c_r
        // with c_0 = x_0, c_r = y_0 = (b + sum a) x_0 (lib/recfilter.cpp:330-336), i.e. x~_r = x_r + g_r x_0 for the first
        // samples in scan direction.  The plan uses this form for scans it has split into sections (plan.cpp).
        if (T_valid >= 1) {
            const S x0 = v[sc.causal ? 0 : T_valid - 1];
            for (int r = 0; r < sc.mod_n && r < T_valid; r++) {
                const int m = sc.causal ? r : T_valid - 1 - r;
                v[m] = v[m] + sc.mod_g[r] * x0;
            }
        }
        clamp_first = false;
    }
    S hist[RF_MAX_ORDER];
    for (int j = 0; j < k; j++) hist[j] = carry ? carry[j] : S(0);
    S y0 = S(0);
    for (int p = 0; p < T_valid; p++) {
        int m = sc.causal ? p : T_valid - 1 - p;
        S x = v[m];
        S acc = sc.b * x;
        for (int j = 0; j < k; j++) {
            S g;
            if (p > j) g = hist[j];
            else if (clamp_first) g = (p == 0) ? x : y0;
            else g = hist[j];  // carry from the previous tile (zero when there is none)
            acc = acc + sc.a[j] * g;
        }
        for (int j = k - 1; j > 0; j--) hist[j] = hist[j - 1];
        hist[0] = acc;
        if (p == 0) y0 = acc;
        v[m] = acc;
    }
}

template <typename S>
struct DimTables {
    int T = 0, k = 0, n = 0;
    std::vector<ScanS<S>> scans;
    // prop[((v*n + q)*n + s)] is a T*k row-major matrix (valid for s >= q)
    std::vector<std::vector<S>> prop;
    std::vector<std::vector<S>> A;   // n matrices k*k, A[s][r*k + j]
    std::vector<std::vector<S>> W;   // 4*n*n matrices k*k, W[(v*n+q)*n+s][r*k + o] (q < s)

    const std::vector<S> &P(int v, int q, int s) const { return prop[(v * n + q) * n + s]; }
    const std::vector<S> &Wm(int v, int q, int s) const { return W[(v * n + q) * n + s]; }
};

inline bool variant_clamps(int v, bool causal) { return causal ? (v & 1) != 0 : (v & 2) != 0; }

// Memory position a tail entry of scan direction `causal` is read from.  An anticausal scan enters the image in the last
// tile, at its last EXISTING sample (memory position Tv - 1), and leaves it at position 0; its tail is positions 0 .. k-1.
// When the last tile has fewer samples than the order (Tv < k), the entries Tv .. k-1 are history from before the border:
// zero for a zero border (the buffer is zero there), and for a clamped border the scan's first output -- taps that reach
// before the first sample read that output (lib/recfilter.cpp:302-343) -- i.e. position Tv - 1.
inline int tail_position(int m, bool causal, int variant, int Tv, bool clamped) {
    if (!causal && (variant & 2) && clamped && m >= Tv) return Tv - 1;
    return m;
}

// T_last: number of samples of the dimension's last tile (== T unless the extent is not a multiple of T); the
// variants with bit1 set describe that tile.
template <typename S>
DimTables<S> build_dim_tables(const std::vector<ScanS<S>> &scans, int k, int T, bool clamped, int T_last = -1) {
    if (T_last < 0 || T_last > T) T_last = T;
    DimTables<S> t;
    t.T = T; t.k = k; t.n = (int)scans.size(); t.scans = scans;
    const int n = t.n;
    t.prop.assign(4 * n * n, {});
    t.W.assign(4 * n * n, {});
    t.A.assign(n, std::vector<S>(k * k, S(0)));
    std::vector<S> col(T);
    for (int v = 0; v < 4; v++) {
        const int Tv = (v & 2) ? T_last : T;      // samples that exist in a tile of this variant
        for (int q = 0; q < n; q++) {
            // R_q: response of an all-zero tile to a unit carry e_o
            std::vector<std::vector<S>> cols(k, std::vector<S>(T, S(0)));
            for (int o = 0; o < k; o++) {
                S carry[RF_MAX_ORDER];
                for (int j = 0; j < k; j++) carry[j] = (j == o) ? S(1) : S(0);
                scan_tile<S>(cols[o].data(), T, k, scans[q], false, carry, Tv);
            }
            for (int s = q; s < n; s++) {
                if (s > q) {
                    bool cl = clamped && variant_clamps(v, scans[s].causal);
                    for (int o = 0; o < k; o++) scan_tile<S>(cols[o].data(), T, k, scans[s], cl, nullptr, Tv);
                }
                std::vector<S> &P = t.prop[(v * n + q) * n + s];
                P.assign((size_t)T * k, S(0));
                for (int m = 0; m < T; m++)
                    for (int o = 0; o < k; o++) P[(size_t)m * k + o] = cols[o][m];
                // tail rows in the direction of scan s
                std::vector<S> tail(k * k);
                for (int r = 0; r < k; r++) {
                    int p = T - 1 - r;
                    int m = scans[s].causal ? p : T - 1 - p;
                    m = tail_position(m, scans[s].causal, v, Tv, clamped);
                    for (int o = 0; o < k; o++) tail[r * k + o] = cols[o][m];
                }
                if (s == q) { if (v == 0) t.A[q] = tail; }
                else t.W[(v * n + q) * n + s] = tail;
            }
        }
    }
    return t;
}

// Impulse responses of the tile-local tails: H[((v*n + s)*k + r)*T + m] = tail r of scan s (after the
// tile-local scans 0..s with zero incoming carries) per unit input at memory position m.
template <typename S>
std::vector<S> build_tail_responses(const std::vector<ScanS<S>> &scans, int k, int T, bool clamped, int T_last = -1) {
    const int n = (int)scans.size();
    if (T_last < 0 || T_last > T) T_last = T;
    std::vector<S> H((size_t)4 * n * k * T, S(0));
    std::vector<S> vec(T);
    for (int v = 0; v < 4; v++) {
        const int Tv = (v & 2) ? T_last : T;      // samples beyond Tv do not exist: their responses stay zero
        for (int m = 0; m < Tv; m++) {
            for (int i = 0; i < T; i++) vec[i] = (i == m) ? S(1) : S(0);
            for (int s = 0; s < n; s++) {
                scan_tile<S>(vec.data(), T, k, scans[s], clamped && variant_clamps(v, scans[s].causal), nullptr, Tv);
                for (int r = 0; r < k; r++) {
                    const int p = T - 1 - r;
                    const int mm = tail_position(scans[s].causal ? p : T - 1 - p, scans[s].causal, v, Tv, clamped);
                    H[(((size_t)v * n + s) * k + r) * T + m] = vec[mm];
                }
            }
        }
    }
    return H;
}

// k x k helpers ------------------------------------------------------------------------
template <typename S>
inline std::vector<S> mat_mul(const std::vector<S> &X, const std::vector<S> &Y, int k) {
    std::vector<S> Z(k * k, S(0));
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            S acc = S(0);
            for (int l = 0; l < k; l++) acc = acc + X[i * k + l] * Y[l * k + j];
            Z[i * k + j] = acc;
        }
    return Z;
}

template <typename S>
inline std::vector<S> mat_identity(int k) {
    std::vector<S> I(k * k, S(0));
    for (int i = 0; i < k; i++) I[i * k + i] = S(1);
    return I;
}

template <typename S>
inline std::vector<S> mat_pow(const std::vector<S> &X, long long e, int k) {
    std::vector<S> R = mat_identity<S>(k), B = X;
    while (e > 0) {
        if (e & 1) R = mat_mul(R, B, k);
        B = mat_mul(B, B, k);
        e >>= 1;
    }
    return R;
}

}  // namespace rf
