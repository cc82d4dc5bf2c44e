// plan_clamp1d.h -- clamped 1-D signals on the fused kernels.
//
// The fused path folds a long 1-D signal into rows that are chained through their entering states (plan_fused.cpp): a
// zero-border machine, because a clamped border would make ONE row of a tile differ from the others in every kernel.  A
// clamped scan, though, differs from the zero-border scan of the same signal by a term that is linear in ONE sample:
//     C_s(v) = Z_s(v) + v[b_s] * g_s            b_s = the border sample of scan s (0 for a causal scan, N-1 otherwise)
// (lib/recfilter.cpp:330-336: at r = 0 every tap reads the old f[b], from r = 1 on the taps beyond the border read the
// scan's first output; the difference to the zero border is a recurrence driven by v[b] alone).  g_s = C_s(e_b) - Z_s(e_b)
// decays like the filter's impulse response.  Unrolled over the scans s = 0..n-1 in application order:
//     out = Z_{n-1..0}(x) + sum_s beta_s * G_s,            G_s = Z_{n-1..s+1}(g_s)
//     beta_s = v_{s-1}[b_s] = w_s . x + sum_{q<s} beta_q * H[q][s]
//     w_s = Z_0^T .. Z_{s-1}^T e_{b_s}   (the transpose of a zero-border scan is the scan of opposite causality)
//     H[q][s] = (Z_{s-1..q+1} g_q)[b_s]  (zero across the two borders of a long signal)
// Everything on the right but x is a property of the FILTER: computed here, on the host, in double, on windows of L samples
// at the two ends of the signal (L = where every sequence has decayed below 1e-13 of its peak).  So a clamped 1-D signal is
// the zero-border fused plan plus two small launches: n dot products of the input's ends with the w_s (before the passes),
// and the rank-one corrections of the output's ends (behind them).  Filters that do not decay inside a quarter of the signal
// (a pole at 1: running sums; integer pixels) keep the generic path.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

#include "rf_internal.h"
#include "tables.h"

namespace rf {

struct Clamp1DTables {
    int n = 0;                    // scans
    int L = 0;                    // window length at either end of the signal
    std::vector<int32_t> side;    // [n]   0: the scan's border is sample 0, 1: sample N-1
    std::vector<double> w;        // [n][L]  window-local (index 0 = the window's first sample in memory)
    std::vector<double> G;        // [n][L]
    std::vector<double> H;        // [n][n]  H[q*n + s], q < s
};

inline bool build_clamp1d_tables(const std::vector<Scan> &scans, int64_t N, Clamp1DTables &t) {
    const int n = (int)scans.size();
    if (n < 1 || N < 4096) return false;
    int64_t Lw = std::min<int64_t>(N / 2, 1 << 16);          // work window
    Lw -= Lw % 256;
    if (Lw < 1024) return false;
    const int L0 = (int)Lw;
    auto ts = [&](int s) {
        ScanS<double> sc;
        sc.causal = scans[(size_t)s].causal;
        sc.b = scans[(size_t)s].b;
        for (int j = 0; j < RF_MAX_ORDER; j++) sc.a[j] = j < scans[(size_t)s].order ? scans[(size_t)s].a[j] : 0.0;
        return sc;
    };
    auto zero_scan = [&](std::vector<double> &v, int s, bool transpose) {
        ScanS<double> sc = ts(s);
        if (transpose) sc.causal = !sc.causal;
        scan_tile<double>(v.data(), L0, scans[(size_t)s].order, sc, false, nullptr);
    };
    t.n = n;
    t.side.assign((size_t)n, 0);
    for (int s = 0; s < n; s++) t.side[(size_t)s] = scans[(size_t)s].causal ? 0 : 1;
    auto border = [&](int s) { return t.side[(size_t)s] == 0 ? 0 : L0 - 1; };
    std::vector<std::vector<double>> w((size_t)n), G((size_t)n);
    t.H.assign((size_t)n * n, 0.0);
    for (int q = 0; q < n; q++) {
        // g_q = C_q(e_b) - Z_q(e_b)
        std::vector<double> c((size_t)L0, 0.0), z((size_t)L0, 0.0);
        c[(size_t)border(q)] = 1.0; z[(size_t)border(q)] = 1.0;
        scan_tile<double>(c.data(), L0, scans[(size_t)q].order, ts(q), true, nullptr);
        scan_tile<double>(z.data(), L0, scans[(size_t)q].order, ts(q), false, nullptr);
        for (int i = 0; i < L0; i++) c[(size_t)i] -= z[(size_t)i];
        for (int s = q + 1; s < n; s++) {
            if (t.side[(size_t)s] == t.side[(size_t)q]) t.H[(size_t)q * n + s] = c[(size_t)border(s)];
            zero_scan(c, s, false);
        }
        G[(size_t)q] = c;
        std::vector<double> v((size_t)L0, 0.0);
        v[(size_t)border(q)] = 1.0;
        for (int m = q - 1; m >= 0; m--) zero_scan(v, m, true);
        w[(size_t)q] = v;
    }
    // the window every sequence has decayed in: distance from its border beyond which it stays below 1e-13 of its peak
    int need = 1;
    for (int s = 0; s < n; s++)
        for (const std::vector<double> *seq : {&w[(size_t)s], &G[(size_t)s]}) {
            double peak = 0.0;
            for (double v : *seq) { if (!std::isfinite(v)) return false; peak = std::max(peak, std::fabs(v)); }
            if (peak == 0.0) continue;
            for (int d = L0 - 1; d >= 0; d--) {
                const int i = t.side[(size_t)s] == 0 ? d : L0 - 1 - d;
                if (std::fabs((*seq)[(size_t)i]) > 1e-13 * peak) { need = std::max(need, d + 1); break; }
            }
        }
    if (need > L0 / 2 || (int64_t)need * 4 > N) return false;     // not decayed: the two ends would see each other
    const int L = (need + 255) / 256 * 256;
    t.L = L;
    t.w.assign((size_t)n * L, 0.0);
    t.G.assign((size_t)n * L, 0.0);
    for (int s = 0; s < n; s++)
        for (int i = 0; i < L; i++) {
            const int src = t.side[(size_t)s] == 0 ? i : L0 - L + i;          // the L samples next to the border
            t.w[(size_t)s * L + i] = w[(size_t)s][(size_t)src];
            t.G[(size_t)s * L + i] = G[(size_t)s][(size_t)src];
        }
    return true;
}

// kernels_generic.hip
template <typename P>
int launch_clamp1d_dots(const P *in, int64_t N, int L, int n, const int32_t *side, const double *w, double *dots, hipStream_t stream);
template <typename P>
int launch_clamp1d_fix(P *out, int64_t N, int L, int n, const int32_t *side, const double *H, const double *G, const double *dots,
                       hipStream_t stream);

}  // namespace rf
