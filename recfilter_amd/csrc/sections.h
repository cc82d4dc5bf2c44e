// sections.h -- a recursive scan of order 4..RF_MAX_ORDER as first/second/third-order scans with the same transfer
// function (host side; the plan rewrite that keeps such filters on the fused kernels).
//
// y[i] = b x[i] + sum_j a_j y[i-1-j] has the poles of z^k - a_0 z^(k-1) - ... - a_(k-1).  Conjugate pairs become second-order
// sections (2 Re p, -|p|^2); a pair takes a real pole along (third order) while there are any, the remaining real poles go
// three, two or one to a section -- as few scans as real coefficients allow.  Applied one after the other in the same direction with a ZERO
// border the sections reproduce the original scan up to rounding (the inverse of overlap_feedback_coeff,
// lib/iir_coeff.cpp:236-263).  The caller keeps the original when the recomposed denominator differs from it by more than
// rounding (clustered poles) or when the border is clamped (the clamped prologue of a high-order scan is not the
// composition of the sections' prologues).
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#include "rf_internal.h"

namespace rf {

// roots of the monic polynomial z^k + c[0] z^(k-1) + ... + c[k-1] (Durand-Kerner); false when it does not converge
inline bool monic_roots(const std::vector<double> &c, std::vector<std::complex<double>> &roots) {
    const int k = (int)c.size();
    double bound = 0.0;
    for (double v : c) bound = std::max(bound, std::fabs(v));
    bound = 1.0 + bound;                                          // Cauchy bound
    roots.resize(k);
    const std::complex<double> seed(0.4, 0.9);
    std::complex<double> p(1.0, 0.0);
    for (int i = 0; i < k; i++) { roots[i] = p * (0.5 * bound); p *= seed; }
    auto eval = [&](std::complex<double> z) {
        std::complex<double> v(1.0, 0.0);
        for (int i = 0; i < k; i++) v = v * z + c[i];
        return v;
    };
    for (int it = 0; it < 500; it++) {
        double step = 0.0;
        for (int i = 0; i < k; i++) {
            std::complex<double> den(1.0, 0.0);
            for (int j = 0; j < k; j++) if (j != i) den *= roots[i] - roots[j];
            if (std::abs(den) < 1e-300) return false;
            const std::complex<double> d = eval(roots[i]) / den;
            roots[i] -= d;
            step = std::max(step, std::abs(d));
        }
        if (step < 1e-15 * bound) return true;
    }
    return false;
}

// denominators are stored as feedback taps a[] (1 - a_0 z^-1 - a_1 z^-2 - ...); product of two of them
inline std::vector<double> multiply_feedback(const std::vector<double> &a, const std::vector<double> &b) {
    std::vector<double> pa(a.size() + 1, 1.0), pb(b.size() + 1, 1.0);
    for (size_t i = 0; i < a.size(); i++) pa[i + 1] = -a[i];
    for (size_t i = 0; i < b.size(); i++) pb[i + 1] = -b[i];
    std::vector<double> pc(pa.size() + pb.size() - 1, 0.0);
    for (size_t i = 0; i < pa.size(); i++)
        for (size_t j = 0; j < pb.size(); j++) pc[i + j] += pa[i] * pb[j];
    std::vector<double> out(pc.size() - 1);
    for (size_t i = 0; i + 1 < pc.size(); i++) out[i] = -pc[i + 1];
    return out;
}

// `cast`: rounding of a coefficient through the pixel type.  Sections of order <= max_order, first one carries b.
template <typename Cast>
bool split_into_sections(const Scan &s, int max_order, Cast cast, std::vector<Scan> &out) {
    const int k = s.order;
    std::vector<double> c(k);
    for (int i = 0; i < k; i++) c[i] = -s.a[i];
    std::vector<std::complex<double>> roots;
    if (!monic_roots(c, roots)) return false;
    std::vector<std::vector<double>> quad;
    std::vector<bool> used(k, false);
    std::vector<double> reals;
    for (int i = 0; i < k; i++) {
        if (used[i]) continue;
        const double tol = 1e-9 * std::max(1.0, std::abs(roots[i]));
        if (std::fabs(roots[i].imag()) <= tol) { reals.push_back(roots[i].real()); used[i] = true; continue; }
        int best = -1;                                            // its conjugate
        for (int j = 0; j < k; j++)
            if (!used[j] && j != i && (best < 0 || std::abs(roots[j] - std::conj(roots[i])) < std::abs(roots[best] - std::conj(roots[i])))) best = j;
        if (best < 0 || std::abs(roots[best] - std::conj(roots[i])) > 1e-6 * std::max(1.0, std::abs(roots[i]))) return false;
        const std::complex<double> p = 0.5 * (roots[i] + std::conj(roots[best]));
        quad.push_back({2.0 * p.real(), -std::norm(p)});
        used[i] = used[best] = true;
    }
    // pack into as few sections of order <= max_order as real coefficients allow: a conjugate pair takes a real pole along
    // (third order) while there are any; the real poles left over go three, two or one to a section
    std::sort(reals.begin(), reals.end(), [](double x, double y) { return std::fabs(x) > std::fabs(y); });
    std::vector<std::vector<double>> sections;
    size_t r = 0;
    for (auto &q : quad) {
        if (max_order >= 3 && r < reals.size()) sections.push_back(multiply_feedback(q, {reals[r++]}));
        else sections.push_back(q);
    }
    while (r < reals.size()) {
        const size_t take = std::min<size_t>(reals.size() - r, (size_t)std::min(max_order, 3));
        std::vector<double> sec = {reals[r]};
        for (size_t i = 1; i < take; i++) sec = multiply_feedback(sec, {reals[r + i]});
        sections.push_back(sec);
        r += take;
    }
    // fewer, larger sections while they fit: 2 + 2 + 2 stays, but nothing larger than max_order is built
    std::vector<double> recomposed;
    out.clear();
    for (size_t i = 0; i < sections.size(); i++) {
        if ((int)sections[i].size() > max_order) return false;
        Scan t = s;
        t.order = (int)sections[i].size();
        t.b = i == 0 ? s.b : cast(1.0);
        for (int j = 0; j < RF_MAX_ORDER; j++) t.a[j] = j < t.order ? cast(sections[i][j]) : 0.0;
        out.push_back(t);
        std::vector<double> rounded(t.a, t.a + t.order);
        recomposed = recomposed.empty() ? rounded : multiply_feedback(recomposed, rounded);
    }
    if ((int)recomposed.size() != k) return false;
    double scale = 1.0, err = 0.0;
    for (int i = 0; i < k; i++) { scale = std::max(scale, std::fabs(s.a[i])); err = std::max(err, std::fabs(recomposed[i] - s.a[i])); }
    return err <= 2e-6 * scale;                                   // sections rounded to the pixel type still give this filter
}

// Conditioning of a cascade of sections in the pixel's arithmetic.  The recomposed denominator may match to rounding while the
// cascade is still a poor way to EVALUATE the filter: poles spread over a circle make every section a sharp resonator whose
// gain the next one cancels (the dummy polynomial of apps/audio/audio_filter_high_order.cpp:41-42 -- all feedback 0.01 --
// loses 5e-5 at order 15 in f32).  So the plan tries it: a probe signal (an impulse plus a fixed pseudo-random sequence)
// through the sections in float arithmetic against the direct form in double; the sections are kept only if they stay
// within `bar` of it (relative to the peak).  f64 pixels pass by construction (the probe would measure 1e-13).
inline bool sections_well_conditioned(const Scan &direct, const std::vector<Scan> &sections, double bar = 2e-5) {
    const int L = 512;
    std::vector<double> x(L), ref(L);
    uint32_t lcg = 12345u;
    for (int i = 0; i < L; i++) { lcg = lcg * 1664525u + 1013904223u; x[i] = (double)(lcg >> 8) / 16777216.0 - 0.5; }
    x[0] += 1.0;
    for (int i = 0; i < L; i++) {
        double acc = direct.b * x[i];
        for (int j = 0; j < direct.order && j < i; j++) acc += direct.a[j] * ref[i - 1 - j];
        ref[i] = acc;
    }
    std::vector<float> cur(L);
    for (int i = 0; i < L; i++) cur[i] = (float)x[i];
    for (const Scan &sec : sections) {
        std::vector<float> nxt(L);
        for (int i = 0; i < L; i++) {
            float acc = (float)sec.b * cur[i];
            for (int j = 0; j < sec.order && j < i; j++) acc = std::fmaf((float)sec.a[j], nxt[i - 1 - j], acc);
            nxt[i] = acc;
        }
        cur.swap(nxt);
    }
    double peak = 0.0, err = 0.0;
    for (int i = 0; i < L; i++) {
        if (!std::isfinite(ref[i]) || !std::isfinite((double)cur[i])) return false;
        peak = std::max(peak, std::fabs(ref[i]));
        err = std::max(err, std::fabs((double)cur[i] - ref[i]));
    }
    return err <= bar * std::max(peak, 1e-30);
}

// The other direction (plan.cpp, "merged runs"): consecutive scans of one direction with a ZERO border are one scan whose
// transfer function is the product of theirs (overlap_feedback_coeff, lib/iir_coeff.cpp:236-263; the reference's
// overlap_to_higher_order_filter, lib/reorder.cpp:231-381, does this on request).  Whether the direct form of the product,
// evaluated in f32, is a good way to compute the cascade is again tried, not assumed: the cascade in double against the merged
// scan in float on the probe signal.
// The probe runs until the cascade's response has DIED OUT (round 6): an impulse plus 768 samples of noise, then
// silence, both forms evaluated until the double-precision response has stayed below 1e-7 of its peak for 256 samples.  A cascade
// with slow poles (time constants of thousands of samples: audio biquads near DC) is then compared over its whole memory --
// rounding the product polynomial to f32 moves clustered poles, and the drift shows only after many time constants, which a
// fixed 768-sample window passed.  A response that has not died out after 2^17 samples is not merged.
inline bool merged_well_conditioned(const std::vector<Scan> &cascade, const Scan &merged, double bar = 2e-5) {
    const int L0 = 768, Lmax = 1 << 17, quiet_needed = 256;
    uint32_t lcg = 4242u;
    auto input = [&](int i) -> double {
        if (i >= L0) return 0.0;
        lcg = lcg * 1664525u + 1013904223u;
        return (double)(lcg >> 8) / 16777216.0 - 0.5 + (i == 0 ? 1.0 : 0.0);
    };
    // running state of every scan of the cascade (double) and of the merged scan (float): the last `order` outputs, newest first
    std::vector<std::vector<double>> hist(cascade.size());
    for (size_t q = 0; q < cascade.size(); q++) hist[q].assign((size_t)std::max(cascade[q].order, 1), 0.0);
    std::vector<float> hist32((size_t)std::max(merged.order, 1), 0.0f);
    double peak = 0.0, err = 0.0;
    int quiet = 0;
    for (int i = 0; i < Lmax; i++) {
        const double x = input(i);
        double v = x;
        for (size_t q = 0; q < cascade.size(); q++) {
            const Scan &sc = cascade[q];
            double acc = sc.b * v;
            for (int jj = 0; jj < sc.order; jj++) acc += sc.a[jj] * hist[q][(size_t)jj];
            for (int jj = sc.order - 1; jj > 0; jj--) hist[q][(size_t)jj] = hist[q][(size_t)jj - 1];
            hist[q][0] = acc;
            v = acc;
        }
        float acc32 = (float)merged.b * (float)x;
        for (int jj = 0; jj < merged.order; jj++) acc32 = std::fmaf((float)merged.a[jj], hist32[(size_t)jj], acc32);
        for (int jj = merged.order - 1; jj > 0; jj--) hist32[(size_t)jj] = hist32[(size_t)jj - 1];
        hist32[0] = acc32;
        if (!std::isfinite(v) || !std::isfinite((double)acc32)) return false;
        peak = std::max(peak, std::fabs(v));
        err = std::max(err, std::fabs((double)acc32 - v));
        if (err > bar * std::max(peak, 1e-30) && i >= L0) return false;       // (already beyond the bar: no need to go on)
        if (i >= L0) {
            quiet = std::fabs(v) <= 1e-7 * peak ? quiet + 1 : 0;
            if (quiet >= quiet_needed) return err <= bar * std::max(peak, 1e-30);
        }
    }
    return false;           // the response outlives the probe: keep the scans as given
}

}  // namespace rf
