// plan_generic.h -- builds the step list of ONE dimension on the generic tiled path
// (pass 1, per-scan carry stages, pass 2).  Shared by plan.cpp (every dimension) and
// plan_fused.cpp (the z dimension of 3-D filters).
#pragma once

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "kernels_fused.h"
#include "plan.h"

namespace rf {

template <typename Acc>
DevScan<Acc> make_dev_scan(const Scan &s) {
    DevScan<Acc> d;
    d.causal = s.causal ? 1 : 0;
    d.order = s.order;
    if constexpr (std::is_same<Acc, uint32_t>::value) {
        d.b = (uint32_t)(int64_t)s.b;
        for (int j = 0; j < RF_MAX_ORDER; j++) d.a[j] = (uint32_t)(int64_t)s.a[j];
    } else {
        d.b = (Acc)s.b;
        for (int j = 0; j < RF_MAX_ORDER; j++) d.a[j] = (Acc)s.a[j];
    }
    return d;
}

template <typename S>
ScanS<S> make_table_scan(const Scan &s) {
    ScanS<S> t;
    t.causal = s.causal;
    if constexpr (std::is_same<S, uint64_t>::value) {
        t.b = (uint64_t)(int64_t)s.b;
        for (int j = 0; j < RF_MAX_ORDER; j++) t.a[j] = (uint64_t)(int64_t)s.a[j];
    } else {
        t.b = (S)s.b;
        for (int j = 0; j < RF_MAX_ORDER; j++) t.a[j] = (S)s.a[j];
        t.mod_n = s.mod_n;
        for (int j = 0; j < RF_MAX_ORDER; j++) t.mod_g[j] = (S)s.mod_g[j];
    }
    return t;
}

template <typename S, typename Acc>
Acc table_to_acc(S v) {
    if constexpr (std::is_same<Acc, uint32_t>::value) return (uint32_t)v;
    else return (Acc)v;
}

template <typename S>
double table_to_double(S v) {
    if constexpr (std::is_same<S, uint64_t>::value) return (double)(int64_t)v;
    else return (double)v;
}

// One scan as the fused / line kernels read it: coefficients plus, with_segment_tables, the tables of the 16-sample
// segment scan of the x phase (scan_device.h: R = effect of the entering state on the segment, in MEMORY order;
// P = segment exit-state transfer over 1, 2, 4, 8 segments) -- the tables of the causal twin at T = 16.
template <typename S, typename Acc>
FusedScan<Acc> make_fused_scan(const Scan &scan, int K, bool with_segment_tables) {
    FusedScan<Acc> f;
    std::memset(&f, 0, sizeof(f));
    ScanS<S> ts = make_table_scan<S>(scan);
    f.causal = ts.causal ? 1 : 0;
    f.b = table_to_acc<S, Acc>(ts.b);
    for (int j = 0; j < K && j < kFusedMaxK; j++) f.a[j] = table_to_acc<S, Acc>(ts.a[j]);
    f.mod_n = scan.mod_n;
    if constexpr (!std::is_same<S, uint64_t>::value) {
        for (int j = 0; j < kFusedMaxMod && j < RF_MAX_ORDER; j++) f.mod_g[j] = (Acc)ts.mod_g[j];
    }
    if (with_segment_tables) {
        ScanS<S> twin = ts;
        twin.causal = true;
        DimTables<S> seg = build_dim_tables<S>({twin}, K, kFusedSeg, false);
        const std::vector<S> &R = seg.P(0, 0, 0);
        for (int p = 0; p < kFusedSeg; p++)
            for (int j = 0; j < K; j++)     // direction position p -> memory position inside the segment
                f.R[j][ts.causal ? p : kFusedSeg - 1 - p] = table_to_acc<S, Acc>(R[(size_t)p * K + j]);
        std::vector<S> Pw = seg.A[0];
        for (int step = 0; step < 4; step++) {
            for (int r = 0; r < K; r++)
                for (int j = 0; j < K; j++) f.P[step][r][j] = table_to_acc<S, Acc>(Pw[r * K + j]);
            Pw = mat_mul<S>(Pw, Pw, K);
        }
    }
    return f;
}

// (A[s])^(i+1) for i = 0..M-1, flattened [s][i][r][j] in the kernels' arithmetic type (GenericDimArgs::Apow)
template <typename S, typename Acc>
std::vector<Acc> carry_apply_powers(const std::vector<std::vector<S>> &A, int64_t M, int k) {
    std::vector<Acc> out((size_t)A.size() * M * k * k);
    for (size_t s = 0; s < A.size(); s++) {
        std::vector<S> pw = A[s];
        for (int64_t i = 0; i < M; i++) {
            for (int e = 0; e < k * k; e++) out[((size_t)s * M + i) * k * k + e] = table_to_acc<S, Acc>(pw[e]);
            pw = mat_mul<S>(pw, A[s], k);
        }
    }
    return out;
}

// ---- merged exchange (one all-gather per sharded dimension) ------------------------------------------------
// With zero entering carries every slab completes ALL its scans locally; what the true entering carries in_q add
// afterwards is linear in them.  Y[q][s][t] (k x k, q <= s, t in memory order) is what a unit carry entering scan q
// adds to the completed tail of scan s at tile t: for q == s the plain propagation A_s^(i+1), for q < s the chaining
// terms W (create_tail_residual_term, lib/split.cpp:912-1004) followed by scan s's own recurrence.  The tables
// depend on which ends of the slab are image borders.  Flattened [q][s][t][r][o]; entries with q > s are zero.
template <typename S>
std::vector<S> cross_scan_transfer(const DimTables<S> &tab, int64_t M, bool first_is_border, bool last_is_border) {
    const int n = tab.n, k = tab.k, kk = k * k;
    std::vector<S> Y((size_t)n * n * M * kk, S(0));
    auto causal = [&](int s) { return tab.scans[s].causal; };
    auto variant = [&](int64_t t) { return ((t == 0 && first_is_border) ? 1 : 0) | ((t == M - 1 && last_is_border) ? 2 : 0); };
    std::vector<std::vector<S>> C(n, std::vector<S>((size_t)M * k));     // C[s][t*k + r]: effect on scan s, tile t
    for (int q = 0; q < n; q++)
        for (int o = 0; o < k; o++) {
            for (int s = q; s < n; s++) {
                std::vector<S> &Cs = C[s];
                std::fill(Cs.begin(), Cs.end(), S(0));
                if (s > q) {      // chaining on the scans q .. s-1 (their effects are already in C)
                    for (int64_t t = 0; t < M; t++) {
                        const int v = variant(t);
                        for (int qq = q; qq < s; qq++) {
                            const bool first = causal(qq) ? (t == 0) : (t == M - 1);
                            const std::vector<S> &Wm = tab.Wm(v, qq, s);
                            for (int r = 0; r < k; r++) {
                                S acc = S(0);
                                for (int j = 0; j < k; j++) {
                                    S c;
                                    if (first) c = (qq == q && j == o) ? S(1) : S(0);
                                    else c = C[qq][(size_t)(causal(qq) ? t - 1 : t + 1) * k + j];
                                    acc = acc + Wm[r * k + j] * c;
                                }
                                Cs[(size_t)t * k + r] = Cs[(size_t)t * k + r] + acc;
                            }
                        }
                    }
                }
                // scan s's own recurrence over the tiles; for s == q the entering state is the unit carry
                std::vector<S> x(k, S(0));
                if (s == q) x[o] = S(1);
                for (int64_t i = 0; i < M; i++) {
                    const int64_t t = causal(s) ? i : M - 1 - i;
                    for (int r = 0; r < k; r++) {
                        S acc = Cs[(size_t)t * k + r];
                        for (int j = 0; j < k; j++) acc = acc + tab.A[s][r * k + j] * x[j];
                        Cs[(size_t)t * k + r] = acc;
                    }
                    for (int r = 0; r < k; r++) x[r] = Cs[(size_t)t * k + r];
                }
                for (int64_t t = 0; t < M; t++)
                    for (int r = 0; r < k; r++)
                        Y[(((size_t)q * n + s) * M + t) * kk + r * k + o] = Cs[(size_t)t * k + r];
            }
        }
    return Y;
}

// A_s^(tiles of slab h) for every scan and every slab of the sharded dimension, [s][h][k x k]: what carries the state
// entering slab h to its exit (per-scan exchange; slabs may have different extents).
template <typename S, typename Acc>
std::vector<Acc> slab_powers(const rf_plan *plan, const std::vector<std::vector<S>> &A, int64_t T, int k) {
    const int n = (int)A.size(), world = plan->shard_world;
    std::vector<Acc> out((size_t)n * world * k * k, Acc(0));
    for (int s = 0; s < n; s++)
        for (int h = 0; h < world; h++) {
            std::vector<S> am = mat_pow<S>(A[s], plan->slab_tiles(h, T), k);
            for (int e = 0; e < k * k; e++) out[((size_t)s * world + h) * k * k + e] = table_to_acc<S, Acc>(am[e]);
        }
    return out;
}

inline bool merged_exchange_applies(int n_scans, int k, int world) {
    return world >= 1 && n_scans >= 1 && n_scans <= 4 && k >= 1 && k <= 3 && n_scans * world * k <= 128;
}

// Exchange structure of a sharded dimension with ONE all-gather: the local step completes every scan with zero
// entering carries and publishes all exit carries ([plane][s][r][line]); the apply step derives every scan's true
// entering carry from the gathered exits and corrects all tails in one pass.
// (T: the tile width, the same on every rank; slab h has plan->slab_tiles(h, T) tiles.)
template <typename S, typename Acc, typename ArgsFn>
int add_merged_exchange(rf_plan *plan, const DimTables<S> &tab, const std::string &dn, int64_t M, int64_t T, int64_t lines,
                        uint32_t causal_mask, ArgsFn gargs, Acc *incoming, size_t inc_pp, const Acc *d_AC, int C,
                        const std::string &carry_name, const Acc **apply_in_final_pass = nullptr) {
    int status = RF_OK;
    const int n = tab.n, K = tab.k, kk = K * K, np = plan->n_planes;
    const int world = plan->shard_world, rank = plan->shard_rank;
    // X[h]: the exit-tile rows of slab h's transfer (its own tile count and border type); Y: this slab's, every tile
    std::vector<Acc> hY, hX((size_t)world * n * n * kk, Acc(0));
    std::vector<double> dY, dX(hX.size(), 0.0);
    for (int h = 0; h < world; h++) {
        const int64_t Mh = plan->slab_tiles(h, T);
        std::vector<S> Y = cross_scan_transfer<S>(tab, Mh, h == 0, h == world - 1);
        for (int q = 0; q < n; q++)
            for (int s = q; s < n; s++) {
                const int64_t t_exit = tab.scans[s].causal ? Mh - 1 : 0;
                for (int e = 0; e < kk; e++) {
                    const S v = Y[(((size_t)q * n + s) * Mh + t_exit) * kk + e];
                    hX[(((size_t)h * n + q) * n + s) * kk + e] = table_to_acc<S, Acc>(v);
                    dX[(((size_t)h * n + q) * n + s) * kk + e] = table_to_double<S>(v);
                }
            }
        if (h == rank) {
            hY.resize(Y.size());
            dY.resize(Y.size());
            for (size_t e = 0; e < Y.size(); e++) { hY[e] = table_to_acc<S, Acc>(Y[e]); dY[e] = table_to_double<S>(Y[e]); }
        }
    }
    plan->tables["Y_" + dn] = dY;
    plan->tables["X_" + dn] = dX;
    const Acc *d_Y = (const Acc *)plan->upload(hY.data(), hY.size() * sizeof(Acc), &status);
    const Acc *d_X = (const Acc *)plan->upload(hX.data(), hX.size() * sizeof(Acc), &status);

    const int64_t plane_stride = (int64_t)n * K * lines, rank_stride = (int64_t)np * plane_stride;
    const int ex_index = (int)plan->exchanges.size();
    rf_plan::Exchange ex;
    ex.bytes = (size_t)rank_stride * sizeof(Acc);
    ex.scratch = plan->alloc(ex.bytes, true, &status);
    if (status != RF_OK) return status;
    ex.send = ex.scratch;
    ex.form_incoming = [plan, gargs, rank_stride, plane_stride, d_X](const void *gathered) {
        for (int pl = 0; pl < plan->n_planes; pl++) {
            int rc = launch_merged_gather<Acc>(gargs(pl), (const Acc *)gathered, rank_stride, pl * plane_stride,
                                               plan->shard_rank, plan->shard_world, d_X, plan->stream);
            if (rc) return rc;
        }
        return (int)RF_OK;
    };
    plan->exchanges.push_back(ex);

    // the local pass sees zero entering carries: a buffer of zeros of its own (the plan's `incoming` holds the true
    // carries of the previous execute until the gather overwrites them)
    const Acc *zeros = (const Acc *)plan->alloc(inc_pp * np * sizeof(Acc), true, &status);
    if (status != RF_OK) return status;
    Step cs;
    cs.name = carry_name;
    cs.run = [plan, gargs, K, n, causal_mask, d_AC, C, ex_index, plane_stride, zeros, inc_pp](int pl) {
        Acc *send = (Acc *)plan->exchanges[ex_index].send;
        auto a = gargs(pl);
        a.incoming = const_cast<Acc *>(zeros) + (size_t)pl * inc_pp;
        return launch_carry_block<Acc>(K, a, causal_mask, 0, n, send ? send + pl * plane_stride : nullptr, d_AC, C,
                                       plan->stream);
    };
    plan->exchange_local_steps.push_back({cs});
    if (apply_in_final_pass != nullptr) {
        // the caller's final pass adds Y * in as it loads a carry (fused pass 2): the tails are not rewritten
        *apply_in_final_pass = d_Y;
        plan->exchange_apply_steps.push_back({});
        return status;
    }
    Step ap;
    ap.name = carry_name + "_apply";
    ap.run = [plan, gargs, d_Y](int pl) { return launch_merged_apply<Acc>(gargs(pl), d_Y, plan->stream); };
    plan->exchange_apply_steps.push_back({ap});
    return status;
}

inline int pick_generic_tile(int64_t N, int k, int hint) {
    if (hint > 0 && hint <= kGenericMaxTile && N % hint == 0 && hint >= k) return hint;
    int cap = hint > 0 ? kGenericMaxTile : 64;
    for (int T = (int)std::min<int64_t>(cap, N); T >= std::max(k, 1); T--)
        if (N % T == 0) return T;
    return 0;
}


// Appends the steps of dimension d.  from_input: this is the first filtered dimension, its
// passes read the caller's input planes; otherwise they read (and overwrite) the output planes.
template <typename P, typename S>
int add_generic_dimension(rf_plan *plan, int tile_hint, int d, bool from_input_flag) {
    using Acc = typename PixelTraits<P>::Acc;
    int status = RF_OK;
    const int outer = plan->ndim - 1;
    const bool first_dim = from_input_flag;
    {
        DimInfo &di = plan->dims[d];
        int T = pick_generic_tile(plan->tile_basis(d), di.k, tile_hint);       // (sharded: every slab must tile alike)
        if (T == 0) {
            set_error("no tile width <= %d divides extent %lld of dimension %d", kGenericMaxTile, (long long)di.N, d);
            return RF_ERR_UNSUPPORTED;
        }
        di.T = T;
        di.M = di.N / T;
        const int n = (int)di.scan_ids.size();
        const int k = di.k;

        // tables
        std::vector<ScanS<S>> ts;
        std::vector<DevScan<Acc>> ds;
        for (int id : di.scan_ids) {
            ts.push_back(make_table_scan<S>(plan->scans[id]));
            DevScan<Acc> dv = make_dev_scan<Acc>(plan->scans[id]);
            dv.order = k;  // shorter scans are zero padded to the dimension's order (lib/split.cpp:575-578)
            ds.push_back(dv);
        }
        DimTables<S> tab = build_dim_tables<S>(ts, k, T, plan->clamped);
        std::vector<Acc> hW((size_t)4 * n * n * k * k, Acc(0)), hA((size_t)n * k * k, Acc(0));
        std::vector<double> dW(hW.size(), 0.0), dA(hA.size(), 0.0);
        for (int v = 0; v < 4; v++)
            for (int q = 0; q < n; q++)
                for (int s = q + 1; s < n; s++)
                    for (int e = 0; e < k * k; e++) {
                        size_t idx = (((size_t)v * n + q) * n + s) * k * k + e;
                        hW[idx] = table_to_acc<S, Acc>(tab.Wm(v, q, s)[e]);
                        dW[idx] = table_to_double<S>(tab.Wm(v, q, s)[e]);
                    }
        for (int s = 0; s < n; s++)
            for (int e = 0; e < k * k; e++) {
                hA[(size_t)s * k * k + e] = table_to_acc<S, Acc>(tab.A[s][e]);
                dA[(size_t)s * k * k + e] = table_to_double<S>(tab.A[s][e]);
            }
        std::string dn(1, "xyz"[d]);
        plan->tables["W_" + dn] = dW;
        plan->tables["A_" + dn] = dA;
        {
            std::vector<double> dP;
            for (int v = 0; v < 4; v++)
                for (int q = 0; q < n; q++)
                    for (int s = 0; s < n; s++) {
                        if (s >= q) for (S x : tab.P(v, q, s)) dP.push_back(table_to_double<S>(x));
                        else dP.insert(dP.end(), (size_t)T * k, 0.0);
                    }
            plan->tables["prop_" + dn] = dP;
        }

        // A^M for the exchange (sharded outermost dimension), A^C for the blocked carry scan
        const int C = carry_chunk_length(di.M, di.lines, k);
        std::vector<Acc> hAM = slab_powers<S, Acc>(plan, tab.A, T, k), hAC((size_t)n * k * k, Acc(0));
        uint32_t causal_mask = 0;
        for (int s = 0; s < n; s++) {
            std::vector<S> ac = mat_pow<S>(tab.A[s], C, k);
            for (int e = 0; e < k * k; e++) {
                hAC[(size_t)s * k * k + e] = table_to_acc<S, Acc>(ac[e]);
            }
            if (ts[s].causal) causal_mask |= 1u << s;
        }

        const DevScan<Acc> *dScans = (const DevScan<Acc> *)plan->upload(ds.data(), ds.size() * sizeof(DevScan<Acc>), &status);
        const Acc *dWp = (const Acc *)plan->upload(hW.data(), hW.size() * sizeof(Acc), &status);
        const Acc *dAp = (const Acc *)plan->upload(hA.data(), hA.size() * sizeof(Acc), &status);
        const Acc *dAMp = (const Acc *)plan->upload(hAM.data(), hAM.size() * sizeof(Acc), &status);
        const Acc *dACp = (const Acc *)plan->upload(hAC.data(), hAC.size() * sizeof(Acc), &status);
        const Acc *dApow = nullptr;
        if (d == outer && plan->sharded()) {
            std::vector<Acc> hApow = carry_apply_powers<S, Acc>(tab.A, di.M, k);
            dApow = (const Acc *)plan->upload(hApow.data(), hApow.size() * sizeof(Acc), &status);
        }
        size_t tails_per_plane = (size_t)n * di.M * k * di.lines;
        size_t inc_per_plane = (size_t)n * k * di.lines;
        Acc *tails = (Acc *)plan->alloc(tails_per_plane * plan->n_planes * sizeof(Acc), false, &status);
        Acc *incoming = (Acc *)plan->alloc(inc_per_plane * plan->n_planes * sizeof(Acc), true, &status);
        if (status != RF_OK) return status;

        const bool sharded_dim = (d == outer) && plan->sharded();
        GenericDimArgs<Acc> base{};
        base.g = LineGeom{di.N, di.stride, di.lines};
        base.T = T; base.M = (int32_t)di.M; base.k = k; base.n_scans = n;
        base.clamped = plan->clamped ? 1 : 0;
        base.first_is_border = (!sharded_dim || plan->shard_rank == 0) ? 1 : 0;
        base.last_is_border = (!sharded_dim || plan->shard_rank == plan->shard_world - 1) ? 1 : 0;
        base.scans = dScans; base.W = dWp; base.A = dAp; base.Apow = dApow;
        auto args_for = [base, tails, incoming, tails_per_plane, inc_per_plane](int pl) {
            GenericDimArgs<Acc> a = base;
            a.tails = tails + (size_t)pl * tails_per_plane;
            a.incoming = incoming + (size_t)pl * inc_per_plane;
            return a;
        };

        const bool from_input = first_dim;
        const bool is_exchange_dim = (d == outer);   // its carry stage is exposed through the stepping API

        Step p1;
        p1.name = "generic_pass1_" + dn;
        p1.run = [plan, args_for, from_input](int pl) {
            const P *src = from_input ? (const P *)plan->in[pl] : (const P *)plan->out[pl];
            return launch_generic_pass1<P>(src, args_for(pl), plan->stream);
        };
        plan->begin_steps.push_back(p1);

        const bool merged = is_exchange_dim && merged_exchange_applies(n, k, plan->shard_world);
        if (merged) {
            int rc = add_merged_exchange<S, Acc>(plan, tab, dn, di.M, T, di.lines, causal_mask, args_for, incoming, inc_per_plane,
                                                 dACp, C, "generic_carry_" + dn);
            if (rc != RF_OK) return rc;
        }
        for (int s = 0; s < n && !merged; s++) {
            int ex_index = -1;
            if (is_exchange_dim) {
                ex_index = (int)plan->exchanges.size();
                rf_plan::Exchange ex;
                ex.bytes = (size_t)plan->n_planes * k * di.lines * sizeof(Acc);
                ex.scratch = plan->alloc(ex.bytes, true, &status);
                if (status != RF_OK) return status;
                ex.send = ex.scratch;
                const Acc *AMs = dAMp + (size_t)s * plan->shard_world * k * k;      // [slab][k x k]
                int64_t rank_stride = (int64_t)plan->n_planes * k * di.lines;
                int64_t plane_stride = (int64_t)k * di.lines;
                ex.form_incoming = [plan, args_for, s, rank_stride, plane_stride, AMs](const void *gathered) {
                    for (int pl = 0; pl < plan->n_planes; pl++) {
                        int rc = launch_gather_incoming<Acc>(args_for(pl), s, (const Acc *)gathered, rank_stride,
                                                             pl * plane_stride, plan->shard_rank, plan->shard_world,
                                                             AMs, plan->stream);
                        if (rc) return rc;
                    }
                    return (int)RF_OK;
                };
                plan->exchanges.push_back(ex);
            }
            Step cs;
            cs.name = "generic_carry_" + dn + std::to_string(s);
            int64_t plane_stride = (int64_t)k * di.lines;
            // the blocked parallel scan of kernels_carry.hip: parallel over lines AND over chunks of tiles, so a 1-D
            // signal (one line) does not degenerate into one thread walking every tile
            cs.run = [plan, args_for, s, ex_index, plane_stride, k, causal_mask, dACp, C](int pl) {
                Acc *send = ex_index >= 0 ? (Acc *)plan->exchanges[ex_index].send : nullptr;
                if (k > kCarryBlockMaxOrder)        // (orders 9..32: one thread per line, kernels_generic.hip)
                    return launch_generic_carry_serial<Acc>(args_for(pl), causal_mask, s, s + 1, send ? send + pl * plane_stride : nullptr,
                                                            plan->stream);
                return launch_carry_block<Acc>(k, args_for(pl), causal_mask, s, s + 1, send ? send + pl * plane_stride : nullptr,
                                               dACp, C, plan->stream);
            };
            if (is_exchange_dim) {
                plan->exchange_local_steps.push_back({cs});
                Step ap;
                ap.name = "generic_carry_apply_" + dn + std::to_string(s);
                ap.run = [plan, args_for, s](int pl) { return launch_generic_carry_apply<Acc>(args_for(pl), s, plan->stream); };
                plan->exchange_apply_steps.push_back({ap});
            } else {
                plan->begin_steps.push_back(cs);
            }
        }

        Step p2;
        p2.name = "generic_pass2_" + dn;
        p2.run = [plan, args_for, from_input](int pl) {
            const P *src = from_input ? (const P *)plan->in[pl] : (const P *)plan->out[pl];
            return launch_generic_pass2<P>(src, (P *)plan->out[pl], args_for(pl), plan->stream);
        };
        if (is_exchange_dim) plan->finish_steps.push_back(p2);
        else plan->begin_steps.push_back(p2);
    }
    return status;
}

}  // namespace rf
