// plan_overlap.cpp -- plan of RF_PATH_TILED_OVERLAPPED: every tiled dimension in ONE pass 1 and ONE pass 2, the
// carries dimension by dimension with the cross-dimension residuals of lib/split.cpp:1215-1633 between them
// (kernels_overlap.hip).  Steps of one execute:
//     overlap_pass1, { [overlap_residual_d,] carry_d }  for d = x, y, z,  overlap_pass2
#include <algorithm>
#include <cstring>

#include "kernels_fused.h"       // launch_carry_block / carry_chunk_length
#include "kernels_overlap.h"
#include "plan.h"
#include "plan_generic.h"

namespace rf {

bool overlap_plan_applicable(const rf_plan *plan, const rf_filter_desc *desc, std::string *why) {
    auto no = [&](const char *msg) { if (why) *why = msg; return false; };
    if (plan->sharded()) return no("the overlapped path runs on one device");
    int filtered = 0;
    int64_t vol = 1;
    for (int d = 0; d < plan->ndim; d++) {
        const DimInfo &di = plan->dims[d];
        if (di.scan_ids.empty()) continue;
        filtered++;
        const int T = desc->tile[d];
        if (T <= 0) return no("every filtered dimension needs an explicit tile width (RecFilter::split)");
        if (di.N % T != 0) return no("tile width does not divide the extent");
        if (T < di.k) return no("tile narrower than the filter order");
        if (di.k > kOvMaxOrder) return no("orders above 8 run on the matrix / generic paths");
        vol *= T;
    }
    if (filtered < 1) return no("no scans");
    if (vol > kOvMaxTile) return no("tile volume above 4096 samples");
    return true;
}

namespace {

template <typename P, typename S>
int build_overlap(rf_plan *plan, const rf_filter_desc *desc) {
    using Acc = typename PixelTraits<P>::Acc;
    int status = RF_OK;
    OvArgs<Acc> base{};
    base.ndim = plan->ndim;
    base.clamped = plan->clamped ? 1 : 0;
    struct DimCarry {
        GenericDimArgs<Acc> g{};
        uint32_t causal_mask = 0;
        const Acc *AC = nullptr;
        int C = 1;
        size_t tails_pp = 0, inc_pp = 0;
        Acc *tails = nullptr, *incoming = nullptr;
    } carry[3];
    for (int d = 0; d < 3; d++) {
        OvDim<Acc> &od = base.d[d];
        od.N = d < plan->ndim ? plan->dims[d].N : 1;
        od.lines = d < plan->ndim ? plan->dims[d].lines : plan->total;
        od.T = 1; od.M = (int32_t)od.N; od.n = 0; od.k = 0;
        od.scans = nullptr; od.G = nullptr; od.tails = nullptr;
        if (d >= plan->ndim || plan->dims[d].scan_ids.empty()) {
            if (od.N >= (1ll << 31)) { set_error("overlapped path: extent too large"); return RF_ERR_UNSUPPORTED; }
            continue;
        }
        DimInfo &di = plan->dims[d];
        const int T = desc->tile[d], n = (int)di.scan_ids.size(), k = di.k;
        di.T = T; di.M = di.N / T;
        od.T = T; od.M = (int32_t)di.M; od.n = n; od.k = k;

        std::vector<ScanS<S>> ts;
        std::vector<DevScan<Acc>> ds;
        for (int id : di.scan_ids) {
            ts.push_back(make_table_scan<S>(plan->scans[id]));
            DevScan<Acc> dv = make_dev_scan<Acc>(plan->scans[id]);
            dv.order = k;            // shorter scans are zero padded to the dimension's order (lib/split.cpp:575-578)
            ds.push_back(dv);
        }
        DimTables<S> tab = build_dim_tables<S>(ts, k, T, plan->clamped);
        std::vector<Acc> hW((size_t)4 * n * n * k * k, Acc(0)), hA((size_t)n * k * k, Acc(0)), hG((size_t)4 * n * T * k, Acc(0));
        std::vector<double> dW(hW.size(), 0.0), dA(hA.size(), 0.0), dG(hG.size(), 0.0);
        for (int v = 0; v < 4; v++)
            for (int q = 0; q < n; q++) {
                for (int s = q + 1; s < n; s++)
                    for (int e = 0; e < k * k; e++) {
                        const size_t idx = (((size_t)v * n + q) * n + s) * k * k + e;
                        hW[idx] = table_to_acc<S, Acc>(tab.Wm(v, q, s)[e]);
                        dW[idx] = table_to_double<S>(tab.Wm(v, q, s)[e]);
                    }
                const std::vector<S> &Pm = tab.P(v, q, n - 1);          // [pos][o]: after ALL scans of the dimension
                for (size_t e = 0; e < (size_t)T * k; e++) {
                    hG[((size_t)v * n + q) * T * k + e] = table_to_acc<S, Acc>(Pm[e]);
                    dG[((size_t)v * n + q) * T * k + e] = table_to_double<S>(Pm[e]);
                }
            }
        for (int s = 0; s < n; s++)
            for (int e = 0; e < k * k; e++) {
                hA[(size_t)s * k * k + e] = table_to_acc<S, Acc>(tab.A[s][e]);
                dA[(size_t)s * k * k + e] = table_to_double<S>(tab.A[s][e]);
            }
        const std::string dn(1, "xyz"[d]);
        plan->tables["W_" + dn] = dW;
        plan->tables["A_" + dn] = dA;
        plan->tables["G_" + dn] = dG;

        DimCarry &c = carry[d];
        c.C = carry_chunk_length(di.M, di.lines, k);
        std::vector<Acc> hAC((size_t)n * k * k, Acc(0));
        for (int s = 0; s < n; s++) {
            std::vector<S> ac = mat_pow<S>(tab.A[s], c.C, k);
            for (int e = 0; e < k * k; e++) hAC[(size_t)s * k * k + e] = table_to_acc<S, Acc>(ac[e]);
            if (ts[s].causal) c.causal_mask |= 1u << s;
        }
        const DevScan<Acc> *dScans = (const DevScan<Acc> *)plan->upload(ds.data(), ds.size() * sizeof(DevScan<Acc>), &status);
        const Acc *dWp = (const Acc *)plan->upload(hW.data(), hW.size() * sizeof(Acc), &status);
        const Acc *dAp = (const Acc *)plan->upload(hA.data(), hA.size() * sizeof(Acc), &status);
        od.G = (const Acc *)plan->upload(hG.data(), hG.size() * sizeof(Acc), &status);
        c.AC = (const Acc *)plan->upload(hAC.data(), hAC.size() * sizeof(Acc), &status);
        od.scans = dScans;
        c.tails_pp = (size_t)n * di.M * k * di.lines;
        c.inc_pp = (size_t)n * k * di.lines;
        c.tails = (Acc *)plan->alloc(c.tails_pp * plan->n_planes * sizeof(Acc), false, &status);
        c.incoming = (Acc *)plan->alloc(c.inc_pp * plan->n_planes * sizeof(Acc), true, &status);     // zeros: image borders
        if (status != RF_OK) return status;
        c.g.g = LineGeom{di.N, di.stride, di.lines};
        c.g.T = T; c.g.M = (int32_t)di.M; c.g.k = k; c.g.n_scans = n;
        c.g.clamped = base.clamped; c.g.first_is_border = 1; c.g.last_is_border = 1;
        c.g.scans = dScans; c.g.W = dWp; c.g.A = dAp; c.g.Apow = nullptr;
    }
    // (dimensions without scans are "tiled" one index at a time: check the tile count fits)
    auto args_for = [base, carry](int pl) {
        OvArgs<Acc> a = base;
        for (int d = 0; d < 3; d++)
            if (a.d[d].n > 0) a.d[d].tails = carry[d].tails + (size_t)pl * carry[d].tails_pp;
        return a;
    };

    Step p1;
    p1.name = "overlap_pass1";
    p1.run = [plan, args_for](int pl) { return launch_overlap_pass1<P>((const P *)plan->in[pl], args_for(pl), plan->stream); };
    plan->begin_steps.push_back(p1);
    bool earlier = false;
    for (int d = 0; d < plan->ndim; d++) {
        if (base.d[d].n == 0) continue;
        const std::string dn(1, "xyz"[d]);
        if (earlier) {
            Step rs;
            rs.name = "overlap_residual_" + dn;
            rs.run = [plan, args_for, d](int pl) { return launch_overlap_residual<Acc>(args_for(pl), d, plan->stream); };
            plan->begin_steps.push_back(rs);
        }
        const DimCarry c = carry[d];
        Step cs;
        cs.name = "carry_" + dn;
        cs.run = [plan, c](int pl) {
            GenericDimArgs<Acc> g = c.g;
            g.tails = c.tails + (size_t)pl * c.tails_pp;
            g.incoming = c.incoming + (size_t)pl * c.inc_pp;
            return launch_carry_block<Acc>(g.k, g, c.causal_mask, 0, g.n_scans, (Acc *)nullptr, c.AC, c.C, plan->stream);
        };
        plan->begin_steps.push_back(cs);
        earlier = true;
    }
    Step p2;
    p2.name = "overlap_pass2";
    p2.run = [plan, args_for](int pl) {
        return launch_overlap_pass2<P>((const P *)plan->in[pl], (P *)plan->out[pl], args_for(pl), plan->stream);
    };
    plan->begin_steps.push_back(p2);
    return status;
}

}  // namespace

int build_overlap_plan(rf_plan *plan, const rf_filter_desc *desc) {
    switch (plan->dtype) {
        case RF_F32: return build_overlap<float, double>(plan, desc);
        case RF_F64: return build_overlap<double, double>(plan, desc);
        case RF_I32: return build_overlap<int32_t, uint64_t>(plan, desc);
        case RF_I16: return build_overlap<int16_t, uint64_t>(plan, desc);
    }
    set_error("overlapped path: unsupported pixel type");
    return RF_ERR_UNSUPPORTED;
}

}  // namespace rf
