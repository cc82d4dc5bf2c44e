// plan_strided.h -- step list of ONE strided dimension on the register-column path
// (kernels_strided.hip): pass 1, blocked carry scan, pass 2.  Used by plan_fused.cpp for the z
// dimension of 3-D filters.  Same exchange structure as plan_generic.h for the sharded dimension.
#pragma once

#include <cstring>

#include "kernels_fused.h"
#include "plan.h"
#include "plan_generic.h"

namespace rf {

inline int strided_tile(const rf_plan *plan, int d) {
    const DimInfo &di = plan->dims[d];
    if (di.scan_ids.empty() || di.k > kFusedMaxK || (int)di.scan_ids.size() > kFusedMaxScans) return 0;
    if (plan->dtype != RF_F32 && plan->dtype != RF_I32 && plan->dtype != RF_I16) return 0;
    const int64_t basis = plan->tile_basis(d);       // sharded dimension: every rank's slab must tile alike
    if (const int want = plan->strided_tile_planes() ? plan->strided_tile_planes() : RF_KNOB("RF_STRIDED_TZ") ? atoi(RF_KNOB("RF_STRIDED_TZ")) : 0) {
        // RF_PLAN_TILE_PLANES(n): the caller's tile width of the strided stage, where it divides the extent
        if ((want == 32 || want == 64 || want == 128) && basis % want == 0) return want;
    }
    // large volumes: 128 samples per thread halve the tails of this dimension and its carry scan (2048^3: carry_z 1.21 ->
    // 0.53 ms, the two passes unchanged within their run-to-run spread)
    if (basis % 128 == 0 && di.lines % 256 == 0 && di.stride % 256 == 0 && di.lines * di.N >= (int64_t)1 << 28) return 128;
    if (basis % 64 == 0) return 64;
    if (basis % 32 == 0) return 32;
    return 0;
}

// What pass 1 of the x/y stage needs to know when it also forms this dimension's tails (kernels_tails_walk.hip): the strided
// dimension then has no first pass of its own.  Filled by add_strided_dimension.
struct WalkHook {
    float *zt = nullptr;          // the dimension's tails, [s][t][r][line] (plane 0)
    size_t zt_stride = 0;         // elements between the tails of consecutive Tuple planes
};

// The x/y filter F over `planes` carry planes of dimension d, in place: a fused x/y plan of its own (null: cannot be built).
inline rf_plan *build_carry_planes_plan(const rf_plan *plan, const rf_filter_desc *desc, int d, int64_t planes) {
    std::vector<rf_scan_desc> xy;
    for (int i = 0; i < desc->n_scans; i++)
        if (desc->scans[i].dim != d) xy.push_back(desc->scans[i]);
    rf_filter_desc cd = *desc;
    cd.scans = xy.data();
    cd.n_scans = (int32_t)xy.size();
    cd.extent[2] = planes;
    cd.n_planes = 1;
    cd.tile[2] = 0;
    std::memset(&cd.pointwise, 0, sizeof(cd.pointwise));      // (F alone: an epilogue of the filter is not part of it)
    cd.path = RF_PATH_TILED_FUSED;
    cd.device = plan->host_only ? RF_DEVICE_HOST_ONLY : plan->device;
    cd.shard_rank = 0; cd.shard_world = 1; cd.shard_extents = nullptr;
    cd.flags = (desc->flags & (RF_PLAN_STREAM_PASS1 | RF_PLAN_STAGED_PASS1 | 0x0000ff00u)) | RF_PLAN_TILED_ONLY | RF_PLAN_NO_CASCADE;
    rf_plan *child = nullptr;
    if (build_plan(&cd, &child) != RF_OK) return nullptr;
    return child;
}

// Whether dimension d of a sharded plan takes the early exchange (below), before the helper plan has been tried.
template <typename P>
inline bool early_exchange_possible(const rf_plan *plan, int d, const rf_filter_desc *desc) {
    using Acc = typename PixelTraits<P>::Acc;
    const DimInfo &di = plan->dims[d];
    return d == plan->ndim - 1 && plan->sharded() && desc != nullptr && d == 2 &&
           merged_exchange_applies((int)di.scan_ids.size(), di.k, plan->shard_world) && !plan->pw.pre && !plan->pw.post &&
           !plan->pw.in_u8 && !(plan->flags & RF_PLAN_LATE_EXCHANGE) && di.lines == plan->dims[0].N * plan->dims[1].N &&
           sizeof(P) == sizeof(Acc);      // (the carry planes are filtered as pixels: f32 / i32)
}

// Early exchange (a z-sharded volume whose x/y stage precedes this dimension).  The operators of this dimension -- tail
// extraction, carry recurrence, the correction by the entering carries -- act along z alone and identically on every
// (x, y) line; the x/y filter F acts on every z plane alone and identically: they commute, borders included (everything
// is linear).  So the carries of the x/y-FILTERED volume are F applied to the carry planes of the RAW volume:
//     begin      pass 1 of this dimension on the raw input, slab-local carries, exit carries -> send
//     <all-gather>   ||   interior: the whole x/y stage (what rf_plan_interior runs beside the collective)
//     apply      entering carries from the gathered exits, correction of the (raw) tails, then F over the k * scans *
//                (tiles + 1) carry planes, in place -- a fused x/y plan of its own over those few planes
//     finish     pass 2 of this dimension on the x/y-filtered output with the filtered carries
// The exchange no longer waits for the x/y stage and the x/y stage no longer waits for the exchange: a rank's step is
// max(kernels, exchange) instead of their sum, for (tiles + 1) * scans * k / planes of extra x/y work (cfg5 on 8 GPUs:
// 12 carry planes beside 256).  `xy_begin` .. end of plan->begin_steps are the x/y stage's steps at the time of the call.
// Needs: the merged exchange, no pointwise stages (a prologue's bias is not linear), the x/y stage in front.
// `walk` (unsharded volumes, kernels_tails_walk.hip): the same commutation without an exchange -- pass 1 of the x/y stage has
// formed this dimension's tails from the raw input as it went, so the dimension is: carry scan, F over the scans * k * tiles
// carry planes (`walk_child`, built by the caller), pass 2 on the x/y-filtered output.
template <typename P, typename S>
int add_strided_dimension(rf_plan *plan, int d, bool from_input, const rf_filter_desc *desc = nullptr, size_t xy_begin = (size_t)-1,
                          WalkHook *walk = nullptr, rf_plan *walk_child = nullptr) {
    using Acc = typename PixelTraits<P>::Acc;
    int status = RF_OK;
    DimInfo &di = plan->dims[d];
    const int TZ = strided_tile(plan, d);
    if (TZ == 0) { set_error("strided path not applicable to dimension %d", d); return RF_ERR_UNSUPPORTED; }
    di.T = TZ;
    di.M = di.N / TZ;
    const int n = (int)di.scan_ids.size(), K = di.k, M = (int)di.M;
    const int outer = plan->ndim - 1;
    const bool sharded = (d == outer) && plan->sharded();
    const int np = plan->n_planes;
    std::string dn(1, "xyz"[d]);

    std::vector<ScanS<S>> ts;
    std::vector<DevScan<Acc>> ds;
    StridedArgs<Acc> base{};
    std::memset(base.scans, 0, sizeof(base.scans));
    uint32_t mask = 0;
    for (int i = 0; i < n; i++) {
        const Scan &sc = plan->scans[di.scan_ids[i]];
        ScanS<S> t = make_table_scan<S>(sc);
        ts.push_back(t);
        DevScan<Acc> dv = make_dev_scan<Acc>(sc);
        dv.order = K;
        ds.push_back(dv);
        base.scans[i].causal = t.causal ? 1 : 0;
        base.scans[i].b = table_to_acc<S, Acc>(t.b);
        for (int j = 0; j < kFusedMaxK; j++) base.scans[i].a[j] = j < K ? table_to_acc<S, Acc>(t.a[j]) : Acc(0);
        base.scans[i].mod_n = sc.mod_n;
        if constexpr (!std::is_same<S, uint64_t>::value) {
            for (int j = 0; j < kFusedMaxMod && j < RF_MAX_ORDER; j++) base.scans[i].mod_g[j] = (Acc)t.mod_g[j];
        }
        if (t.causal) mask |= 1u << i;
    }
    DimTables<S> tab = build_dim_tables<S>(ts, K, TZ, plan->clamped);
    const int C = carry_chunk_length(M, di.lines, K);
    std::vector<Acc> hW((size_t)4 * n * n * K * K, Acc(0)), hA((size_t)n * K * K), hAC(hA.size()), hAM(hA.size());
    std::vector<double> dW(hW.size(), 0.0), dA(hA.size(), 0.0);
    for (int v = 0; v < 4; v++)
        for (int q = 0; q < n; q++)
            for (int s = q + 1; s < n; s++)
                for (int e = 0; e < K * K; e++) {
                    size_t idx = (((size_t)v * n + q) * n + s) * K * K + e;
                    hW[idx] = table_to_acc<S, Acc>(tab.Wm(v, q, s)[e]);
                    dW[idx] = table_to_double<S>(tab.Wm(v, q, s)[e]);
                }
    for (int s = 0; s < n; s++) {
        std::vector<S> ac = mat_pow<S>(tab.A[s], C, K);
        for (int e = 0; e < K * K; e++) {
            hA[(size_t)s * K * K + e] = table_to_acc<S, Acc>(tab.A[s][e]);
            dA[(size_t)s * K * K + e] = table_to_double<S>(tab.A[s][e]);
            hAC[(size_t)s * K * K + e] = table_to_acc<S, Acc>(ac[e]);
        }
    }
    plan->tables["W_" + dn] = dW;
    plan->tables["A_" + dn] = dA;

    const DevScan<Acc> *d_scans = (const DevScan<Acc> *)plan->upload(ds.data(), ds.size() * sizeof(DevScan<Acc>), &status);
    const Acc *d_W = (const Acc *)plan->upload(hW.data(), hW.size() * sizeof(Acc), &status);
    const Acc *d_A = (const Acc *)plan->upload(hA.data(), hA.size() * sizeof(Acc), &status);
    const Acc *d_AC = (const Acc *)plan->upload(hAC.data(), hAC.size() * sizeof(Acc), &status);
    hAM = slab_powers<S, Acc>(plan, tab.A, TZ, K);                 // [s][slab][K x K]
    const Acc *d_AM = (const Acc *)plan->upload(hAM.data(), hAM.size() * sizeof(Acc), &status);
    const Acc *d_Apow = nullptr;
    if (sharded) {
        std::vector<Acc> hApow = carry_apply_powers<S, Acc>(tab.A, M, K);
        d_Apow = (const Acc *)plan->upload(hApow.data(), hApow.size() * sizeof(Acc), &status);
    }
    const size_t tails_pp = (size_t)n * M * K * di.lines, inc_pp = (size_t)n * K * di.lines;
    bool early = sharded && !from_input && xy_begin != (size_t)-1 && early_exchange_possible<P>(plan, d, desc);
    // The early exchange needs a helper plan: F over the carry planes = the x/y scans of this filter on a volume of
    // (tiles + 1) * scans * k planes, in place.  It is built BEFORE anything of the early layout is committed: a helper that
    // cannot be built (unsupported shape, out of memory) leaves the plan on the late exchange instead of failing it.
    // (a sharded plan whose pass 1 walks has had both checked by its builder: its helper is `walk_child`)
    rf_plan *child = walk ? walk_child : nullptr;
    if (walk && sharded && !early) { set_error("one-read pass 1 of a sharded volume needs the early exchange"); return RF_ERR_UNSUPPORTED; }
    if (early && !walk) {
        child = build_carry_planes_plan(plan, desc, d, (int64_t)n * K * (M + 1));
        if (!child) early = false;
    }
    std::unique_ptr<rf_plan> child_owner(child);
    // early exchange: the tails and the entering carries of a plane are ONE run of (tiles + 1) * scans * k carry planes,
    // which the x/y filter then takes as a volume of that many z planes
    const size_t chunk_pp = early ? tails_pp + inc_pp : 0;
    Acc *tails, *incoming;
    size_t tails_stride = tails_pp, inc_stride = inc_pp;
    if (early) {
        tails = (Acc *)plan->alloc(chunk_pp * np * sizeof(Acc), true, &status);
        incoming = tails ? tails + tails_pp : nullptr;
        tails_stride = inc_stride = chunk_pp;
    } else {
        tails = (Acc *)plan->alloc(tails_pp * np * sizeof(Acc), false, &status);
        incoming = (Acc *)plan->alloc(inc_pp * np * sizeof(Acc), true, &status);
    }
    if (status != RF_OK) return status;

    base.n = di.N; base.inner = di.stride; base.lines = di.lines; base.M = M; base.n_scans = n;
    base.clamped = plan->clamped ? 1 : 0;
    base.mod_form = plan->mod_form ? 1 : 0;
    base.first_is_border = (!sharded || plan->shard_rank == 0) ? 1 : 0;
    base.last_is_border = (!sharded || plan->shard_rank == plan->shard_world - 1) ? 1 : 0;
    auto sargs = [=](int pl) {
        StridedArgs<Acc> a = base;
        a.tails = tails + (size_t)pl * tails_stride;
        a.incoming = incoming + (size_t)pl * inc_stride;
        return a;
    };
    GenericDimArgs<Acc> gb{};
    gb.g = LineGeom{di.N, di.stride, di.lines};
    gb.T = TZ; gb.M = M; gb.k = K; gb.n_scans = n; gb.clamped = base.clamped;
    gb.first_is_border = base.first_is_border; gb.last_is_border = base.last_is_border;
    gb.scans = d_scans; gb.W = d_W; gb.A = d_A; gb.Apow = d_Apow;
    auto gargs = [=](int pl) {
        GenericDimArgs<Acc> a = gb;
        a.tails = tails + (size_t)pl * tails_stride;
        a.incoming = incoming + (size_t)pl * inc_stride;
        return a;
    };

    Step p1;
    p1.name = "strided_pass1_" + dn;
    if (walk) { walk->zt = reinterpret_cast<float *>(tails); walk->zt_stride = tails_stride; }
    p1.run = [plan, sargs, K, TZ, from_input, early](int pl) {
        const P *src = (from_input || early) ? (const P *)plan->in[pl] : (const P *)plan->xy_result(pl);
        return launch_strided_pass<P>(false, K, TZ, src, (P *)plan->out[pl], sargs(pl), plan->stream);
    };
    if (early) {
        // the x/y stage leaves the begin phase: it is what runs beside the all-gather (all of it but a pass 1 that also forms
        // this dimension's tails, which the exchange waits for)
        const size_t keep = xy_begin + (walk ? 1 : 0);
        plan->interior_steps.assign(plan->begin_steps.begin() + (std::ptrdiff_t)keep, plan->begin_steps.end());
        plan->begin_steps.resize(keep);
    }
    if (!walk) plan->begin_steps.push_back(p1);

    if (!sharded) {
        Step cs;
        cs.name = "carry_" + dn;
        cs.run = [plan, gargs, K, n, d_AC, C, mask](int pl) {
            return launch_carry_block<Acc>(K, gargs(pl), mask, 0, n, (Acc *)nullptr, d_AC, C, plan->stream);
        };
        plan->begin_steps.push_back(cs);
        if (walk) {
            plan->helpers.emplace_back(child_owner.release());
            plan->workspace_bytes += child->workspace_bytes;
            std::vector<const Step *> steps;
            for (const Step &st : child->begin_steps) steps.push_back(&st);
            for (const Step &st : child->finish_steps) steps.push_back(&st);
            Step w;
            w.name = "carry_planes_xy";
            w.run = [plan, child, steps, tails, tails_stride](int pl) {
                child->in[0] = child->orig_in[0] = tails + (size_t)pl * tails_stride;
                child->out[0] = tails + (size_t)pl * tails_stride;
                child->stream = plan->stream;
                for (const Step *sp : steps) {
                    const int rc = sp->run(0);
                    if (rc != RF_OK) return rc;
                }
                return (int)RF_OK;
            };
            plan->begin_steps.push_back(w);
        }
    } else if (merged_exchange_applies(n, K, plan->shard_world)) {
        int rc = add_merged_exchange<S, Acc>(plan, tab, dn, M, TZ, di.lines, mask, gargs, incoming, inc_pp, d_AC, C, "carry_" + dn);
        if (rc != RF_OK) return rc;
        if (early) {
            plan->helpers.emplace_back(child_owner.release());
            plan->workspace_bytes += child->workspace_bytes;
            // ONE step: the helper has one workspace, so its launches for a plane run back to back (the steps of an execute
            // run plane by plane inside every step)
            std::vector<const Step *> steps;
            for (const Step &s : child->begin_steps) steps.push_back(&s);
            for (const auto &ex : child->exchange_local_steps)
                for (const Step &s : ex) steps.push_back(&s);
            for (const Step &s : child->finish_steps) steps.push_back(&s);
            Step w;
            w.name = "carry_planes_xy";
            w.run = [plan, child, steps, tails, chunk_pp](int pl) {
                // the helper's context: this plane's run of carry planes, filtered in place
                child->in[0] = child->orig_in[0] = tails + (size_t)pl * chunk_pp;
                child->out[0] = tails + (size_t)pl * chunk_pp;
                child->stream = plan->stream;
                for (const Step *sp : steps) {
                    const int rc = sp->run(0);
                    if (rc != RF_OK) return rc;
                }
                return (int)RF_OK;
            };
            plan->exchange_apply_steps.back().push_back(w);
        }
    } else {
        for (int s = 0; s < n; s++) {
            const int64_t plane_stride = (int64_t)K * di.lines, rank_stride = (int64_t)np * K * di.lines;
            const int ex_index = (int)plan->exchanges.size();
            rf_plan::Exchange ex;
            ex.bytes = (size_t)np * K * di.lines * sizeof(Acc);
            ex.scratch = plan->alloc(ex.bytes, true, &status);
            if (status != RF_OK) return status;
            ex.send = ex.scratch;
            const Acc *AMs = d_AM + (size_t)s * plan->shard_world * K * K;
            ex.form_incoming = [plan, gargs, s, rank_stride, plane_stride, AMs](const void *gathered) {
                for (int pl = 0; pl < plan->n_planes; pl++) {
                    int rc = launch_gather_incoming<Acc>(gargs(pl), s, (const Acc *)gathered, rank_stride, pl * plane_stride,
                                                         plan->shard_rank, plan->shard_world, AMs, plan->stream);
                    if (rc) return rc;
                }
                return (int)RF_OK;
            };
            plan->exchanges.push_back(ex);
            Step cs;
            cs.name = "carry_" + dn + std::to_string(s);
            cs.run = [plan, gargs, K, s, d_AC, C, mask, ex_index, plane_stride](int pl) {
                Acc *send = (Acc *)plan->exchanges[ex_index].send;
                return launch_carry_block<Acc>(K, gargs(pl), mask, s, s + 1, send ? send + pl * plane_stride : nullptr,
                                               d_AC, C, plan->stream);
            };
            plan->exchange_local_steps.push_back({cs});
            Step ap;
            ap.name = "carry_apply_" + dn + std::to_string(s);
            ap.run = [plan, gargs, s](int pl) { return launch_generic_carry_apply<Acc>(gargs(pl), s, plan->stream); };
            plan->exchange_apply_steps.push_back({ap});
        }
    }

    (void)inc_pp;
    Step p2;
    p2.name = "strided_pass2_" + dn;
    p2.run = [plan, sargs, K, TZ, from_input](int pl) {
        const P *src = from_input ? (const P *)plan->in[pl] : (const P *)plan->xy_result(pl);
        return launch_strided_pass<P>(true, K, TZ, src, (P *)plan->out[pl], sargs(pl), plan->stream);
    };
    if (d == outer) plan->finish_steps.push_back(p2);
    else plan->begin_steps.push_back(p2);
    return status;
}

}  // namespace rf
