// plan.cpp -- plan construction: validation, scan grouping, path choice, tables, step lists.
//
// Reference behaviour restated here (host side only):
//   * argument checks of RecFilter::add_filter      lib/recfilter.cpp:274-300
//   * group_scans_by_dimension                      lib/split.cpp:215-242  (with the
//     coefficient permutation the reference forgets)
//   * split() preconditions (tile divides extent)   lib/recfilter.h:311, lib/split.cpp:1879-1931
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rf {

int build_fused_plan(rf_plan *plan, const rf_filter_desc *desc);  // plan_fused.cpp
bool fused_plan_applicable(const rf_plan *plan, const rf_filter_desc *desc, std::string *why);

namespace {

double cast_coeff(double c, int dtype) {
    switch (dtype) {
        case RF_F32: return (double)(float)c;
        case RF_F64: return c;
        case RF_I32: return (double)(int32_t)c;   // Cast::make(Int(32), float): truncation
        case RF_I16: return (double)(int16_t)c;
    }
    return c;
}

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

int pick_generic_tile(int64_t N, int k, int hint) {
    if (hint > 0 && hint <= kGenericMaxTile && N % hint == 0 && hint >= k) return hint;
    int cap = hint > 0 ? kGenericMaxTile : 64;
    for (int T = (int)std::min<int64_t>(cap, N); T >= std::max(k, 1); T--)
        if (N % T == 0) return T;
    return 0;
}

// ---------------------------------------------------------------------------------------
template <typename P>
int build_untiled(rf_plan *plan) {
    using Acc = typename PixelTraits<P>::Acc;
    if (plan->shard_world > 1) {
        set_error("the untiled path cannot be sharded across devices");
        return RF_ERR_UNSUPPORTED;
    }
    bool first = true;
    for (size_t i = 0; i < plan->scans.size(); i++) {
        const Scan &s = plan->scans[i];
        const DimInfo &d = plan->dims[s.dim];
        LineGeom g{d.N, d.stride, d.lines};
        DevScan<Acc> sc = make_dev_scan<Acc>(s);
        bool from_input = first;
        bool clamped = plan->clamped;
        first = false;
        Step st;
        st.name = std::string("untiled_scan_") + "xyz"[s.dim] + (s.causal ? "+" : "-");
        st.run = [plan, g, sc, from_input, clamped](int pl) {
            const P *src = from_input ? (const P *)plan->in[pl] : (const P *)plan->out[pl];
            return launch_untiled_scan<P>(src, (P *)plan->out[pl], g, sc, clamped, plan->stream);
        };
        plan->begin_steps.push_back(st);
    }
    if (plan->scans.empty()) {
        Step st;
        st.name = "copy";
        st.run = [plan](int pl) -> int {
            if (plan->in[pl] != plan->out[pl])
                RF_HIP_CHECK(hipMemcpyAsync(plan->out[pl], plan->in[pl], plan->total * sizeof(P),
                                            hipMemcpyDeviceToDevice, plan->stream));
            return (int)RF_OK;
        };
        plan->begin_steps.push_back(st);
    }
    return RF_OK;
}

// ---------------------------------------------------------------------------------------
template <typename P, typename S>
int build_generic(rf_plan *plan, const rf_filter_desc *desc) {
    using Acc = typename PixelTraits<P>::Acc;
    int status = RF_OK;
    const int outer = plan->ndim - 1;
    bool first_dim = true;
    int last_scan_dim = -1;
    for (int d = 0; d < plan->ndim; d++)
        if (!plan->dims[d].scan_ids.empty()) last_scan_dim = d;
    if (last_scan_dim < 0) return build_untiled<P>(plan);
    if (plan->shard_world > 1 && plan->ndim < 2) {
        set_error("sharding needs at least two dimensions");
        return RF_ERR_UNSUPPORTED;
    }

    for (int d = 0; d < plan->ndim; d++) {
        DimInfo &di = plan->dims[d];
        if (di.scan_ids.empty()) continue;
        int T = pick_generic_tile(di.N, di.k, desc->tile[d]);
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

        // A^M for the exchange (sharded outermost dimension)
        std::vector<Acc> hAM((size_t)n * k * k, Acc(0));
        for (int s = 0; s < n; s++) {
            std::vector<S> am = mat_pow<S>(tab.A[s], di.M, k);
            for (int e = 0; e < k * k; e++) hAM[(size_t)s * k * k + e] = table_to_acc<S, Acc>(am[e]);
        }

        const DevScan<Acc> *dScans = (const DevScan<Acc> *)plan->upload(ds.data(), ds.size() * sizeof(DevScan<Acc>), &status);
        const Acc *dWp = (const Acc *)plan->upload(hW.data(), hW.size() * sizeof(Acc), &status);
        const Acc *dAp = (const Acc *)plan->upload(hA.data(), hA.size() * sizeof(Acc), &status);
        const Acc *dAMp = (const Acc *)plan->upload(hAM.data(), hAM.size() * sizeof(Acc), &status);
        size_t tails_per_plane = (size_t)n * di.M * k * di.lines;
        size_t inc_per_plane = (size_t)n * k * di.lines;
        Acc *tails = (Acc *)plan->alloc(tails_per_plane * plan->n_planes * sizeof(Acc), false, &status);
        Acc *incoming = (Acc *)plan->alloc(inc_per_plane * plan->n_planes * sizeof(Acc), true, &status);
        if (status != RF_OK) return status;

        const bool sharded_dim = (d == outer) && plan->shard_world > 1;
        GenericDimArgs<Acc> base{};
        base.g = LineGeom{di.N, di.stride, di.lines};
        base.T = T; base.M = (int32_t)di.M; base.k = k; base.n_scans = n;
        base.clamped = plan->clamped ? 1 : 0;
        base.first_is_border = (!sharded_dim || plan->shard_rank == 0) ? 1 : 0;
        base.last_is_border = (!sharded_dim || plan->shard_rank == plan->shard_world - 1) ? 1 : 0;
        base.scans = dScans; base.W = dWp; base.A = dAp;
        auto args_for = [base, tails, incoming, tails_per_plane, inc_per_plane](int pl) {
            GenericDimArgs<Acc> a = base;
            a.tails = tails + (size_t)pl * tails_per_plane;
            a.incoming = incoming + (size_t)pl * inc_per_plane;
            return a;
        };

        const bool from_input = first_dim;
        first_dim = false;
        const bool is_exchange_dim = (d == outer);   // its carry stage is exposed through the stepping API

        Step p1;
        p1.name = "generic_pass1_" + dn;
        p1.run = [plan, args_for, from_input](int pl) {
            const P *src = from_input ? (const P *)plan->in[pl] : (const P *)plan->out[pl];
            return launch_generic_pass1<P>(src, args_for(pl), plan->stream);
        };
        plan->begin_steps.push_back(p1);

        for (int s = 0; s < n; s++) {
            int ex_index = -1;
            if (is_exchange_dim) {
                ex_index = (int)plan->exchanges.size();
                rf_plan::Exchange ex;
                ex.bytes = (size_t)plan->n_planes * k * di.lines * sizeof(Acc);
                ex.scratch = plan->alloc(ex.bytes, true, &status);
                if (status != RF_OK) return status;
                ex.send = ex.scratch;
                const Acc *AMs = dAMp + (size_t)s * k * k;
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
            cs.run = [plan, args_for, s, ex_index, plane_stride](int pl) {
                Acc *send = ex_index >= 0 ? (Acc *)plan->exchanges[ex_index].send : nullptr;
                return launch_generic_carry_scan<Acc>(args_for(pl), s, send ? send + pl * plane_stride : nullptr,
                                                      plan->stream);
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
    // a filter whose outermost dimension has no scans but whose data still has to reach `out`
    // is covered: the last filtered dimension's pass 2 wrote `out`.
    return status;
}

template <typename P>
int build_for_pixel(rf_plan *plan, const rf_filter_desc *desc, int path) {
    using S = typename std::conditional<PixelTraits<P>::is_integer, uint64_t, double>::type;
    if (path == RF_PATH_UNTILED) return build_untiled<P>(plan);
    return build_generic<P, S>(plan, desc);
}

}  // namespace

int build_plan(const rf_filter_desc *desc, rf_plan **out) {
    if (!desc || !out) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    *out = nullptr;
    if (desc->ndim < 1 || desc->ndim > RF_MAX_DIMS) { set_error("ndim must be 1..%d", RF_MAX_DIMS); return RF_ERR_INVALID_ARG; }
    if (desc->n_planes < 1 || desc->n_planes > RF_MAX_PLANES) { set_error("n_planes must be 1..%d", RF_MAX_PLANES); return RF_ERR_INVALID_ARG; }
    if (desc->dtype < RF_F32 || desc->dtype > RF_I16) { set_error("unknown dtype %d", desc->dtype); return RF_ERR_INVALID_ARG; }
    if (desc->n_scans < 0 || desc->n_scans > RF_MAX_SCANS || (desc->n_scans > 0 && !desc->scans)) {
        set_error("n_scans must be 0..%d", RF_MAX_SCANS);
        return RF_ERR_INVALID_ARG;
    }
    if (desc->border != RF_BORDER_ZERO && desc->border != RF_BORDER_CLAMP) { set_error("unknown border mode"); return RF_ERR_INVALID_ARG; }
    for (int d = 0; d < desc->ndim; d++) {
        if (desc->extent[d] < 1) { set_error("extent[%d] must be positive", d); return RF_ERR_INVALID_ARG; }
        if (desc->tile[d] < 0) { set_error("tile[%d] must be >= 0", d); return RF_ERR_INVALID_ARG; }
        if (desc->tile[d] > 0 && desc->extent[d] % desc->tile[d] != 0) {
            // lib/recfilter.h:311: tile width must divide the image width
            set_error("tile %d does not divide extent %lld of dimension %d", desc->tile[d], (long long)desc->extent[d], d);
            return RF_ERR_INVALID_ARG;
        }
    }
    for (int i = 0; i < desc->n_scans; i++) {
        const rf_scan_desc &s = desc->scans[i];
        if (s.dim < 0 || s.dim >= desc->ndim) {   // lib/recfilter.cpp:296-300
            set_error("scan %d: dimension %d is not one of the filter's %d dimensions", i, s.dim, desc->ndim);
            return RF_ERR_INVALID_ARG;
        }
        if (s.order < 1 || s.order > RF_MAX_ORDER) {   // lib/recfilter.cpp:274-278
            set_error("scan %d: needs a feedforward and 1..%d feedback coefficients", i, RF_MAX_ORDER);
            return RF_ERR_INVALID_ARG;
        }
    }
    int world = desc->shard_world < 1 ? 1 : desc->shard_world;
    if (desc->shard_rank < 0 || desc->shard_rank >= world) { set_error("shard_rank out of range"); return RF_ERR_INVALID_ARG; }

    const bool host_only = desc->device == RF_DEVICE_HOST_ONLY;
    int device = desc->device;
    if (!host_only) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
            set_error("no HIP device available (this library has no CPU fallback)");
            return RF_ERR_HIP;
        }
        if (device < 0) RF_HIP_CHECK(hipGetDevice(&device));
        if (device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return RF_ERR_INVALID_ARG; }
        RF_HIP_CHECK(hipSetDevice(device));
    }

    std::unique_ptr<rf_plan> plan(new rf_plan);
    plan->host_only = host_only;
    plan->ndim = desc->ndim;
    plan->dtype = desc->dtype;
    plan->n_planes = desc->n_planes;
    plan->clamped = desc->border == RF_BORDER_CLAMP;
    plan->device = device;
    plan->shard_rank = desc->shard_rank;
    plan->shard_world = world;
    plan->total = 1;
    for (int d = 0; d < desc->ndim; d++) {
        plan->dims[d].N = desc->extent[d];
        plan->dims[d].stride = plan->total;
        plan->total *= desc->extent[d];
    }
    for (int d = 0; d < desc->ndim; d++) plan->dims[d].lines = plan->total / plan->dims[d].N;

    // group scans by dimension, stable (lib/split.cpp:215-242); coefficients travel with their scan
    for (int d = 0; d < desc->ndim; d++) {
        for (int i = 0; i < desc->n_scans; i++) {
            const rf_scan_desc &r = desc->scans[i];
            if (r.dim != d) continue;
            Scan s;
            s.dim = d;
            s.causal = r.causal != 0;
            s.order = r.order;
            s.b = cast_coeff(r.feedfwd, desc->dtype);
            for (int j = 0; j < r.order; j++) s.a[j] = cast_coeff(r.feedback[j], desc->dtype);
            plan->dims[d].k = std::max(plan->dims[d].k, r.order);
            plan->dims[d].scan_ids.push_back((int)plan->scans.size());
            plan->scans.push_back(s);
        }
    }

    int path = desc->path;
    std::string why;
    if (path == RF_PATH_AUTO) {
        if (fused_plan_applicable(plan.get(), desc, &why)) path = RF_PATH_TILED_FUSED;
        else path = RF_PATH_TILED_GENERIC;
    }
    if (path == RF_PATH_TILED_FUSED && !fused_plan_applicable(plan.get(), desc, &why)) {
        set_error("fused tiled path not applicable: %s", why.c_str());
        return RF_ERR_UNSUPPORTED;
    }

    auto build = [&](int p) -> int {
        plan->path = p;
        if (p == RF_PATH_TILED_FUSED) return build_fused_plan(plan.get(), desc);
        switch (desc->dtype) {
            case RF_F32: return build_for_pixel<float>(plan.get(), desc, p);
            case RF_F64: return build_for_pixel<double>(plan.get(), desc, p);
            case RF_I32: return build_for_pixel<int32_t>(plan.get(), desc, p);
            case RF_I16: return build_for_pixel<int16_t>(plan.get(), desc, p);
        }
        return RF_ERR_INVALID_ARG;
    };
    int rc = build(path);
    if (rc == RF_ERR_UNSUPPORTED && desc->path == RF_PATH_AUTO && path != RF_PATH_UNTILED && world == 1) {
        // auto mode: a shape no tile fits falls back to the untiled recurrence (still on the GPU)
        std::unique_ptr<rf_plan> fresh(new rf_plan);
        // rebuild the description part
        fresh->ndim = plan->ndim; fresh->dtype = plan->dtype; fresh->n_planes = plan->n_planes;
        fresh->clamped = plan->clamped; fresh->device = plan->device; fresh->host_only = plan->host_only;
        fresh->shard_rank = plan->shard_rank; fresh->shard_world = plan->shard_world;
        fresh->scans = plan->scans; fresh->total = plan->total;
        for (int d = 0; d < RF_MAX_DIMS; d++) { fresh->dims[d] = plan->dims[d]; fresh->dims[d].T = 0; fresh->dims[d].M = 0; }
        plan.swap(fresh);
        rc = build(RF_PATH_UNTILED);
    }
    if (rc != RF_OK) return rc;
    if (!host_only) RF_HIP_CHECK(hipDeviceSynchronize());   // uploads done before the first execute
    *out = plan.release();
    return RF_OK;
}

}  // namespace rf

rf_plan::~rf_plan() {
    for (auto &b : buffers)
        if (b.ptr) (void)hipFree(b.ptr);
}

void *rf_plan::alloc(size_t bytes, bool zero, int *status) {
    if (*status != RF_OK) return nullptr;
    if (bytes == 0) bytes = 16;
    if (host_only) { workspace_bytes += bytes; return nullptr; }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        rf::set_error("hipMalloc of %zu bytes failed", bytes);
        *status = RF_ERR_NOMEM;
        return nullptr;
    }
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) {
        rf::set_error("hipMemset failed");
        *status = RF_ERR_HIP;
    }
    buffers.push_back({p, bytes});
    workspace_bytes += bytes;
    return p;
}

void *rf_plan::upload(const void *host, size_t bytes, int *status) {
    void *p = alloc(bytes, false, status);
    if (!p) return nullptr;
    if (bytes && hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        rf::set_error("hipMemcpy (table upload) failed");
        *status = RF_ERR_HIP;
    }
    return p;
}
