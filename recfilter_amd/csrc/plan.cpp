// plan.cpp -- plan construction: validation, scan grouping, path choice, tables, step lists.
//
// Reference behaviour restated here (host side only):
//   * argument checks of RecFilter::add_filter      lib/recfilter.cpp:274-300
//   * group_scans_by_dimension                      lib/split.cpp:215-242  (with the
//     coefficient permutation the reference forgets)
//   * split() preconditions (tile divides extent)   lib/recfilter.h:311, lib/split.cpp:1879-1931
#include "plan.h"
#include "sections.h"
#include "plan_generic.h"
#include "plan_clamp1d.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstring>

namespace rf {

int build_fused_plan(rf_plan *plan, const rf_filter_desc *desc);  // plan_fused.cpp
bool fused_plan_applicable(const rf_plan *plan, const rf_filter_desc *desc, std::string *why);
int build_overlap_plan(rf_plan *plan, const rf_filter_desc *desc);  // plan_overlap.cpp
bool overlap_plan_applicable(const rf_plan *plan, const rf_filter_desc *desc, std::string *why);
int build_matrix_plan(rf_plan *plan, const rf_filter_desc *desc);   // plan_matrix.cpp
bool matrix_plan_applicable(const rf_plan *plan, const rf_filter_desc *desc, std::string *why);

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

// The untiled operator with the line-parallel kernels (kernels_lines.hip): pixel types of 2 / 4 bytes in the 32-bit
// arithmetic, orders <= 3, <= 4 scans per dimension, extents that are multiples of 16 (a lane owns 16 samples; the
// lines of a strided dimension are moved 16 at a time).
bool line_scans_applicable(const rf_plan *plan) {
    if (plan->sharded() || plan->scans.empty()) return false;
    if ((plan->flags & RF_PLAN_SERIAL_UNTILED) || RF_KNOB("RF_NO_LINE_SCANS") != nullptr) return false;     // the one-thread-per-line kernel
    bool strided = false;
    for (int d = 0; d < plan->ndim; d++) {
        const DimInfo &di = plan->dims[d];
        if (di.scan_ids.empty()) continue;
        if (di.k > kFusedMaxK || (int)di.scan_ids.size() > kFusedMaxScans || di.N % 16 != 0) return false;
        if (d > 0) strided = true;
    }
    if (strided && plan->dims[0].N % 16 != 0) return false;
    return true;
}

template <typename P, typename S>
int build_line_scans(rf_plan *plan) {
    using Acc = typename PixelTraits<P>::Acc;
    bool first = true;
    plan->vector_access = true;          // load16 / store16 / load_lines16 of kernels_lines.hip
    for (int d = 0; d < plan->ndim; d++) {
        const DimInfo &di = plan->dims[d];
        if (di.scan_ids.empty()) continue;
        LineScanArgs<Acc> a{};
        a.n = di.N; a.inner = di.stride; a.lines = di.lines;
        a.n_scans = (int)di.scan_ids.size();
        a.clamped = plan->clamped ? 1 : 0;
        for (int i = 0; i < a.n_scans; i++) a.scans[i] = make_fused_scan<S, Acc>(plan->scans[di.scan_ids[i]], di.k, true);
        const bool from_input = first, strided = d > 0;
        const int K = di.k;
        first = false;
        Step st;
        st.name = std::string("line_scans_") + "xyz"[d];
        st.run = [plan, a, K, strided, from_input](int pl) {
            const P *src = from_input ? (const P *)plan->in[pl] : (const P *)plan->out[pl];
            return launch_line_scans<P>(K, strided, src, (P *)plan->out[pl], a, plan->stream);
        };
        plan->begin_steps.push_back(st);
    }
    return RF_OK;
}

// ---------------------------------------------------------------------------------------
template <typename P>
int build_untiled(rf_plan *plan) {
    using Acc = typename PixelTraits<P>::Acc;
    if (plan->sharded()) {
        set_error("the untiled path cannot be sharded across devices");
        return RF_ERR_UNSUPPORTED;
    }
    {
        using S = typename std::conditional<PixelTraits<P>::is_integer, uint64_t, double>::type;
        if (line_scans_applicable(plan)) return build_line_scans<P, S>(plan);
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
    int status = RF_OK;
    bool first_dim = true;
    int last_scan_dim = -1;
    for (int d = 0; d < plan->ndim; d++)
        if (!plan->dims[d].scan_ids.empty()) last_scan_dim = d;
    if (last_scan_dim < 0) return build_untiled<P>(plan);
    if (plan->sharded() && plan->ndim < 2) {
        set_error("sharding needs at least two dimensions");
        return RF_ERR_UNSUPPORTED;
    }

    for (int d = 0; d < plan->ndim; d++) {
        if (plan->dims[d].scan_ids.empty()) continue;
        int rc = add_generic_dimension<P, S>(plan, desc->tile[d], d, first_dim);
        if (rc != RF_OK) return rc;
        first_dim = false;
    }
    // a filter whose outermost dimension has no scans but whose data still has to reach `out`
    // is covered: the last filtered dimension's pass 2 wrote `out`.
    return status;
}

// The pointwise stages the chosen path does not fuse run as stand-alone elementwise kernels around it:
// pointwise_pre writes x' into the output planes (every later stage then filters the output planes in
// place), pointwise_post combines the filtered output with the caller's input.
template <typename P>
void add_pointwise_steps_typed(rf_plan *plan) {
    const Pointwise pw = plan->pw;
    if (pw.pre && !pw.pre_fused) {
        Step st;
        st.name = "pointwise_pre";
        st.run = [plan, pw](int pl) {
            int rc = RF_OK;
            if constexpr (std::is_same<P, float>::value) {
                if (pw.in_u8) rc = launch_pointwise_from<P, uint8_t>(nullptr, (const uint8_t *)plan->orig_in[pl], (P *)plan->out[pl],
                                                                     plan->total, 0.0, pw.pre_s, pw.pre_b, plan->stream);
            }
            if (!pw.in_u8) rc = launch_pointwise<P>((const P *)plan->orig_in[pl], (const P *)nullptr, (P *)plan->out[pl], plan->total,
                                                    pw.pre_s, 0.0, pw.pre_b, plan->stream);
            plan->in[pl] = plan->out[pl];
            return rc;
        };
        plan->begin_steps.insert(plan->begin_steps.begin(), st);
    }
    if (pw.post && !pw.post_fused) {
        Step st;
        st.name = "pointwise_post";
        // out = post_f * F + post_i * (pre_s * in + pre_b) + post_b
        const double c1 = pw.post_i * (pw.pre ? pw.pre_s : 1.0);
        const double c2 = pw.post_b + pw.post_i * (pw.pre ? pw.pre_b : 0.0);
        st.run = [plan, pw, c1, c2](int pl) {
            if constexpr (std::is_same<P, float>::value) {
                if (pw.in_u8) return launch_pointwise_from<P, uint8_t>((const P *)plan->out[pl], (const uint8_t *)plan->orig_in[pl],
                                                                       (P *)plan->out[pl], plan->total, pw.post_f, c1, c2, plan->stream);
            }
            return launch_pointwise<P>((const P *)plan->out[pl], (const P *)plan->orig_in[pl], (P *)plan->out[pl], plan->total,
                                       pw.post_f, c1, c2, plan->stream);
        };
        plan->finish_steps.push_back(st);
    }
}

void add_pointwise_steps(rf_plan *plan) {
    if (plan->dtype == RF_F32) add_pointwise_steps_typed<float>(plan);
    else if (plan->dtype == RF_F64) add_pointwise_steps_typed<double>(plan);
}

template <typename P>
int build_for_pixel(rf_plan *plan, const rf_filter_desc *desc, int path) {
    using S = typename std::conditional<PixelTraits<P>::is_integer, uint64_t, double>::type;
    if (path == RF_PATH_UNTILED) return build_untiled<P>(plan);
    return build_generic<P, S>(plan, desc);
}

// ---- in-plan cascade ---------------------------------------------------------------------------------------------
// Scans of different dimensions commute and the scans of one dimension are successive in-place passes by definition
// (lib/recfilter.cpp:302-343), so any split of the scan list that keeps every dimension's order is the same filter
// (the reference's cascade(), lib/reorder.cpp:28-229, is this split done by the caller).  Stage of every scan of `desc`:
// at most kFusedMaxScans sections per dimension and stage (a scan of order n > 3 becomes up to ceil(n/2) sections,
// sections.h), and for a 1-D signal that gets zero-padded (length not a multiple of 8192) a new stage before an
// anticausal scan that follows a causal one -- every stage copies only the signal out of its padded buffer, so the
// ringing a causal scan leaves in the padding never reaches a later stage.
std::vector<int> cascade_stage_of_scans(const rf_filter_desc *desc) {
    const bool sections_ok = desc->border == RF_BORDER_ZERO && (desc->dtype == RF_F32 || desc->dtype == RF_F64) &&
                             !(desc->flags & RF_PLAN_NO_SECTIONS);
    const bool padded_1d = desc->ndim == 1 && desc->border == RF_BORDER_ZERO && desc->extent[0] % 8192 != 0;
    std::vector<int> stage((size_t)desc->n_scans, 0);
    for (int d = 0; d < desc->ndim; d++) {
        int cur = 0, used = 0;
        bool seen_causal = false;
        for (int i = 0; i < desc->n_scans; i++) {
            const rf_scan_desc &r = desc->scans[i];
            if (r.dim != d) continue;
            if (r.order > kFusedMaxK && (!sections_ok || r.order > kFusedMaxMod)) return {};
            const int w = r.order <= kFusedMaxK ? 1 : (r.order + 1) / 2;
            if (w > kFusedMaxScans) return {};
            if (used + w > kFusedMaxScans || (padded_1d && !r.causal && seen_causal)) { cur++; used = 0; seen_causal = false; }
            stage[(size_t)i] = cur;
            used += w;
            seen_causal = seen_causal || r.causal != 0;
        }
    }
    return stage;
}

int build_cascade(const rf_filter_desc *desc, const std::vector<int> &stage_of, rf_plan *parent) {
    int n_stages = 0;
    for (int st : stage_of) n_stages = std::max(n_stages, st + 1);
    std::vector<std::unique_ptr<rf_plan>> children;
    for (int st = 0; st < n_stages; st++) {
        std::vector<rf_scan_desc> sub;
        for (int i = 0; i < desc->n_scans; i++)
            if (stage_of[(size_t)i] == st) sub.push_back(desc->scans[i]);
        rf_filter_desc cd = *desc;
        cd.scans = sub.data();
        cd.n_scans = (int32_t)sub.size();
        if (st > 0) { cd.pointwise.flags &= ~RF_POINTWISE_PRE; cd.pointwise.in_dtype = RF_IN_PIXEL; }    // stage > 0 reads the output planes
        if (st + 1 < n_stages) cd.pointwise.flags &= ~RF_POINTWISE_POST;
        rf_plan *child = nullptr;
        int rc = build_plan(&cd, &child);
        if (rc != RF_OK) return rc;
        children.emplace_back(child);
        // the point of the split is the fused (or, for small images, the line-parallel) kernels: a stage that would still
        // run on the per-dimension generic passes means something else keeps this filter off them
        // (an untiled stage must be the line-parallel kind: the one-thread-per-line kernel is what the split is meant to avoid)
        if (child->path == RF_PATH_TILED_GENERIC || child->path == RF_PATH_TILED_OVERLAPPED ||
            (child->path == RF_PATH_UNTILED && !child->vector_access)) {
            set_error("cascade: stage %d would not run on the fused kernels", st);
            return RF_ERR_UNSUPPORTED;
        }
    }
    bool any_fused = false;
    parent->workspace_bytes = 0;
    for (auto &c : children) {
        any_fused = any_fused || c->path == RF_PATH_TILED_FUSED;
        parent->vector_access = parent->vector_access || c->vector_access;
        parent->workspace_bytes += c->workspace_bytes;
        for (auto &ex : c->exchanges) ex.send = ex.scratch;       // single device: nothing leaves the stage
    }
    parent->path = any_fused ? (int)RF_PATH_TILED_FUSED : children[0]->path;
    // rf_plan_tiles / rf_plan_table report the first FUSED stage (stage 0 may be an untiled line-kernel stage without tiles)
    const rf_plan *shown = children[0].get();
    for (auto &c : children)
        if (c->path == RF_PATH_TILED_FUSED) { shown = c.get(); break; }
    for (int d = 0; d < RF_MAX_DIMS; d++) { parent->dims[d].T = shown->dims[d].T; parent->dims[d].M = shown->dims[d].M; }
    parent->tables = shown->tables;
    for (int st = 0; st < n_stages; st++) {
        rf_plan *child = children[(size_t)st].get();
        std::vector<const Step *> steps;
        for (const Step &s : child->begin_steps) steps.push_back(&s);
        for (const auto &ex : child->exchange_local_steps)
            for (const Step &s : ex) steps.push_back(&s);
        for (const Step &s : child->finish_steps) steps.push_back(&s);
        for (size_t k = 0; k < steps.size(); k++) {
            const Step *sp = steps[k];
            Step w;
            w.name = st == 0 ? sp->name : "stage" + std::to_string(st) + "." + sp->name;
            const bool first = k == 0;
            w.run = [parent, child, sp, st, first](int pl) {
                if (first && pl == 0) {        // the stage's context, all planes (a batched stage launches them from plane 0)
                    for (int q = 0; q < parent->n_planes; q++) {
                        child->in[q] = child->orig_in[q] = st == 0 ? parent->orig_in[q] : (const void *)parent->out[q];
                        child->out[q] = parent->out[q];
                    }
                    child->stream = parent->stream;
                }
                return sp->run(pl);
            };
            parent->begin_steps.push_back(w);
        }
    }
    parent->stages = std::move(children);
    return RF_OK;
}

// ---- clamped 1-D signals (plan_clamp1d.h) ----------------------------------------------------------------------------
// The zero-border fused plan of the same scans as a child, a launch of n dot products in front of it and the corrections of
// the two ends of the output behind it.  RF_ERR_UNSUPPORTED (the caller goes on to the other paths) when the filter does not
// decay inside a quarter of the signal or its zero-border form does not run on the fused kernels.
int build_clamped_1d(const rf_filter_desc *desc, rf_plan *parent) {
    Clamp1DTables t;
    if (!build_clamp1d_tables(parent->scans, parent->dims[0].N, t)) {
        set_error("clamped 1-D: the filter does not decay inside a quarter of the signal");
        return RF_ERR_UNSUPPORTED;
    }
    rf_filter_desc cd = *desc;
    cd.border = RF_BORDER_ZERO;
    cd.path = RF_PATH_TILED_FUSED;
    rf_plan *child = nullptr;
    int rc = build_plan(&cd, &child);
    if (rc != RF_OK) return RF_ERR_UNSUPPORTED;
    std::unique_ptr<rf_plan> holder(child);
    if (child->path != RF_PATH_TILED_FUSED) return RF_ERR_UNSUPPORTED;
    int status = RF_OK;
    const int n = t.n, L = t.L;
    const int32_t *d_side = (const int32_t *)parent->upload(t.side.data(), t.side.size() * sizeof(int32_t), &status);
    const double *d_w = (const double *)parent->upload(t.w.data(), t.w.size() * sizeof(double), &status);
    const double *d_G = (const double *)parent->upload(t.G.data(), t.G.size() * sizeof(double), &status);
    const double *d_H = (const double *)parent->upload(t.H.data(), t.H.size() * sizeof(double), &status);
    double *d_dots = (double *)parent->alloc((size_t)n * parent->n_planes * sizeof(double), true, &status);
    if (status != RF_OK) return status;
    const int64_t N = parent->dims[0].N;
    Step dots;
    dots.name = "clamp1d_dots";
    dots.run = [parent, N, L, n, d_side, d_w, d_dots](int pl) {
        return launch_clamp1d_dots<float>((const float *)parent->orig_in[pl], N, L, n, d_side, d_w, d_dots + (size_t)pl * n, parent->stream);
    };
    parent->begin_steps.push_back(dots);
    std::vector<const Step *> steps;
    for (const Step &s : child->begin_steps) steps.push_back(&s);
    for (const auto &ex : child->exchange_local_steps)
        for (const Step &s : ex) steps.push_back(&s);
    for (const Step &s : child->finish_steps) steps.push_back(&s);
    for (size_t k = 0; k < steps.size(); k++) {
        const Step *sp = steps[k];
        Step w;
        w.name = sp->name;
        const bool first = k == 0;
        w.run = [parent, child, sp, first](int pl) {
            if (first && pl == 0) {          // the child's context, all planes
                for (int q = 0; q < parent->n_planes; q++) {
                    child->in[q] = child->orig_in[q] = parent->orig_in[q];
                    child->out[q] = parent->out[q];
                }
                child->stream = parent->stream;
            }
            return sp->run(pl);
        };
        parent->begin_steps.push_back(w);
    }
    Step fix;
    fix.name = "clamp1d_fix";
    fix.run = [parent, N, L, n, d_side, d_H, d_G, d_dots](int pl) {
        return launch_clamp1d_fix<float>((float *)parent->out[pl], N, L, n, d_side, d_H, d_G, d_dots + (size_t)pl * n, parent->stream);
    };
    parent->begin_steps.push_back(fix);
    for (auto &ex : child->exchanges) ex.send = ex.scratch;
    parent->path = RF_PATH_TILED_FUSED;
    parent->vector_access = true;
    parent->workspace_bytes += child->workspace_bytes;
    for (int d = 0; d < RF_MAX_DIMS; d++) { parent->dims[d].T = child->dims[d].T; parent->dims[d].M = child->dims[d].M; }
    parent->tables = child->tables;
    parent->tables["clamp1d_w"] = t.w;
    parent->tables["clamp1d_G"] = t.G;
    parent->tables["clamp1d_H"] = t.H;
    parent->tables["clamp1d_L"] = {(double)L};
    parent->stages.push_back(std::move(holder));
    return RF_OK;
}

// what a replica of this plan is built from (concurrent executions, capi.cpp)
void save_desc(rf_plan *plan, const rf_filter_desc *desc) {
    plan->saved.d = *desc;
    plan->saved.scans.assign(desc->scans, desc->scans + desc->n_scans);
    plan->saved.d.scans = nullptr;           // (re-pointed at the copies when a replica is built)
    plan->saved.extents.clear();
    if (desc->shard_extents != nullptr && desc->shard_world > 1)
        plan->saved.extents.assign(desc->shard_extents, desc->shard_extents + desc->shard_world);
    plan->saved.d.shard_extents = nullptr;
}

}  // namespace

int build_plan(const rf_filter_desc *desc, rf_plan **out) {
    if (!desc || !out) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    *out = nullptr;
    if (desc->abi != RF_ABI) {
        set_error("rf_filter_desc.abi is %u, this library speaks revision %u of include/recfilter_amd.h (set .abi = RF_ABI; a caller built "
                  "against another revision of the header must be rebuilt)", desc->abi, (unsigned)RF_ABI);
        return RF_ERR_INVALID_ARG;
    }
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
    const rf_pointwise_desc &pwd = desc->pointwise;
    if (pwd.flags & ~(RF_POINTWISE_PRE | RF_POINTWISE_POST)) { set_error("unknown pointwise flags 0x%x", pwd.flags); return RF_ERR_INVALID_ARG; }
    if (pwd.in_dtype != RF_IN_PIXEL && pwd.in_dtype != RF_IN_U8) { set_error("unknown pointwise input type %d", pwd.in_dtype); return RF_ERR_INVALID_ARG; }
    if (pwd.in_dtype == RF_IN_U8 && desc->dtype != RF_F32) { set_error("unsigned-byte input needs f32 pixels"); return RF_ERR_UNSUPPORTED; }
    if (pwd.flags != 0 && desc->dtype != RF_F32 && desc->dtype != RF_F64) {
        set_error("pointwise stages need a floating-point pixel type");
        return RF_ERR_UNSUPPORTED;
    }
    if (desc->flags & ~(RF_PLAN_ALL_FLAGS | 0x00ffff00u)) { set_error("unknown plan flags 0x%x", desc->flags); return RF_ERR_INVALID_ARG; }
    for (int v : {(int)((desc->flags >> 8) & 0xffu), (int)((desc->flags >> 16) & 0xffu)})
        if (v != 0 && v != 32 && v != 64 && v != 128) { set_error("RF_PLAN_TILE_ROWS / RF_PLAN_TILE_PLANES take 32, 64 or 128"); return RF_ERR_INVALID_ARG; }
    if ((desc->flags & RF_PLAN_STREAM_PASS1) && (desc->flags & RF_PLAN_STAGED_PASS1)) { set_error("RF_PLAN_STREAM_PASS1 and RF_PLAN_STAGED_PASS1 exclude each other"); return RF_ERR_INVALID_ARG; }
    int world = desc->shard_world < 1 ? 1 : desc->shard_world;
    if (desc->shard_rank < 0 || desc->shard_rank >= world) { set_error("shard_rank out of range"); return RF_ERR_INVALID_ARG; }

    // ---- merged runs (1-D signals) -------------------------------------------------------------------------------------
    // Consecutive scans of one direction with a zero border are ONE scan whose transfer function is the product of theirs --
    // the reference's overlap_to_higher_order_filter (lib/reorder.cpp:231-381), which a caller applies by hand.  On a 1-D
    // signal the fused kernels take four scans per stage (~50 us + 18 us per scan at 10,000,000 samples) while the matrix
    // path takes a scan of ANY order up to 32 in four launches (~68 us): fifteen biquads (apps/audio/audio_filter_biquads.cpp)
    // are four fused stages or one scan of order 30.  Done where it leaves fewer stages AND a probe finds the merged direct form,
    // evaluated in f32, within 2e-5 of the cascade (sections.h); the plan then IS the plan of the merged scan list.
    if (desc->path == RF_PATH_AUTO && desc->ndim == 1 && desc->border == RF_BORDER_ZERO && desc->dtype == RF_F32 && world == 1 &&
        !(desc->flags & (RF_PLAN_FORCE_EXCHANGE | RF_PLAN_NO_OVERLAP)) && desc->n_scans >= 2 && desc->extent[0] >= 8192 && desc->extent[0] % 4 == 0) {
        std::vector<rf_scan_desc> merged;
        bool ok = true, any = false;
        for (int i = 0; i < desc->n_scans && ok; ) {
            int j = i, order = 0;
            std::vector<Scan> run;
            while (j < desc->n_scans && (desc->scans[j].causal != 0) == (desc->scans[i].causal != 0) && order + desc->scans[j].order <= RF_MAX_ORDER) {
                Scan s;
                s.causal = desc->scans[j].causal != 0; s.order = desc->scans[j].order; s.b = (double)desc->scans[j].feedfwd;
                for (int e = 0; e < s.order; e++) s.a[e] = (double)desc->scans[j].feedback[e];
                order += s.order;
                run.push_back(s);
                j++;
            }
            if (run.size() == 1) { merged.push_back(desc->scans[i]); i = j; continue; }
            std::vector<double> fb(run[0].a, run[0].a + run[0].order);
            double b = run[0].b;
            for (size_t q = 1; q < run.size(); q++) {
                fb = multiply_feedback(fb, std::vector<double>(run[q].a, run[q].a + run[q].order));
                b *= run[q].b;
            }
            rf_scan_desc m{};
            m.dim = 0; m.causal = desc->scans[i].causal; m.order = (int32_t)fb.size(); m.feedfwd = (float)b;
            for (size_t e = 0; e < fb.size(); e++) m.feedback[e] = (float)fb[e];
            Scan ms;
            ms.causal = m.causal != 0; ms.order = m.order; ms.b = (double)m.feedfwd;
            for (int e = 0; e < m.order; e++) ms.a[e] = (double)m.feedback[e];
            ok = merged_well_conditioned(run, ms);
            merged.push_back(m);
            any = true;
            i = j;
        }
        // what each form costs, in the units above: a fused stage holds four scans (and a causal scan behind an anticausal one of a
        // padded signal starts a new stage, cascade_stage_of_scans); the matrix path runs every scan of order above 3 by itself
        const int fused_stages = (desc->n_scans + kFusedMaxScans - 1) / kFusedMaxScans;
        const double fused_cost = 50.0 * fused_stages + 18.0 * desc->n_scans, matrix_cost = 68.0 * (double)merged.size();
        if (ok && any && matrix_cost < fused_cost) {
            rf_filter_desc md = *desc;
            md.scans = merged.data();
            md.n_scans = (int32_t)merged.size();
            md.flags |= RF_PLAN_NO_OVERLAP;
            // The merge is kept only if every merged run of order above 3 really runs in its DIRECT form, one matrix stage each
            // (ADVICE r5): a plan that sends the merged scans elsewhere -- back into sections of the f32-rounded product: a second
            // rounding of the poles -- is dropped and the scans are planned as given.
            rf_plan *mp = nullptr;
            const int rc = build_plan(&md, &mp);
            int max_order = 0;
            for (const rf_scan_desc &m : merged) max_order = std::max(max_order, (int)m.order);
            if (rc == RF_OK && mp != nullptr && (mp->path == RF_PATH_TILED_MATRIX || max_order <= kFusedMaxK)) { *out = mp; return RF_OK; }
            delete mp;
            *out = nullptr;
        }
    }

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
    plan->flags = desc->flags;
    plan->pw.pre = (pwd.flags & RF_POINTWISE_PRE) != 0;
    plan->pw.post = (pwd.flags & RF_POINTWISE_POST) != 0;
    plan->pw.in_u8 = pwd.in_dtype == RF_IN_U8;
    if (plan->pw.pre) { plan->pw.pre_s = pwd.pre_scale; plan->pw.pre_b = pwd.pre_bias; }
    if (plan->pw.in_u8) plan->pw.pre = true;        // the conversion is a prologue (scale 1, bias 0 unless given)
    if (plan->pw.post) { plan->pw.post_f = pwd.post_filtered; plan->pw.post_i = pwd.post_input; plan->pw.post_b = pwd.post_bias; }
    {
        const int64_t own = desc->extent[desc->ndim - 1];
        plan->shard_extents.assign((size_t)world, own);
        if (desc->shard_extents != nullptr && world > 1) {
            for (int h = 0; h < world; h++) {
                if (desc->shard_extents[h] < 1) { set_error("shard_extents[%d] must be positive", h); return RF_ERR_INVALID_ARG; }
                plan->shard_extents[(size_t)h] = desc->shard_extents[h];
            }
            if (desc->shard_extents[desc->shard_rank] != own) {
                set_error("shard_extents[%d] = %lld is not this rank's extent %lld", desc->shard_rank,
                          (long long)desc->shard_extents[desc->shard_rank], (long long)own);
                return RF_ERR_INVALID_ARG;
            }
        }
        int64_t g = 0;
        for (int64_t e : plan->shard_extents) { int64_t a = e, b = g; while (b) { int64_t t = a % b; a = b; b = t; } g = a; }
        plan->shard_common = g;
    }
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

    // Orders above 3 (lib/split.cpp:575-578 pads any order; the fused kernels stop at 3): with a zero border and float
    // pixels a scan of order 4..RF_MAX_ORDER is the same filter as its first/second/third-order sections applied one after
    // the other (sections.h), and those the fused kernels take -- as long as no dimension ends up with more than four scans.
    // With a CLAMPED border (2-D / 3-D, f32) the same rewrite works on scans in "mod form": a clamped scan IS the zero-border
    // scan of an input whose first k samples in scan direction are x_r + g_r x_0 (tables.h, scan_tile) --
    //     y_r = b x_r + sum_{j<r} a_j y_{r-1-j} + (sum_{j>=r} a_j) c_r,   c_0 = x_0,  c_r = y_0 = (b + sum a) x_0
    // (lib/recfilter.cpp:330-336), so  g_0 = (sum_j a_j) / b,  g_r = (sum_{j>=r} a_j)(b + sum a) / b -- and the zero-border
    // scan factors into sections exactly.  The first section of every scan carries the modification of the ORIGINAL scan;
    // every scan of the plan (the low-order ones too) is put into that form, and the kernels that run recurrences apply the
    // modification on the tile where the scan enters the image (scan_device.h, border_mod_*).
    // The rewrite is kept only if it makes the fused path applicable; every other path runs the scans as given.
    if ((desc->path == RF_PATH_AUTO || desc->path == RF_PATH_TILED_FUSED) &&
        (plan->dtype == RF_F32 || plan->dtype == RF_F64) && !(desc->flags & RF_PLAN_NO_SECTIONS)) {
        // (orders above kFusedMaxMod = 8 are never sectioned: the direct form on the matrix path is well conditioned
        // where a long cascade of f32 resonators is not, and a border modification touches at most eight samples)
        bool high = false, too_high = false;
        for (const Scan &sc : plan->scans) { high = high || sc.order > kFusedMaxK; too_high = too_high || sc.order > kFusedMaxMod; }
        if (too_high) high = false;
        // ONE scan of a 1-D signal (apps/audio/audio_filter_high_order.cpp): its direct form on the matrix path is four
        // launches whatever the order (10,000,000 samples: 0.065 ms at orders 4..16), its two or three sections on the fused
        // kernels take 0.086-0.115 ms (profiles/r5/matrix_audio_sweep.txt)
        if (high && desc->path == RF_PATH_AUTO && desc->ndim == 1 && desc->n_scans == 1 && matrix_plan_applicable(plan.get(), desc, nullptr)) high = false;
        const bool mod = plan->clamped;
        if (mod && (plan->dtype != RF_F32 || desc->ndim < 2 || plan->sharded())) high = false;
        if (high) {
            std::vector<Scan> rewritten;
            bool ok = true;
            const int dtype = plan->dtype;
            for (const Scan &sc : plan->scans) {
                std::vector<Scan> sec;
                if (sc.order <= kFusedMaxK) sec.push_back(sc);
                else ok = ok && split_into_sections(sc, kFusedMaxK, [dtype](double v) { return cast_coeff(v, dtype); }, sec) &&
                          (dtype != RF_F32 || sections_well_conditioned(sc, sec));
                if (!ok) break;
                if (mod) {
                    // the modification of the scan as given, carried by its first section
                    if (sc.b == 0.0) { ok = false; break; }
                    double suffix[RF_MAX_ORDER + 1] = {0};
                    for (int j = sc.order - 1; j >= 0; j--) suffix[j] = suffix[j + 1] + sc.a[j];
                    const double total = sc.b + suffix[0];
                    for (size_t i = 0; i < sec.size(); i++) {
                        sec[i].mod_n = i == 0 ? sc.order : 0;
                        for (int r = 0; r < RF_MAX_ORDER; r++)
                            sec[i].mod_g[r] = (i == 0 && r < sc.order) ? (r == 0 ? suffix[0] / sc.b : suffix[r] * total / sc.b) : 0.0;
                    }
                }
                rewritten.insert(rewritten.end(), sec.begin(), sec.end());
            }
            if (ok) {
                std::vector<Scan> original = plan->scans;
                DimInfo saved[RF_MAX_DIMS];
                for (int d = 0; d < RF_MAX_DIMS; d++) saved[d] = plan->dims[d];
                plan->scans = rewritten;
                plan->mod_form = mod;
                for (int d = 0; d < desc->ndim; d++) { plan->dims[d].scan_ids.clear(); plan->dims[d].k = 0; }
                for (size_t i = 0; i < plan->scans.size(); i++) {
                    DimInfo &di = plan->dims[plan->scans[i].dim];
                    di.scan_ids.push_back((int)i);
                    di.k = std::max(di.k, plan->scans[i].order);
                }
                std::string unused;
                if (!fused_plan_applicable(plan.get(), desc, &unused)) {
                    plan->scans = original;
                    plan->mod_form = false;
                    for (int d = 0; d < RF_MAX_DIMS; d++) plan->dims[d] = saved[d];
                }
            }
        }
    }

    // A clamped 1-D signal: the zero-border fused plan plus the border corrections (plan_clamp1d.h).
    if ((desc->path == RF_PATH_AUTO || desc->path == RF_PATH_TILED_FUSED) && desc->ndim == 1 && plan->clamped && plan->dtype == RF_F32 &&
        !plan->sharded() && !plan->pw.pre && !plan->pw.post && !plan->pw.in_u8 && desc->n_scans >= 1 &&
        // (one scan of order above 3: the matrix path, as for the zero border above)
        !(desc->path == RF_PATH_AUTO && desc->n_scans == 1 && plan->scans[0].order > kFusedMaxK && matrix_plan_applicable(plan.get(), desc, nullptr))) {
        const int rc = build_clamped_1d(desc, plan.get());
        if (rc == RF_OK) { save_desc(plan.get(), desc); if (int fb = plan->finish_build()) return fb; *out = plan.release(); return RF_OK; }
        if (rc != RF_ERR_UNSUPPORTED) return rc;
        plan->begin_steps.clear();
        plan->stages.clear();
    }

    // What the fused kernels cannot take in one piece because of the NUMBER or the ORDER of its scans runs as a cascade of
    // plans inside this one (build_cascade above) -- not on the generic path, whose carry scans are one thread per line
    // (five biquads over 10,000,000 samples: 59.6 ms there, 0.26 ms as two stages).
    if ((desc->path == RF_PATH_AUTO || desc->path == RF_PATH_TILED_FUSED) && !plan->sharded() && desc->n_scans > 1 &&
        !(plan->pw.post && plan->pw.post_i != 0.0) && !(desc->flags & RF_PLAN_NO_CASCADE)) {
        std::string why_not;
        if (!fused_plan_applicable(plan.get(), desc, &why_not)) {
            const std::vector<int> stage_of = cascade_stage_of_scans(desc);
            int n_stages = 0;
            for (int st : stage_of) n_stages = std::max(n_stages, st + 1);
            if (n_stages > 1) {
                const int rc = build_cascade(desc, stage_of, plan.get());
                if (rc == RF_OK) { save_desc(plan.get(), desc); if (int fb = plan->finish_build()) return fb; *out = plan.release(); return RF_OK; }
                if (desc->path == RF_PATH_TILED_FUSED) return rc;
                plan->begin_steps.clear();          // (auto: the paths below run the scans as given)
                plan->stages.clear();
            }
        }
    }

    int path = desc->path;
    std::string why;
    if (path == RF_PATH_AUTO) {
        int filtered_dims = 0;
        for (int d = 0; d < desc->ndim; d++) filtered_dims += plan->dims[d].scan_ids.empty() ? 0 : 1;
        // small images are launch-bound on the five-kernel tiled pipeline (30-35 us, most of it host enqueue time): the
        // line-parallel untiled kernels need one launch per filtered dimension (kernels_lines.hip)
        int64_t max_extent = 0;
        for (int d = 0; d < desc->ndim; d++) max_extent = std::max<int64_t>(max_extent, plan->dims[d].N);
        // (RF_PLAN_TILED_ONLY: the automatic path never picks the line kernels; the limit itself is a developer knob)
        const int64_t small_limit = (desc->flags & RF_PLAN_TILED_ONLY) ? 0 : RF_KNOB("RF_SMALL_LIMIT") ? atoll(RF_KNOB("RF_SMALL_LIMIT")) : 1024;
        // ... up to 1984 for order-3 filters with four or more scans, whose tiled launches are the heaviest (C++ caller,
        // gaussian_3xy: 1152^2 47 -> 33 us, 1536^2 49 -> 45 us, 1920^2 56 -> 51 us, even at 2048^2; order 1 and 2 cross over
        // at 1024..1152)
        int max_order = 0;
        for (const Scan &sc : plan->scans) max_order = std::max(max_order, sc.order);
        const int64_t long_limit = (small_limit == 1024 && max_order >= 3 && plan->scans.size() >= 4) ? 1984 : small_limit;
        bool user_tiles = false;
        for (int d = 0; d < desc->ndim; d++) user_tiles = user_tiles || desc->tile[d] > 0;
        const bool fused_ok = fused_plan_applicable(plan.get(), desc, &why);
        // (where the fused kernels apply, split() widths are hints -- the tile size never changes the result -- so a small
        // split() filter takes the line kernels too: the reference's own sweep, scripts/profile_app.sh, tiles at 32)
        // (scans in mod form -- clamped sections, above -- exist for the fused kernels only)
        if (fused_ok && !plan->mod_form && !plan->sharded() && max_extent <= long_limit && plan->total * plan->n_planes <= long_limit * long_limit * 4 &&
            plan->pw.pre == false && plan->pw.post == false && line_scans_applicable(plan.get()))
            path = RF_PATH_UNTILED;
        else if (fused_ok) path = RF_PATH_TILED_FUSED;
        // a filter split() along two or more dimensions with small tiles: the fully overlapped tiling (two passes over
        // the image instead of two per dimension)
        else if (filtered_dims >= 2 && overlap_plan_applicable(plan.get(), desc, &why)) path = RF_PATH_TILED_OVERLAPPED;
        // orders above 3 that did not become sections (orders above 8; cascades of f32 resonators the conditioning probe of
        // sections.h rejected; borders and shapes the section rewrite does not take): the direct form on the matrix cores
        else if (max_order > kFusedMaxK && matrix_plan_applicable(plan.get(), desc, &why)) path = RF_PATH_TILED_MATRIX;
        // what the fused kernels do not take (f64 pixels): the line-parallel untiled kernels beat the per-dimension tiled
        // passes of the generic path at every size measured (profiles/r2/paths_4096.txt), unless the user tiled the filter
        else if (!user_tiles && !plan->sharded() && small_limit > 0 && line_scans_applicable(plan.get()))
            path = RF_PATH_UNTILED;
        else path = RF_PATH_TILED_GENERIC;
    }
    if (path == RF_PATH_TILED_OVERLAPPED && !overlap_plan_applicable(plan.get(), desc, &why)) {
        set_error("overlapped tiled path not applicable: %s", why.c_str());
        return RF_ERR_UNSUPPORTED;
    }
    if (path == RF_PATH_TILED_MATRIX && !matrix_plan_applicable(plan.get(), desc, &why)) {
        set_error("matrix path not applicable: %s", why.c_str());
        return RF_ERR_UNSUPPORTED;
    }
    if (path < RF_PATH_UNTILED || path > RF_PATH_TILED_MATRIX) { set_error("unknown path %d", path); return RF_ERR_INVALID_ARG; }
    if (path == RF_PATH_TILED_FUSED && !fused_plan_applicable(plan.get(), desc, &why)) {
        set_error("fused tiled path not applicable: %s", why.c_str());
        return RF_ERR_UNSUPPORTED;
    }

    auto build = [&](int p) -> int {
        plan->path = p;
        if (p == RF_PATH_TILED_FUSED) return build_fused_plan(plan.get(), desc);
        if (p == RF_PATH_TILED_OVERLAPPED) return build_overlap_plan(plan.get(), desc);
        if (p == RF_PATH_TILED_MATRIX) return build_matrix_plan(plan.get(), desc);
        switch (desc->dtype) {
            case RF_F32: return build_for_pixel<float>(plan.get(), desc, p);
            case RF_F64: return build_for_pixel<double>(plan.get(), desc, p);
            case RF_I32: return build_for_pixel<int32_t>(plan.get(), desc, p);
            case RF_I16: return build_for_pixel<int16_t>(plan.get(), desc, p);
        }
        return RF_ERR_INVALID_ARG;
    };
    int rc = build(path);
    if (rc == RF_ERR_UNSUPPORTED && desc->path == RF_PATH_AUTO && path != RF_PATH_UNTILED && !plan->sharded() && !plan->mod_form) {
        // auto mode: a shape no tile fits falls back to the untiled recurrence (still on the GPU)
        std::unique_ptr<rf_plan> fresh(new rf_plan);
        // rebuild the description part
        fresh->ndim = plan->ndim; fresh->dtype = plan->dtype; fresh->n_planes = plan->n_planes;
        fresh->clamped = plan->clamped; fresh->device = plan->device; fresh->host_only = plan->host_only;
        fresh->shard_rank = plan->shard_rank; fresh->shard_world = plan->shard_world; fresh->flags = plan->flags;
        fresh->shard_extents = plan->shard_extents; fresh->shard_common = plan->shard_common;
        fresh->scans = plan->scans; fresh->total = plan->total; fresh->pw = plan->pw;
        fresh->pw.pre_fused = fresh->pw.post_fused = false;
        for (int d = 0; d < RF_MAX_DIMS; d++) { fresh->dims[d] = plan->dims[d]; fresh->dims[d].T = 0; fresh->dims[d].M = 0; }
        plan.swap(fresh);
        rc = build(RF_PATH_UNTILED);
    }
    if (rc != RF_OK) return rc;
    add_pointwise_steps(plan.get());
    {   // the scans as the plan runs them (after the rewrite into sections, grouped by dimension), for rf_plan_table("scans"):
        // per scan [dim, causal, order, b, a[0..7], mod_n, mod_g[0..7]] -- what tests/fused_emulator.py replays
        std::vector<double> flat;
        for (const Scan &sc : plan->scans) {
            flat.push_back(sc.dim); flat.push_back(sc.causal ? 1.0 : 0.0); flat.push_back(sc.order); flat.push_back(sc.b);
            for (int j = 0; j < RF_MAX_ORDER; j++) flat.push_back(sc.a[j]);
            flat.push_back(sc.mod_n);
            for (int j = 0; j < RF_MAX_ORDER; j++) flat.push_back(sc.mod_g[j]);
        }
        plan->tables["scans"] = flat;
    }
    // Table uploads and zero fills are work of a private NON-BLOCKING stream (build_stream below): waiting for it waits
    // neither for the device nor -- as the legacy null stream would -- for every blocking stream of the process.  A replica
    // is built while other streams are busy with executions of this very plan (capi.cpp, acquire_instance), and such a
    // wait would serialise what the replica exists to overlap.
    if (int fb = plan->finish_build()) return fb;
    save_desc(plan.get(), desc);
    *out = plan.release();
    return RF_OK;
}

}  // namespace rf

rf_plan::~rf_plan() {
    replicas.clear();
    stages.clear();
    helpers.clear();
    if (done) (void)hipEventDestroy(done);
    for (auto &b : buffers)
        if (b.ptr) (void)hipFree(b.ptr);
}

namespace {
// One stream per device for everything a plan build puts on the GPU (zero fills, table uploads): created once with
// hipStreamNonBlocking, so that it neither waits for nor is waited for by the null stream, and kept for the life of the
// process.  HIP streams may be used from several host threads.
hipStream_t build_stream(int device, int *status) {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        rf::set_error("hipStreamCreateWithFlags (plan build stream) failed");
        *status = RF_ERR_HIP;
        return nullptr;
    }
    streams[device] = s;
    return s;
}
}  // namespace

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
    buffers.push_back({p, bytes});
    workspace_bytes += bytes;
    if (zero) {
        hipStream_t bs = build_stream(device, status);
        if (*status == RF_OK && hipMemsetAsync(p, 0, bytes, bs) != hipSuccess) {
            rf::set_error("hipMemsetAsync failed");
            *status = RF_ERR_HIP;
        }
    }
    return p;
}

void *rf_plan::upload(const void *host, size_t bytes, int *status) {
    void *p = alloc(bytes, false, status);
    if (!p) return nullptr;
    if (bytes) {
        // the host tables are temporaries of the build: the copy has left them when this returns
        hipStream_t bs = build_stream(device, status);
        if (*status == RF_OK && (hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, bs) != hipSuccess || hipStreamSynchronize(bs) != hipSuccess)) {
            rf::set_error("hipMemcpyAsync (table upload) failed");
            *status = RF_ERR_HIP;
        }
    }
    return p;
}

int rf_plan::finish_build() {
    if (host_only) return RF_OK;
    int status = RF_OK;
    hipStream_t bs = build_stream(device, &status);
    if (status != RF_OK) return status;
    RF_HIP_CHECK(hipStreamSynchronize(bs));
    if (done == nullptr) RF_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    return RF_OK;
}
