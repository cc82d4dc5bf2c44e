// capi.cpp -- extern "C" entry points declared in include/recfilter_amd.h.
#include <cstdint>
#include <cmath>
#include <complex>
#include <cstring>

#include "plan.h"

namespace rf {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {

int run_steps(rf_plan *plan, const std::vector<Step> &steps) {
    for (const Step &st : steps)
        for (int pl = 0; pl < plan->n_planes; pl++) {
            int rc = st.run(pl);
            if (rc != RF_OK) return rc;
        }
    return RF_OK;
}

int set_context(rf_plan *plan, const void *const *in_planes, void *const *out_planes, void *stream) {
    if (!plan || !in_planes || !out_planes) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    if (plan->host_only) { set_error("host-only plan (RF_DEVICE_HOST_ONLY) cannot execute"); return RF_ERR_HIP; }
    for (int pl = 0; pl < plan->n_planes; pl++) {
        if (!in_planes[pl] || !out_planes[pl]) { set_error("plane %d: null image pointer", pl); return RF_ERR_INVALID_ARG; }
        if (plan->pw.post && plan->pw.post_i != 0.0 && in_planes[pl] == out_planes[pl]) {
            set_error("plane %d: a pointwise epilogue that reads the input needs out != in", pl);
            return RF_ERR_INVALID_ARG;
        }
        // the fused kernels and the line-parallel untiled kernels move 16 bytes per lane (4 for unsigned-byte input planes)
        // (an image whose width is not a multiple of 4 has element-aligned rows anyway: any element-aligned plane)
        if (plan->vector_access && !(plan->path == RF_PATH_TILED_FUSED && plan->ndim >= 2 && plan->dims[0].N % 4 != 0)) {
            const uintptr_t in_mask = plan->pw.in_u8 ? 3u : 15u;
            if (((uintptr_t)in_planes[pl] & in_mask) != 0 || ((uintptr_t)out_planes[pl] & 15u) != 0) {
                set_error("plane %d: this plan's kernels (%s) need 16-byte aligned image pointers (4-byte for uint8 inputs)", pl,
                          plan->path == RF_PATH_TILED_FUSED ? "fused path" : "line-parallel untiled kernels");
                return RF_ERR_INVALID_ARG;
            }
        }
        plan->in[pl] = plan->orig_in[pl] = in_planes[pl];
        plan->out[pl] = out_planes[pl];
    }
    plan->stream = (hipStream_t)stream;
    RF_HIP_CHECK(hipSetDevice(plan->device));
    return RF_OK;
}

// every step of a single-device execute, in order
std::vector<const Step *> flat_steps(const rf_plan *plan) {
    std::vector<const Step *> v;
    for (const Step &s : plan->begin_steps) v.push_back(&s);
    for (const auto &ex : plan->exchange_local_steps)
        for (const Step &s : ex) v.push_back(&s);
    for (const Step &s : plan->interior_steps) v.push_back(&s);
    for (const auto &ex : plan->exchange_apply_steps)         // (non-empty only for plans built with the exchange structure)
        for (const Step &s : ex) v.push_back(&s);
    for (const Step &s : plan->finish_steps) v.push_back(&s);
    return v;
}

// ---- concurrent executions -------------------------------------------------------------------------------------------
// The instance of `plan` an execution on `stream` runs on, OWNED by the calling thread (plan.h): the instance that last ran
// on this stream (stream order separates the two executions); else one whose last execution has finished; else a new
// replica.  All pool state -- the `owned` flags included -- lives under the primary's pool_mu; an instance's completion
// event is looked at only while nobody owns the instance, so no thread can be about to re-record it.
// release_instance() records the completion event and hands the instance back.
std::vector<rf_plan *> instances(rf_plan *plan) {
    std::vector<rf_plan *> all{plan};
    for (auto &r : plan->replicas) all.push_back(r.get());
    return all;
}

rf_plan *acquire_instance(rf_plan *plan, hipStream_t stream) {
    std::unique_lock<std::mutex> pool(plan->pool_mu);
    if (plan->host_only) {
        plan->pool_cv.wait(pool, [&] { return !plan->owned; });
        plan->owned = true;
        return plan;
    }
    for (;;) {
        rf_plan *same = nullptr;
        for (rf_plan *inst : instances(plan))
            if (inst->used && inst->last_stream == stream) { same = inst; break; }
        if (same != nullptr) {
            if (!same->owned) { same->owned = true; return same; }
            // another host thread is enqueueing on it (or stepping through it): wait for a release, then look again --
            // the instance may have moved to another stream meanwhile
            plan->pool_cv.wait(pool);
            continue;
        }
        for (rf_plan *inst : instances(plan)) {
            if (inst->owned) continue;
            const bool idle = !inst->used || hipEventQuery(inst->done) == hipSuccess;
            if (idle) { inst->owned = true; inst->used = true; inst->last_stream = stream; return inst; }
        }
        (void)hipGetLastError();                 // (hipEventQuery reports "not ready" as an error code)
        break;
    }
    // every instance is busy on another stream: build a replica, without the pool lock (a plan build takes host time)
    rf_filter_desc d = plan->saved.d;
    d.scans = plan->saved.scans.data();
    d.shard_extents = plan->saved.extents.empty() ? nullptr : plan->saved.extents.data();
    d.device = plan->device;
    pool.unlock();
    rf_plan *fresh = nullptr;
    if (build_plan(&d, &fresh) != RF_OK) return nullptr;
    fresh->used = true;
    fresh->owned = true;
    fresh->last_stream = stream;
    pool.lock();
    plan->replicas.emplace_back(fresh);
    return fresh;
}

// `enqueued`: at least one launch of this execution may have been attempted on inst->stream -- also when the execution
// then failed: what was enqueued still runs on the instance's workspace, and the completion event must cover it.
void release_instance(rf_plan *plan, rf_plan *inst, bool enqueued) {
    if (enqueued && !inst->host_only && inst->done != nullptr) (void)hipEventRecord(inst->done, inst->stream);
    {
        std::lock_guard<std::mutex> pool(plan->pool_mu);
        inst->owned = false;
    }
    plan->pool_cv.notify_all();
}

// stepping API: the instance rf_plan_begin acquired for this host thread, until rf_plan_finish / rf_plan_abort.  The map
// lives in the plan (not in thread-local storage), so it goes away with the plan and never outlives it.
rf_plan *stepping_instance(rf_plan *plan) {
    std::lock_guard<std::mutex> pool(plan->pool_mu);
    auto it = plan->stepping.find(std::this_thread::get_id());
    return it == plan->stepping.end() ? nullptr : it->second;
}

// the execute this thread began is over, finished or not: the instance goes back to the pool (what was enqueued so far
// still runs: the completion event covers it) and the next rf_plan_begin starts afresh
int end_stepping(rf_plan *plan, rf_plan *inst, int rc) {
    inst->phase = 0;
    inst->interior_pending = false;
    {
        std::lock_guard<std::mutex> pool(plan->pool_mu);
        plan->stepping.erase(std::this_thread::get_id());
    }
    release_instance(plan, inst, true);
    return rc;
}
int abort_stepping(rf_plan *plan, rf_plan *inst, int rc) { return end_stepping(plan, inst, rc); }

// the exchange-independent work of this execute, if the caller has not asked for it yet
int run_pending_interior(rf_plan *plan) {
    if (!plan->interior_pending) return RF_OK;
    plan->interior_pending = false;
    return run_steps(plan, plan->interior_steps);
}

}  // namespace
}  // namespace rf

using namespace rf;

extern "C" {

int rf_plan_create(const rf_filter_desc *desc, rf_plan **plan_out) { return build_plan(desc, plan_out); }

int rf_plan_destroy(rf_plan *plan) {
    // (ownership of instances is a flag inside the plan and the stepping map lives in the plan: an execute abandoned between
    // rf_plan_begin and rf_plan_finish leaves nothing behind that outlives the plan.  The caller must not destroy a plan
    // while another host thread is inside one of its calls, as for any object.)
    if (plan) {
        if (!plan->host_only) (void)hipSetDevice(plan->device);
        delete plan;
    }
    return RF_OK;
}

size_t rf_plan_workspace_bytes(const rf_plan *plan) { return plan ? plan->workspace_bytes : 0; }
int rf_plan_num_instances(rf_plan *plan) {
    if (!plan) return 0;
    std::lock_guard<std::mutex> pool(plan->pool_mu);
    return 1 + (int)plan->replicas.size();
}
int rf_plan_path(const rf_plan *plan) { return plan ? plan->path : -1; }

int rf_plan_tiles(const rf_plan *plan, int32_t tile_out[RF_MAX_DIMS]) {
    if (!plan || !tile_out) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    for (int d = 0; d < RF_MAX_DIMS; d++) tile_out[d] = d < plan->ndim ? plan->dims[d].T : 0;
    return RF_OK;
}

int rf_plan_num_kernels(const rf_plan *plan) { return plan ? (int)flat_steps(plan).size() : 0; }
int rf_plan_num_exchanges(const rf_plan *plan) { return plan ? (int)plan->exchanges.size() : 0; }

int rf_plan_execute(rf_plan *plan, const void *const *in_planes, void *const *out_planes, void *stream) {
    if (!plan) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    if (plan->sharded()) { set_error("a sharded plan must be driven through rf_plan_begin/exchange/finish"); return RF_ERR_STATE; }
    if (stepping_instance(plan) != nullptr) { set_error("rf_plan_execute between this thread's rf_plan_begin and rf_plan_finish"); return RF_ERR_STATE; }
    rf_plan *inst = acquire_instance(plan, (hipStream_t)stream);
    if (!inst) return RF_ERR_NOMEM;
    int rc = set_context(inst, in_planes, out_planes, stream);
    const bool context_set = rc == RF_OK;        // from here on launches are attempted on inst->stream
    if (rc == RF_OK) {
        for (auto &ex : inst->exchanges) ex.send = ex.scratch;
        for (const Step *st : flat_steps(inst)) {
            for (int pl = 0; pl < inst->n_planes && rc == RF_OK; pl++) rc = st->run(pl);
            if (rc != RF_OK) break;
        }
    }
    release_instance(plan, inst, context_set);
    return rc;
}

int rf_plan_execute_timed(rf_plan *plan, const void *const *in_planes, void *const *out_planes, void *stream,
                          float *ms_out, const char **names_out, int capacity) {
    if (!plan) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    if (plan->sharded()) { set_error("a sharded plan must be driven through rf_plan_begin/exchange/finish"); return RF_ERR_STATE; }
    if (stepping_instance(plan) != nullptr) { set_error("rf_plan_execute between this thread's rf_plan_begin and rf_plan_finish"); return RF_ERR_STATE; }
    rf_plan *inst = acquire_instance(plan, (hipStream_t)stream);
    if (!inst) return RF_ERR_NOMEM;
    struct Release {
        rf_plan *plan, *inst; bool ran = false;
        ~Release() { release_instance(plan, inst, ran); }
    } guard{plan, inst};
    int rc = set_context(inst, in_planes, out_planes, stream);
    if (rc) return rc;
    for (auto &ex : inst->exchanges) ex.send = ex.scratch;
    auto steps = flat_steps(inst);
    if (capacity < (int)steps.size() || !ms_out) { set_error("ms_out too small: need %zu", steps.size()); return RF_ERR_INVALID_ARG; }
    // events are destroyed on every return path; outputs are fully written even when a step fails
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    } events;
    // (names: the primary's steps have the same names and live as long as the plan)
    auto names = flat_steps(plan);
    for (size_t i = 0; i < steps.size(); i++) {
        ms_out[i] = 0.0f;
        if (names_out) names_out[i] = names[i]->name.c_str();
    }
    for (size_t i = 0; i < steps.size() + 1; i++) {
        hipEvent_t e;
        RF_HIP_CHECK(hipEventCreate(&e));
        events.ev.push_back(e);
    }
    std::vector<hipEvent_t> &ev = events.ev;
    RF_HIP_CHECK(hipEventRecord(ev[0], inst->stream));
    guard.ran = true;
    for (size_t i = 0; i < steps.size() && rc == RF_OK; i++) {
        for (int pl = 0; pl < inst->n_planes && rc == RF_OK; pl++) rc = steps[i]->run(pl);
        if (rc == RF_OK && hipEventRecord(ev[i + 1], inst->stream) != hipSuccess) rc = RF_ERR_HIP;
    }
    if (rc == RF_OK && hipEventSynchronize(ev.back()) != hipSuccess) rc = RF_ERR_HIP;
    for (size_t i = 0; i < steps.size() && rc == RF_OK; i++)
        if (hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]) != hipSuccess) rc = RF_ERR_HIP;
    if (rc == RF_ERR_HIP) set_error("HIP event timing failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

int rf_plan_begin(rf_plan *plan, const void *const *in_planes, void *const *out_planes, void *stream) {
    if (!plan) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    // An execute this thread began and never finished (its caller failed between the calls -- a collective that raised, an
    // exception on the way) is abandoned here: begin always starts afresh.
    if (rf_plan *stale = stepping_instance(plan)) (void)abort_stepping(plan, stale, RF_OK);
    rf_plan *inst = acquire_instance(plan, (hipStream_t)stream);
    if (!inst) return RF_ERR_NOMEM;
    int rc = set_context(inst, in_planes, out_planes, stream);
    const bool context_set = rc == RF_OK;
    if (rc == RF_OK) rc = run_steps(inst, inst->begin_steps);
    if (rc != RF_OK) { release_instance(plan, inst, context_set); return rc; }
    inst->phase = 1;
    inst->interior_pending = !inst->interior_steps.empty();
    {
        std::lock_guard<std::mutex> pool(plan->pool_mu);
        plan->stepping[std::this_thread::get_id()] = inst;      // owned by this thread until rf_plan_finish / rf_plan_abort
    }
    return RF_OK;
}

int rf_plan_abort(rf_plan *plan) {
    if (!plan) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    if (rf_plan *inst = stepping_instance(plan)) (void)abort_stepping(plan, inst, RF_OK);
    return RF_OK;
}

int rf_plan_has_interior(const rf_plan *plan) { return plan && !plan->interior_steps.empty() ? 1 : 0; }

int rf_plan_interior(rf_plan *plan) {
    rf_plan *inst = plan ? stepping_instance(plan) : nullptr;
    if (!inst || inst->phase != 1) { set_error("rf_plan_interior before rf_plan_begin"); return RF_ERR_STATE; }
    const int rc = run_pending_interior(inst);
    return rc == RF_OK ? rc : abort_stepping(plan, inst, rc);
}

size_t rf_plan_exchange_bytes(const rf_plan *plan, int exchange) {
    if (!plan || exchange < 0 || exchange >= (int)plan->exchanges.size()) return 0;
    return plan->exchanges[exchange].bytes;
}

int rf_plan_exchange_local(rf_plan *plan, int exchange, void *send) {
    rf_plan *inst = plan ? stepping_instance(plan) : nullptr;
    if (!inst || inst->phase != 1) { set_error("rf_plan_exchange_local before rf_plan_begin"); return RF_ERR_STATE; }
    if (exchange < 0 || exchange >= (int)inst->exchanges.size()) { set_error("exchange index out of range"); return RF_ERR_INVALID_ARG; }
    if (!send && inst->sharded()) { set_error("null send buffer"); return RF_ERR_INVALID_ARG; }
    inst->exchanges[exchange].send = send ? send : inst->exchanges[exchange].scratch;
    const int rc = run_steps(inst, inst->exchange_local_steps[exchange]);
    return rc == RF_OK ? rc : abort_stepping(plan, inst, rc);
}

int rf_plan_exchange_apply(rf_plan *plan, int exchange, const void *gathered) {
    rf_plan *inst = plan ? stepping_instance(plan) : nullptr;
    if (!inst || inst->phase != 1) { set_error("rf_plan_exchange_apply before rf_plan_begin"); return RF_ERR_STATE; }
    if (exchange < 0 || exchange >= (int)inst->exchanges.size()) { set_error("exchange index out of range"); return RF_ERR_INVALID_ARG; }
    if (!inst->sharded()) return RF_OK;   // nothing comes in from a neighbour
    if (!gathered) { set_error("null gathered buffer"); return RF_ERR_INVALID_ARG; }
    int rc = run_pending_interior(inst);      // (a caller that never called rf_plan_interior: nothing overlaps, same result)
    if (rc == RF_OK) rc = inst->exchanges[exchange].form_incoming(gathered);
    if (rc == RF_OK) rc = run_steps(inst, inst->exchange_apply_steps[exchange]);
    return rc == RF_OK ? rc : abort_stepping(plan, inst, rc);
}

int rf_plan_finish(rf_plan *plan) {
    rf_plan *inst = plan ? stepping_instance(plan) : nullptr;
    if (!inst || inst->phase != 1) { set_error("rf_plan_finish before rf_plan_begin"); return RF_ERR_STATE; }
    int rc = run_pending_interior(inst);
    if (rc == RF_OK) rc = run_steps(inst, inst->finish_steps);
    return end_stepping(plan, inst, rc);
}

int rf_plan_table(const rf_plan *plan, const char *name, double *out, size_t capacity, size_t *n_out) {
    if (!plan || !name) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    auto it = plan->tables.find(name);
    if (it == plan->tables.end()) { set_error("no table named '%s'", name); return RF_ERR_INVALID_ARG; }
    if (n_out) *n_out = it->second.size();
    if (out) {
        if (capacity < it->second.size()) { set_error("table '%s' needs %zu doubles", name, it->second.size()); return RF_ERR_INVALID_ARG; }
        std::memcpy(out, it->second.data(), it->second.size() * sizeof(double));
    }
    return RF_OK;
}

int rf_plan_debug_buffer(const rf_plan *plan, int index, void **ptr_out, size_t *bytes_out) {
    if (!plan || index < 0 || index >= (int)plan->buffers.size()) { set_error("no such buffer"); return RF_ERR_INVALID_ARG; }
    if (ptr_out) *ptr_out = plan->buffers[index].ptr;
    if (bytes_out) *bytes_out = plan->buffers[index].bytes;
    return RF_OK;
}

// ---- coefficient design ------------------------------------------------------------------
// Recursive Gaussian of van Vliet, Young and Verbeek: the poles of a fixed prototype are
// rescaled to the requested sigma, d -> |d|^(1/q) e^(i arg(d)/q) with q = 0.00399341 +
// 0.4715161 sigma (lib/iir_coeff.cpp:38-63,83-85).  Orders 1 and 2 come from one real / one
// complex-conjugate pole pair (:103-136); order 3 is their cascade (:150-159).  The mix of
// float and double below mirrors the reference's declared types because its published
// coefficient values depend on where it rounds.
namespace {
struct Gauss1 { float b0, a1; };
struct Gauss2 { float b0, a1, a2; };

float pole_scale(float sigma) { return (float)(0.00399341 + 0.4715161 * sigma); }

Gauss1 gauss_order1(float sigma) {
    double q = pole_scale(sigma);
    float d = (float)std::pow(1.86543f, 1.0 / q);
    return {(float)(-(1.0 - d) / d), (float)(-1.0 / d)};
}

Gauss2 gauss_order2(float sigma) {
    double q = pole_scale(sigma);
    std::complex<double> proto(1.41650, 1.00829);
    std::complex<double> d = std::polar(std::pow(std::abs(proto), 1.0 / q), std::arg(proto) / q);
    float n2 = (float)std::abs(d);
    n2 *= n2;
    float re = (float)d.real();
    return {(float)((1.0 - 2.0 * re + n2) / n2), (float)(-2.0 * re / n2), (float)(1.0 / n2)};
}
}  // namespace

int rf_gaussian_weights(float sigma, int order, float *coeff_out) {
    if (!coeff_out || order < 1 || order > 3) { set_error("gaussian_weights: order must be 1..3"); return RF_ERR_INVALID_ARG; }
    if (order == 1) {
        Gauss1 g = gauss_order1(sigma);
        coeff_out[0] = g.b0; coeff_out[1] = -g.a1;
    } else if (order == 2) {
        Gauss2 g = gauss_order2(sigma);
        coeff_out[0] = g.b0; coeff_out[1] = -g.a1; coeff_out[2] = -g.a2;
    } else {
        Gauss1 g1 = gauss_order1(sigma);
        Gauss2 g2 = gauss_order2(sigma);
        coeff_out[0] = g1.b0 * g2.b0;
        coeff_out[1] = -(g1.a1 + g2.a1);
        coeff_out[2] = -(g1.a1 * g2.a1 + g2.a2);
        coeff_out[3] = -(g1.a1 * g2.a2);
    }
    return RF_OK;
}

int rf_integral_image_coeff(int n, float *coeff_out) {
    // feedback = -(binomial expansion of (1-x)^n without the constant term), lib/iir_coeff.cpp:222-234
    if (!coeff_out || n < 1 || n > RF_MAX_ORDER) { set_error("integral_image_coeff: n must be 1..%d", RF_MAX_ORDER); return RF_ERR_INVALID_ARG; }
    coeff_out[0] = 1.0f;
    double binom = 1.0;
    for (int i = 1; i <= n; i++) {
        binom = binom * (double)(n - i + 1) / (double)i;
        coeff_out[i] = (float)((i % 2 == 1) ? binom : -binom);
    }
    return RF_OK;
}

int rf_overlap_feedback_coeff(const float *a, int na, const float *b, int nb, float *c_out) {
    // (1 - sum a_i z^-i)(1 - sum b_i z^-i) = 1 - sum c_i z^-i, lib/iir_coeff.cpp:236-263
    if (!a || !b || !c_out || na < 1 || nb < 1) { set_error("overlap_feedback_coeff: bad arguments"); return RF_ERR_INVALID_ARG; }
    std::vector<float> pa(na + 1), pb(nb + 1), pc(na + nb + 1, 0.0f);
    pa[0] = pb[0] = 1.0f;
    for (int i = 0; i < na; i++) pa[i + 1] = -a[i];
    for (int i = 0; i < nb; i++) pb[i + 1] = -b[i];
    for (int i = 0; i <= na; i++)
        for (int j = 0; j <= nb; j++) pc[i + j] += pa[i] * pb[j];
    for (int i = 1; i <= na + nb; i++) c_out[i - 1] = -pc[i];
    return RF_OK;
}

int rf_gaussian_box_filter(int k, float sigma, int *width_out) {
    // width of k iterated box filters approximating a Gaussian, lib/iir_coeff.cpp:205-220
    if (!width_out || k < 1 || k > 12) { set_error("gaussian_box_filter: k must be 1..12"); return RF_ERR_INVALID_ARG; }
    auto fact = [](int n) { int r = 1; for (int i = 2; i <= n; i++) r *= i; return r; };
    float sum = 0.0f;
    int limit = (int)std::floor(((float)k - 1.0f) / 2.0f);
    for (int i = 0; i <= limit; i++) {
        float f = (float)(fact(k) / (fact(i) * fact(k - i)));
        float p = (float)(std::pow(-1.0, i) / (float)fact(k - 1));
        sum += (float)(p * f * std::pow(((float)k / 2.0 - i), k - 1));
    }
    sum = (float)(std::sqrt(2.0 * M_PI) * (sum + 0.005f) * sigma);
    *width_out = (int)std::ceil(sum);
    return RF_OK;
}

int rf_box_difference(const void *in, void *out, int ndim, const int64_t *extent, int dtype, int radius, const int32_t *order,
                      void *stream) {
    if (!in || !out || !extent || !order) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    if (in == out) { set_error("box_difference gathers: out must differ from in"); return RF_ERR_INVALID_ARG; }
    if (ndim < 1 || ndim > RF_MAX_DIMS) { set_error("ndim must be 1..%d", RF_MAX_DIMS); return RF_ERR_INVALID_ARG; }
    if (radius < 0) { set_error("radius must be >= 0"); return RF_ERR_INVALID_ARG; }
    BoxDiffArgs a{};
    for (int d = 0; d < RF_MAX_DIMS; d++) {
        a.n[d] = d < ndim ? extent[d] : 1;
        a.order[d] = d < ndim ? order[d] : 0;
        if (a.n[d] < 1) { set_error("extent[%d] must be positive", d); return RF_ERR_INVALID_ARG; }
        if (a.order[d] < 0 || a.order[d] > 2) { set_error("order[%d] must be 0..2", d); return RF_ERR_INVALID_ARG; }
    }
    a.radius = radius;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (this library has no CPU fallback)"); return RF_ERR_HIP; }
    if (dtype == RF_F32) return launch_box_difference<float>((const float *)in, (float *)out, a, (hipStream_t)stream);
    if (dtype == RF_F64) return launch_box_difference<double>((const double *)in, (double *)out, a, (hipStream_t)stream);
    set_error("box_difference needs a floating-point pixel type");
    return RF_ERR_UNSUPPORTED;
}

int rf_stream_copy(const float *src, float *dst, int64_t width, int64_t rows, void *stream) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (this library has no CPU fallback)"); return RF_ERR_HIP; }
    return launch_stream_copy(src, dst, width, rows, (hipStream_t)stream);
}

int rf_tap_filter(const void *const *in_planes, int n_in, void *out, int ndim, const int64_t *extent, int dtype,
                  const rf_tap *taps, int n_taps, void *stream) {
    if (!in_planes || !out || !extent || !taps) { set_error("null argument"); return RF_ERR_INVALID_ARG; }
    if (ndim < 1 || ndim > RF_MAX_DIMS) { set_error("ndim must be 1..%d", RF_MAX_DIMS); return RF_ERR_INVALID_ARG; }
    if (n_in < 1 || n_in > RF_MAX_PLANES) { set_error("1..%d input planes", RF_MAX_PLANES); return RF_ERR_INVALID_ARG; }
    if (n_taps < 1 || n_taps > RF_MAX_TAPS) { set_error("1..%d taps", RF_MAX_TAPS); return RF_ERR_INVALID_ARG; }
    TapArgs a{};
    for (int d = 0; d < RF_MAX_DIMS; d++) {
        a.n[d] = d < ndim ? extent[d] : 1;
        if (a.n[d] < 1) { set_error("extent[%d] must be positive", d); return RF_ERR_INVALID_ARG; }
    }
    for (int p = 0; p < n_in; p++) {
        if (!in_planes[p]) { set_error("input plane %d is null", p); return RF_ERR_INVALID_ARG; }
        if (in_planes[p] == out) { set_error("tap_filter gathers: out must differ from every input"); return RF_ERR_INVALID_ARG; }
        a.in[p] = in_planes[p];
    }
    a.n_taps = n_taps;
    for (int t = 0; t < n_taps; t++) {
        if (taps[t].plane < 0 || taps[t].plane >= n_in) { set_error("tap %d: plane %d out of range", t, taps[t].plane); return RF_ERR_INVALID_ARG; }
        a.plane[t] = taps[t].plane;
        a.weight[t] = taps[t].weight;
        for (int d = 0; d < RF_MAX_DIMS; d++) {
            a.off[t][d] = d < ndim ? taps[t].offset[d] : 0;
            if (d >= ndim && taps[t].offset[d] != 0) { set_error("tap %d: offset along a missing dimension", t); return RF_ERR_INVALID_ARG; }
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (this library has no CPU fallback)"); return RF_ERR_HIP; }
    if (dtype == RF_F32) return launch_tap_filter<float>((float *)out, a, (hipStream_t)stream);
    if (dtype == RF_F64) return launch_tap_filter<double>((double *)out, a, (hipStream_t)stream);
    set_error("tap_filter needs a floating-point pixel type");
    return RF_ERR_UNSUPPORTED;
}

const char *rf_last_error_string(void) { return g_last_error.c_str(); }
const char *rf_version(void) { return "recfilter_amd 0.3 (gfx950; abi 3)"; }

int rf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
