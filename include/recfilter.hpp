// recfilter.hpp -- Halide-free C++ front-end with the surface of the reference's RecFilter
// (/root/reference lib/recfilter.h:68-566), implemented on top of the C ABI in recfilter_amd.h.
//
// Same class and method names, argument meaning and misuse behaviour as the reference.  The two
// unavoidable differences of a runtime without Halide:
//   * the defining right-hand side is a bound device image instead of a Halide::Expr:
//         R(x, y) = RecFilterImage<float>(device_ptr)            // one per Tuple element via {...}
//     (the reference: R(x,y) = image(x,y), lib/recfilter.cpp:150-162, 192-248)
//   * realize() returns RecFilterRealization (device pointers owned by the filter) instead of
//     Halide::Realization (lib/recfilter.cpp:984-989).
// Where the reference prints to cerr and assert(false)s this throws RecFilterError, which carries
// the same message.  Header-only; link against librecfilter_amd.so and the HIP runtime.
#pragma once

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <functional>
#include <algorithm>
#include <cmath>
#include <ostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "recfilter_amd.h"

class RecFilterError : public std::runtime_error {
public:
    explicit RecFilterError(const std::string &m) : std::runtime_error(m) {}
};

/** Filter dimension: a name and the image extent along it (lib/recfilter.h:68-95). */
class RecFilterDim {
    std::string v;
    int e = 0;
public:
    RecFilterDim() {}
    RecFilterDim(std::string var_name, int var_extent) : v(std::move(var_name)), e(var_extent) {}
    const std::string &var() const { return v; }
    int num_pixels() const { return e; }
};

/** Filter dimension augmented with causality (lib/recfilter.h:98-128). */
class RecFilterDimAndCausality {
    RecFilterDim r;
    bool c = true;
public:
    RecFilterDimAndCausality() {}
    RecFilterDimAndCausality(RecFilterDim rec_var, bool causal) : r(std::move(rec_var)), c(causal) {}
    const std::string &var() const { return r.var(); }
    int num_pixels() const { return r.num_pixels(); }
    bool causal() const { return c; }
};

/** +x: causal scan, -x: anticausal scan (lib/recfilter.h:135-139). */
inline RecFilterDimAndCausality operator+(RecFilterDim x) { return RecFilterDimAndCausality(std::move(x), true); }
inline RecFilterDimAndCausality operator-(RecFilterDim x) { return RecFilterDimAndCausality(std::move(x), false); }

template <typename T> struct RecFilterPixel;
template <> struct RecFilterPixel<float>   { static constexpr int dtype = RF_F32; };
template <> struct RecFilterPixel<double>  { static constexpr int dtype = RF_F64; };
template <> struct RecFilterPixel<int32_t> { static constexpr int dtype = RF_I32; };
template <> struct RecFilterPixel<int16_t> { static constexpr int dtype = RF_I16; };

/** A dense x-fastest DEVICE image bound as (one Tuple element of) the filter's definition. */
struct RecFilterImageRef {
    const void *ptr = nullptr;
    int dtype = RF_F32;
    /** affine defining expression, e.g. `cast<float>(in)/255` (demo/demo_gaussian_filter.cpp:51-53): the filter
     *  runs on scale*image + bias, applied when the passes load pixels.  Shared by all Tuple elements. */
    float scale = 1.0f, bias = 0.0f;
    /** the buffer holds unsigned bytes that are converted to float on load (rf_pointwise_desc.in_dtype = RF_IN_U8);
     *  the filter itself is a float filter */
    bool bytes = false;
    RecFilterImageRef operator*(float s) const { RecFilterImageRef r = *this; r.scale *= s; r.bias *= s; return r; }
    RecFilterImageRef operator/(float s) const { return *this * (1.0f / s); }
    RecFilterImageRef operator+(float b) const { RecFilterImageRef r = *this; r.bias += b; return r; }
};

/** A pointwise consumer of a filter for RecFilter::compute_at: `w_filtered * F + w_input * I + bias`, I = the filter's
 *  own (scaled) input -- the unsharp mask `(1+w)*I - w*Blur` of apps/usm/unsharp_mask_optimized.cpp:57. */
struct RecFilterPointwise {
    float w_filtered = 1.0f, w_input = 0.0f, bias = 0.0f;
};
template <typename T>
RecFilterImageRef RecFilterImage(const T *device_ptr) { return RecFilterImageRef{device_ptr, RecFilterPixel<T>::dtype}; }
/** `cast<float>(input(x,y))` of a uint8 image (demo/demo_gaussian_filter.cpp:51-53): the bytes are read directly by
 *  the passes; combine with `/ 255.0f`. */
inline RecFilterImageRef RecFilterImage(const uint8_t *device_ptr) {
    RecFilterImageRef r{device_ptr, RF_F32};
    r.bytes = true;
    return r;
}

/** Result of realize(): one device buffer per Tuple element, owned by the filter. */
struct RecFilterRealization {
    std::vector<void *> planes;
    std::vector<int64_t> extent;   // x first
    int dtype = RF_F32;
    size_t bytes_per_plane = 0;
    /** copy plane i to host memory */
    template <typename T> std::vector<T> to_host(size_t i = 0) const {
        std::vector<T> h(bytes_per_plane / sizeof(T));
        if (hipMemcpy(h.data(), planes.at(i), bytes_per_plane, hipMemcpyDeviceToHost) != hipSuccess)
            throw RecFilterError("hipMemcpy to host failed");
        return h;
    }
};

/** Error summary of a result against a reference, in the reference's terms (lib/recfilter.h:792-823): per-sample
 *  relative error in percent, 100 * |ref - out| / (ref + 1e-9), its maximum and its mean.  Streams like the
 *  reference's CheckResult ("Max relative error = ... %"). */
template <typename T>
struct CheckResult {
    float max_diff = 0.0f, mean_diff = 0.0f;
    CheckResult(const std::vector<T> &ref, const std::vector<T> &out) {
        if (ref.size() != out.size()) throw RecFilterError("CheckResult: images differ in size");
        double sum = 0.0;
        for (size_t i = 0; i < ref.size(); i++) {
            const float re = 100.0f * std::abs((float)ref[i] - (float)out[i]) / ((float)ref[i] + 1e-9f);
            sum += re;
            max_diff = std::max(max_diff, re);
        }
        mean_diff = ref.empty() ? 0.0f : (float)(sum / (double)ref.size());
    }
};
template <typename T>
std::ostream &operator<<(std::ostream &s, const CheckResult<T> &v) {
    return s << "Max  relative error = " << v.max_diff << " % \n" << "Mean relative error = " << v.mean_diff << " % \n\n";
}

class RecFilter;

/** Handle returned by intra_/inter_/full_schedule (lib/recfilter.h:516-566).  The reference's
 *  directives steer Halide's code generator; the kernels here are hand written, so every directive
 *  is accepted and recorded only. */
class RecFilterSchedule {
    std::shared_ptr<std::vector<std::string>> log;
    std::string what;
    RecFilterSchedule &note(const std::string &n) { log->push_back(what + "." + n); return *this; }
public:
    RecFilterSchedule(std::shared_ptr<std::vector<std::string>> l, std::string w) : log(std::move(l)), what(std::move(w)) {}
    template <typename... A> RecFilterSchedule &compute_globally(A...) { return note("compute_globally"); }
    template <typename... A> RecFilterSchedule &compute_locally(A...) { return note("compute_locally"); }
    template <typename... A> RecFilterSchedule &parallel(A...) { return note("parallel"); }
    template <typename... A> RecFilterSchedule &unroll(A...) { return note("unroll"); }
    template <typename... A> RecFilterSchedule &vectorize(A...) { return note("vectorize"); }
    template <typename... A> RecFilterSchedule &gpu_blocks(A...) { return note("gpu_blocks"); }
    template <typename... A> RecFilterSchedule &gpu_threads(A...) { return note("gpu_threads"); }
    template <typename... A> RecFilterSchedule &fuse(A...) { return note("fuse"); }
    template <typename... A> RecFilterSchedule &split(A...) { return note("split"); }
    template <typename... A> RecFilterSchedule &reorder(A...) { return note("reorder"); }
    template <typename... A> RecFilterSchedule &storage_layout(A...) { return note("storage_layout"); }
    template <typename... A> RecFilterSchedule &reorder_storage(A...) { return note("reorder_storage"); }
};

class RecFilterRefVar;

/** Recursive filter (lib/recfilter.h:146-510).  Copies alias the same contents
 *  (lib/recfilter.cpp:141-144). */
class RecFilter {
    struct Scan { int dim; bool causal; std::vector<float> coeff; };
    struct Contents {
        std::string name;
        std::vector<RecFilterDim> dims;
        std::vector<RecFilterImageRef> inputs;
        std::shared_ptr<Contents> source;       // cascaded stage: reads the previous stage's result
        std::vector<Scan> scans;
        std::map<std::string, int> tile;
        bool clamped = false, tiled = false, compiled = false, has_consumer = false;
        bool merged = false;                     // the plan holds the scans of the whole cascade this stage ends (compile_jit)
        bool merge_at_compile = true;            // merge_cascades() when the plan was built (toggled since: execute_chain plans again)
        RecFilterPointwise consumer;
        rf_plan *plan = nullptr;
        void *stream = nullptr;                  // HIP stream of enqueue() / realize() (set_stream); null = default stream
        std::vector<void *> out;                 // device buffers of the last realization
        int shard_rank = 0, shard_world = 1;     // shard(): this object describes ONE rank's slab of a larger image
        uint32_t plan_flags = 0;                 // plan_options(): rf_filter_desc.flags (RF_PLAN_*)
        std::vector<int64_t> shard_extents;      // ... the slab extents of all ranks (empty: all equal)
        std::vector<void *> xsend, xgathered;    // exchange buffers of realize_sharded(), one pair per exchange
        hipStream_t xstream = nullptr;           // side stream of the collective when the plan has exchange-independent work
        hipEvent_t xev_sent = nullptr, xev_got = nullptr;
        std::shared_ptr<std::vector<std::string>> schedule_log = std::make_shared<std::vector<std::string>>();
        ~Contents() {
            if (plan) rf_plan_destroy(plan);
            for (void *p : out) if (p) (void)hipFree(p);
            for (void *p : xsend) if (p) (void)hipFree(p);
            for (void *p : xgathered) if (p) (void)hipFree(p);
            if (xev_sent) (void)hipEventDestroy(xev_sent);
            if (xev_got) (void)hipEventDestroy(xev_got);
            if (xstream) (void)hipStreamDestroy(xstream);
        }
    };
    std::shared_ptr<Contents> c;
    static int &counter() { static int n = 0; return n; }
    static int &max_threads() { static int v = 128; return v; }
    static int &vec_width() { static int v = 8; return v; }
    [[noreturn]] static void fail(const std::string &m) { throw RecFilterError(m); }

    int dim_index(const std::string &var) const {
        for (size_t i = 0; i < c->dims.size(); i++) if (c->dims[i].var() == var) return (int)i;
        return -1;
    }
    size_t plane_elems() const { size_t n = 1; for (auto &d : c->dims) n *= (size_t)d.num_pixels(); return n; }
    int dtype() const { return c->source ? RecFilter(c->source).dtype() : c->inputs.at(0).dtype; }
    size_t n_planes() const { return c->source ? RecFilter(c->source).n_planes() : c->inputs.size(); }
    static size_t dtype_size(int dt) { return dt == RF_F64 ? 8 : (dt == RF_I16 ? 2 : 4); }
    explicit RecFilter(std::shared_ptr<Contents> p) : c(std::move(p)) {}

public:
    explicit RecFilter(std::string name = "") : c(std::make_shared<Contents>()) {
        c->name = (name.empty() ? std::string("R") : name) + "_" + std::to_string(counter()++);
    }
    std::string name() const { return c->name; }

    /** R(x), R(x,y), R(x,y,z) = image   (lib/recfilter.h:207-210) */
    RecFilterRefVar operator()(RecFilterDim x);
    RecFilterRefVar operator()(RecFilterDim x, RecFilterDim y);
    RecFilterRefVar operator()(RecFilterDim x, RecFilterDim y, RecFilterDim z);
    RecFilterRefVar operator()(std::vector<RecFilterDim> x);

    /** lib/recfilter.cpp:192-248 */
    void define(std::vector<RecFilterDim> pure_args, std::vector<RecFilterImageRef> pure_def) {
        if (pure_args.empty() || pure_def.empty()) fail("empty filter definition");
        for (auto &i : pure_def)
            if (i.dtype != pure_def[0].dtype) fail("Type of all Tuple elements in filter definition must be same");
        if (!c->dims.empty()) fail("Recursive filter " + c->name + " already defined");
        if (pure_args.size() > RF_MAX_DIMS) fail("at most 3 dimensions are supported");
        c->dims = std::move(pure_args);
        c->inputs = std::move(pure_def);
    }
    /** definition that reads another filter's result: f2(x,y) = f1.as_func()(x,y) in the reference */
    void define(std::vector<RecFilterDim> pure_args, const RecFilter &source) {
        if (!c->dims.empty()) fail("Recursive filter " + c->name + " already defined");
        c->dims = std::move(pure_args);
        c->source = source.c;
    }

    /** lib/recfilter.cpp:252-258 */
    void set_clamped_image_border() {
        if (!c->dims.empty()) fail("Recursive filter " + c->name + " already defined");
        c->clamped = true;
    }

    /** lib/recfilter.cpp:260-392; coeff = {feedforward, feedback_1 .. feedback_k} */
    void add_filter(RecFilterDim x, std::vector<float> coeff) { add_filter(RecFilterDimAndCausality(x, true), std::move(coeff)); }
    void add_filter(RecFilterDimAndCausality x, std::vector<float> coeff) {
        if (c->dims.empty())
            fail("Cannot add scans to recursive filter " + c->name + " before specifying an initial definition using RecFilter::define()");
        if (coeff.size() < 2)
            fail("Cannot add scan to recursive filter " + c->name + " without feed forward and feedback coefficients");
        int d = dim_index(x.var());
        if (d < 0) fail("Variable " + x.var() + " is not one of the dimensions of the recursive filter " + c->name);
        if (c->compiled) fail("cannot add scans after the filter is compiled");
        c->scans.push_back({d, x.causal(), std::move(coeff)});
    }

    /** lib/split.cpp:1850-2080 */
    void split(std::map<std::string, int> dims) {
        if (c->tiled) fail("Recursive filter cannot be tiled twice");
        for (auto &kv : dims) {
            int d = dim_index(kv.first);
            if (d < 0) fail("Variable " + kv.first + " is not a dimension of " + c->name);
            bool has = false;
            for (auto &s : c->scans) has = has || s.dim == d;
            if (!has) fail("Cannot tile dimension " + kv.first + " without any scans in it");
            if (kv.second <= 0 || c->dims[d].num_pixels() % kv.second) fail("tile does not divide the extent of " + kv.first);
        }
        c->tile = std::move(dims);
        c->tiled = true;
    }
    void split(RecFilterDim x, int tx) { split({{x.var(), tx}}); }
    void split(RecFilterDim x, int tx, RecFilterDim y, int ty) { split({{x.var(), tx}, {y.var(), ty}}); }
    void split(RecFilterDim x, int tx, RecFilterDim y, int ty, RecFilterDim z, int tz) {
        split({{x.var(), tx}, {y.var(), ty}, {z.var(), tz}});
    }
    void split_all_dimensions(int tx) {
        std::map<std::string, int> m;
        for (size_t d = 0; d < c->dims.size(); d++)
            for (auto &s : c->scans) if (s.dim == (int)d) { m[c->dims[d].var()] = tx; break; }
        split(m);
    }

    /** lib/reorder.cpp:28-229 */
    std::vector<RecFilter> cascade(std::vector<std::vector<int>> lists) {
        if (c->tiled || c->compiled)
            fail("Cascading directive cascade() cannot be used after the filter is already tiled, compiled or realized");
        std::vector<int> flat;
        for (auto &g : lists) for (int s : g) flat.push_back(s);
        const int n = (int)c->scans.size();
        for (int s : flat) if (s < 0 || s >= n) fail("Scan " + std::to_string(s) + " not found in recursive filter");
        for (size_t u = 0; u < flat.size(); u++)
            for (size_t v = u + 1; v < flat.size(); v++) {
                const Scan &a = c->scans[flat[u]], &b = c->scans[flat[v]];
                if (a.dim == b.dim && a.causal != b.causal && flat[v] < flat[u])
                    fail("Scans " + std::to_string(flat[u]) + " " + std::to_string(flat[v]) +
                         " cannot be reordered during cascading because they have opposite causality");
            }
        for (int s = 0; s < n; s++) {
            int count = 0;
            for (int f : flat) count += (f == s);
            if (count == 0) fail("Scan " + std::to_string(s) + " does not appear in the list of scans for cascading");
            if (count > 1) fail("Scan " + std::to_string(s) + " appears multiple times in the list of scans for cascading");
        }
        std::vector<RecFilter> out;
        for (size_t i = 0; i < lists.size(); i++) {
            RecFilter rf(c->name + "_" + std::to_string(i));
            if (c->clamped) rf.set_clamped_image_border();
            if (i == 0) { rf.c->dims = c->dims; rf.c->inputs = c->inputs; rf.c->source = c->source; }
            else rf.define(c->dims, out[i - 1]);
            for (int s : lists[i]) rf.c->scans.push_back(c->scans[s]);
            out.push_back(rf);
        }
        return out;
    }
    std::vector<RecFilter> cascade(std::vector<int> a, std::vector<int> b) { return cascade({std::move(a), std::move(b)}); }
    std::vector<RecFilter> cascade_by_causality() {
        std::vector<int> causal, anti;
        for (size_t d = 0; d < c->dims.size(); d++)
            for (size_t i = 0; i < c->scans.size(); i++)
                if (c->scans[i].dim == (int)d) (c->scans[i].causal ? causal : anti).push_back((int)i);
        return cascade({causal, anti});
    }
    std::vector<RecFilter> cascade_by_dimension() {
        std::vector<std::vector<int>> groups;
        for (size_t d = 0; d < c->dims.size(); d++) {
            std::vector<int> g;
            for (size_t i = 0; i < c->scans.size(); i++) if (c->scans[i].dim == (int)d) g.push_back((int)i);
            if (!g.empty()) groups.push_back(g);
        }
        return cascade(groups);
    }

    /** lib/reorder.cpp:231-381: `this` reads fA's result; merge both into one higher-order filter */
    RecFilter overlap_to_higher_order_filter(RecFilter fA, std::string name = "O") {
        if (c->tiled || fA.c->tiled) fail("overlap_to_higher_order_filter cannot be used on tiled filters");
        if (c->dims.size() != fA.c->dims.size()) fail("filters must have the same dimensions");
        RecFilter rf(std::move(name));
        if (fA.c->clamped) rf.set_clamped_image_border();
        rf.c->dims = fA.c->dims; rf.c->inputs = fA.c->inputs; rf.c->source = fA.c->source;
        for (size_t d = 0; d < c->dims.size(); d++) {
            std::vector<Scan> sa, sb;
            for (auto &s : fA.c->scans) if (s.dim == (int)d) sa.push_back(s);
            for (auto &s : c->scans) if (s.dim == (int)d) sb.push_back(s);
            if (sa.size() != sb.size()) fail("each dimension must have the same number of scans in both filters");
            for (size_t i = 0; i < sa.size(); i++) {
                if (sa[i].causal != sb[i].causal) fail("each scan of each dimension must have the same causality in both filters");
                std::vector<float> a(sa[i].coeff.begin() + 1, sa[i].coeff.end()), b(sb[i].coeff.begin() + 1, sb[i].coeff.end());
                std::vector<float> fb(a.size() + b.size());
                if (rf_overlap_feedback_coeff(a.data(), (int)a.size(), b.data(), (int)b.size(), fb.data()) != RF_OK)
                    fail(rf_last_error_string());
                fb.insert(fb.begin(), sa[i].coeff[0] * sb[i].coeff[0]);
                rf.c->scans.push_back({(int)d, sa[i].causal, fb});
            }
        }
        return rf;
    }

    /** schedule handles and auto-schedules: accepted, not needed (lib/recfilter.cpp:396-870) */
    RecFilterSchedule intra_schedule(int id = 0) { return RecFilterSchedule(c->schedule_log, "intra" + std::to_string(id)); }
    RecFilterSchedule inter_schedule() { return RecFilterSchedule(c->schedule_log, "inter"); }
    RecFilterSchedule full_schedule() {
        if (c->tiled) fail("Filter is tiled, use RecFilter::intra_schedule() and RecFilter::inter_schedule()");
        return RecFilterSchedule(c->schedule_log, "full");
    }
    /** lib/recfilter.cpp:473-573: compute the result inside a consumer's tiles.  With a pointwise consumer the final
     *  pass applies it to every sample before its only store; a RecFilter consumer needs nothing here (cascaded stages
     *  already read their producer's device buffer). */
    void compute_at(RecFilterPointwise consumer) {
        if (c->has_consumer) fail("Cannot compute " + c->name + " at another consumer because it already has a consumer");
        if (c->compiled) fail("compute_at must be called before the filter is compiled or realized");
        c->consumer = consumer;
        c->has_consumer = true;
    }
    /** lib/recfilter.cpp:473-573, compute_at(RecFilter external): a SCHEDULE directive there -- this filter's result is
     *  computed inside the tiles of the RecFilter that consumes it instead of going through memory; the result is the same
     *  with or without it.  Here a filter defined on another filter's output (cascade stages, R2(x,y) = R1) is realized as
     *  ONE plan over the head's input (cascade_chain below): the producer's result never exists in memory, which is what the
     *  directive asks for, so it is recorded and needs nothing else (no app or test of the reference uses it). */
    void compute_at(RecFilter external) { c->schedule_log->push_back("compute_at(" + external.c->name + "): the consumer's plan holds this filter's scans"); }
    void gpu_auto_schedule(int = 32) {}
    void gpu_auto_full_schedule(int = 32) {}
    void gpu_auto_inter_schedule() {}
    void gpu_auto_intra_schedule(int = 0) {}
    void cpu_auto_schedule() {}
    void cpu_auto_full_schedule() {}
    void cpu_auto_inter_schedule() {}
    void cpu_auto_intra_schedule() {}
    static void set_max_threads_per_cuda_warp(int v) {
        if (v % 32) fail("max threads per warp must be a multiple of 32");
        max_threads() = v;
    }
    static void set_vectorization_width(int v) {
        if (v < 2 || v > 64 || (v & (v - 1))) fail("vectorization width must be a power of two <= 64");
        vec_width() = v;
    }

    /** The stages of the cascade this filter ends, head first, when the whole chain IS one filter: the scans of a cascade
     *  are the scans of the filter it was made from (lib/reorder.cpp:100-176 distributes them over Funcs that read one
     *  another), scans of different dimensions commute and the scans of a dimension stay in order -- so the last stage's
     *  result is the result of ONE plan on the head's input with all the scans in stage order, which moves every sample once
     *  per pass instead of once per pass and stage (gaussian_1xy_2xy, apps/gaussian/gaussian_filter_1xy_2xy.cpp:44-54: four
     *  scans per dimension of order <= 2 = one fused stage).  Empty when a stage boundary carries something the merged plan
     *  cannot express (a consumer fused into an upstream stage, stages of different borders or extents) or merge_cascades()
     *  is off. */
    static bool &merge_cascades() { static bool on = true; return on; }
    std::vector<const Contents *> cascade_chain() const {
        std::vector<const Contents *> chain;
        if (!merge_cascades() || !c->source || c->shard_world > 1) return chain;
        for (const Contents *p = c.get(); p; p = p->source.get()) chain.insert(chain.begin(), p);
        for (const Contents *p : chain) {
            bool same = p->dims.size() == c->dims.size() && p->clamped == c->clamped && (p == c.get() || !p->has_consumer);
            for (size_t i = 0; same && i < p->dims.size(); i++)
                same = p->dims[i].var() == c->dims[i].var() && p->dims[i].num_pixels() == c->dims[i].num_pixels();
            if (!same) return {};
        }
        return chain;
    }

    /** lib/recfilter.cpp:918-930: builds the plan (tiling tables + kernels), replaces Func::compile_jit */
    void compile_jit(std::string = "") {
        if (c->dims.empty()) fail("filter has no definition");
        if (c->plan) { rf_plan_destroy(c->plan); c->plan = nullptr; }
        // The merged plan of a cascade may not exist where every stage's own plan does (more than RF_MAX_SCANS scans in all, a
        // combination no path accepts): then this stage is planned by itself and reads its source's result, as before the
        // merge existed.
        if (!cascade_chain().empty() && build_plan_for(cascade_chain())) return;
        if (!build_plan_for({})) fail(rf_last_error_string());
    }

    /** one attempt of compile_jit: the whole cascade `chain` as one plan, or (empty chain) this stage alone */
    bool build_plan_for(const std::vector<const Contents *> &chain) {
        c->merged = !chain.empty();
        std::vector<Scan> all;                              // the stage's scans, or those of the whole cascade in stage order
        if (c->merged) { for (const Contents *p : chain) all.insert(all.end(), p->scans.begin(), p->scans.end()); }
        else all = c->scans;
        const Contents *head = c->merged ? chain.front() : c.get();
        bool tiled = c->tiled;
        for (const Contents *p : chain) tiled = tiled || p->tiled;
        std::vector<rf_scan_desc> sd(all.size());
        for (size_t i = 0; i < sd.size(); i++) {
            const Scan &s = all[i];
            if ((int)s.coeff.size() - 1 > RF_MAX_ORDER) fail("filter order above RF_MAX_ORDER");
            sd[i].dim = s.dim; sd[i].causal = s.causal; sd[i].order = (int)s.coeff.size() - 1; sd[i].feedfwd = s.coeff[0];
            for (size_t j = 1; j < s.coeff.size(); j++) sd[i].feedback[j - 1] = s.coeff[j];
        }
        rf_filter_desc d{};
        d.abi = RF_ABI;
        d.ndim = (int)c->dims.size();
        for (int i = 0; i < d.ndim; i++) {
            d.extent[i] = c->dims[i].num_pixels();
            auto it = c->tile.find(c->dims[i].var());
            d.tile[i] = it == c->tile.end() ? 0 : it->second;
        }
        d.dtype = dtype(); d.n_planes = (int)n_planes();
        d.border = c->clamped ? RF_BORDER_CLAMP : RF_BORDER_ZERO;
        d.n_scans = (int)sd.size(); d.scans = sd.data();
        d.path = tiled ? RF_PATH_AUTO : RF_PATH_UNTILED;
        d.device = -1; d.shard_rank = c->shard_rank; d.shard_world = c->shard_world;
        d.shard_extents = c->shard_extents.empty() ? nullptr : c->shard_extents.data();
        d.flags = c->plan_flags;
        // (the defining expression belongs to the stage that reads the image: this one, or the head of the merged cascade)
        if (!head->source && !head->inputs.empty() && head->inputs[0].bytes) d.pointwise.in_dtype = RF_IN_U8;
        if (!head->source && !head->inputs.empty() && (head->inputs[0].scale != 1.0f || head->inputs[0].bias != 0.0f)) {
            d.pointwise.flags |= RF_POINTWISE_PRE;
            d.pointwise.pre_scale = head->inputs[0].scale; d.pointwise.pre_bias = head->inputs[0].bias;
        }
        if (c->has_consumer) {
            d.pointwise.flags |= RF_POINTWISE_POST;
            d.pointwise.post_filtered = c->consumer.w_filtered; d.pointwise.post_input = c->consumer.w_input;
            d.pointwise.post_bias = c->consumer.bias;
        }
        if (rf_plan_create(&d, &c->plan) != RF_OK) { c->plan = nullptr; c->merged = false; return false; }
        c->compiled = true;
        c->merge_at_compile = merge_cascades();
        return true;
    }

    /** launches every upstream cascade stage, then this one, without synchronising: Func::realize on the last stage of
     *  a cascade recomputes all of its producers (they are compute_root Funcs) */
    void execute_chain() {
        if (c->compiled && c->merge_at_compile != merge_cascades()) c->compiled = false;      // toggled since: plan again
        if (!c->compiled) compile_jit();
        std::vector<const void *> in;
        if (c->merged) {                          // the whole cascade is this one plan: it reads the head's image
            const Contents *head = c.get();
            while (head->source) head = head->source.get();
            for (auto &i : head->inputs) in.push_back(i.ptr);
        } else if (c->source) { RecFilter up(c->source); up.execute_chain(); for (void *p : up.c->out) in.push_back(p); }
        else for (auto &i : c->inputs) in.push_back(i.ptr);
        const size_t bytes = plane_elems() * dtype_size(dtype());
        if (c->out.size() != in.size()) {
            for (void *p : c->out) (void)hipFree(p);
            c->out.assign(in.size(), nullptr);
            for (auto &p : c->out) if (hipMalloc(&p, bytes) != hipSuccess) fail("hipMalloc failed");
        }
        if (rf_plan_execute(c->plan, in.data(), c->out.data(), c->stream) != RF_OK) fail(rf_last_error_string());
    }

    /** Not in the reference (it has no multi-device path): this object filters ONE rank's slab of an image that is
     *  partitioned along its outermost dimension over `world` GPUs, one process (or thread) per GPU.  The extents of the
     *  RecFilterDims are the LOCAL slab's; `extents` lists every rank's extent along the outermost dimension when they
     *  differ (whole tiles; the same list on every rank). */
    void shard(int rank, int world, std::vector<int64_t> extents = {}) {
        if (world < 1 || rank < 0 || rank >= world) fail("shard: rank out of range");
        if (!extents.empty() && (int)extents.size() != world) fail("shard: one extent per rank");
        c->shard_rank = rank; c->shard_world = world; c->shard_extents = std::move(extents);
        c->compiled = false;
    }

    /** Not in the reference: options of the plan behind realize() (rf_filter_desc.flags, RF_PLAN_* of recfilter_amd.h) --
     *  e.g. RF_PLAN_TILED_ONLY, or RF_PLAN_FORCE_EXCHANGE to drive a one-rank shard through enqueue_sharded(). */
    void plan_options(uint32_t flags) { c->plan_flags = flags; c->compiled = false; }

    /** `all_gather(send, gathered, bytes_per_rank, stream)`: every rank contributes `bytes_per_rank` device bytes and
     *  receives all ranks' contributions rank-major, ordered on `stream` (ncclAllGather(send, gathered, bytes, ncclChar,
     *  comm, (hipStream_t)stream) is exactly that). */
    using AllGather = std::function<void(const void *send, void *gathered, size_t bytes_per_rank, void *stream)>;

    /** enqueue() for a sharded filter: pass 1 and the slab-local carries, then per exchange (ONE for all scans of the
     *  sharded dimension up to order 3) the slab's exit carries, the caller's all-gather and the entering carries, then
     *  the final pass -- rf_plan_begin / exchange_local / exchange_apply / finish (include/recfilter_amd.h).  No
     *  synchronisation; the output planes are valid once the stream has drained. */
    RecFilterRealization enqueue_sharded(const AllGather &all_gather) {
        if (c->source) fail("a cascade cannot be sharded stage by stage: shard the merged filter");
        if (!c->compiled) compile_jit();
        std::vector<const void *> in;
        for (auto &i : c->inputs) in.push_back(i.ptr);
        const size_t bytes = plane_elems() * dtype_size(dtype());
        if (c->out.size() != in.size()) {
            for (void *p : c->out) (void)hipFree(p);
            c->out.assign(in.size(), nullptr);
            for (auto &p : c->out) if (hipMalloc(&p, bytes) != hipSuccess) fail("hipMalloc failed");
        }
        if (c->shard_world <= 1 && !(c->plan_flags & RF_PLAN_FORCE_EXCHANGE)) {
            if (rf_plan_execute(c->plan, in.data(), c->out.data(), c->stream) != RF_OK) fail(rf_last_error_string());
        } else {
            if (rf_plan_begin(c->plan, in.data(), c->out.data(), c->stream) != RF_OK) fail(rf_last_error_string());
            // whatever throws between here and rf_plan_finish (fail(), the caller's collective) hands the execution instance
            // back to the plan, so that the next realize starts afresh
            struct AbortUnlessFinished {
                rf_plan *plan; bool finished = false;
                ~AbortUnlessFinished() { if (!finished) (void)rf_plan_abort(plan); }
            } stepping_guard{c->plan};
            const int nex = rf_plan_num_exchanges(c->plan);
            if ((int)c->xsend.size() != nex) {
                for (void *p : c->xsend) (void)hipFree(p);
                for (void *p : c->xgathered) (void)hipFree(p);
                c->xsend.assign((size_t)nex, nullptr); c->xgathered.assign((size_t)nex, nullptr);
                for (int e = 0; e < nex; e++) {
                    const size_t xb = rf_plan_exchange_bytes(c->plan, e);
                    if (hipMalloc(&c->xsend[(size_t)e], xb) != hipSuccess ||
                        hipMalloc(&c->xgathered[(size_t)e], xb * (size_t)c->shard_world) != hipSuccess)
                        fail("hipMalloc failed");
                }
            }
            for (int e = 0; e < nex; e++) {
                if (rf_plan_exchange_local(c->plan, e, c->xsend[(size_t)e]) != RF_OK) fail(rf_last_error_string());
                if (e == nex - 1 && rf_plan_has_interior(c->plan)) {
                    // exchange-independent work (the x/y stage of a z-sharded volume): the collective goes to a side stream
                    // behind the exit carries, rf_plan_interior runs beside it, the apply step waits for it
                    if (!c->xstream) {
                        if (hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&c->xev_sent, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&c->xev_got, hipEventDisableTiming) != hipSuccess)
                            fail("hipStreamCreate failed");
                    }
                    if (hipEventRecord(c->xev_sent, (hipStream_t)c->stream) != hipSuccess || hipStreamWaitEvent(c->xstream, c->xev_sent, 0) != hipSuccess)
                        fail("hipEventRecord failed");
                    all_gather(c->xsend[(size_t)e], c->xgathered[(size_t)e], rf_plan_exchange_bytes(c->plan, e), c->xstream);
                    if (hipEventRecord(c->xev_got, c->xstream) != hipSuccess) fail("hipEventRecord failed");
                    if (rf_plan_interior(c->plan) != RF_OK) fail(rf_last_error_string());
                    if (hipStreamWaitEvent((hipStream_t)c->stream, c->xev_got, 0) != hipSuccess) fail("hipStreamWaitEvent failed");
                } else {
                    all_gather(c->xsend[(size_t)e], c->xgathered[(size_t)e], rf_plan_exchange_bytes(c->plan, e), c->stream);
                }
                if (rf_plan_exchange_apply(c->plan, e, c->xgathered[(size_t)e]) != RF_OK) fail(rf_last_error_string());
            }
            stepping_guard.finished = true;      // (rf_plan_finish ends the execute whether it succeeds or not)
            if (rf_plan_finish(c->plan) != RF_OK) fail(rf_last_error_string());
        }
        RecFilterRealization r;
        r.planes = c->out; r.dtype = dtype(); r.bytes_per_plane = bytes;
        for (auto &dm : c->dims) r.extent.push_back(dm.num_pixels());
        return r;
    }

    /** realize() for a sharded filter: enqueue_sharded, then this rank's stream is drained */
    RecFilterRealization realize_sharded(const AllGather &all_gather) {
        RecFilterRealization r = enqueue_sharded(all_gather);
        if (hipStreamSynchronize((hipStream_t)c->stream) != hipSuccess) fail("stream synchronisation failed");
        return r;
    }

    /** Not in the reference (Halide owns its streams): the HIP stream this filter's kernels run on (a hipStream_t).  Every
     *  stage of a cascade launches on its own setting: give all stages the same stream. */
    void set_stream(void *hip_stream) { c->stream = hip_stream; }

    /** realize() without the device synchronisation: launches the filter (and its upstream cascade stages) on its stream
     *  and returns the output planes, valid once the stream has drained.  Several RecFilter objects enqueued on
     *  different streams run beside each other -- the carry kernels of one under the passes of another. */
    RecFilterRealization enqueue() {
        execute_chain();
        RecFilterRealization r;
        r.planes = c->out; r.dtype = dtype(); r.bytes_per_plane = plane_elems() * dtype_size(dtype());
        for (auto &dm : c->dims) r.extent.push_back(dm.num_pixels());
        return r;
    }

    /** lib/recfilter.cpp:984-989 */
    RecFilterRealization realize() {
        execute_chain();
        if (hipDeviceSynchronize() != hipSuccess) fail("device synchronisation failed");
        RecFilterRealization r;
        r.planes = c->out; r.dtype = dtype(); r.bytes_per_plane = plane_elems() * dtype_size(dtype());
        for (auto &dm : c->dims) r.extent.push_back(dm.num_pixels());
        return r;
    }

    /** lib/recfilter.cpp:991-1016: one warm-up, then the mean time of `iterations` runs in ms, every run including the
     *  upstream stages of a cascade (unlike the reference the device is synchronised before the clock is read) */
    float profile(int iterations) {
        realize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iterations; i++) execute_chain();
        if (hipDeviceSynchronize() != hipSuccess) fail("device synchronisation failed");
        std::chrono::duration<double, std::milli> dt = std::chrono::steady_clock::now() - t0;
        return (float)(dt.count() / (iterations > 0 ? iterations : 1));
    }

    std::string print_synopsis() const {
        std::ostringstream s;
        s << "RecFilter " << c->name << ":";
        for (auto &d : c->dims) s << " " << d.var() << "=" << d.num_pixels();
        s << "\n";
        for (size_t i = 0; i < c->scans.size(); i++) {
            s << "  scan " << i << ": " << (c->scans[i].causal ? "+" : "-") << c->dims[c->scans[i].dim].var() << " {";
            for (float v : c->scans[i].coeff) s << " " << v;
            s << " }\n";
        }
        if (c->plan) {
            int32_t t[RF_MAX_DIMS];
            rf_plan_tiles(c->plan, t);
            s << "  plan: path " << rf_plan_path(c->plan) << " tiles " << t[0] << " " << t[1] << " " << t[2] << "\n";
        }
        return s.str();
    }
    const std::vector<std::string> &schedule_log() const { return *c->schedule_log; }

    friend class RecFilterRefVar;
};

/** R(x,y) = image  (lib/recfilter.h:580-600) */
class RecFilterRefVar {
    RecFilter rf;
    std::vector<RecFilterDim> args;
public:
    RecFilterRefVar(RecFilter r, std::vector<RecFilterDim> a) : rf(std::move(r)), args(std::move(a)) {}
    void operator=(RecFilterImageRef image) { rf.define(args, std::vector<RecFilterImageRef>{image}); }
    void operator=(std::vector<RecFilterImageRef> tuple) { rf.define(args, std::move(tuple)); }
    void operator=(const RecFilter &source) { rf.define(args, source); }
};

inline RecFilterRefVar RecFilter::operator()(RecFilterDim x) { return RecFilterRefVar(*this, {x}); }
inline RecFilterRefVar RecFilter::operator()(RecFilterDim x, RecFilterDim y) { return RecFilterRefVar(*this, {x, y}); }
inline RecFilterRefVar RecFilter::operator()(RecFilterDim x, RecFilterDim y, RecFilterDim z) { return RecFilterRefVar(*this, {x, y, z}); }
inline RecFilterRefVar RecFilter::operator()(std::vector<RecFilterDim> x) { return RecFilterRefVar(*this, std::move(x)); }

inline std::ostream &operator<<(std::ostream &s, const RecFilter &r) { return s << r.print_synopsis(); }

/** coefficient helpers of lib/iir_coeff.h */
inline std::vector<float> gaussian_weights(float sigma, int order) {
    std::vector<float> c(order + 1);
    if (rf_gaussian_weights(sigma, order, c.data()) != RF_OK) throw RecFilterError(rf_last_error_string());
    return c;
}
inline std::vector<float> integral_image_coeff(int n) {
    std::vector<float> c(n + 1);
    if (rf_integral_image_coeff(n, c.data()) != RF_OK) throw RecFilterError(rf_last_error_string());
    return c;
}
inline std::vector<float> overlap_feedback_coeff(std::vector<float> a, std::vector<float> b) {
    std::vector<float> c(a.size() + b.size());
    if (rf_overlap_feedback_coeff(a.data(), (int)a.size(), b.data(), (int)b.size(), c.data()) != RF_OK)
        throw RecFilterError(rf_last_error_string());
    return c;
}
inline int gaussian_box_filter(int k, float sigma) {
    int w = 0;
    if (rf_gaussian_box_filter(k, sigma, &w) != RF_OK) throw RecFilterError(rf_last_error_string());
    return w;
}

/** The pointwise-with-offsets Funcs the reference's apps define on top of a filter's result (Halide expressions there):
 *  box_difference -- apps/box/box_filter.h:36-39, 128-139; tap_filter -- any clamped tap combination, e.g. diff_op_x /
 *  diff_op_y / diff_op_xy of apps/DoG/diff_gauss.cpp:176-197.  Device planes in, device plane out, on `stream`. */
inline void box_difference(const RecFilterRealization &table, int plane, void *out, int radius, std::vector<int32_t> order,
                           void *stream = nullptr) {
    order.resize(table.extent.size(), 0);
    if (rf_box_difference(table.planes.at((size_t)plane), out, (int)table.extent.size(), table.extent.data(), table.dtype, radius,
                          order.data(), stream) != RF_OK) throw RecFilterError(rf_last_error_string());
}
inline void tap_filter(const std::vector<const void *> &in_planes, void *out, const std::vector<int64_t> &extent, int dtype,
                       const std::vector<rf_tap> &taps, void *stream = nullptr) {
    if (rf_tap_filter(in_planes.data(), (int)in_planes.size(), out, (int)extent.size(), extent.data(), dtype, taps.data(),
                      (int)taps.size(), stream) != RF_OK) throw RecFilterError(rf_last_error_string());
}
