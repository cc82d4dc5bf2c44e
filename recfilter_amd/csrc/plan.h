// plan.h -- the execution plan behind the C ABI (host side).
#pragma once

#include <functional>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "kernels.h"
#include "rf_internal.h"
#include "tables.h"

namespace rf {

struct DimInfo {
    int64_t N = 1;        // local extent
    int64_t stride = 1;   // elements between neighbours along the dimension
    int64_t lines = 1;    // total / N
    int T = 0;            // tile width (0 = dimension not tiled / no scans)
    int64_t M = 0;        // tiles
    int k = 0;            // max feedback order of the scans in this dimension
    std::vector<int> scan_ids;  // indices into Plan::scans (application order)
};

// One kernel launch (or a small fixed group of them) of a plan.
struct Step {
    std::string name;
    std::function<int(int plane)> run;   // launches on Plan::stream for one plane
};

// rf_pointwise_desc as the plan keeps it; *_fused = applied inside the fused kernels, otherwise a
// stand-alone elementwise step runs it (pointwise_pre first, pointwise_post last).
struct Pointwise {
    bool pre = false, post = false;
    bool in_u8 = false;            // input planes are unsigned bytes, converted on load (implies a prologue)
    bool pre_fused = false, post_fused = false;
    double pre_s = 1.0, pre_b = 0.0, post_f = 1.0, post_i = 0.0, post_b = 0.0;
};

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace rf

struct rf_plan {
    // description
    int ndim = 0;
    int dtype = RF_F32;
    int n_planes = 1;
    bool clamped = false;
    // the scans are in zero-border form behind border modifications (Scan::mod_n >= 0 for every scan; plan.cpp, "clamped
    // sections"): a clamped filter whose scans of order 4..8 were split into sections of order <= 3
    bool mod_form = false;
    int path = RF_PATH_UNTILED;
    int device = 0;
    bool host_only = false;               // tables only, no device memory, cannot execute
    int shard_rank = 0, shard_world = 1;
    uint32_t flags = 0;                   // rf_filter_desc.flags (RF_PLAN_*)
    // built with the exchange structure of a sharded dimension: several slabs, or one slab driven like one of several
    bool sharded() const { return shard_world > 1 || (flags & RF_PLAN_FORCE_EXCHANGE) != 0; }
    int fused_tile_rows() const { return (int)((flags >> 8) & 0xffu); }     // RF_PLAN_TILE_ROWS(n), 0 = automatic
    int strided_tile_planes() const { return (int)((flags >> 16) & 0xffu); }  // RF_PLAN_TILE_PLANES(n), 0 = automatic
    std::vector<int64_t> shard_extents;   // extent of every rank's slab along the outermost dimension (size shard_world)
    int64_t shard_common = 0;             // their greatest common divisor: what the tile width of that dimension must divide
    // extent that decides the tile width of dimension d: the slabs' common divisor for the sharded one
    int64_t tile_basis(int d) const { return (sharded() && d == ndim - 1) ? shard_common : dims[d].N; }
    // tiles of slab h along the sharded dimension, given the tile width
    int64_t slab_tiles(int h, int64_t T) const { return shard_extents[(size_t)h] / T; }
    rf::Pointwise pw;
    // the kernels of this plan (or of a stage of its cascade) move 16 bytes per lane: the fused path and the line-parallel
    // untiled kernels; rf_plan_execute / rf_plan_begin then refuse planes that are not 16-byte aligned
    bool vector_access = false;
    std::vector<rf::Scan> scans;          // grouped by dimension, otherwise in call order
    rf::DimInfo dims[RF_MAX_DIMS];
    int64_t total = 1;                    // elements per plane

    // device memory owned by the plan
    std::vector<rf::DeviceBuffer> buffers;
    size_t workspace_bytes = 0;

    // steps
    std::vector<rf::Step> begin_steps;                       // pass 1 + slab-local carry stages
    std::vector<std::vector<rf::Step>> exchange_local_steps; // per exchange
    std::vector<std::vector<rf::Step>> exchange_apply_steps; // per exchange (after gather)
    std::vector<rf::Step> finish_steps;                      // final correction pass
    // Work of a sharded execute that does NOT depend on the exchange (rf_plan_interior): a z-sharded volume exchanges the
    // carries of the RAW input first -- the z operators commute with the x/y filter -- so its whole x/y stage runs beside
    // the all-gather (plan_strided.h, "early exchange").  Run by rf_plan_interior, or by the first exchange_apply / finish
    // of an execute whose caller never asked for it.
    std::vector<rf::Step> interior_steps;
    bool interior_pending = false;
    // plans this one drives as steps of its own (the x/y filter of the carry planes after an early exchange)
    std::vector<std::unique_ptr<rf_plan>> helpers;
    struct Exchange {
        void *send = nullptr;      // caller's buffer for the current call, [plane][r][line]
        void *scratch = nullptr;   // plan-owned buffer used when the caller passes none (single device)
        size_t bytes = 0;          // per rank
        std::function<int(const void *gathered)> form_incoming;  // launches gather kernels
    };
    std::vector<Exchange> exchanges;

    // per-execute context
    const void *in[RF_MAX_PLANES] = {nullptr};       // what the first filter stage reads (== out after pointwise_pre)
    const void *orig_in[RF_MAX_PLANES] = {nullptr};  // the caller's input planes
    void *out[RF_MAX_PLANES] = {nullptr};
    // 3-D on the fused path (plan_fused.cpp, "intermediate volume"): where the x/y stage writes and the strided z stage
    // reads -- a workspace volume per plane, or null: the output planes themselves (the z stage then runs in place)
    void *mid[RF_MAX_PLANES] = {nullptr};
    void *xy_result(int pl) const { return mid[pl] ? mid[pl] : out[pl]; }
    hipStream_t stream = nullptr;
    int phase = 0;   // 0 idle, 1 begun
    // 1-D signals whose length is not a multiple of 8192 on the fused path: zero-padded copies the kernels run on
    int64_t padded_len = 0;
    void *pad_in[RF_MAX_PLANES] = {nullptr}, *pad_out[RF_MAX_PLANES] = {nullptr};

    // In-plan cascade (plan.cpp, build_cascade): a filter the fused kernels cannot take in one piece -- more than four scans in
    // a dimension, or a 1-D signal of any length whose anticausal scans follow causal ones -- runs as successive plans
    // ("stages"), each on a subset of the scans; stage 0 reads the input, every later stage filters the output in place.
    // The parent's steps are the stages' steps in order (so every entry point of the C ABI drives it like any plan).
    std::vector<std::unique_ptr<rf_plan>> stages;

    // host tables exposed through rf_plan_table
    std::map<std::string, std::vector<double>> tables;

    // ---- concurrent executions (capi.cpp, acquire_instance) ------------------------------------------------------------
    // A plan is a description plus tables; an EXECUTION needs a tail/carry workspace and a context (plane pointers, stream,
    // stepping phase).  The plan object holds one of each.  An execute that arrives on another stream while the last one
    // enqueued here may still be in flight runs on a REPLICA -- the same description built again: own workspace, own
    // context -- created on first need and kept.  Executes on one stream share an instance (stream order separates them).
    struct SavedDesc {
        rf_filter_desc d{};
        std::vector<rf_scan_desc> scans;
        std::vector<int64_t> extents;
    } saved;
    // Everything below is guarded by the PRIMARY plan's pool_mu (a replica's own pool_mu / pool_cv / stepping are unused).
    // An instance is OWNED by the host thread that is enqueueing an execution on it (rf_plan_execute: for the duration of
    // the call; the stepping calls: from rf_plan_begin to rf_plan_finish / rf_plan_abort).  Ownership is a flag, not a
    // mutex held across API calls: a plan may be destroyed, and a stepping execute abandoned, from any state.
    std::mutex pool_mu;
    std::condition_variable pool_cv;                     // signalled whenever an instance is released
    std::vector<std::unique_ptr<rf_plan>> replicas;
    std::map<std::thread::id, rf_plan *> stepping;       // the instance a host thread's rf_plan_begin acquired
    bool owned = false;                                  // a host thread is enqueueing on this instance
    hipStream_t last_stream = nullptr;                   // stream of the last execution enqueued on this instance
    bool used = false;
    // recorded behind every execution; created once when the plan is built (never rewritten), queried only while the
    // instance is not owned -- i.e. never while another thread may be about to record it
    hipEvent_t done = nullptr;

    ~rf_plan();
    void *alloc(size_t bytes, bool zero, int *status);
    void *upload(const void *host, size_t bytes, int *status);
    int finish_build();      // waits for the uploads and zero fills (the build stream), creates `done`
};

namespace rf {
int build_plan(const rf_filter_desc *desc, rf_plan **out);
}
