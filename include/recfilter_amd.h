/*
 * recfilter_amd.h -- C ABI of the MI355X-native tiled recursive-filter runtime.
 *
 * This is the drop-in boundary for the hot path of mit-gfx/recfilter: everything
 * the reference hands to Halide at
 *      RecFilter::realize()   lib/recfilter.cpp:984-989  (Func::realize)
 *      RecFilter::profile()   lib/recfilter.cpp:991-1016
 *      RecFilter::compile_jit lib/recfilter.cpp:918-930  (Func::compile_jit)
 * after it has built the scan list with RecFilter::add_filter
 * (lib/recfilter.cpp:260-392) and tiled it with RecFilter::split
 * (lib/split.cpp:1850-2080).  A plan replaces the Func graph that split() builds and
 * the schedule that RecFilterSchedule (lib/schedule.cpp) attaches to it: the tiling
 * algebra becomes host-side tables, the four stages (intra-tile scans, tail extraction,
 * cross-tile carry recurrence, final correction) are hand-written gfx950 kernels.
 *
 * Conventions
 *   - plain C types only; every function returns an rf_status code, 0 = ok, and never throws;
 *     rf_last_error_string() describes the last failure on the calling thread.
 *   - images are dense, planar, x fastest (Halide's layout, lib/recfilter.cpp:969-981);
 *     a Halide Tuple is n_planes separate buffers that share the filter.
 *   - all image pointers are DEVICE pointers (hipMalloc or torch); `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream).  rf_plan_execute is
 *     asynchronous on that stream.
 *   - the library has no CPU fallback: without a usable HIP device every entry point that
 *     needs one fails with RF_ERR_HIP.
 */
#ifndef RECFILTER_AMD_H
#define RECFILTER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_MAX_DIMS    3    /* the reference's auto-schedules stop at 3 (lib/recfilter.cpp:6,734) */
#define RF_MAX_ORDER   32   /* feedback taps per scan: RecFilter::add_filter takes any order (lib/recfilter.cpp:260-343);
                             * the reference's own sweep stops at 29 (apps/audio/audio_filter_high_order.cpp:14,38) */
#define RF_MAX_SCANS   32
#define RF_MAX_PLANES  16
/* desc.device value that builds the plan's host tables without touching a device; such a plan
 * answers rf_plan_table / rf_plan_tiles / rf_plan_path but refuses to execute (RF_ERR_HIP). */
#define RF_DEVICE_HOST_ONLY (-2)

typedef enum {
    RF_OK = 0,
    RF_ERR_INVALID_ARG = 1,   /* the misuse cases the reference asserts on (lib/recfilter.cpp:274,296) */
    RF_ERR_UNSUPPORTED = 2,
    RF_ERR_HIP = 3,           /* HIP runtime error or no device */
    RF_ERR_NOMEM = 4,
    RF_ERR_STATE = 5          /* stepping API called out of order */
} rf_status;

/* pixel type P = type of the defining expression (lib/recfilter.cpp:197); coefficients are
 * cast to P (lib/recfilter.cpp:324,335) */
typedef enum { RF_F32 = 0, RF_F64 = 1, RF_I32 = 2, RF_I16 = 3 } rf_dtype;

/* default border is zero; RecFilter::set_clamped_image_border (lib/recfilter.cpp:252-258) */
typedef enum { RF_BORDER_ZERO = 0, RF_BORDER_CLAMP = 1 } rf_border;

/* how a plan executes */
typedef enum {
    RF_PATH_AUTO = 0,      /* fastest path the shape admits                                        */
    RF_PATH_UNTILED = 1,   /* one serial recurrence per line (a filter that was never split())      */
    RF_PATH_TILED_GENERIC = 2, /* tiled, any tile width dividing the extent (split(x,tx,..))        */
    RF_PATH_TILED_FUSED = 3,   /* tiled, LDS-staged fused x/y tiles with fixed MI355X tile shapes; a filter with more
                                * than four scans in a dimension runs as successive stages of such plans inside the
                                * one plan (stage 0 reads the input, later stages filter the output planes in place) */
    RF_PATH_TILED_OVERLAPPED = 4, /* tiled, ALL dimensions in one pass 1 / one pass 2 with the cross-dimension
                                  * residuals of lib/split.cpp:1215-1633 between every pair of dimensions (x->y, x->z,
                                  * y->z): the reference's fully overlapped N-D tiling.  Needs an explicit tile width
                                  * for every filtered dimension, tile volume <= 4096 samples; RF_PATH_AUTO picks it
                                  * for such filters when the fused path does not apply */
    RF_PATH_TILED_MATRIX = 5     /* tiled, one scan at a time, every stage a dense f32 GEMM on the matrix cores
                                  * (kernels_matrix.hip): tail extraction H[k x T] . tile, the carry recurrence as a chain
                                  * of k x k products, the final pass as 32 x 32 impulse-response blocks.  Scans of ANY
                                  * order up to RF_MAX_ORDER in their direct form (the reference's apps sweep orders up to
                                  * 29); f32 pixels, any extents whose width is a multiple of 4 samples.  RF_PATH_AUTO picks it for
                                  * filters with a scan of order above 3 that the fused path (sections) does not take */
} rf_path;

/* one RecFilter::add_filter(+-dim, {feedfwd, fb1..fbk}) call, lib/recfilter.cpp:264-343 */
typedef struct {
    int32_t dim;                    /* 0 = x (fastest), 1 = y, 2 = z                      */
    int32_t causal;                 /* 1 = +dim, 0 = -dim                                 */
    int32_t order;                  /* number of feedback coefficients, 1..RF_MAX_ORDER   */
    float   feedfwd;
    float   feedback[RF_MAX_ORDER]; /* y[i] = feedfwd*x[i] + sum_j feedback[j]*y[i-j-1]   */
} rf_scan_desc;

/* Pointwise stages fused around the filter.  The reference writes them as Halide expressions: the
 * defining expression of the filter (R(x,y) = cast<float>(in(x,y))/255, demo/demo_gaussian_filter.cpp:51-57)
 * and a consumer computed at the filter's tiles with RecFilter::compute_at (lib/recfilter.cpp:473-573),
 * e.g. USM = (1+w)*in - w*Blur in apps/usm/unsharp_mask_optimized.cpp:57-61.  Here they are two affine maps:
 *     x'  = pre_scale * in + pre_bias                               before the first scan
 *     out = post_filtered * F(x') + post_input * x' + post_bias      after the last scan (F = the filter)
 * flags = 0 (a zeroed struct) means both are the identity.  Float pixel types only.  On the fused path both
 * are applied inside pass 1 / pass 2 (no extra pass over the image); the other paths run them as separate
 * elementwise kernels.  RF_POINTWISE_POST with post_input != 0 needs out != in. */
#define RF_POINTWISE_PRE   1
#define RF_POINTWISE_POST  2
/* element type of the INPUT planes: the pixel type (default), or unsigned bytes converted on load -- the
 * `cast<float>(input(x,y,c)) / 255.0f` of demo/demo_gaussian_filter.cpp:51-53 with the uint8 image read directly
 * (1 byte per sample instead of 4 in both passes).  RF_IN_U8 needs dtype RF_F32; outputs stay f32. */
typedef enum { RF_IN_PIXEL = 0, RF_IN_U8 = 1 } rf_input_dtype;
typedef struct {
    int32_t flags;                    /* RF_POINTWISE_PRE | RF_POINTWISE_POST */
    float   pre_scale, pre_bias;
    float   post_filtered, post_input, post_bias;
    int32_t in_dtype;                 /* rf_input_dtype */
} rf_pointwise_desc;

/* Plan options (rf_filter_desc.flags).  The library reads NO environment variable: what a caller (or a test) wants
 * to choose about a plan is chosen here.  (A/B timing switches of the kernel developers exist only in builds made
 * with -DRF_AB_KNOBS, tools/; the shipped library has none.)
 *   RF_PLAN_FORCE_EXCHANGE  build the sharded structure -- per-scan launches around the exchange points, exit carries,
 *                           the gather walk, the correction of the slab -- even for shard_world == 1, and insist on the
 *                           stepping calls.  The all-gather of one rank is the identity, so the result is the plain
 *                           filter; it lets a box with ONE GPU drive begin / exchange_local / all-gather (RCCL) /
 *                           exchange_apply / finish exactly as every rank of an N-GPU run does.
 *   RF_PLAN_TILED_ONLY      RF_PATH_AUTO never resolves to the line-parallel untiled kernels it prefers for images up
 *                           to 1024^2 (launch-bound regime): small images take the tiled kernels too.
 *   RF_PLAN_NO_CASCADE      a filter the fused kernels cannot take in one piece (more than four scans per dimension)
 *                           is not split into successive fused stages inside the plan; it runs as given on another path.
 *   RF_PLAN_NO_SECTIONS     scans of order 4..8 are not rewritten into sections of order <= 3.
 *   RF_PLAN_NO_OVERLAP      the consecutive same-direction scans of a 1-D signal (zero border, f32) are not merged into one
 *                           scan of their product transfer function (overlap_feedback_coeff, lib/iir_coeff.cpp:236-263) for
 *                           the matrix path -- what RF_PATH_AUTO does where a host-side probe finds the merged direct form
 *                           within 2e-5 of the cascade and fewer stages result (five biquads: two fused stages -> one scan).
 *   RF_PLAN_NO_PLANE_BATCH  the planes of a 2-D Tuple run as separate launches instead of one batched launch per step.
 *   RF_PLAN_STREAM_PASS1 /  pass 1 of the fused path as the LDS-DMA streaming kernel wherever its shape rules allow,
 *   RF_PLAN_STAGED_PASS1    whatever the image size / never (default: single planes of at least 2048 tiles).
 *                           STAGED also keeps the vector-ALU contraction (fused_tails_kernel) for every order.
 *   RF_PLAN_MFMA_PASS1      pass 1 with its x-tail contraction on the matrix cores (kernels_tails_mfma.hip) wherever its
 *                           shape rules allow -- f32 images of whole tiles, at most two scans per dimension -- whatever
 *                           the order (default: orders 2 and 3).  The streaming kernel keeps what it takes.
 *   RF_PLAN_WALK_PASS1      3-D: pass 1 reads the volume ONCE and forms the x, y and z tails together (the z operators commuted
 *                           in front of the x/y filter, kernels_tails_walk.hip; 20 instead of 24 bytes per sample) wherever
 *                           its shape rules allow -- f32 volumes whose depth is whole z tiles and whose width is a multiple of
 *                           four (partial tiles along x and y load as zeros; z slabs: with the early exchange), a prologue
 *                           x' = s x + b applied as the samples arrive (not 8-bit input), orders <= 3 along x / y and <= 2 along z,
 *                           one or two scans in each of the three dimensions -- whatever the size (default: volumes of at
 *                           least 256 patches of 256 x 32 samples x one z tile, one per compute unit);
 *                           RF_PLAN_STAGED_PASS1 keeps the two first passes of the x/y and z stages.
 *   RF_PLAN_LATE_EXCHANGE   a z-sharded volume exchanges the carries of the x/y-FILTERED data, after its x/y stage
 *                           (nothing runs beside the all-gather); default: the carries of the raw input first, the x/y
 *                           stage beside the all-gather (rf_plan_interior below).
 *   RF_PLAN_SERIAL_UNTILED  RF_PATH_UNTILED as one serial recurrence per line for every filter (the literal operator of
 *                           lib/recfilter.cpp:302-343; an independent on-device reference) instead of the line-parallel
 *                           kernels it uses where they apply.
 *   RF_PLAN_TILE_ROWS(n) /  n = 32, 64 or 128: tile height of the fused x/y stage / tile width of the strided z stage,
 *   RF_PLAN_TILE_PLANES(n)  where the shape admits it (default: chosen from the image size; rf_plan_tiles reports it).
 *                           rf_filter_desc.tile[] stays what RecFilter::split passes: binding on the generic and
 *                           overlapped paths, a hint on the fused path (the tile size never changes the result).
 *   RF_PLAN_INPLACE_Z       3-D on the fused path: the z stage filters the x/y stage's result where it lies, in the output
 *                           planes.  Default for volumes of at least 2^28 samples: the x/y stage writes a plan-owned
 *                           INTERMEDIATE VOLUME and the z stage reads it (rf_plan_workspace_bytes grows by one volume) --
 *                           a final z pass that reads and writes the same addresses runs 4 % slower (its read and write
 *                           fronts chase each other through the same DRAM banks: tools/microbench/zpass_shape.hip) -- as
 *                           long as that volume is at most a third of the device memory free when the plan is built.
 *                           Same kernels, same results either way. */
#define RF_PLAN_FORCE_EXCHANGE  0x01u
#define RF_PLAN_TILED_ONLY      0x02u
#define RF_PLAN_NO_CASCADE      0x04u
#define RF_PLAN_NO_SECTIONS     0x08u
#define RF_PLAN_NO_PLANE_BATCH  0x10u
#define RF_PLAN_STREAM_PASS1    0x20u
#define RF_PLAN_STAGED_PASS1    0x40u
#define RF_PLAN_LATE_EXCHANGE   0x80u
#define RF_PLAN_SERIAL_UNTILED  0x01000000u
#define RF_PLAN_MFMA_PASS1      0x02000000u
#define RF_PLAN_WALK_PASS1      0x04000000u
#define RF_PLAN_NO_OVERLAP      0x08000000u
#define RF_PLAN_INPLACE_Z       0x10000000u
#define RF_PLAN_ALL_FLAGS       0x1f0000ffu
#define RF_PLAN_TILE_ROWS(n)    (((uint32_t)(n) & 0xffu) << 8)
#define RF_PLAN_TILE_PLANES(n)  (((uint32_t)(n) & 0xffu) << 16)

/* Revision of the binary interface: the layout of the structs below as this header declares them.  rf_plan_create refuses a
 * descriptor whose `abi` is not RF_ABI -- a caller compiled against another revision of the header (revision 3: RF_MAX_ORDER
 * went from 8 to 32 in round 5, which changed sizeof(rf_scan_desc) and the row layout of rf_plan_table("scans")) gets
 * RF_ERR_INVALID_ARG instead of a mis-strided scans array.  The field sits in what used to be the padding behind `ndim`. */
#define RF_ABI 3u

typedef struct {
    int32_t  ndim;                    /* 1..RF_MAX_DIMS                                          */
    uint32_t abi;                     /* RF_ABI                                                  */
    int64_t  extent[RF_MAX_DIMS];     /* extent[0] = width (x)                                   */
    int32_t  dtype;                   /* rf_dtype                                                */
    int32_t  n_planes;                /* Tuple size, >= 1                                        */
    int32_t  border;                  /* rf_border                                               */
    int32_t  n_scans;
    const rf_scan_desc *scans;        /* in add_filter call order                                */
    int32_t  tile[RF_MAX_DIMS];       /* RecFilter::split widths; 0 = let the plan choose        */
    int32_t  path;                    /* rf_path                                                 */
    int32_t  device;                  /* HIP device ordinal, -1 = current, RF_DEVICE_HOST_ONLY   */
    /* outermost-dimension shard (multi-GPU); world = 1 for a single GPU.  extent[] is the LOCAL
     * slab; the slab of rank r follows the slab of rank r-1 along dimension ndim-1. */
    int32_t  shard_rank;
    int32_t  shard_world;
    rf_pointwise_desc pointwise;      /* zeroed = none                                           */
    /* slabs of different extents: shard_world extents along dimension ndim-1, one per rank, in rank order
     * (shard_extents[shard_rank] == extent[ndim-1]); NULL = every slab has this rank's extent.  All ranks pass
     * the same array: the tile width along the sharded dimension is chosen from their common divisor so that
     * every rank tiles alike. */
    const int64_t *shard_extents;
    uint32_t flags;                   /* RF_PLAN_* options below; 0 = the defaults                 */
} rf_filter_desc;

typedef struct rf_plan rf_plan;

/* ---- plan lifetime (replaces split()+schedule+compile_jit) ------------------------------- */
int rf_plan_create(const rf_filter_desc *desc, rf_plan **plan_out);
int rf_plan_destroy(rf_plan *plan);

/* bytes of device workspace the plan allocated for tails/carries (owned by the plan) */
size_t rf_plan_workspace_bytes(const rf_plan *plan);
/* execution instances the plan holds right now: 1 + the replicas concurrent executes made it build (see rf_plan_execute) */
int rf_plan_num_instances(rf_plan *plan);
/* which rf_path the plan resolved to, and the tile widths it uses (0 for an untiled dim) */
int rf_plan_path(const rf_plan *plan);
int rf_plan_tiles(const rf_plan *plan, int32_t tile_out[RF_MAX_DIMS]);
/* number of kernels one execute launches, and their names (for profilers) */
int rf_plan_num_kernels(const rf_plan *plan);

/* ---- execution (replaces Func::realize) --------------------------------------------------- */
/* Concurrent executions.  A plan is a description plus tables (immutable after rf_plan_create); an EXECUTION needs a
 * tail/carry workspace and a context (plane pointers, stream, stepping phase).  The plan holds one of each; an execute
 * that arrives on another stream while the previous one may still be in flight runs on a replica of the plan -- the same
 * description built again, with its own workspace (~8 % of one image) -- created on first need and kept for later
 * executes; an execute finds the instance that last ran on its stream (stream order separates the two), else one whose
 * last execution has finished (hipEventQuery), else builds a replica.  So rf_plan_execute may be called on distinct
 * streams, and from distinct host threads, without the executions waiting for one another (the call that builds a replica
 * takes the host time of a plan creation once; it does not wait for the device).  rf_plan_workspace_bytes reports one instance.  The stepping calls below belong to the
 * host thread that called rf_plan_begin, until its rf_plan_finish (or rf_plan_abort).
 *
 * in_planes/out_planes: n_planes device pointers each.  in == out (same pointers) is allowed.  A plan whose kernels
 * move 16 bytes per lane -- the fused path (rf_plan_path() == RF_PATH_TILED_FUSED) and the line-parallel untiled
 * kernels RF_PATH_AUTO / RF_PATH_UNTILED use for orders <= 3 with extents that are multiples of 16 -- needs 16-byte
 * aligned planes (4-byte for RF_IN_U8 input planes) and returns RF_ERR_INVALID_ARG otherwise (hipMalloc and torch
 * allocations are 256-byte aligned; only offset views are affected).  The generic and overlapped tiled paths take any
 * element-aligned pointer. */
int rf_plan_execute(rf_plan *plan, const void *const *in_planes, void *const *out_planes,
                    void *stream);

/* Same, but brackets every kernel with HIP events on `stream` and returns per-kernel
 * milliseconds in ms_out[0..n) (n = rf_plan_num_kernels) and their names in names_out
 * (pointers stay valid for the life of the plan).  Synchronises the stream. */
int rf_plan_execute_timed(rf_plan *plan, const void *const *in_planes, void *const *out_planes,
                          void *stream, float *ms_out, const char **names_out, int capacity);

/* ---- sharded execution: the same work as rf_plan_execute split at the exchange points ----- */
/* For a plan with shard_world > 1 the scans along the outermost dimension need the carry of
 * the neighbouring slab.  Protocol, per execute (all calls asynchronous on the begin() stream):
 *     rf_plan_begin(...)                         pass 1 + every slab-local carry stage
 *     for e in 0 .. rf_plan_num_exchanges()-1:   ONE for all scans of the sharded dimension (orders <= 3, at most
 *                                                4 scans, scans * world * order <= 128), else one per scan
 *                                                (RF_PATH_TILED_MATRIX: always one per scan along the sharded dimension)
 *         rf_plan_exchange_local(e, send)        slab-local recurrence; writes this slab's exit
 *                                                carries (rf_plan_exchange_bytes(e) bytes) to `send`
 *         -- caller all-gathers `send` over the ranks into `gathered` (world * bytes, rank-major;
 *            RCCL all-gather on the same stream) --
 *         rf_plan_exchange_apply(e, gathered)    forms the incoming carry; what it adds to the slab's tails is
 *                                                applied here, or by the final pass as it loads a carry (fused 2-D)
 *     rf_plan_finish(...)                        final correction pass
 * Work that does not depend on the exchange: between ISSUING the last all-gather and WAITING for it, call
 *     rf_plan_interior(plan)
 * A z-sharded volume (fused x/y stage + strided z stage, merged exchange, no pointwise stages) exchanges the z carries
 * of the RAW input -- the z operators commute with the x/y filter -- so its whole x/y stage is such work and runs beside
 * the collective; afterwards exchange_apply filters the few carry planes along x/y.  A caller whose all-gather is
 * asynchronous with respect to the begin() stream (RCCL on its own stream; torch.distributed with async_op=True) gets a
 * step of max(kernels, exchange) instead of their sum.  rf_plan_has_interior() tells whether a plan has such work; the
 * call is optional -- exchange_apply / finish run whatever is still pending -- and a no-op for plans without any.
 * `send` and `gathered` are caller-owned device buffers.  With shard_world == 1 the apply step
 * is a no-op and may be skipped.  Slabs may have different extents along the sharded dimension
 * (rf_filter_desc.shard_extents: a slab's exit carry is propagated across the slabs between it and the
 * receiver with one transfer table per slab); the exchanged bytes per rank do not depend on them. */
int rf_plan_num_exchanges(const rf_plan *plan);
size_t rf_plan_exchange_bytes(const rf_plan *plan, int exchange);
int rf_plan_begin(rf_plan *plan, const void *const *in_planes, void *const *out_planes, void *stream);
int rf_plan_exchange_local(rf_plan *plan, int exchange, void *send);
int rf_plan_exchange_apply(rf_plan *plan, int exchange, const void *gathered);
int rf_plan_has_interior(const rf_plan *plan);
int rf_plan_interior(rf_plan *plan);
int rf_plan_finish(rf_plan *plan);
/* Abandons the execute this host thread began with rf_plan_begin and did not finish (the caller's collective failed, an
 * exception unwound between the calls): the execution instance goes back to the plan's pool; kernels already enqueued still
 * run, the output planes are undefined.  A no-op without such an execute.  rf_plan_begin does the same to an unfinished
 * execute of the calling thread before it starts a new one, and rf_plan_destroy may be called in any state. */
int rf_plan_abort(rf_plan *plan);

/* ---- plan tables (host side of the tiling algebra; also what the CPU tests inspect) ------- */
/* Copies a named table into out (as doubles) and returns its element count through n_out;
 * pass out = NULL to query the size.  Names are documented in DESIGN.md ("plan tables").
 * "scans" lists the scans the plan EXECUTES, one row of 5 + 2 * RF_MAX_ORDER doubles each: where the plan rewrote the filter --
 * merged runs of a 1-D signal (one scan per run, RF_PLAN_NO_OVERLAP forbids it), sections of order <= 3 in place of a scan of
 * order 4..8 (RF_PLAN_NO_SECTIONS), the first stage of an in-plan cascade -- these are not the scans the caller gave. */
int rf_plan_table(const rf_plan *plan, const char *name, double *out, size_t capacity, size_t *n_out);

/* Debugging aid: device pointer and size of the i-th buffer the plan owns (tables, tails, carries,
 * in allocation order); RF_ERR_INVALID_ARG past the last one. */
int rf_plan_debug_buffer(const rf_plan *plan, int index, void **ptr_out, size_t *bytes_out);

/* ---- coefficient design (lib/iir_coeff.cpp:162-177, 222-234, 236-263, 205-220) ------------ */
int rf_gaussian_weights(float sigma, int order, float *coeff_out /* order+1 */);
int rf_integral_image_coeff(int n, float *coeff_out /* n+1 */);
int rf_overlap_feedback_coeff(const float *a, int na, const float *b, int nb, float *c_out /* na+nb */);
int rf_gaussian_box_filter(int k, float sigma, int *width_out);

/* ---- finite differences of summed-area tables (iterated box filters) --------------------- */
/* The consumer of the reference's box-filter apps (apps/box/box_filter.h:36-39, 128-139; apps/DoG/diff_gauss.cpp:
 * 132-150): along every dimension d, order[d] times,
 *     out(i) = (s(min(i + radius, N-1)) - s(max(i - radius - 1, 0))) / (2*radius + 1)
 * with s a summed-area table of matching order (rf_integral_image_coeff).  The nested clamps are evaluated exactly
 * as written there.  order[d] in 0..2.  One elementwise kernel over dense x-fastest device planes; float pixel types.
 * in == out is NOT allowed (the operator gathers). */
int rf_box_difference(const void *in, void *out, int ndim, const int64_t *extent, int dtype, int radius,
                      const int32_t *order, void *stream);

/* ---- clamped tap combinations (the other pointwise-with-offsets Funcs of the reference's apps) -------- */
/* HARNESS UTILITY, OUTSIDE THE HOT PATH: SURVEY.md 2c marks apps/DoG out of scope; this entry point exists so that
 * tools/profile_app.py can run the reference's diff_gauss app end to end on device buffers.  Nothing of the tiled
 * recursive-filter path (plans, kernels, sharding) uses it; a drop-in integration does not need to bind it. */
/* out(p) = sum_t weight_t * in_planes[plane_t]( clamp(p + offset_t) ),  clamp per dimension to [0, extent-1].
 * Covers every difference operator the apps put behind a summed-area table that rf_box_difference does not:
 * apps/DoG/diff_gauss.cpp:176-197 (diff_op_x / diff_op_y: taps +B, -1 (twice), -2B-2, one division; diff_op_xy with two
 * radii on one table; the final difference of two planes).  At most RF_MAX_TAPS taps over at most RF_MAX_PLANES input
 * planes; dense x-fastest device planes of one floating-point type; `out` must not alias an input.  One gather kernel. */
#define RF_MAX_TAPS 16
typedef struct {
    int32_t plane;                    /* index into in_planes                                      */
    int32_t offset[RF_MAX_DIMS];      /* per dimension, x first                                    */
    float   weight;
} rf_tap;
int rf_tap_filter(const void *const *in_planes, int n_in, void *out, int ndim, const int64_t *extent, int dtype,
                  const rf_tap *taps, int n_taps, void *stream);

/* ---- measured copy ceiling ("also report against a measured stream-copy ceiling") ---------- */
/* MEASUREMENT UTILITY: the final pass of the fused path with its arithmetic taken out -- every 256 x 128 tile of a dense f32
 * image of `rows` x `width` samples (width a multiple of 256, rows a multiple of 128) read with 16-byte non-temporal loads and
 * written with 16-byte non-temporal stores by one workgroup of 256 threads, eight loads per thread in flight.  What a
 * read-once / write-once pass over an image reaches on the device it runs on: bench.py times it beside the filter
 * (roofline.copy_ceiling_gbps).  src != dst.  Nothing of the filter path calls it. */
int rf_stream_copy(const float *src, float *dst, int64_t width, int64_t rows, void *stream);

/* ---- misc ------------------------------------------------------------------------------- */
const char *rf_last_error_string(void);
const char *rf_version(void);
/* number of visible HIP devices (0 when there is none); never fails */
int rf_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RECFILTER_AMD_H */
