// plan_fused.cpp -- plan for the LDS-staged fused x/y path (kernels_fused.hip).
//
// Stages of one execute (2-D; a 3-D filter runs this per z plane as a batch and then filters z
// with the generic dimension builder):
//   fused_tails      pass 1: tile-local tails of every x scan (per row) and the y tails' combined rows, by
//                    contraction with precomputed impulse responses
//   carry_x          x carry recurrence, all x scans (same-dimension chaining included)
//   xscan_rows       finishes the y tails: tile-local x scans of the combined rows + the cross-dimension
//                    residual of the completed x carries (lib/split.cpp:1215-1633)
//   carry_y          y carry recurrence, all y scans (one launch per scan around the exchanges when sharded)
//   fused_pass2      final correction pass
#include <cstring>
#include <memory>

#include "kernels_fused.h"
#include "plan.h"
#include "plan_generic.h"
#include "plan_strided.h"

namespace rf {

namespace {

int fused_order(const rf_plan *plan) {
    return std::max(plan->dims[0].k, plan->ndim > 1 ? plan->dims[1].k : 0);
}

// A long 1-D signal runs on the fused path folded into rows: N = NY rows of NX samples, every row tiled like an
// image row and the rows chained through their entering states ("chained rows").  Longest row first: fewer rows
// to chain, and the blocked carry scan still finds NY * MX/16 waves of work.
// A zero-border 1-D signal of any length is the same filter on the signal padded with zeros to the next multiple of 8192, as
// long as no anticausal scan follows a causal one: a causal scan never sees what follows it, and an anticausal scan that runs
// on untouched padding enters the signal with the zero state the padding leaves it in (fused_plan_applicable checks).
// The padded length is chosen so that the signal folds into LONG rows (few rows to chain): the longest row length whose
// 32-row granule costs at most 1/16 of padding.
int64_t chained_padded_length(int64_t N) {
    if (N % 8192 == 0) return N;
    for (int64_t nx = 16384; nx >= kFusedTX; nx /= 2) {
        const int64_t g = nx * 32, np = (N + g - 1) / g * g;
        if (np - N <= N / 16) return np;
    }
    return (N + 8191) / 8192 * 8192;
}

int64_t chained_row_length(int64_t N) {
    for (int64_t nx = 16384; nx >= kFusedTX; nx /= 2)
        if (N % nx == 0 && (N / nx) % 64 == 0) return nx;
    for (int64_t nx = 16384; nx >= kFusedTX; nx /= 2)
        if (N % nx == 0 && (N / nx) % 32 == 0) return nx;
    return 0;
}

template <typename P, typename S>
int build_fused(rf_plan *plan, const rf_filter_desc *desc) {
    using Acc = typename PixelTraits<P>::Acc;
    int status = RF_OK;
    plan->vector_access = true;                  // 16-byte chunks per lane in both passes
    const int K = fused_order(plan);
    const bool chained = plan->ndim == 1;        // 1-D signal folded into chained rows
    DimInfo &dx = plan->dims[0];
    DimInfo no_y;
    DimInfo &dy = chained ? no_y : plan->dims[1];
    const int64_t N1 = chained ? chained_padded_length(dx.N) : dx.N;     // chained: the length the kernels see
    // ... and whether they run on zero-padded COPIES (an epilogue that re-reads the input, RF_PAD_COPIES=1 for A/B runs) or on
    // the caller's buffers with the samples behind the signal's end masked (FusedArgs::lin_limit: no copy in, no copy out --
    // 34 us of 98 for one biquad over 10,000,000 samples)
    static const bool pad_copies_env = RF_KNOB("RF_PAD_COPIES") != nullptr;
    const bool in_place_tail = chained && N1 != dx.N && !pad_copies_env && !(plan->pw.post && plan->pw.post_i != 0.0);
    const bool padded = chained && N1 != dx.N && !in_place_tail;
    const int64_t NX = chained ? chained_row_length(N1) : dx.N;
    // Tuple planes of a 2-D filter ride in ONE launch per step, as the z planes of a volume whose planes are separate
    // buffers (FusedArgs::plane_batch): 5 launches instead of 5 per plane, which is most of the time of a small RGB image.
    const bool batch = plan->ndim == 2 && plan->n_planes > 1 && plan->n_planes <= kFusedMaxPlanes && !plan->sharded() &&
                       !(plan->flags & RF_PLAN_NO_PLANE_BATCH) && RF_KNOB("RF_NO_PLANE_BATCH") == nullptr;
    const int64_t NY = chained ? N1 / NX : dy.N, NZ = batch ? plan->n_planes : (plan->ndim > 2 ? plan->dims[2].N : 1);
    const size_t first_begin_step = plan->begin_steps.size(), first_finish_step = plan->finish_steps.size();
    // tile height: 64 rows unless only 32 divides the height; any other height runs 64-row tiles (32 below 33 rows)
    // with a partial last tile row
    // (row shards: decided on the slabs' common divisor, so that every rank tiles alike)
    const bool rows_sharded = plan->ndim == 2 && plan->sharded();
    const int64_t NYB = rows_sharded ? plan->shard_common : NY;
    int TY = (NYB % 64 == 0) ? 64 : (NYB % 32 == 0 || NYB < 32) ? 32 : 64;
    // a small image has too few 256 x 64 tiles to fill 256 CUs: half-height tiles double the workgroups
    // (2048^2: 43.5 -> 40.8 us, 1024^2: 38 -> 34.6 us; at 4096^2, 1024 tiles, the 64-row tiles win again)
    // (order 3 keeps 64 rows when 32-row tiles would be more than 64 per column: its carry scan along y then runs in two
    //  blocks -- 2112^2: 28 against 17 us, 80.7 against 75.1 us per filter, tools/mid_probe.py)
    if (TY == 64 && NY % 32 == 0 && !rows_sharded && ((NX + kFusedTX - 1) / kFusedTX) * (NY / 64) * NZ <= 384 &&
        !(K >= 3 && NY / 32 > 64))
        TY = 32;
    // ... and a mid-size image whose height is not a multiple of 64 runs the EDGE variants of its kernels either way: 32-row
    // tiles then waste less of the partial last tile row and cost the EDGE final pass less (tools/ty_probe.py, 64 -> 32 rows:
    // 5000^2 order 2 0.127 -> 0.123 ms, order 3 0.175 -> 0.156; 6000 x 8000 0.181 -> 0.172 / 0.234 -> 0.224; from ~4000 tiles
    // on the 64-row tiles are ahead again: 9000 x 12000 0.330 against 0.358)
    if (TY == 64 && NY % 64 != 0 && NY >= 32 && !rows_sharded && !chained &&
        ((NX + kFusedTX - 1) / kFusedTX) * ((NY + 63) / 64) * NZ <= 3200)
        TY = 32;
    // ... except at order 3 with 65..128 32-row tiles per column, where the y carry scan leaves its register-resident form
    // (kernels_carry.hip): 64-row tiles with a partial last row are ahead there (2160 x 3840: 74 against 85 us, 3000 x 4000:
    // 98 / 110, 4000^2: 112 / 123; above 128 tiles per column the 32-row tiles win again, 5000^2: 157 / 175 us)
    if (TY == 32 && K >= 3 && !rows_sharded && !chained && NY >= 64 && (NY + 31) / 32 > 64 && (NY + 31) / 32 <= 128)
        TY = 64;
    // Large images: 128-row tiles halve the y tails and the kernels that walk them between the passes; the final pass
    // takes such a tile through the LDS in two halves and keeps its columns in registers (kernels_fused_tall.hip).
    // Needs enough of them to fill the chip several times over (row shards: whole 128-row tiles per slab).  A height that is
    // not a multiple of 128 leaves a partial last tile row, which the final pass runs as a strip of its own on the EDGE
    // variant (kernels_fused_tall.hip): 16380 x 16384, 0.70 ms on 64-row tiles with the EDGE kernel everywhere.
    const int nx_early = (int)dx.scan_ids.size(), ny_early = (int)dy.scan_ids.size();
    // (end of round 2, same box, 64 against 128 rows: order 1, cfg4a 1.660 -> 1.636 ms, one bicubic plane 0.593 -> 0.575; order 2,
    // cfg3 0.624 -> 0.601; order 3, cfg4b 2.25 -> 1.87 ms; below ~4096 tiles -- cfg2, 8192^2 -- the 64-row tiles stay ahead)
    // (integer pixels keep 64 rows: their final pass needs more registers than the 128-sample column leaves)
    if ((TY == 64 || (TY == 32 && NY >= 4096)) && !chained && ny_early > 0 && nx_early > 0 && !PixelTraits<P>::is_integer &&
        (NYB % 128 == 0 || !rows_sharded) &&
        !(plan->pw.post && plan->pw.post_i != 0.0 && K <= 2) &&     // (orders 1, 2: an epilogue with an input operand keeps the input
                                                                    // column in registers, which a 128-sample column leaves no room for)
        RF_KNOB("RF_NO_TALL_TILES") == nullptr &&
        ((NX + kFusedTX - 1) / kFusedTX) * ((NY + 127) / 128) * NZ >= 4096)
        TY = 128;
    if (const int want = plan->fused_tile_rows() ? plan->fused_tile_rows() : RF_KNOB("RF_FUSED_TY") ? atoi(RF_KNOB("RF_FUSED_TY")) : 0) {
        // RF_PLAN_TILE_ROWS(n): the caller's tile height, where the shape admits it
        if (want == 32 || want == 64 || (want == 128 && !chained && (!rows_sharded || NYB % 128 == 0))) TY = want;
    }
    // scans in mod form (plan.cpp, "clamped sections"): whole tile rows, so that a modification sits at compile-time rows of
    // the column (fused_plan_applicable has checked that 32 divides the height)
    if (plan->mod_form && ny_early > 0)
        while (TY > 32 && NY % TY != 0) TY /= 2;
    // f64 pixels: a 256 x 32 tile is the 64 KiB of LDS a 256 x 64 tile of f32 takes (any height: partial last tile row)
    if (sizeof(Acc) == 8) {
        if (rows_sharded && NYB % 32 != 0) { set_error("row-sharded f64 slabs must be multiples of 32 rows"); return RF_ERR_UNSUPPORTED; }
        TY = 32;
    }
    const int nx = (int)dx.scan_ids.size(), ny = (int)dy.scan_ids.size();
    // The width only has to be a multiple of 4 (rows stay 16-byte aligned): the last tile of a row may be partial.  Its
    // missing samples are loaded as zeros and never stored; the tables of the "last tile" variants are built for the
    // samples that exist (tables.h, T_last), so a clamped anticausal scan enters at the true image border.
    // The height is arbitrary: the last tile row may be partial in the same way (rows loaded as zeros, never stored,
    // an anticausal y scan enters at the last existing row).
    const int MX = (int)((NX + kFusedTX - 1) / kFusedTX), MY = (int)((NY + TY - 1) / TY);
    const int TVx = (int)(NX - (int64_t)(MX - 1) * kFusedTX);      // samples of the last tile, (0, 256]
    const int TVy = (int)(NY - (int64_t)(MY - 1) * TY);            // rows of the last tile row, (0, TY]
    const int64_t NXP = (int64_t)MX * kFusedTX;                     // padded width: pitch of everything indexed by column
    const int64_t NYP = (int64_t)MY * TY;                           // padded height: pitch of everything indexed by row
    dx.T = kFusedTX; dx.M = chained ? N1 / kFusedTX : MX;
    dy.T = TY;       dy.M = MY;
    const int64_t Lx = NYP * NZ, Ly = NXP * NZ;
    const int outer = plan->ndim - 1;
    const bool y_is_exchange_dim = (outer == 1);
    const bool y_sharded = y_is_exchange_dim && plan->sharded();

    // ---- tables -------------------------------------------------------------------------
    auto table_scans = [&](const std::vector<int> &ids) {
        std::vector<ScanS<S>> v;
        for (int id : ids) v.push_back(make_table_scan<S>(plan->scans[id]));
        return v;
    };
    auto fused_scans = [&](const std::vector<int> &ids, bool with_segment_tables) {
        std::vector<FusedScan<Acc>> v;
        for (int id : ids) v.push_back(make_fused_scan<S, Acc>(plan->scans[id], K, with_segment_tables));
        return v;
    };
    auto dev_scans = [&](const std::vector<int> &ids) {
        std::vector<DevScan<Acc>> v;
        for (int id : ids) {
            DevScan<Acc> d = make_dev_scan<Acc>(plan->scans[id]);
            d.order = K;
            v.push_back(d);
        }
        return v;
    };
    auto flatten_W = [&](const DimTables<S> &tab, int n, std::vector<Acc> &hW, std::vector<Acc> &hA,
                         const std::string &dn) {
        hW.assign((size_t)4 * n * n * K * K, Acc(0));
        hA.assign((size_t)n * K * K, Acc(0));
        std::vector<double> dW(hW.size(), 0.0), dA(hA.size(), 0.0);
        for (int v = 0; v < 4; v++)
            for (int q = 0; q < n; q++)
                for (int s = q + 1; s < n; s++)
                    for (int e = 0; e < K * K; e++) {
                        size_t idx = (((size_t)v * n + q) * n + s) * K * K + e;
                        hW[idx] = table_to_acc<S, Acc>(tab.Wm(v, q, s)[e]);
                        dW[idx] = table_to_double<S>(tab.Wm(v, q, s)[e]);
                    }
        for (int s = 0; s < n; s++)
            for (int e = 0; e < K * K; e++) {
                hA[(size_t)s * K * K + e] = table_to_acc<S, Acc>(tab.A[s][e]);
                dA[(size_t)s * K * K + e] = table_to_double<S>(tab.A[s][e]);
            }
        plan->tables["W_" + dn] = dW;
        plan->tables["A_" + dn] = dA;
    };

    std::vector<FusedScan<Acc>> hxs = fused_scans(dx.scan_ids, true), hys = fused_scans(dy.scan_ids, false);
    if (nx > 0) {
        // segment tables as the x phase reads them (float pixels: exact values of the kernel constants)
        std::vector<double> sr, sp;
        for (const auto &f : hxs) {
            for (int p = 0; p < kFusedSeg; p++)
                for (int j = 0; j < K; j++) sr.push_back((double)f.R[j][f.causal ? p : kFusedSeg - 1 - p]);
            for (int step = 0; step < 4; step++)
                for (int r = 0; r < K; r++)
                    for (int j = 0; j < K; j++) sp.push_back((double)f.P[step][r][j]);
        }
        plan->tables["seg_R_x"] = sr;
        plan->tables["seg_P_x"] = sp;
    }
    std::vector<DevScan<Acc>> hxd = dev_scans(dx.scan_ids), hyd = dev_scans(dy.scan_ids);
    std::vector<Acc> hWx, hAx, hWy, hAy, hG, hAMy, hACx, hACy, hHx, hHy, hAMx, hAMSx, hApowX, hApowY;
    const int chain_S = (int)((NY + 63) / 64);     // rows per lane of the row-chain kernel
    const int Cx = carry_chunk_length(MX, Lx, K), Cy = carry_chunk_length(MY, Ly, K);
    if (nx > 0) {
        DimTables<S> tx = build_dim_tables<S>(table_scans(dx.scan_ids), K, kFusedTX, plan->clamped, TVx);
        flatten_W(tx, nx, hWx, hAx, "x");
        if (chained) hApowX = carry_apply_powers<S, Acc>(tx.A, MX, K);
        {
            std::vector<S> H = build_tail_responses<S>(table_scans(dx.scan_ids), K, kFusedTX, plan->clamped, TVx);
            std::vector<double> dH(H.size());
            hHx.resize(H.size());
            for (size_t e = 0; e < H.size(); e++) { hHx[e] = table_to_acc<S, Acc>(H[e]); dH[e] = table_to_double<S>(H[e]); }
            plan->tables["H_x"] = dH;
        }
        hACx.assign((size_t)nx * K * K, Acc(0));
        hAMx.assign((size_t)nx * K * K, Acc(0));
        hAMSx.assign((size_t)nx * K * K, Acc(0));
        for (int s = 0; s < nx; s++) {
            std::vector<S> ac = mat_pow<S>(tx.A[s], Cx, K), am = mat_pow<S>(tx.A[s], MX, K);
            std::vector<S> ams = mat_pow<S>(am, chain_S, K);
            for (int e = 0; e < K * K; e++) {
                hACx[(size_t)s * K * K + e] = table_to_acc<S, Acc>(ac[e]);
                hAMx[(size_t)s * K * K + e] = table_to_acc<S, Acc>(am[e]);
                hAMSx[(size_t)s * K * K + e] = table_to_acc<S, Acc>(ams[e]);
            }
        }
        // G[v][q][o][xi]: what the carry entering x scan q adds to the tile after ALL x scans
        hG.assign((size_t)4 * nx * kFusedTX * K, Acc(0));
        std::vector<double> dG(hG.size());
        for (int v = 0; v < 4; v++)
            for (int q = 0; q < nx; q++) {
                const std::vector<S> &Pm = tx.P(v, q, nx - 1);          // [xi][o]
                for (int xi = 0; xi < kFusedTX; xi++)
                    for (int o = 0; o < K; o++) {
                        size_t idx = (((size_t)v * nx + q) * K + o) * kFusedTX + xi;
                        hG[idx] = table_to_acc<S, Acc>(Pm[(size_t)xi * K + o]);
                        dG[idx] = table_to_double<S>(Pm[(size_t)xi * K + o]);
                    }
            }
        plan->tables["G_x"] = dG;
    }
    DimTables<S> ty;
    if (ny > 0) {
        ty = build_dim_tables<S>(table_scans(dy.scan_ids), K, TY, plan->clamped, TVy);
        flatten_W(ty, ny, hWy, hAy, "y");
        if (y_sharded) hApowY = carry_apply_powers<S, Acc>(ty.A, MY, K);
        {
            std::vector<S> H = build_tail_responses<S>(table_scans(dy.scan_ids), K, TY, plan->clamped, TVy);
            std::vector<double> dH(H.size());
            hHy.resize(H.size());
            for (size_t e = 0; e < H.size(); e++) { hHy[e] = table_to_acc<S, Acc>(H[e]); dH[e] = table_to_double<S>(H[e]); }
            plan->tables["H_y"] = dH;
        }
        if (y_sharded) hAMy = slab_powers<S, Acc>(plan, ty.A, TY, K);          // [j][slab][K x K], for the per-scan exchange
        hACy.assign((size_t)ny * K * K, Acc(0));
        for (int j = 0; j < ny; j++) {
            std::vector<S> ac = mat_pow<S>(ty.A[j], Cy, K);
            for (int e = 0; e < K * K; e++) hACy[(size_t)j * K * K + e] = table_to_acc<S, Acc>(ac[e]);
        }
    }

    // ---- device memory --------------------------------------------------------------------
    auto up = [&](const auto &vec) {
        using T = typename std::decay<decltype(vec)>::type::value_type;
        return (const T *)plan->upload(vec.data(), vec.size() * sizeof(T), &status);
    };
    const DevScan<Acc> *d_xd = up(hxd);
    const DevScan<Acc> *d_yd = up(hyd);
    const Acc *d_Wx = up(hWx), *d_Ax = up(hAx), *d_Wy = up(hWy), *d_Ay = up(hAy), *d_G = up(hG), *d_AMy = up(hAMy);
    const Acc *d_ACx = up(hACx), *d_ACy = up(hACy), *d_Hx = up(hHx), *d_Hy = up(hHy);
    const Acc *d_AMx = up(hAMx), *d_AMSx = up(hAMSx);
    const Acc *d_ApowX = up(hApowX), *d_ApowY = up(hApowY);      // only filled when carry_apply runs (chained rows, row shards)

    const size_t xt_pp = (size_t)nx * MX * K * Lx, yt_pp = (size_t)ny * MY * K * Ly;
    const size_t xin_pp = (size_t)nx * K * Lx, yin_pp = (size_t)ny * K * Ly;
    const int np = batch ? 1 : plan->n_planes;         // batched planes are inside Lx / Ly already (NZ)
    // few tiles per row: xscan_rows completes the x tails itself, into a second array (kernels_tails.hip, XC)
    const bool merged_cx = !chained && xscan_completes_x_tails(K, TY, (int)MX, nx, ny, sizeof(Acc), (int64_t)MY * NZ);
    Acc *xt = (Acc *)plan->alloc(xt_pp * np * sizeof(Acc), false, &status);
    Acc *xt_done = merged_cx ? (Acc *)plan->alloc(xt_pp * np * sizeof(Acc), false, &status) : nullptr;
    Acc *yt = (Acc *)plan->alloc(yt_pp * np * sizeof(Acc), false, &status);
    Acc *xin = (Acc *)plan->alloc(xin_pp * np * sizeof(Acc), true, &status);
    Acc *yin = (Acc *)plan->alloc(yin_pp * np * sizeof(Acc), true, &status);
    // (two of them: with the chain of scan s folded into the carry launch of scan s + 1 that launch reads one and writes the other)
    Acc *row_exit = chained ? (Acc *)plan->alloc((size_t)2 * K * Lx * np * sizeof(Acc), true, &status) : nullptr;
    if (status != RF_OK) return status;

    // ---- 3-D volumes: pass 1 in ONE read (kernels_tails_walk.hip) ------------------------------------
    // The z tails are taken from the raw input by the pass that extracts the x/y tails (the z operators commute with the x/y
    // filter: plan_strided.h); the z stage then has no first pass.  f32 volumes of whole tiles without a prologue (its bias is not
    // linear) -- whole
    // volumes, and z slabs that take the early exchange (the pass is then the slab's begin step) -- of at least one patch column (256 x 32 samples x one z tile: a workgroup of 1024 threads) per compute unit --
    // measured, one read against two: 256^3 (32 patch columns) 0.214 against 0.124 ms, 512^3 (256) 0.645 against 0.680,
    // 768^3 1.97 / 2.05, 1024^3 4.5 / 4.95, 2048^3 34.0 / 37.0 (profiles/r4/walk_tails_sizes.txt); RF_PLAN_WALK_PASS1: whatever
    // the size; RF_PLAN_STAGED_PASS1 keeps the two first passes.
    WalkArgs walk_args{};
    std::shared_ptr<WalkHook> walk_hook;
    std::unique_ptr<rf_plan> walk_child;             // F over the z carry planes (plan_strided.h), handed to the z stage below
    if constexpr (std::is_same<P, float>::value) {
        static const char *walk_knob = RF_KNOB("RF_WALK");                        // A/B: 0 = never
        const bool wanted = !(plan->flags & RF_PLAN_STAGED_PASS1) && !(walk_knob && atoi(walk_knob) == 0);
        const bool z_slabs = plan->sharded();           // z slabs: with the early exchange (plan_strided.h), whose first step this pass then is
        if (wanted && plan->ndim == 3 && (!z_slabs || early_exchange_possible<P>(plan, 2, desc)) && !batch && !chained && !plan->mod_form &&
            !plan->pw.in_u8 && nx > 0 && ny > 0 && !plan->dims[2].scan_ids.empty() &&      // (a prologue x' = s x + b is applied as the samples arrive; an epilogue runs behind the z stage either way)
            plan->dims[2].lines == NX * NY) {
            const DimInfo &dz = plan->dims[2];
            const int TZ = strided_tile(plan, 2), nz = (int)dz.scan_ids.size(), KZ = dz.k;
            const int64_t patch_columns = TZ > 0 ? (int64_t)MX * ((NY + 31) / 32) * (dz.N / TZ) : 0;
            // (widths that are not multiples of four: the pass takes them since round 6 -- 4-byte loads, kernels_tails_walk.hip U4 --
            //  but loses to the two first passes there, 1024 x 1021 x 1021: walk 2.27 ms against 1.15 + 0.94, 1022 wide: 2.09 against
            //  1.13 + 1.04 (profiles/r6/walk_odd_widths.txt); only RF_PLAN_WALK_PASS1 selects it for such volumes)
            if (TZ > 0 && dz.N % TZ == 0 && walk_tails_applicable(K, TY, nx, ny, nz, KZ, TZ, TVx, TVy) &&
                ((patch_columns >= 256 && NX % 4 == 0) || (plan->flags & RF_PLAN_WALK_PASS1))) {
                const int MZ = (int)(dz.N / TZ);
                walk_child.reset(build_carry_planes_plan(plan, desc, 2, (int64_t)nz * KZ * (MZ + (z_slabs ? 1 : 0))));
                if (walk_child) {
                    // impulse responses of the z tails, transposed: [variant][z][4]
                    std::vector<S> H = build_tail_responses<S>(table_scans(dz.scan_ids), KZ, TZ, plan->clamped);
                    std::vector<float> hHz((size_t)4 * TZ * 4, 0.0f);
                    std::vector<double> dHz(H.size());
                    for (size_t e = 0; e < H.size(); e++) dHz[e] = table_to_double<S>(H[e]);
                    for (int v = 0; v < 4; v++)
                        for (int j = 0; j < nz * KZ; j++)
                            for (int z = 0; z < TZ; z++)
                                hHz[((size_t)v * TZ + z) * 4 + j] = table_to_acc<S, Acc>(H[((size_t)v * nz * KZ + j) * TZ + z]);
                    plan->tables["H_z"] = dHz;
                    // tall patches (128 columns x 64 rows, round 6): 128-row y tiles, at most four x tails
                    static const bool tall_off = RF_KNOB("RF_WALK_NO_TALL") != nullptr;       // A/B
                    const bool tall = !tall_off && TY == 128 && K <= 2 && nx * K <= 4;
                    const int parts = tall ? TY / 64 : TY / 32;
                    walk_args.tall = tall ? 1 : 0;
                    walk_args.xt2 = tall ? (float *)plan->alloc(xt_pp * np * sizeof(float), false, &status) : nullptr;      // (per Tuple plane)
                    walk_args.HzT = (const float *)plan->upload(hHz.data(), hHz.size() * sizeof(float), &status);
                    walk_args.TY = TY; walk_args.TZ = TZ; walk_args.MZ = MZ; walk_args.nzk = nz * KZ; walk_args.KZ = KZ;
                    walk_args.parts_log2 = parts == 4 ? 2 : parts == 2 ? 1 : 0;
                    walk_args.z_first_border = (!z_slabs || plan->shard_rank == 0) ? 1 : 0;
                    walk_args.z_last_border = (!z_slabs || plan->shard_rank == plan->shard_world - 1) ? 1 : 0;
                    walk_args.part_stride = (int64_t)yt_pp;
                    walk_args.ytp = parts > 1 ? (float *)plan->alloc(yt_pp * parts * np * sizeof(float), false, &status) : nullptr;      // (per Tuple plane)
                    walk_hook = std::make_shared<WalkHook>();
                    if (status != RF_OK) return status;
                }
            }
        }
    }
    const bool walk = (bool)walk_hook;

    FusedArgs<Acc> fbase{};
    fbase.NX = NX; fbase.NY = NY; fbase.NZ = NZ; fbase.MX = MX; fbase.MY = MY; fbase.nx = nx; fbase.ny = ny;
    fbase.NXP = NXP; fbase.last_lane = (TVx - 1) / kFusedSeg; fbase.last_cols = TVx;
    fbase.NYP = NYP; fbase.last_rows = TVy;
    fbase.row_bytes = (uint32_t)(NX * (int64_t)sizeof(P));
    fbase.clamped = plan->clamped ? 1 : 0;
    fbase.mod_form = plan->mod_form ? 1 : 0;
    fbase.y_first_border = (!y_sharded || plan->shard_rank == 0) ? 1 : 0;
    fbase.y_last_border = (!y_sharded || plan->shard_rank == plan->shard_world - 1) ? 1 : 0;
    if constexpr (!PixelTraits<P>::is_integer) {
        const bool z_follows = plan->ndim > 2 && !plan->dims[2].scan_ids.empty();
        plan->pw.pre_fused = plan->pw.pre;
        plan->pw.post_fused = plan->pw.post && !z_follows;     // with a z stage the epilogue runs after it
        fbase.pw_flags = (plan->pw.pre_fused ? 1 : 0) | (plan->pw.post_fused ? 2 : 0);
        fbase.pre_s = (Acc)plan->pw.pre_s; fbase.pre_b = (Acc)plan->pw.pre_b;
        fbase.post_f = (Acc)plan->pw.post_f; fbase.post_i = (Acc)plan->pw.post_i; fbase.post_b = (Acc)plan->pw.post_b;
    }
    // The y tails of an unsharded plan are tile-major, [tile row][tile column][scan][r][256]: the rows pass 1 stores for
    // one tile, and the ones pass 2 loads, are then one run of ny * K * 1 KiB instead of ny * K runs a whole image width
    // apart (FusedArgs::yt_index).  Slabs keep [scan][tile row][r][line], which the exchange kernels address, and so do
    // order-3 filters: measured on 16384^2, order 2 gains 0.005 ms of 0.62 ms and order 3 loses 0.01-0.03 ms of 1.98 ms
    // (its carry scan reads six rows per tile and line), order 1 is unchanged either way.
    static const bool yt_row_major = RF_KNOB("RF_YT_ROW_MAJOR") != nullptr;      // A/B runs
    static const bool yt_force_tile = RF_KNOB("RF_YT_TILE_MAJOR") != nullptr;    // A/B runs: order 3 too
    // (the one-read pass 1 of a volume stores a tile's combined rows as one run: tile-major for order 3 as well)
    const bool yt_tile_major = !yt_row_major && !y_sharded && ny > 0 && (K <= 2 || yt_force_tile || walk) && Ly % kFusedTX == 0 &&
                               Ly == NXP * (int64_t)NZ;
    fbase.yt_tile_major = yt_tile_major ? 1 : 0;
    fbase.lin_limit = in_place_tail ? dx.N : 0;
    if constexpr (std::is_same<P, float>::value) {
        // (pass 1 left the combined rows in parts: xscan_rows adds them up)
        if (walk && walk_args.ytp) { fbase.yt_parts = walk_args.tall ? TY / 64 : TY / 32; fbase.yt_part_stride = walk_args.part_stride; fbase.ytp = walk_args.ytp; }
    }
    std::memset(fbase.xs, 0, sizeof(fbase.xs));
    std::memset(fbase.ys, 0, sizeof(fbase.ys));
    for (int s = 0; s < nx; s++) fbase.xs[s] = hxs[s];
    for (int j = 0; j < ny; j++) {
        fbase.ys[j].causal = hys[j].causal;
        fbase.ys[j].b = hys[j].b;
        for (int e = 0; e < kFusedMaxK; e++) fbase.ys[j].a[e] = hys[j].a[e];
        fbase.ys[j].mod_n = hys[j].mod_n;
        for (int e = 0; e < kFusedMaxMod; e++) fbase.ys[j].mod_g[e] = hys[j].mod_g[e];
    }
    auto fargs = [=](int pl) {
        FusedArgs<Acc> a = fbase;
        a.xt = xt + (size_t)pl * xt_pp;
        a.yt = yt + (size_t)pl * yt_pp;
        if (a.ytp != nullptr) a.ytp = a.ytp + (size_t)pl * yt_pp * (size_t)a.yt_parts;       // (the parts of the one-read pass 1)
        a.y_incoming = yin + (size_t)pl * yin_pp;
        a.x_incoming = xin + (size_t)pl * xin_pp;
        a.plane_batch = batch ? 1 : 0;
        if (batch)
            for (int i = 0; i < plan->n_planes; i++) { a.in_planes[i] = plan->in[i]; a.out_planes[i] = plan->xy_result(i); }
        return a;
    };
    GenericDimArgs<Acc> gx{};
    gx.g = LineGeom{NX, 1, Lx};
    gx.T = kFusedTX; gx.M = MX; gx.k = K; gx.n_scans = nx; gx.clamped = fbase.clamped;
    gx.first_is_border = 1; gx.last_is_border = 1;
    gx.scans = d_xd; gx.W = d_Wx; gx.A = d_Ax; gx.Apow = hApowX.empty() ? nullptr : d_ApowX;
    auto gxargs = [=](int pl) {
        GenericDimArgs<Acc> a = gx;
        a.tails = xt + (size_t)pl * xt_pp;
        a.incoming = xin + (size_t)pl * xin_pp;
        if constexpr (std::is_same<P, float>::value) {
            if (walk_args.xt2) a.tails_part2 = walk_args.xt2 + (size_t)pl * xt_pp;       // (tall patches: the x tails in two parts)
        }
        return a;
    };
    GenericDimArgs<Acc> gy{};
    gy.g = LineGeom{NY, NXP, Ly};
    gy.T = TY; gy.M = MY; gy.k = K; gy.n_scans = ny; gy.clamped = fbase.clamped;
    gy.first_is_border = fbase.y_first_border; gy.last_is_border = fbase.y_last_border;
    gy.scans = d_yd; gy.W = d_Wy; gy.A = d_Ay; gy.Apow = hApowY.empty() ? nullptr : d_ApowY;
    gy.tile_major = fbase.yt_tile_major;
    auto gyargs = [=](int pl) {
        GenericDimArgs<Acc> a = gy;
        a.tails = yt + (size_t)pl * yt_pp;
        a.incoming = yin + (size_t)pl * yin_pp;
        return a;
    };

    uint32_t xmask = 0, ymask = 0;
    for (int s = 0; s < nx; s++) if (hxs[s].causal) xmask |= 1u << s;
    for (int j = 0; j < ny; j++) if (hys[j].causal) ymask |= 1u << j;

    // ---- steps -----------------------------------------------------------------------------
    // pass 1: tail extraction by contraction with the impulse responses (kernels_tails.hip)
    Step p1;
    p1.name = "fused_tails";
    const int stream_mode = (plan->flags & RF_PLAN_STREAM_PASS1) ? 1 : (plan->flags & RF_PLAN_STAGED_PASS1) ? -1 : 0;
    const int mfma_mode = (plan->flags & RF_PLAN_MFMA_PASS1) ? 1 : (plan->flags & RF_PLAN_STAGED_PASS1) ? -1 : 0;
    if (walk) p1.name = "walk_tails";
    p1.run = [plan, fargs, K, TY, d_Hx, d_Hy, padded, stream_mode, mfma_mode, walk_args, walk_hook, xt_pp](int pl) {
        (void)xt_pp;
        const FusedArgs<Acc> a = fargs(pl);
        (void)stream_mode;
        (void)mfma_mode;
        (void)walk_args;
        if constexpr (std::is_same<P, float>::value) {
            if (walk_hook) {
                WalkArgs wa = walk_args;
                wa.zt = walk_hook->zt + (size_t)pl * walk_hook->zt_stride;
                if (wa.xt2) wa.xt2 += (size_t)pl * xt_pp;
                if (wa.ytp) wa.ytp = const_cast<float *>(a.ytp);       // this plane's parts (fargs)
                else wa.ytp = a.yt;                         // one patch per y tile: the combined rows go where they belong
                return launch_walk_tails(K, (const float *)plan->in[pl], a, wa, d_Hx, d_Hy, plan->stream);
            }
        }
        // images of whole 256 x 64 tiles stream through the LDS-DMA ring (kernels_stream.hip)
        if constexpr (std::is_same<P, float>::value) {
            if (a.lin_limit == 0 && stream_tails_applicable(K, TY, plan->pw.in_u8, a.pw_flags, a.last_cols, a.last_rows, (int64_t)a.MX * a.MY * a.NZ, a.MX,
                                        a.NZ, a.nx * K, a.ny * K, stream_mode))
                return launch_stream_tails(K, (const float *)(padded ? plan->pad_in[pl] : plan->in[pl]), a, d_Hx, d_Hy, plan->stream);
            // ... other f32 images of whole tiles contract their x tails on the matrix cores (kernels_tails_mfma.hip)
            if (mfma_tails_applicable(K, TY, plan->pw.in_u8, a.pw_flags, a.last_cols, a.last_rows, a.lin_limit, a.nx, a.ny, mfma_mode))
                return launch_mfma_tails(K, TY, (const float *)(padded ? plan->pad_in[pl] : plan->in[pl]), a, d_Hx, d_Hy, plan->stream);
        }
        return launch_fused_tails<P>(K, TY, padded ? plan->pad_in[pl] : plan->in[pl], plan->pw.in_u8, a, d_Hx, d_Hy, plan->stream);
    };
    if (padded) {
        // the zero-padded copies the kernels run on (the padding of the input copy is written once, here, and never again)
        plan->padded_len = N1;
        const size_t user_bytes = (size_t)dx.N * sizeof(P);
        for (int pl = 0; pl < plan->n_planes; pl++) {
            plan->pad_in[pl] = plan->alloc((size_t)N1 * sizeof(P), true, &status);
            plan->pad_out[pl] = plan->alloc((size_t)N1 * sizeof(P), false, &status);
        }
        if (status != RF_OK) return status;
        Step ci;
        ci.name = "pad_copy_in";
        ci.run = [plan, user_bytes](int pl) -> int {
            RF_HIP_CHECK(hipMemcpyAsync(plan->pad_in[pl], plan->in[pl], user_bytes, hipMemcpyDeviceToDevice, plan->stream));
            return (int)RF_OK;
        };
        plan->begin_steps.push_back(ci);
    }
    plan->begin_steps.push_back(p1);
    if (nx > 0 && !chained && !merged_cx) {
        Step cx;
        cx.name = "carry_x";
        cx.run = [plan, gxargs, K, nx, d_ACx, Cx, xmask](int pl) {
            return launch_carry_block<Acc>(K, gxargs(pl), xmask, 0, nx, (Acc *)nullptr, d_ACx, Cx, plan->stream);
        };
        plan->begin_steps.push_back(cx);
    }
    if (chained) {
        // Chained rows: per scan, (1) the blocked carry scan of every row with a zero entering state, publishing the
        // rows' exit states, (2) the chain over the rows -> state entering every row, (3) that state propagated through
        // the row's tails.  The same three steps as a sharded dimension (exchange_local / gather / exchange_apply),
        // with rows in the role of slabs.  Scan s+1 chains on scan s's completed carries, hence scan by scan.
        // Where the rows' entering states fit the LDS the chain and its propagation are not launches of their own: scan s's
        // are done by the carry launch of scan s + 1 before its own scan (carry_block_kernel PRE), the last scan's by
        // chain_apply_kernel -- n + 1 launches for n scans.
        static const bool no_pre = RF_KNOB("RF_NO_CHAIN_PRE") != nullptr;        // A/B runs: chain_apply after every scan
        const bool one_launch_chain = chain_apply_applies(K, Lx, sizeof(Acc)) && !hApowX.empty();
        const size_t exit_pp = (size_t)K * Lx;            // one buffer of exit states (two per plane, by scan parity)
        for (int s = 0; s < nx; s++) {
            const bool causal = hxs[s].causal != 0;
            const bool fold_prev = one_launch_chain && !no_pre && s > 0;
            Step cs;
            cs.name = "carry_x" + std::to_string(s);
            ChainPre<Acc> pre{};
            if (fold_prev) {
                pre.AM = d_AMx + (size_t)(s - 1) * K * K; pre.AMS = d_AMSx + (size_t)(s - 1) * K * K;
                pre.Apow = d_ApowX + (size_t)(s - 1) * MX * K * K;
                pre.S = chain_S; pre.causal_prev = hxs[s - 1].causal != 0 ? 1 : 0;
            }
            cs.run = [plan, gxargs, K, s, d_ACx, Cx, xmask, row_exit, exit_pp, xin, xin_pp, Lx, pre, fold_prev](int pl) {
                Acc *mine = row_exit + ((size_t)pl * 2 + (s & 1)) * exit_pp;
                if (!fold_prev)
                    return launch_carry_block<Acc>(K, gxargs(pl), xmask, s, s + 1, mine, d_ACx, Cx, plan->stream);
                ChainPre<Acc> p = pre;
                p.exit_states = row_exit + ((size_t)pl * 2 + ((s - 1) & 1)) * exit_pp;
                p.incoming_prev = xin + (size_t)pl * xin_pp + (size_t)(s - 1) * K * Lx;
                return launch_carry_block<Acc>(K, gxargs(pl), xmask, s, s + 1, mine, d_ACx, Cx, plan->stream, &p);
            };
            plan->begin_steps.push_back(cs);
            if (one_launch_chain) {
                if (!no_pre && s + 1 < nx) continue;          // the next scan's carry launch finishes this one
                // the chain over the rows and the propagation through their tails in one launch (kernels_carry.hip)
                Step ca;
                ca.name = "chain_apply" + std::to_string(s);
                const Acc *AMs = d_AMx + (size_t)s * K * K, *AMSs = d_AMSx + (size_t)s * K * K;
                ca.run = [plan, gxargs, K, s, causal, row_exit, exit_pp, xin, xin_pp, Lx, AMs, AMSs, chain_S](int pl) {
                    Acc *inc = xin + (size_t)pl * xin_pp + (size_t)s * K * Lx;
                    return launch_chain_apply<Acc>(K, gxargs(pl), s, row_exit + ((size_t)pl * 2 + (s & 1)) * exit_pp, inc, causal, AMs,
                                                   AMSs, chain_S, plan->stream);
                };
                plan->begin_steps.push_back(ca);
                continue;
            }
            Step rc;
            rc.name = "row_chain" + std::to_string(s);
            const Acc *AMs = d_AMx + (size_t)s * K * K, *AMSs = d_AMSx + (size_t)s * K * K;
            rc.run = [plan, K, s, causal, row_exit, exit_pp, xin, xin_pp, Lx, AMs, AMSs, chain_S](int pl) {
                Acc *inc = xin + (size_t)pl * xin_pp + (size_t)s * K * Lx;
                return launch_row_chain<Acc>(K, row_exit + ((size_t)pl * 2 + (s & 1)) * exit_pp, inc, (int)Lx, causal, AMs, AMSs, chain_S,
                                             plan->stream);
            };
            plan->begin_steps.push_back(rc);
            Step ap;
            ap.name = "carry_x_apply" + std::to_string(s);
            ap.run = [plan, gxargs, s](int pl) { return launch_generic_carry_apply<Acc>(gxargs(pl), s, plan->stream); };
            plan->begin_steps.push_back(ap);
        }
    }
    if (nx > 0 && ny > 0) {
        // finishes the y tails: tile-local x scans of the combined rows + the cross-dimension residual of the
        // completed x carries (lib/split.cpp:1215-1633)
        Step xs;
        xs.name = "xscan_rows";
        xs.run = [plan, fargs, K, TY, d_Hy, d_G, merged_cx, d_Wx, d_Ax, xt_done, xt_pp](int pl) {
            if (merged_cx)
                return launch_xscan_rows<Acc>(K, TY, fargs(pl), d_Hy, d_G, plan->stream, d_Wx, d_Ax, xt_done + (size_t)pl * xt_pp);
            return launch_xscan_rows<Acc>(K, TY, fargs(pl), d_Hy, d_G, plan->stream);
        };
        plan->begin_steps.push_back(xs);
    }
    const Acc *d_Yapply = nullptr;
    if (!y_sharded) {   // one launch for every y scan; per-scan launches only around the exchanges
        if (ny > 0) {
            Step cy;
            cy.name = "carry_y";
            cy.run = [plan, gyargs, K, ny, d_ACy, Cy, ymask](int pl) {
                return launch_carry_block<Acc>(K, gyargs(pl), ymask, 0, ny, (Acc *)nullptr, d_ACy, Cy, plan->stream);
            };
            plan->begin_steps.push_back(cy);
        }
    } else if (merged_exchange_applies(ny, K, plan->shard_world)) {
        // one all-gather for all y scans (plan_generic.h, "merged exchange")
        // ... whose correction of the tails is left to pass 2 (FusedArgs::y_apply): no launch between gather and pass 2
        static const bool separate_apply = RF_KNOB("RF_SHARD_SEPARATE_APPLY") != nullptr;     // A/B runs
        int rc = add_merged_exchange<S, Acc>(plan, ty, "y", MY, TY, Ly, ymask, gyargs, yin, yin_pp, d_ACy, Cy, "carry_y",
                                             separate_apply ? nullptr : &d_Yapply);
        if (rc != RF_OK) return rc;
    } else {
        for (int j = 0; j < ny; j++) {
            const int64_t plane_stride = (int64_t)K * Ly;
            const int ex_index = (int)plan->exchanges.size();
            rf_plan::Exchange ex;
            ex.bytes = (size_t)np * K * Ly * sizeof(Acc);
            ex.scratch = plan->alloc(ex.bytes, true, &status);
            if (status != RF_OK) return status;
            ex.send = ex.scratch;
            const Acc *AMj = d_AMy + (size_t)j * plan->shard_world * K * K;
            const int64_t rank_stride = (int64_t)np * K * Ly;
            ex.form_incoming = [plan, gyargs, j, rank_stride, plane_stride, AMj](const void *gathered) {
                for (int pl = 0; pl < plan->n_planes; pl++) {
                    int rc = launch_gather_incoming<Acc>(gyargs(pl), j, (const Acc *)gathered, rank_stride,
                                                         pl * plane_stride, plan->shard_rank, plan->shard_world, AMj,
                                                         plan->stream);
                    if (rc) return rc;
                }
                return (int)RF_OK;
            };
            plan->exchanges.push_back(ex);
            Step cy;
            cy.name = "carry_y" + std::to_string(j);
            cy.run = [plan, gyargs, K, j, d_ACy, Cy, ex_index, plane_stride, ymask](int pl) {
                Acc *send = (Acc *)plan->exchanges[ex_index].send;
                return launch_carry_block<Acc>(K, gyargs(pl), ymask, j, j + 1, send ? send + pl * plane_stride : nullptr,
                                               d_ACy, Cy, plan->stream);
            };
            plan->exchange_local_steps.push_back({cy});
            Step ap;
            ap.name = "carry_y_apply" + std::to_string(j);
            ap.run = [plan, gyargs, j](int pl) { return launch_generic_carry_apply<Acc>(gyargs(pl), j, plan->stream); };
            plan->exchange_apply_steps.push_back({ap});
        }
    }
    Step p2;
    p2.name = "fused_pass2";
    p2.run = [plan, fargs, K, TY, d_Yapply, padded, merged_cx, xt_done, xt_pp](int pl) {
        FusedArgs<Acc> a = fargs(pl);
        a.y_apply = d_Yapply;
        if (merged_cx) a.xt = xt_done + (size_t)pl * xt_pp;
        if constexpr (sizeof(Acc) == 4) {
            if (TY == 128) return launch_fused_pass2_tall<P>(K, plan->in[pl], plan->pw.in_u8, (P *)plan->xy_result(pl), a, plan->stream);
        }
        return launch_fused_pass2<P>(K, TY, padded ? plan->pad_in[pl] : plan->in[pl], plan->pw.in_u8, (P *)(padded ? plan->pad_out[pl] : plan->xy_result(pl)), a,
                                     plan->stream);
    };
    if (y_is_exchange_dim) plan->finish_steps.push_back(p2);
    else plan->begin_steps.push_back(p2);
    if (padded) {
        const size_t user_bytes = (size_t)dx.N * sizeof(P);
        Step co;
        co.name = "pad_copy_out";
        co.run = [plan, user_bytes](int pl) -> int {
            RF_HIP_CHECK(hipMemcpyAsync(plan->out[pl], plan->pad_out[pl], user_bytes, hipMemcpyDeviceToDevice, plan->stream));
            return (int)RF_OK;
        };
        plan->begin_steps.push_back(co);
    }

    // ---- z (3-D): filtered after the fused x/y stage, reading and writing the output planes ----
    if (plan->ndim > 2 && !plan->dims[2].scan_ids.empty()) {
        // Intermediate volume (RF_PLAN_INPLACE_Z forbids it): a final z pass that reads and writes the SAME addresses is 4 %
        // slower than one from one volume to another -- its write front follows its read front through the same DRAM banks
        // (tools/microbench/zpass_shape.hip: 2.95 against 2.82 ms per 512 planes of 2048^2; config 5 at 2048^3: 12.3 -> 11.8 ms).
        // Large volumes on the strided kernels therefore get a plan-owned volume between the two stages, as long as it is at
        // most a third of the memory the device has free now.  The x/y stage writes it, the z stage reads it and writes the
        // output planes; nothing else looks at the x/y stage's result.
        if constexpr (sizeof(Acc) == 4) {
            const size_t mid_bytes = (size_t)plan->total * sizeof(P);
            if (!(plan->flags & RF_PLAN_INPLACE_Z) && !plan->host_only && !padded && strided_tile(plan, 2) > 0 &&
                plan->total >= ((int64_t)1 << 28)) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && mid_bytes * (size_t)plan->n_planes <= free_b / 3) {
                    for (int pl = 0; pl < plan->n_planes && status == RF_OK; pl++) plan->mid[pl] = plan->alloc(mid_bytes, false, &status);
                    if (status != RF_OK) return status;
                }
            }
        }
        int rc;
        if constexpr (sizeof(Acc) == 4)
            rc = strided_tile(plan, 2) > 0 ? add_strided_dimension<P, S>(plan, 2, /*from_input=*/false, desc, first_begin_step,
                                                                         walk_hook.get(), walk_hook ? walk_child.release() : nullptr)
                                           : add_generic_dimension<P, S>(plan, desc->tile[2], 2, /*from_input=*/false);
        else rc = add_generic_dimension<P, S>(plan, desc->tile[2], 2, /*from_input=*/false);      // (f64: no strided kernels)
        if (rc != RF_OK) return rc;
    }
    if (batch) {
        // every step above covers all planes in its one launch: it runs for plane 0 and is skipped for the others
        auto once = [](std::vector<Step> &steps, size_t first) {
            for (size_t i = first; i < steps.size(); i++) {
                auto inner = steps[i].run;
                steps[i].run = [inner](int pl) { return pl > 0 ? (int)RF_OK : inner(0); };
            }
        };
        once(plan->begin_steps, first_begin_step);
        once(plan->finish_steps, first_finish_step);
    }
    return status;
}

}  // namespace

bool fused_plan_applicable(const rf_plan *plan, const rf_filter_desc *, std::string *why) {
    auto no = [&](const char *msg) { if (why) *why = msg; return false; };
    if (plan->dtype != RF_F32 && plan->dtype != RF_I32 && plan->dtype != RF_I16 && plan->dtype != RF_F64)
        return no("pixel type must be f32, f64, i32 or i16");
    if (plan->dtype == RF_F64 && (plan->ndim < 2 || plan->pw.in_u8)) return no("f64 pixels: 2-D / 3-D images of f64 samples");
    if (plan->ndim == 1) {
        // a long 1-D signal folded into chained rows (zero border only: the clamped prologue would differ per row)
        if (plan->clamped) return no("1-D: clamped border not supported on the fused path");
        if (plan->sharded()) return no("1-D: cannot be sharded");
        if (plan->dims[0].scan_ids.empty()) return no("no scans");
        if (plan->dims[0].N < 8192 || chained_row_length(chained_padded_length(plan->dims[0].N)) == 0)
            return no("1-D: at least 8192 samples");
        if (chained_padded_length(plan->dims[0].N) != plan->dims[0].N) {
            // zero padding behind the signal: a causal scan rings on into it, and an anticausal scan AFTER a causal one
            // would pick that ringing up -- such filters (and prologues, which turn the padding into their bias) run as given
            if (plan->pw.in_u8 || plan->pw.pre) return no("1-D: a length that is not a multiple of 8192 cannot take a prologue");
            bool seen_causal = false;
            for (int id : plan->dims[0].scan_ids) {
                if (plan->scans[id].causal) seen_causal = true;
                else if (seen_causal) return no("1-D: an anticausal scan behind a causal one needs a length that is a multiple of 8192");
            }
        }
        if (plan->dims[0].k > kFusedMaxK) return no("feedback order above 3");
        if ((int)plan->dims[0].scan_ids.size() > kFusedMaxScans) return no("more than 4 scans");
        return true;
    }
    if (plan->dims[0].scan_ids.empty() && plan->dims[1].scan_ids.empty()) return no("no scans along x or y");
    if (plan->mod_form) {
        // scans in zero-border form behind border modifications (plan.cpp, "clamped sections"): the kernels position a
        // modification at compile-time indices of the entry segment / the tile's first or last rows
        if (plan->dtype != RF_F32 || plan->sharded()) return no("clamped sections: unsharded f32 images");
        if (plan->pw.post && plan->pw.post_i != 0.0) return no("clamped sections: no epilogue with an input operand");
        if (plan->dims[0].N % 16 != 0) return no("clamped sections: width must be a multiple of 16");
        if (!plan->dims[1].scan_ids.empty() && plan->dims[1].N % 32 != 0) return no("clamped sections: height must be a multiple of 32");
        if (plan->ndim > 2 && !plan->dims[2].scan_ids.empty() && strided_tile(plan, 2) == 0) return no("clamped sections: the z stage needs the strided kernels");
        for (int d = 0; d < plan->ndim; d++)
            if ((int)plan->dims[d].scan_ids.size() > kFusedMaxScans) return no("clamped sections: more than 4 scans in a dimension");
    }
    // rows of 4- and 8-byte pixels only have to be element-aligned: a width that is not a multiple of 4 ends every row in a
    // partial chunk, loaded sample by sample (scan_device.h, load_chunk_cols); 2-byte pixels and unsigned-byte inputs are
    // moved in 8- and 4-byte pieces and keep the rule
    if (plan->dims[0].N % 4 != 0 && (plan->dtype == RF_I16 || plan->pw.in_u8))
        return no("int16 pixels / uint8 inputs: width must be a multiple of 4");
    if (plan->ndim == 2 && plan->sharded() && plan->shard_common % 32 != 0)
        return no("row-sharded slabs must be whole tiles (height a multiple of 32)");
    const int K = fused_order(plan);
    if (K > kFusedMaxK) return no("feedback order above 3");
    if ((int)plan->dims[0].scan_ids.size() > kFusedMaxScans || (int)plan->dims[1].scan_ids.size() > kFusedMaxScans)
        return no("more than 4 scans along x or y");
    const int64_t NZ = plan->ndim > 2 ? plan->dims[2].N : 1;
    if (NZ > 65535 || (plan->dims[1].N + 31) / 32 > 65535) return no("grid too large");
    if (plan->ndim > 2 && !plan->dims[2].scan_ids.empty()) {
        // the z stage runs on the generic dimension builder
        if (strided_tile(plan, 2) == 0 && pick_generic_tile(plan->dims[2].N, plan->dims[2].k, 0) == 0)
            return no("no tile divides the z extent");
    }
    return true;
}

int build_fused_plan(rf_plan *plan, const rf_filter_desc *desc) {
    if (plan->dtype == RF_F32) return build_fused<float, double>(plan, desc);
    if (plan->dtype == RF_I32) return build_fused<int32_t, uint64_t>(plan, desc);
    if (plan->dtype == RF_I16) return build_fused<int16_t, uint64_t>(plan, desc);
    if (plan->dtype == RF_F64) return build_fused<double, double>(plan, desc);
    set_error("fused path: unsupported pixel type");
    return RF_ERR_UNSUPPORTED;
}

}  // namespace rf
