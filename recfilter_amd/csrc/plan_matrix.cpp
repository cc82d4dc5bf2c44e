// plan_matrix.cpp -- RF_PATH_TILED_MATRIX: the host side of kernels_matrix.hip.
//
// One stage per scan, in the plan's scan order (grouped by dimension; scans of different dimensions commute and the scans
// of one dimension are successive in-place passes, lib/recfilter.cpp:302-343): pass 1 (tail extraction), the carry chain
// with its levels, pass 2.  Every matrix the kernels multiply with is built here by running the scan recurrence (scan_tile,
// tables.h) on unit vectors in double and rounding once to f32:
//   G  (32 x 32)   column j = the zero-border scan of a 32-sample sub-block whose only non-zero sample is j
//   R  (32 x 32)   column i = what a unit OUTPUT at row i of the previous sub-block (in scan direction) adds to this one
//   dG (32)        clamped border: what the prologue of lib/recfilter.cpp:330-336 adds to the first sub-block per unit x_0
//   H  (k x T)     the tile-local tail per unit input sample (extract_tails_from_each_scan, lib/split.cpp:256-499); dH as dG
//   A  (k x k)     carry -> next carry across one tile (matrix_R's tail rows, lib/coefficients.cpp:51-83, lib/split.cpp:770)
// and the chain levels' A^(16^l) and their powers 1..16 (tabulated in double: no chain of f32 products).
#include <algorithm>
#include <cstring>

#include "kernels_matrix.h"
#include "plan.h"
#include "plan_generic.h"

namespace rf {

bool matrix_plan_applicable(const rf_plan *plan, const rf_filter_desc *desc, std::string *why) {
    auto no = [&](const char *msg) { if (why) *why = msg; return false; };
    (void)desc;
    if (plan->dtype != RF_F32) return no("the matrix cores take f32 pixels (integer and f64 filters run on the generic path)");
    if (plan->scans.empty()) return no("no scans");
    if (plan->sharded()) {
        // slabs of the outermost dimension (one carry exchange per scan along it): equal slabs of whole tiles
        if (plan->ndim < 2) return no("a 1-D signal is not sharded");
        for (int64_t e : plan->shard_extents)
            if (e != plan->dims[plan->ndim - 1].N) return no("the matrix path shards into slabs of equal extent");
        if (plan->dims[plan->ndim - 1].N % kMxSB != 0) return no("a slab of the sharded dimension is not a multiple of 32 samples");
        if (plan->pw.pre || plan->pw.post) return no("pointwise stages of a sharded plan run on the fused path");
    }
    for (int d = 0; d < plan->ndim; d++) {
        const DimInfo &di = plan->dims[d];
        if (di.scan_ids.empty()) continue;
        // (the tiles need not divide the extent -- kernels_matrix.hip, mx_element -- but the passes move 16 bytes per lane)
        if (d == 0 && di.N % 4 != 0) return no("the width is not a multiple of 4 samples");
        if (d > 0 && di.stride % 4 != 0) return no("rows are not 16-byte aligned");
        if (d > 0 && di.lines / di.stride > 65535) return no("too many planes");
        // (lane = line / column: the tile index is a grid dimension; 1-D signals -- lane = tile -- have no such bound)
        const bool tile_in_grid = d > 0 || di.lines >= 32;
        if (tile_in_grid && di.N / kMxSB > 65535) return no("extent too large");
    }
    return true;
}

namespace {

inline int mx_row(int t, int h) { return ((t >> 2) << 3) + (h << 2) + (t & 3); }

// A-operand fragments of a 32 x 32 matrix (row-major, out row x K index): frag[t][lane] = Mat[lane & 31][row(t, lane >> 5)]
void pack_fragments(const double *mat, std::vector<float> &out) {
    for (int t = 0; t < 16; t++)
        for (int lane = 0; lane < 64; lane++) out.push_back((float)mat[(lane & 31) * 32 + mx_row(t, lane >> 5)]);
}

std::vector<double> pad32(const std::vector<double> &m, int k) {
    std::vector<double> p(32 * 32, 0.0);
    for (int r = 0; r < k; r++)
        for (int j = 0; j < k; j++) p[r * 32 + j] = m[r * k + j];
    return p;
}

struct StageTables {
    std::vector<double> G, R, dG, H, dH, A;      // dense, as documented above (H: 32 x T, rows >= k zero; A: k x k)
};

StageTables build_stage_tables(const Scan &scan, int T, bool clamped) {
    const int k = scan.order;
    ScanS<double> ts = make_table_scan<double>(scan);
    ts.mod_n = -1;
    StageTables t;
    t.G.assign(32 * 32, 0.0); t.R.assign(32 * 32, 0.0); t.dG.assign(32, 0.0);
    t.H.assign((size_t)32 * T, 0.0); t.dH.assign(32, 0.0); t.A.assign((size_t)k * k, 0.0);
    std::vector<double> v(std::max(T, 32));
    for (int j = 0; j < 32; j++) {
        std::fill(v.begin(), v.end(), 0.0);
        v[j] = 1.0;
        scan_tile<double>(v.data(), 32, k, ts, false, nullptr);
        for (int i = 0; i < 32; i++) t.G[i * 32 + j] = v[i];
    }
    for (int i = 0; i < 32; i++) {
        const int j = ts.causal ? 31 - i : i;          // row i of the previous sub-block is y[-1-j] of this one
        if (j >= k) continue;
        double carry[RF_MAX_ORDER] = {0};
        carry[j] = 1.0;
        std::fill(v.begin(), v.end(), 0.0);
        scan_tile<double>(v.data(), 32, k, ts, false, carry);
        for (int m = 0; m < 32; m++) t.R[m * 32 + i] = v[m];
    }
    auto tail_pos = [&](int r) { return ts.causal ? T - 1 - r : r; };
    for (int m = 0; m < T; m++) {
        std::fill(v.begin(), v.end(), 0.0);
        v[m] = 1.0;
        scan_tile<double>(v.data(), T, k, ts, false, nullptr);
        for (int r = 0; r < k; r++) t.H[(size_t)r * T + m] = v[tail_pos(r)];
    }
    for (int j = 0; j < k; j++) {
        double carry[RF_MAX_ORDER] = {0};
        carry[j] = 1.0;
        std::fill(v.begin(), v.end(), 0.0);
        scan_tile<double>(v.data(), T, k, ts, false, carry);
        for (int r = 0; r < k; r++) t.A[(size_t)r * k + j] = v[tail_pos(r)];
    }
    if (clamped) {
        // the clamped prologue reads x_0 and y_0 = (b + sum a) x_0 where the zero border reads zeros: the difference of the
        // two scans is linear in the first sample alone
        std::vector<double> vc(std::max(T, 32), 0.0);
        const int m32 = ts.causal ? 0 : 31, mT = ts.causal ? 0 : T - 1;
        vc[m32] = 1.0;
        scan_tile<double>(vc.data(), 32, k, ts, true, nullptr);
        for (int i = 0; i < 32; i++) t.dG[i] = vc[i] - t.G[i * 32 + m32];
        std::fill(vc.begin(), vc.end(), 0.0);
        vc[mT] = 1.0;
        scan_tile<double>(vc.data(), T, k, ts, true, nullptr);
        for (int r = 0; r < k; r++) t.dH[r] = vc[tail_pos(r)] - t.H[(size_t)r * T + mT];
    }
    return t;
}

}  // namespace

int build_matrix_plan(rf_plan *plan, const rf_filter_desc *desc) {
    (void)desc;
    int status = RF_OK;
    plan->vector_access = true;
    const int np = plan->n_planes;

    struct Level {
        int64_t M = 0;           // elements per line
        bool top = false;
        bool zero = false;       // the level's transfer matrix is all zeros once rounded to f32: its chain is the identity
        const float *A = nullptr, *P = nullptr;
    };
    struct Stage {
        MxPassArgs pass{};
        std::vector<Level> levels;
        int scan = 0;
        bool sharded_dim = false;           // a scan along the sharded dimension: an exchange of its exit carries follows its chain
        const float *AM = nullptr, *PM = nullptr;      // fragments of A^M (slab transfer) and of A^1 .. A^M
    };
    std::vector<Stage> stages;
    size_t tails_floats = 0;                 // per plane, max over the stages (they run one after the other)
    std::vector<size_t> level_floats;        // [l-1] per plane, max over the stages

    // the last x pass of a 2-D image can hand the first y scan its tile-local tails (MxPassArgs::next == 2, below)
    bool xy_handover = false;
    if (plan->ndim == 2 && !plan->sharded() && !plan->dims[0].scan_ids.empty() && !plan->dims[1].scan_ids.empty() && RF_KNOB("RF_MX_NO_NEXT2") == nullptr) {
        const DimInfo &dx = plan->dims[0], &dy = plan->dims[1];
        const int ky = plan->scans[(size_t)dy.scan_ids.front()].order;
        xy_handover = dx.lines >= 32 && dx.N % kMxUnits == 0 && dy.N % kMxUnits == 0 && ky <= 16;
    }
    for (int d = 0; d < plan->ndim; d++) {
        DimInfo &di = plan->dims[d];
        if (di.scan_ids.empty()) continue;
        // tile width: the widest of 256 / 224 / .. / 32 that divides the extent (wide tiles halve the tails and their chain; a tile of
        // 128 at most while the launch would otherwise not fill the CUs, and for the y scans of a 2-D image that take their
        // tile-local tails from the last x pass -- its 128-line workgroups are their tiles -- when those tails are short enough
        // for that to pay: 16384^2, orders 12 / 32, round 5); an extent that is no multiple of 32 takes tiles of that width
        // (one tile of 32 .. when it is shorter) and pads the last one where the scan leaves the image (MxPassArgs::off)
        int nb_cap = kMxMaxNB;
        const int64_t wide_groups = (di.lines * ((di.N + kMxSB * kMxMaxNB - 1) / (kMxSB * kMxMaxNB)) + kMxUnits - 1) / kMxUnits;
        if (wide_groups < 256) nb_cap = 4;
        if (xy_handover && d == 1) nb_cap = 4;
        if (RF_KNOB("RF_MX_NB")) nb_cap = atoi(RF_KNOB("RF_MX_NB"));       // A/B: narrower / wider tiles
        int NB = nb_cap;
        if (di.N % kMxSB == 0) {
            const int64_t blocks = di.N / kMxSB;
            for (int nb = nb_cap; nb >= 1; nb--)
                if (blocks % nb == 0) { NB = nb; break; }
        } else if (di.N < kMxSB * nb_cap) {
            NB = (int)((di.N + kMxSB - 1) / kMxSB);
        }
        const int T = kMxSB * NB;
        di.T = T;
        di.M = (di.N + T - 1) / T;
        const int mode = d > 0 ? MX_Y : (di.lines >= 32 ? MX_XL : MX_X1);
        for (int id : di.scan_ids) {
            const Scan &scan = plan->scans[(size_t)id];
            const int k = scan.order;
            StageTables tb = build_stage_tables(scan, T, plan->clamped);
            const std::string tag = std::to_string(id);
            plan->tables["mx_G_" + tag] = tb.G;
            plan->tables["mx_R_" + tag] = tb.R;
            plan->tables["mx_dG_" + tag] = tb.dG;
            plan->tables["mx_H_" + tag] = tb.H;
            plan->tables["mx_dH_" + tag] = tb.dH;
            plan->tables["mx_A_" + tag] = tb.A;

            Stage st;
            st.scan = id;
            MxPassArgs &pa = st.pass;
            pa.mode = mode; pa.T = T; pa.NB = NB; pa.M = (int32_t)di.M; pa.k = k;
            pa.causal = scan.causal ? 1 : 0;
            pa.clamped = plan->clamped ? 1 : 0;
            pa.N = di.N; pa.inner = di.stride; pa.lines = di.lines; pa.units = di.lines * di.M;
            pa.off = scan.causal ? 0 : di.M * T - di.N;
            pa.ragged = di.M * T != di.N ? 1 : 0;
            const bool sharded_dim = plan->sharded() && d == plan->ndim - 1;
            pa.slab_first = (!sharded_dim || plan->shard_rank == 0) ? 1 : 0;
            pa.slab_last = (!sharded_dim || plan->shard_rank == plan->shard_world - 1) ? 1 : 0;
            pa.incoming = nullptr;
            st.sharded_dim = sharded_dim;
            if (sharded_dim) {
                // what carries a slab's entering state to its exit, and to each of its tiles: A^M, A^1 .. A^M (in double, rounded once)
                std::vector<float> fAM, fPM;
                std::vector<double> pw = tb.A;
                for (int64_t i = 0; i < di.M; i++) {
                    pack_fragments(pad32(pw, k).data(), fPM);
                    if (i + 1 < di.M) pw = mat_mul<double>(pw, tb.A, k);
                }
                pack_fragments(pad32(pw, k).data(), fAM);
                st.AM = (const float *)plan->upload(fAM.data(), fAM.size() * sizeof(float), &status);
                st.PM = (const float *)plan->upload(fPM.data(), fPM.size() * sizeof(float), &status);
            }
            std::vector<float> fG, fR, fH, fdG(32), fdH(32);
            pack_fragments(tb.G.data(), fG);
            pack_fragments(tb.R.data(), fR);
            for (int b = 0; b < NB; b++) {
                std::vector<double> Hb(32 * 32, 0.0);
                for (int r = 0; r < 32; r++)
                    for (int i = 0; i < 32; i++) Hb[r * 32 + i] = tb.H[(size_t)r * T + 32 * b + i];
                pack_fragments(Hb.data(), fH);
            }
            for (int i = 0; i < 32; i++) { fdG[i] = (float)tb.dG[i]; fdH[i] = (float)tb.dH[i]; }
            pa.G = (const float *)plan->upload(fG.data(), fG.size() * sizeof(float), &status);
            pa.R = (const float *)plan->upload(fR.data(), fR.size() * sizeof(float), &status);
            pa.H = (const float *)plan->upload(fH.data(), fH.size() * sizeof(float), &status);
            pa.dG = (const float *)plan->upload(fdG.data(), fdG.size() * sizeof(float), &status);
            pa.dH = (const float *)plan->upload(fdH.data(), fdH.size() * sizeof(float), &status);
            const size_t KP = 8 * (size_t)((k + 7) / 8);       // rows of a stored k-vector
            tails_floats = std::max(tails_floats, KP * (size_t)pa.units);

            // chain levels: the sequence of a level is the chunk exits of the level below, its transfer matrix that level's
            // to the power of the chunk length
            std::vector<double> B = tb.A;
            std::vector<double> levels_info;
            for (int64_t Ml = di.M;; Ml = (Ml + kMxChunk - 1) / kMxChunk) {
                Level lv;
                lv.M = Ml;
                lv.top = Ml <= kMxTopMax;
                std::vector<float> fA;
                pack_fragments(pad32(B, k).data(), fA);
                lv.zero = std::all_of(fA.begin(), fA.end(), [](float v) { return v == 0.0f; });
                lv.A = (const float *)plan->upload(fA.data(), fA.size() * sizeof(float), &status);
                levels_info.push_back((double)Ml);
                levels_info.push_back(lv.zero ? 1.0 : 0.0);
                if (!lv.top) {
                    std::vector<float> fP;
                    std::vector<double> pw = B;
                    for (int j = 0; j < kMxChunk; j++) {
                        pack_fragments(pad32(pw, k).data(), fP);
                        if (j + 1 < kMxChunk) pw = mat_mul<double>(pw, B, k);
                    }
                    lv.P = (const float *)plan->upload(fP.data(), fP.size() * sizeof(float), &status);
                    B = pw;                                  // B^16: the transfer matrix of the level above
                }
                const size_t l = st.levels.size();
                if (l >= 1) {
                    if (level_floats.size() < l) level_floats.resize(l, 0);
                    level_floats[l - 1] = std::max(level_floats[l - 1], KP * (size_t)di.lines * (size_t)Ml);
                }
                st.levels.push_back(lv);
                if (lv.top) break;
            }
            plan->tables["mx_levels_" + tag] = levels_info;
            plan->tables["mx_geom_" + tag] = {(double)T, (double)di.M, (double)pa.off};
            stages.push_back(st);
        }
    }
    if (status != RF_OK) return status;

    // Which stages hand the next one its tile-local tails (MxPassArgs::next): consecutive scans of one dimension always; the
    // last x scan to the first y scan of a 2-D image whose 128 x 128 blocks coincide.  The receiving stage has no pass 1.
    for (size_t i = 0; i + 1 < stages.size(); i++) {
        MxPassArgs &a = stages[i].pass;
        const MxPassArgs &b = stages[i + 1].pass;
        const int da = plan->scans[(size_t)stages[i].scan].dim, db = plan->scans[(size_t)stages[i + 1].scan].dim;
        int next = 0;
        if (da == db && a.off == b.off) next = 1;            // (the same tiles: a causal / anticausal pair only when they divide the extent)
        else if (xy_handover && da == 0 && db == 1 && a.mode == MX_XL && !a.ragged && b.T == kMxUnits) next = 2;
        // (not in a sharded plan: a slab's y tiles have their own border rules, MxPassArgs::slab_first / slab_last)
        if (RF_KNOB("RF_MX_NO_NEXT") != nullptr) next = 0;      // A/B: every stage with its own pass 1
        a.next = next;
        if (next) { a.next_k = b.k; a.next_causal = b.causal; a.next_NB = b.NB; a.next_H = b.H; a.next_dH = b.dH; }
    }
    // (two tail buffers: a final pass reads its own stage's completed tails while it writes the next stage's local ones)
    float *tails_ab[2];
    tails_ab[0] = (float *)plan->alloc(tails_floats * np * sizeof(float), false, &status);
    tails_ab[1] = stages.size() > 1 ? (float *)plan->alloc(tails_floats * np * sizeof(float), false, &status) : tails_ab[0];
    std::vector<float *> level_buf;
    for (size_t l = 0; l < level_floats.size(); l++) level_buf.push_back((float *)plan->alloc(level_floats[l] * np * sizeof(float), false, &status));
    if (status != RF_OK) return status;

    // Steps accumulate in `pending`; a scan along the sharded dimension cuts the list at its exchange: what came before goes to
    // the begin phase (first exchange) or in front of the next exchange's local step, what follows the last exchange is the finish.
    std::vector<Step> pending;
    size_t max_lines_kp = 0;
    for (const Stage &st : stages)
        if (st.sharded_dim) max_lines_kp = std::max(max_lines_kp, (size_t)st.pass.lines * 8 * (size_t)((st.pass.k + 7) / 8));
    const int world = plan->shard_world;
    float *incoming_buf = nullptr, *rank_scratch = nullptr;
    if (plan->sharded()) {
        incoming_buf = (float *)plan->alloc(max_lines_kp * np * sizeof(float), true, &status);
        rank_scratch = (float *)plan->alloc(max_lines_kp * np * (size_t)world * sizeof(float), true, &status);
        if (status != RF_OK) return status;
    }
    bool first_stage = true;
    for (size_t si = 0; si < stages.size(); si++) {
        const Stage &st = stages[si];
        float *tails = tails_ab[si & 1], *tails_next = tails_ab[(si + 1) & 1];
        const bool has_pass1 = si == 0 || stages[si - 1].pass.next == 0;
        const Scan &scan = plan->scans[(size_t)st.scan];
        const std::string nm = std::string(1, "xyz"[scan.dim]) + (scan.causal ? "+" : "-") + std::to_string(st.scan);
        const bool from_input = first_stage;
        first_stage = false;
        MxPassArgs base = st.pass;
        const size_t tails_pp = tails_floats;
        const size_t lines_kp = (size_t)base.lines * 8 * (size_t)((base.k + 7) / 8);       // one k-vector per line
        const bool sharded_dim = st.sharded_dim;
        auto pass_args = [base, tails, tails_next, tails_pp, sharded_dim, incoming_buf, lines_kp](int pl) {
            MxPassArgs a = base;
            a.tails = tails + (size_t)pl * tails_pp;
            a.next_tails = tails_next + (size_t)pl * tails_pp;
            if (sharded_dim) a.incoming = incoming_buf + (size_t)pl * lines_kp;
            return a;
        };
        if (has_pass1) {
            Step p1;
            p1.name = "mx_pass1_" + nm;
            p1.run = [plan, pass_args, from_input](int pl) {
                const float *src = from_input ? (const float *)plan->in[pl] : (const float *)plan->out[pl];
                return launch_mx_pass1(src, pass_args(pl), plan->stream);
            };
            pending.push_back(p1);
        }

        // the chain: up the levels, then the propagation down
        const int k = base.k, nlev = (int)st.levels.size();
        const bool x1 = base.mode == MX_X1, causal = base.causal != 0;
        const int64_t lines = base.lines;
        std::vector<MxChainArgs> chain((size_t)nlev);
        for (int l = 0; l < nlev; l++) {
            const Level &lv = st.levels[(size_t)l];
            MxChainArgs c{};
            c.A = lv.A; c.P = lv.P; c.k = k;
            c.Mtot = lv.M;
            c.C = lv.top ? (int32_t)lv.M : kMxChunk;
            const int64_t nch = lv.top ? 1 : (lv.M + kMxChunk - 1) / kMxChunk;
            c.ncols = lines * nch;
            const bool reversed = l == 0 && !causal;      // level 0 lives in memory order, the levels above in scan order
            if (x1) {
                c.chunk_is_lo = 1; c.cdiv = nch;
                c.s_hi = lv.M; c.s_lo = reversed ? -(int64_t)c.C : c.C; c.s_j = reversed ? -1 : 1;
                c.base = reversed ? lv.M - 1 : 0;
                c.e_hi = nch; c.e_lo = 1;
            } else {
                c.chunk_is_lo = 0; c.cdiv = lines;
                c.s_hi = (reversed ? -(int64_t)c.C : (int64_t)c.C) * lines; c.s_lo = 1; c.s_j = reversed ? -lines : lines;
                c.base = reversed ? (lv.M - 1) * lines : 0;
                c.e_hi = lines; c.e_lo = 1;
            }
            chain[(size_t)l] = c;
        }
        std::vector<size_t> lf = level_floats;
        auto chain_args = [chain, tails, tails_pp, level_buf, lf](int l, int pl) {
            MxChainArgs c = chain[(size_t)l];
            c.seq = l == 0 ? tails + (size_t)pl * tails_pp : level_buf[(size_t)l - 1] + (size_t)pl * lf[(size_t)l - 1];
            c.exits = (size_t)l + 1 < chain.size() ? level_buf[(size_t)l] + (size_t)pl * lf[(size_t)l] : nullptr;
            return c;
        };
        // A transfer matrix that is all zeros in f32 -- a filter that has decayed below the smallest float across a tile (the
        // audio app's 0.01 taps after 128 samples), or across 16, 256 ... tiles on the levels above -- makes that level's chain
        // the identity and everything above it a sum of zeros: those launches are left out (bit-identical: they would add +0).
        int live = 0;                         // levels whose chain runs
        while (live < nlev && !st.levels[(size_t)live].zero) live++;
        for (int l = 0; l < live; l++) {
            Step cs;
            cs.name = "mx_chain" + std::to_string(l) + "_" + nm;
            cs.run = [plan, chain_args, l](int pl) { return launch_mx_chain(chain_args(l, pl), plan->stream); };
            pending.push_back(cs);
        }
        for (int l = std::min(live - 1, nlev - 2); l >= 0; l--) {
            Step as;
            as.name = "mx_apply" + std::to_string(l) + "_" + nm;
            as.run = [plan, chain_args, l](int pl) { return launch_mx_apply(chain_args(l, pl), plan->stream); };
            pending.push_back(as);
        }

        if (sharded_dim) {
            // ---- the exchange of this scan: the slab's exit carry (its completed tail of the last tile in scan direction, with
            // zero entering state) -> all-gather -> every rank chains the exits of the slabs before it with A^M (the same chain
            // kernel, the ranks as its steps) -> the carry entering this slab, propagated through its tiles with A^1 .. A^M;
            // the final pass takes it in the first tile's y_(-1).
            const int64_t M = base.M;
            const int rank = plan->shard_rank;
            const int ex_index = (int)plan->exchanges.size();
            rf_plan::Exchange ex;
            ex.bytes = lines_kp * np * sizeof(float);
            ex.scratch = plan->alloc(ex.bytes, true, &status);
            if (status != RF_OK) return status;
            ex.send = ex.scratch;
            const float *AM = st.AM;
            const int kk = base.k;
            ex.form_incoming = [plan, rank_scratch, incoming_buf, lines_kp, np, world, rank, causal, AM, kk, lines](const void *gathered) -> int {
                // (the gathered buffer is the caller's: the chain runs on a copy; rank-major, [rank][plane][line][KP])
                RF_HIP_CHECK(hipMemcpyAsync(rank_scratch, gathered, lines_kp * np * (size_t)world * sizeof(float), hipMemcpyDeviceToDevice, plan->stream));
                const int j = causal ? rank : world - 1 - rank;                  // this slab's place in scan order
                for (int pl = 0; pl < np; pl++) {
                    if (world > 1) {
                        MxChainArgs c{};
                        c.seq = rank_scratch; c.exits = nullptr; c.A = AM; c.P = nullptr; c.k = kk; c.C = world; c.Mtot = world;
                        c.chunk_is_lo = 0; c.enter_fixed = 0; c.ncols = lines; c.cdiv = lines;
                        c.s_lo = 1; c.s_hi = 0; c.s_j = (causal ? 1 : -1) * (int64_t)np * lines;
                        c.base = (int64_t)pl * lines + (causal ? 0 : (int64_t)(world - 1) * np * lines);
                        c.e_hi = 0; c.e_lo = 0;
                        if (int rc = launch_mx_chain(c, plan->stream)) return rc;
                    }
                    float *inc = incoming_buf + (size_t)pl * lines_kp;
                    if (j == 0) RF_HIP_CHECK(hipMemsetAsync(inc, 0, lines_kp * sizeof(float), plan->stream));
                    else {
                        const int h = causal ? rank - 1 : rank + 1;          // the slab before this one in scan direction
                        RF_HIP_CHECK(hipMemcpyAsync(inc, rank_scratch + ((size_t)h * np + pl) * lines_kp, lines_kp * sizeof(float),
                                                    hipMemcpyDeviceToDevice, plan->stream));
                    }
                }
                return (int)RF_OK;
            };
            plan->exchanges.push_back(ex);
            Step xs;
            xs.name = "mx_exits_" + nm;
            xs.run = [plan, tails, tails_pp, lines_kp, M, causal, ex_index, lines](int pl) -> int {
                float *send = (float *)plan->exchanges[(size_t)ex_index].send;
                const float *exit_tail = tails + (size_t)pl * tails_pp + (size_t)(causal ? M - 1 : 0) * lines_kp;      // [tile][line][KP]
                (void)lines;
                RF_HIP_CHECK(hipMemcpyAsync(send + (size_t)pl * lines_kp, exit_tail, lines_kp * sizeof(float), hipMemcpyDeviceToDevice, plan->stream));
                return (int)RF_OK;
            };
            if (ex_index == 0) {
                for (Step &p : pending) plan->begin_steps.push_back(p);
                plan->exchange_local_steps.push_back({xs});
            } else {
                pending.push_back(xs);
                plan->exchange_local_steps.push_back(pending);
            }
            pending.clear();
            // propagation of the entering carry through the slab's tiles: tail(t) += A^(i+1) incoming
            MxChainArgs ap = chain[0];
            ap.P = st.PM; ap.C = (int32_t)M; ap.Mtot = M; ap.ncols = lines; ap.cdiv = lines; ap.chunk_is_lo = 0; ap.enter_fixed = 1;
            ap.s_hi = 0; ap.s_lo = 1; ap.s_j = causal ? lines : -lines; ap.base = causal ? 0 : (M - 1) * lines;
            ap.e_hi = 0; ap.e_lo = 1;
            Step as;
            as.name = "mx_incoming_" + nm;
            as.run = [plan, ap, tails, tails_pp, incoming_buf, lines_kp](int pl) {
                MxChainArgs c = ap;
                c.seq = tails + (size_t)pl * tails_pp;
                c.exits = incoming_buf + (size_t)pl * lines_kp;
                return launch_mx_apply(c, plan->stream);
            };
            plan->exchange_apply_steps.push_back({as});
        }

        Step p2;
        p2.name = "mx_pass2_" + nm;
        p2.run = [plan, pass_args, from_input](int pl) {
            const float *src = from_input ? (const float *)plan->in[pl] : (const float *)plan->out[pl];
            return launch_mx_pass2(src, (float *)plan->out[pl], pass_args(pl), plan->stream);
        };
        pending.push_back(p2);
    }
    // what follows the last exchange is the finish phase (an unsharded plan: everything is its begin phase)
    if (plan->exchanges.empty()) for (Step &p : pending) plan->begin_steps.push_back(p);
    else for (Step &p : pending) plan->finish_steps.push_back(p);
    return status;
}

}  // namespace rf
