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

// What a PAIR stage (a causal scan c, then the anticausal scan n along the same dimension; MxPassArgs::pair) needs beyond the two
// scans' own tables, all by running the scans on unit vectors in double:
//   H21 (32 x T, rows >= k_n zero)  tile-local tail of n applied to the tile-local result of c:  tail_n(B_c x)
//   v21 (32)                         what the clamped border of c adds to that, per first sample of the image's first tile
//   W21 (k_n x k_c)                  what the causal carry entering a tile adds to n's tile-local tail: tail_n(Rt_c carry)
struct PairTables {
    std::vector<double> H21, v21, W21;
};

PairTables build_pair_tables(const Scan &c, const Scan &n, int T, bool clamped) {
    const int kc = c.order, kn = n.order;
    ScanS<double> tc = make_table_scan<double>(c), tn = make_table_scan<double>(n);
    tc.mod_n = -1; tn.mod_n = -1;
    PairTables t;
    t.H21.assign((size_t)32 * T, 0.0); t.v21.assign(32, 0.0); t.W21.assign((size_t)kn * kc, 0.0);
    std::vector<double> v((size_t)T);
    for (int m = 0; m < T; m++) {
        std::fill(v.begin(), v.end(), 0.0);
        v[(size_t)m] = 1.0;
        scan_tile<double>(v.data(), T, kc, tc, false, nullptr);
        scan_tile<double>(v.data(), T, kn, tn, false, nullptr);
        for (int r = 0; r < kn; r++) t.H21[(size_t)r * T + m] = v[(size_t)r];       // (anticausal: tail r is sample r)
    }
    for (int j = 0; j < kc; j++) {
        double carry[RF_MAX_ORDER] = {0};
        carry[j] = 1.0;
        std::fill(v.begin(), v.end(), 0.0);
        scan_tile<double>(v.data(), T, kc, tc, false, carry);
        scan_tile<double>(v.data(), T, kn, tn, false, nullptr);
        for (int r = 0; r < kn; r++) t.W21[(size_t)r * kc + j] = v[(size_t)r];
    }
    if (clamped) {
        std::vector<double> vz((size_t)T, 0.0);
        std::fill(v.begin(), v.end(), 0.0);
        v[0] = 1.0; vz[0] = 1.0;
        scan_tile<double>(v.data(), T, kc, tc, true, nullptr);
        scan_tile<double>(vz.data(), T, kc, tc, false, nullptr);
        for (int i = 0; i < T; i++) v[(size_t)i] -= vz[(size_t)i];
        scan_tile<double>(v.data(), T, kn, tn, false, nullptr);
        for (int r = 0; r < kn; r++) t.v21[(size_t)r] = v[(size_t)r];
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
        int pair = 0;                        // 1: the causal scan of a pair stage, 2: its anticausal scan (MxPassArgs::pair)
        const float *W21 = nullptr, *Dlast = nullptr;      // pair, first scan: fragments of W21; of dH(anticausal) (x) e_0 (clamped border)
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
    // Pairs: a causal scan directly followed by an anticausal one along the same dimension, tails of the same number of pieces and
    // short enough (ceil(k / 8) <= 2: above that the chain of the narrower tile costs what the saved pass gains), tiles that
    // divide the extent, not the sharded dimension (its scans have an exchange each).
    auto pair_at = [&](const DimInfo &di, int d, size_t i) {
        if (RF_KNOB("RF_MX_NO_PAIR") != nullptr || i + 1 >= di.scan_ids.size()) return false;
        if (plan->sharded() && d == plan->ndim - 1) return false;
        const Scan &c = plan->scans[(size_t)di.scan_ids[i]], &n = plan->scans[(size_t)di.scan_ids[i + 1]];
        return c.causal && !n.causal && (c.order + 7) / 8 == (n.order + 7) / 8 && (c.order + 7) / 8 <= 2 && di.N % kMxSB == 0;
    };
    std::vector<bool> dim_pairs((size_t)plan->ndim, false);
    std::vector<int> pair_role(plan->scans.size(), 0);        // 1: the causal scan of a pair, 2: its anticausal scan
    for (int d = 0; d < plan->ndim; d++) {
        const DimInfo &di = plan->dims[d];
        for (size_t i = 0; i + 1 < di.scan_ids.size(); i++)
            if (pair_at(di, d, i)) {
                pair_role[(size_t)di.scan_ids[i]] = 1;
                pair_role[(size_t)di.scan_ids[i + 1]] = 2;
                dim_pairs[(size_t)d] = true;
                i++;
            }
    }
    if (xy_handover && pair_role[(size_t)plan->dims[0].scan_ids.back()] != 0) xy_handover = false;      // (the last x pass is a pair's: no hand-over)
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
        if (dim_pairs[(size_t)d]) nb_cap = 4;             // (a pair keeps the tile's causal result in registers)
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
        if (NB > 4)                                        // (an A/B build asked for wider tiles: no pairs)
            for (int id : di.scan_ids) pair_role[(size_t)id] = 0;
        for (size_t idx = 0; idx < di.scan_ids.size(); idx++) {
            const int id = di.scan_ids[idx];
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
            st.pair = pair_role[(size_t)id];
            pa.pair = st.pair;
            if (st.pair == 1) {
                const int id2 = di.scan_ids[idx + 1];
                const Scan &scan2 = plan->scans[(size_t)id2];
                const PairTables pt = build_pair_tables(scan, scan2, T, plan->clamped);
                plan->tables["mx_pair_" + tag] = {(double)id2};
                plan->tables["mx_H21_" + tag] = pt.H21;
                plan->tables["mx_v21_" + tag] = pt.v21;
                plan->tables["mx_W21_" + tag] = pt.W21;
                std::vector<float> fH21, fv21(32), fW;
                // pass 1 contracts once for both tails: the causal scan's H in the rows 0 .. 15, H21 in the rows 16 .. 31
                for (int b = 0; b < NB; b++) {
                    std::vector<double> Hb(32 * 32, 0.0);
                    for (int r = 0; r < 16; r++)
                        for (int i = 0; i < 32; i++) {
                            Hb[r * 32 + i] = tb.H[(size_t)r * T + 32 * b + i];
                            Hb[(16 + r) * 32 + i] = pt.H21[(size_t)r * T + 32 * b + i];
                        }
                    pack_fragments(Hb.data(), fH21);
                }
                for (int i = 0; i < 16; i++) { fv21[i] = (float)tb.dH[(size_t)i]; fv21[16 + i] = (float)pt.v21[(size_t)i]; }
                // W21 as a 32 x 32 operand: k_n rows, k_c columns
                std::vector<double> W32(32 * 32, 0.0);
                for (int r = 0; r < scan2.order; r++)
                    for (int j = 0; j < k; j++) W32[r * 32 + j] = pt.W21[(size_t)r * k + j];
                pack_fragments(W32.data(), fW);
                pa.p_k = scan2.order;
                pa.p_H = (const float *)plan->upload(fH21.data(), fH21.size() * sizeof(float), &status);
                pa.p_dH = (const float *)plan->upload(fv21.data(), fv21.size() * sizeof(float), &status);
                st.W21 = (const float *)plan->upload(fW.data(), fW.size() * sizeof(float), &status);
                if (plan->clamped) {
                    // the anticausal scan's clamped border in the image's last tile: its tile-local tail gains dH_n times its first
                    // sample, which is the causal RESULT's last sample = row 0 of the tile's completed causal tail
                    const StageTables tb2 = build_stage_tables(scan2, T, true);
                    std::vector<double> D32(32 * 32, 0.0);
                    for (int r = 0; r < scan2.order; r++) D32[r * 32 + 0] = tb2.dH[(size_t)r];
                    std::vector<float> fD;
                    pack_fragments(D32.data(), fD);
                    st.Dlast = (const float *)plan->upload(fD.data(), fD.size() * sizeof(float), &status);
                }
            }
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
                // one chain over the whole line where the lines alone fill the chip (lane = line / column, 16384 of them = 512 waves):
                // M dependent steps, but ONE pass over the tails instead of the two of chunks + propagation (16384^2: 69-75 us
                // against 96-113 per scan; 8192 lines: level, profiles/r5/matrix_chain_ab.txt)
                const int64_t top_wide = RF_KNOB("RF_MX_TOP") ? atoi(RF_KNOB("RF_MX_TOP")) : kMxTopWide;
                lv.top = Ml <= ((mode != MX_X1 && di.lines >= 16384) ? top_wide : (int64_t)kMxTopMax);
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
        if (stages[i].pair != 0 || stages[i + 1].pair == 1) next = 0;       // (a pair stage forms both scans' tails in its own pass 1)
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
    bool first_stage = true, pair_from_input = false;
    for (size_t si = 0; si < stages.size(); si++) {
        const Stage &st = stages[si];
        float *tails = tails_ab[si & 1], *tails_next = tails_ab[(si + 1) & 1];
        const bool has_pass1 = st.pair != 2 && (si == 0 || stages[si - 1].pass.next == 0);
        const Scan &scan = plan->scans[(size_t)st.scan];
        const std::string nm = std::string(1, "xyz"[scan.dim]) + (scan.causal ? "+" : "-") + std::to_string(st.scan);
        const bool from_input = st.pair == 2 ? pair_from_input : first_stage;       // (a pair's final pass reads what its pass 1 read)
        if (st.pair == 1) pair_from_input = first_stage;
        first_stage = false;
        MxPassArgs base = st.pass;
        const size_t tails_pp = tails_floats;
        const size_t lines_kp = (size_t)base.lines * 8 * (size_t)((base.k + 7) / 8);       // one k-vector per line
        const bool sharded_dim = st.sharded_dim;
        auto pass_args = [base, tails, tails_next, tails_pp, sharded_dim, incoming_buf, lines_kp](int pl) {
            MxPassArgs a = base;
            a.tails = tails + (size_t)pl * tails_pp;
            a.next_tails = tails_next + (size_t)pl * tails_pp;
            if (a.pair == 1) a.p_tails = tails_next + (size_t)pl * tails_pp;       // (the anticausal scan's tails: the other buffer)
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
        // A pair whose anticausal chain runs over the whole line in one go takes the cross term on the way (MxChainArgs::cross): no
        // launch of its own in front of the chain.  `fold_at(si)`: si is the pair's first stage.
        auto fold_at = [&stages](size_t first) {
            const Stage &n = stages[first + 1];
            return stages[first].pair == 1 && n.levels.size() == 1 && !n.levels[0].zero && n.pass.mode != MX_X1 && RF_KNOB("RF_MX_NO_FOLD") == nullptr;
        };
        if (st.pair == 2 && fold_at(si - 1)) {
            MxChainArgs &c0 = chain[0];
            c0.crossW = stages[si - 1].W21;
            c0.crossD = stages[si - 1].Dlast;
            c0.cross_shift = -lines;                   // [tile][line][KP]: the tile in front
            c0.cross_steps = (int32_t)base.M - 1;      // (the anticausal chain starts at the last tile: step j is tile M - 1 - j, and tile 0 has none in front)
        }
        const bool fold_cross = st.pair == 2 && fold_at(si - 1);
        std::vector<size_t> lf = level_floats;
        auto chain_args = [chain, tails, tails_next, tails_pp, level_buf, lf, fold_cross](int l, int pl) {
            MxChainArgs c = chain[(size_t)l];
            c.seq = l == 0 ? tails + (size_t)pl * tails_pp : level_buf[(size_t)l - 1] + (size_t)pl * lf[(size_t)l - 1];
            c.exits = (size_t)l + 1 < chain.size() ? level_buf[(size_t)l] + (size_t)pl * lf[(size_t)l] : nullptr;
            if (fold_cross && l == 0) c.cross = tails_next + (size_t)pl * tails_pp;      // (the completed causal carries: the other buffer)
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

        if (st.pair == 1 && fold_at(si)) continue;       // (the cross term rides on the anticausal chain; the pair's final pass follows it)
        if (st.pair == 1) {
            // between the two chains: what the completed causal carry entering a tile adds to the anticausal scan's tile-local tail
            // (one product per tile: the propagation kernel with chunks of one tile), and the anticausal scan's clamped border
            MxChainArgs cr{};
            cr.P = st.W21; cr.k = std::max(base.k, base.p_k); cr.C = 1; cr.Mtot = base.M; cr.enter_fixed = 0;
            cr.ncols = lines * base.M;
            if (x1) { cr.chunk_is_lo = 1; cr.cdiv = base.M; cr.s_hi = base.M; cr.s_lo = 1; cr.s_j = 1; cr.base = 0; cr.e_hi = base.M; cr.e_lo = 1; }
            else { cr.chunk_is_lo = 0; cr.cdiv = lines; cr.s_hi = lines; cr.s_lo = 1; cr.s_j = lines; cr.base = 0; cr.e_hi = lines; cr.e_lo = 1; }
            Step xs;
            xs.name = "mx_cross_" + nm;
            xs.run = [plan, cr, tails, tails_next, tails_pp](int pl) {
                MxChainArgs c = cr;
                c.seq = tails_next + (size_t)pl * tails_pp;
                c.exits = tails + (size_t)pl * tails_pp;
                return launch_mx_apply(c, plan->stream);
            };
            pending.push_back(xs);
            if (st.Dlast != nullptr) {
                MxChainArgs bd{};
                bd.P = st.Dlast; bd.k = cr.k; bd.C = 1; bd.Mtot = 1; bd.enter_fixed = 1; bd.ncols = lines; bd.cdiv = lines; bd.chunk_is_lo = 0;
                bd.s_hi = 0; bd.s_j = 0; bd.e_hi = 0;
                const int64_t KPf = 8 * (int64_t)((cr.k + 7) / 8);
                int64_t exit_off;
                if (x1) { bd.base = base.M - 1; bd.s_lo = base.M; bd.e_lo = base.M; exit_off = (base.M - 1) * KPf; }
                else { bd.base = (base.M - 1) * lines; bd.s_lo = 1; bd.e_lo = 1; exit_off = (base.M - 1) * lines * KPf; }
                Step bs;
                bs.name = "mx_cross_border_" + nm;
                bs.run = [plan, bd, tails, tails_next, tails_pp, exit_off](int pl) {
                    MxChainArgs c = bd;
                    c.seq = tails_next + (size_t)pl * tails_pp;
                    c.exits = tails + (size_t)pl * tails_pp + exit_off;
                    return launch_mx_apply(c, plan->stream);
                };
                pending.push_back(bs);
            }
            continue;                                     // (the pair's final pass follows the anticausal scan's chain)
        }
        Step p2;
        if (st.pair == 2) {
            // the pair's final pass: this stage's block describes the anticausal scan, the stage before it the causal one
            const MxPassArgs first = stages[si - 1].pass;
            p2.name = "mx_pass2_pair_" + nm;
            p2.run = [plan, first, base, tails, tails_next, tails_pp, from_input](int pl) {
                MxPassArgs a = first;                      // (the causal scan's geometry, G, R, dG; its completed tails: the other buffer)
                a.tails = tails_next + (size_t)pl * tails_pp;
                a.p_G = base.G; a.p_R = base.R; a.p_dG = base.dG; a.p_k = base.k;
                a.p_tails = tails + (size_t)pl * tails_pp;
                a.next = 0;
                const float *src = from_input ? (const float *)plan->in[pl] : (const float *)plan->out[pl];
                return launch_mx_pass2_pair(src, (float *)plan->out[pl], a, plan->stream);
            };
            pending.push_back(p2);
            continue;
        }
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
