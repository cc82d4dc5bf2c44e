// kernels_carry.hip -- the cross-tile carry recurrence as a blocked parallel scan.
//
// The reference runs this stage (create_complete_tail_term, lib/split.cpp:743-867) as one
// unrolled per-thread loop over all tiles of a line (gpu_auto_inter_schedule,
// lib/recfilter.cpp:763-785): a chain of M dependent steps, each waiting on a global load.
// On MI355X that is pure latency (16k lines = one wave per CU).  Here a workgroup of 16 waves owns
// 64 lines (lane = line) and cuts each line's M tiles into 16 chunks (wave = chunk, so everything
// that depends on the tile index is wave-uniform and lives in scalar registers):
//
//   A  every thread (line, chunk) loads its <=16 tile tails at once (independent loads), adds
//      the same-dimension chaining terms (create_tail_residual_term, lib/split.cpp:912-1004),
//      then runs the recurrence inside its chunk with a zero incoming state
//   B  chunk exit states go through LDS; every thread forms the state entering its chunk with
//      the precomputed chunk transfer matrix A^C (<= 15 k x k steps, no memory traffic)
//   C  the entering state is propagated through the chunk and the completed tails are stored
//
// All scans of the dimension run inside ONE launch (a workgroup owns its lines completely, so a
// barrier orders scan s+1's chaining loads after scan s's stores).  Results are identical to the
// serial recurrence up to f32 rounding; integer pixel types stay bit-exact (ring arithmetic).
#include <cstdlib>

#include "kernels.h"
#include "kernels_fused.h"

namespace rf {

namespace {

constexpr int kCarryLines = 64;    // lines per workgroup = one wave: 256-byte coalesced tail accesses
constexpr int kCarryChunks = 16;   // chunks per line per block of tiles = waves per workgroup
constexpr int kCarryMaxC = 16;     // tiles per chunk (orders <= 3; higher orders keep 4 tiles of k-vectors in registers)
constexpr int kCarryMaxCHigh = 4;
constexpr int kCarryPair3MaxC = 8;  // order 3, two scans: more than 8 tiles per thread take the pair kernel ...
constexpr int kCarryPair3Chunks = 8; // ... as 8 chunk-waves of up to 16 tiles: 158 registers and no scratch, where 16 waves of 8 tiles had 128
                                     // registers and 168 bytes of scratch that its loads waited for (cfg4b y carries: 122 -> 90 us)
constexpr int kCarryChunksHigh = 8; // and at most 8 chunk-waves, which bounds the LDS combine buffer (k = 8 in f64: 32 KiB)

template <typename Acc, int K>
__device__ __forceinline__ void matvec_acc(const Acc *__restrict__ m, const Acc (&x)[K], Acc (&y)[K]) {
#pragma unroll
    for (int r = 0; r < K; r++)
#pragma unroll
        for (int j = 0; j < K; j++) y[r] = y[r] + m[r * K + j] * x[j];
}

struct CarryGeom {
    uint32_t tile_major;     // tails are [tile][line / 256][scan][r][256] instead of [scan][tile][r][line]
    uint32_t lines;          // number of lines (all tails offsets fit 32 bits, checked on the host)
    int32_t M;               // tiles per line
    int32_t n_scans;
    int32_t first_is_border, last_is_border;
    uint32_t causal_mask;    // bit s = scan s is causal
    const void *part2;       // second part of the tile-local tails (GenericDimArgs::tails_part2), or null
};

// element offset of tail (scan s, tile tt, component r) of `line`
template <int K>
__device__ __forceinline__ uint32_t tail_off(const CarryGeom &g, int s, int tt, int r, uint32_t line) {
    if (g.tile_major)
        return ((((uint32_t)tt * (g.lines >> 8) + (line >> 8)) * (uint32_t)g.n_scans + (uint32_t)s) * K + (uint32_t)r) * 256u + (line & 255u);
    return (((uint32_t)s * (uint32_t)g.M + (uint32_t)tt) * K + (uint32_t)r) * g.lines + line;
}

// PRE (long 1-D signals folded into rows, plan_fused.cpp "chained rows"): before its own scan s_begin the launch finishes
// scan s_begin - 1 -- the chain over the rows' exit states and its propagation through the rows' tails, what
// chain_apply_kernel below does as a launch of its own: every workgroup walks the exit states (wave 0), keeps the states
// entering the rows in LDS, and updates the tails of ITS lines.  n scans then take n + 1 launches instead of 2 n.
template <typename Acc, int K, int MAXC, int NCH, bool PRE = false>
__global__ void __launch_bounds__(kCarryLines * NCH)
carry_block_kernel(CarryGeom g, int s_begin, int s_end, Acc *__restrict__ tails, const Acc *__restrict__ incoming,
                   const Acc *__restrict__ Wtab, const Acc *__restrict__ Atab, const Acc *__restrict__ AC,
                   Acc *__restrict__ send, int C, ChainPre<Acc> pre = ChainPre<Acc>{}) {
    __shared__ Acc exits[NCH][kCarryLines][K];
    __shared__ Acc carry_in[kCarryLines][K];
    extern __shared__ __attribute__((aligned(16))) unsigned char pre_raw[];
    Acc *entering_prev = reinterpret_cast<Acc *>(pre_raw);      // PRE: [line][K]

    const int ln = threadIdx.x & (kCarryLines - 1);
    const int ch = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kCarryLines);    // wave-uniform
    const uint32_t L = g.lines;
    const uint32_t line_raw = blockIdx.x * kCarryLines + ln;
    const bool line_ok = line_raw < L;
    const uint32_t line = line_ok ? line_raw : L - 1;     // out-of-range lanes shadow the last line, stores masked
    const int M = g.M;
    const int n_chunks = (int)blockDim.x / kCarryLines;      // <= NCH; fewer when a line has few tiles
    const int tiles_per_block = n_chunks * C;
    const int n_blocks = (M + tiles_per_block - 1) / tiles_per_block;

    if constexpr (PRE) {
        const int NY = (int)L, sp = s_begin - 1;
        const bool pc = pre.causal_prev != 0;
        Acc (*lane_exit)[K] = exits[0];                       // [64][K]: not in use yet
        const int l = (int)threadIdx.x;
        const int i0 = l * pre.S;
        int i1 = i0 + pre.S;
        i1 = i1 > NY ? NY : i1;
        if (l < 64) {
            Acc x[K];
#pragma unroll
            for (int r = 0; r < K; r++) x[r] = Acc(0);
            for (int i = i0; i < i1; i++) {
                const int row = pc ? i : NY - 1 - i;
                Acc nx[K];
#pragma unroll
                for (int r = 0; r < K; r++) nx[r] = pre.exit_states[(size_t)r * NY + row];
                matvec_acc<Acc, K>(pre.AM, x, nx);
#pragma unroll
                for (int r = 0; r < K; r++) x[r] = nx[r];
            }
#pragma unroll
            for (int r = 0; r < K; r++) lane_exit[l][r] = x[r];
        }
        __syncthreads();
        if (l < 64) {
            Acc inc[K];
#pragma unroll
            for (int r = 0; r < K; r++) inc[r] = Acc(0);
            for (int c = 0; c < l; c++) {
                Acc nx[K];
#pragma unroll
                for (int r = 0; r < K; r++) nx[r] = lane_exit[c][r];
                matvec_acc<Acc, K>(pre.AMS, inc, nx);
#pragma unroll
                for (int r = 0; r < K; r++) inc[r] = nx[r];
            }
            for (int i = i0; i < i1; i++) {
                const int row = pc ? i : NY - 1 - i;
                Acc nx[K];
#pragma unroll
                for (int r = 0; r < K; r++) {
                    entering_prev[row * K + r] = inc[r];
                    if (blockIdx.x == 0) pre.incoming_prev[(size_t)r * NY + row] = inc[r];
                    nx[r] = pre.exit_states[(size_t)r * NY + row];
                }
                matvec_acc<Acc, K>(pre.AM, inc, nx);
#pragma unroll
                for (int r = 0; r < K; r++) inc[r] = nx[r];
            }
        }
        __syncthreads();
        // tails of scan sp of this workgroup's lines: tile tt += A^(i+1) * entering state, i = the tile's position in scan order
        Acc xin[K];
#pragma unroll
        for (int j = 0; j < K; j++) xin[j] = entering_prev[line * K + j];
        for (int tt = ch; tt < M; tt += n_chunks) {
            const int i = pc ? tt : M - 1 - tt;
            const Acc *Ap = pre.Apow + (size_t)i * K * K;                  // wave-uniform
            Acc cur[K];
#pragma unroll
            for (int r = 0; r < K; r++) cur[r] = tails[tail_off<K>(g, sp, tt, r, line)];
#pragma unroll
            for (int r = 0; r < K; r++) {
                Acc add = Acc(0);
#pragma unroll
                for (int j = 0; j < K; j++) add = add + Ap[r * K + j] * xin[j];
                if (line_ok) tails[tail_off<K>(g, sp, tt, r, line)] = cur[r] + add;
            }
        }
        __threadfence_block();
        __syncthreads();
    }

    for (int s = s_begin; s < s_end; s++) {
        const bool causal = ((g.causal_mask >> s) & 1u) != 0;
        const Acc *Am = Atab + s * K * K;
        const Acc *ACm = AC + s * K * K;
        if (ch == 0) {
#pragma unroll
            for (int r = 0; r < K; r++) carry_in[ln][r] = Acc(0);
        }
        Acc last_tail[K];
#pragma unroll
        for (int r = 0; r < K; r++) last_tail[r] = Acc(0);

        for (int blk = 0; blk < n_blocks; blk++) {
            const int base_i = blk * tiles_per_block + ch * C;
            int nvalid = M - base_i;
            nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);      // wave-uniform

            // ---- A: load, add chaining, chunk-local recurrence ----
            Acc cur[MAXC][K];
#pragma unroll
            for (int ii = 0; ii < MAXC; ii++) {
#pragma unroll
                for (int r = 0; r < K; r++) cur[ii][r] = Acc(0);
                if (ii < nvalid) {
                    const int tt = causal ? base_i + ii : M - 1 - (base_i + ii);
#pragma unroll
                    for (int r = 0; r < K; r++) cur[ii][r] = tails[tail_off<K>(g, s, tt, r, line)];
                    if (g.part2 != nullptr) {
#pragma unroll
                        for (int r = 0; r < K; r++) cur[ii][r] = cur[ii][r] + static_cast<const Acc *>(g.part2)[tail_off<K>(g, s, tt, r, line)];
                    }
                }
            }
            for (int q = 0; q < s; q++) {
                const bool qc = ((g.causal_mask >> q) & 1u) != 0;
#pragma unroll
                for (int ii = 0; ii < MAXC; ii++) {
                    if (ii < nvalid) {
                        const int tt = causal ? base_i + ii : M - 1 - (base_i + ii);
                        const int v = ((tt == 0 && g.first_is_border) ? 1 : 0) | ((tt == M - 1 && g.last_is_border) ? 2 : 0);
                        const bool q_first = qc ? (tt == 0) : (tt == M - 1);
                        Acc c[K];
                        if (q_first) {
                            bool from_lds = false;
                            if constexpr (PRE) from_lds = q == s_begin - 1;        // (another workgroup may not have stored it yet)
#pragma unroll
                            for (int o = 0; o < K; o++)
                                c[o] = from_lds ? entering_prev[line * K + o] : incoming[(uint32_t)(q * K + o) * L + line];
                        } else {
                            const int tp = qc ? tt - 1 : tt + 1;
#pragma unroll
                            for (int o = 0; o < K; o++) c[o] = tails[tail_off<K>(g, q, tp, o, line)];
                        }
                        const Acc *Wm = Wtab + (((v * g.n_scans + q) * g.n_scans + s) * K) * K;
                        matvec_acc<Acc, K>(Wm, c, cur[ii]);
                    }
                }
            }
            Acc xstate[K];
#pragma unroll
            for (int r = 0; r < K; r++) xstate[r] = Acc(0);
#pragma unroll
            for (int ii = 0; ii < MAXC; ii++) {
                if (ii < nvalid) {
                    matvec_acc<Acc, K>(Am, xstate, cur[ii]);   // cur += A * state of the previous tile
#pragma unroll
                    for (int r = 0; r < K; r++) xstate[r] = cur[ii][r];
                }
            }
#pragma unroll
            for (int r = 0; r < K; r++) exits[ch][ln][r] = xstate[r];
            __syncthreads();

            // ---- B: state entering this chunk ----
            Acc inc[K];
#pragma unroll
            for (int r = 0; r < K; r++) inc[r] = carry_in[ln][r];
            for (int c = 0; c < ch; c++) {
                Acc nx[K];
#pragma unroll
                for (int r = 0; r < K; r++) nx[r] = exits[c][ln][r];
                matvec_acc<Acc, K>(ACm, inc, nx);
#pragma unroll
                for (int r = 0; r < K; r++) inc[r] = nx[r];
            }

            // ---- C: propagate it through the chunk, store the completed tails ----
#pragma unroll
            for (int ii = 0; ii < MAXC; ii++) {
                if (ii < nvalid) {
                    Acc y[K];
#pragma unroll
                    for (int r = 0; r < K; r++) y[r] = Acc(0);
                    matvec_acc<Acc, K>(Am, inc, y);
                    const int tt = causal ? base_i + ii : M - 1 - (base_i + ii);
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        inc[r] = y[r];
                        cur[ii][r] = cur[ii][r] + y[r];
                        last_tail[r] = cur[ii][r];
                        if (line_ok) tails[tail_off<K>(g, s, tt, r, line)] = cur[ii][r];
                    }
                }
            }
            __syncthreads();   // everyone has read exits / carry_in of this block
            // the wave that owns the block's last tile publishes the state entering the next block
            const int last_i = (blk + 1) * tiles_per_block < M ? (blk + 1) * tiles_per_block - 1 : M - 1;
            if (nvalid > 0 && base_i + nvalid - 1 == last_i) {
#pragma unroll
                for (int r = 0; r < K; r++) carry_in[ln][r] = last_tail[r];
                if (send != nullptr && last_i == M - 1 && line_ok) {
#pragma unroll
                    for (int r = 0; r < K; r++) send[(uint32_t)((s - s_begin) * K + r) * L + line] = last_tail[r];
                }
            }
            __syncthreads();
        }
        // scan s+1 chains on the tails just stored by other waves of this workgroup
        __threadfence_block();
        __syncthreads();
    }
}

// completes one scan of the pair in registers (the chaining terms must already be in t) and stores it
// (the owned-tile arrays are [16][KP] with KP = max(K, 2): as [16][1] the compiler turns them into one 16-wide vector
// and every guarded element update into a whole-vector copy through scratch)
template <typename Acc, int K, int KP>
__device__ __forceinline__ void pair_matvec(const Acc *__restrict__ m, const Acc (&x)[K], Acc (&y)[KP]) {
#pragma unroll
    for (int r = 0; r < K; r++)
#pragma unroll
        for (int j = 0; j < K; j++) y[r] = y[r] + m[r * K + j] * x[j];
}

template <typename Acc, int K, int KP, bool CAUSAL, int MAXC>
__device__ __forceinline__ void pair_run_scan(Acc (&t)[MAXC][KP], int s, const CarryGeom &g, Acc *__restrict__ tails,
                                          const Acc *__restrict__ Atab, const Acc *__restrict__ AC, int C,
                                          Acc (*exits)[kCarryLines][K], int ln, int ch, int n_chunks, int t0, int nvalid,
                                          uint32_t line, bool line_ok, Acc *__restrict__ send) {
    const int M = g.M;
    const uint32_t L = g.lines;
    constexpr bool causal = CAUSAL;     // compile time: the owned tiles stay statically indexed registers
    const Acc *Am = Atab + s * K * K;
    const Acc *ACm = AC + s * K * K;
    Acc x[K];
#pragma unroll
    for (int r = 0; r < K; r++) x[r] = Acc(0);
    // chunk-local recurrence in scan direction, zero entering state
#pragma unroll
    for (int p = 0; p < MAXC; p++) {
        const int ii = causal ? p : MAXC - 1 - p;
        if (ii < nvalid) {
            pair_matvec<Acc, K, KP>(Am, x, t[ii]);
#pragma unroll
            for (int r = 0; r < K; r++) x[r] = t[ii][r];
        }
    }
#pragma unroll
    for (int r = 0; r < K; r++) exits[ch][ln][r] = x[r];
    __syncthreads();
    // state entering this chunk: the chunks before it in scan direction (all of them full chunks of C tiles)
    Acc inc[K];
#pragma unroll
    for (int r = 0; r < K; r++) inc[r] = Acc(0);
    if (causal) {
#pragma unroll 1
        for (int c = 0; c < ch; c++) {
            Acc nx[K];
#pragma unroll
            for (int r = 0; r < K; r++) nx[r] = exits[c][ln][r];
            matvec_acc<Acc, K>(ACm, inc, nx);
#pragma unroll
            for (int r = 0; r < K; r++) inc[r] = nx[r];
        }
    } else {
#pragma unroll 1
        for (int c = n_chunks - 1; c > ch; c--) {
            Acc nx[K];
#pragma unroll
            for (int r = 0; r < K; r++) nx[r] = exits[c][ln][r];
            // the chunk at the high end may be partial or empty: an empty one hands over nothing and the
            // transfer across the partial one is folded into its own exit state, so AC applies from the
            // second chunk (in scan direction) on
            const int nv_c = M - c * C;
            if (nv_c >= C) matvec_acc<Acc, K>(ACm, inc, nx);
#pragma unroll
            for (int r = 0; r < K; r++) inc[r] = nx[r];
        }
    }
    // propagate through the chunk and store
#pragma unroll
    for (int p = 0; p < MAXC; p++) {
        const int ii = causal ? p : MAXC - 1 - p;
        if (ii < nvalid) {
            Acc y[K];
#pragma unroll
            for (int r = 0; r < K; r++) y[r] = Acc(0);
            matvec_acc<Acc, K>(Am, inc, y);
#pragma unroll
            for (int r = 0; r < K; r++) {
                inc[r] = y[r];
                t[ii][r] = t[ii][r] + y[r];
                if (line_ok) tails[tail_off<K>(g, s, t0 + ii, r, line)] = t[ii][r];
                // the slab's exit carry: the completed tail of the last tile in scan direction
                if (send != nullptr && line_ok && t0 + ii == (causal ? M - 1 : 0)) send[(uint32_t)r * L + line] = t[ii][r];
            }
        }
    }
}

// ---- two scans of one dimension in one pass over registers -----------------------------------------------------
// The common case -- a causal/anticausal pair (or any two scans) of a dimension whose tiles fit one block of chunks --
// gets a kernel without the second scan's trip through memory.  Every thread owns the SAME C tiles (memory order)
// of its line for both scans, requests both scans' tails up front, completes the first scan, and takes the chaining
// terms of the second scan (create_tail_residual_term, lib/split.cpp:912-1004) from its own registers plus one
// halo tile from each neighbouring chunk (through LDS).  The general kernel above re-reads the first scan's
// completed tails from memory behind a barrier instead.
// MAXC: tiles a thread owns (16).  Order 3: two register-resident scans of 16 tiles do not fit the 128 registers of a
// 16-wave workgroup, so its launches have 8 chunk-waves (NCHB) -- lines of at most 128 tiles, i.e. every image on 128-row tiles
// NCHB: most chunk-waves the launch may have (bounds the registers: 16 waves -> 128, 8 waves -> 256)
template <typename Acc, int K, int MAXC, int NCHB = kCarryChunks>
__global__ void __launch_bounds__(kCarryLines * NCHB)
carry_pair_kernel(CarryGeom g, int s0, Acc *__restrict__ tails, const Acc *__restrict__ incoming,
                  const Acc *__restrict__ Wtab, const Acc *__restrict__ Atab, const Acc *__restrict__ AC, int C,
                  Acc *__restrict__ send) {
    // LDS by the launch's chunk count (pair_lds_bytes): sized for 16 chunks it was 24 KiB -- six workgroups per CU, and where the
    // lines alone fill the chip (a volume's 4 M lines: ONE chunk-wave per workgroup) that meant six WAVES per CU on a kernel
    // that does nothing but wait for its 64 loads (config 5 at 2048^3: x / y / z carries 0.47 / 0.67 / 0.69 ms at 2.3-3.1 TB/s)
    extern __shared__ __attribute__((aligned(16))) unsigned char pair_raw[];
    const int n_chunks = (int)blockDim.x / kCarryLines;
    Acc (*exits)[kCarryLines][K] = reinterpret_cast<Acc (*)[kCarryLines][K]>(pair_raw);                       // [n_chunks]
    Acc (*edge)[2][kCarryLines][K] = reinterpret_cast<Acc (*)[2][kCarryLines][K]>(exits + n_chunks);          // [n_chunks]: first / last completed tail of every chunk (scan s0)

    const int ln = threadIdx.x & (kCarryLines - 1);
    const int ch = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kCarryLines);    // wave-uniform
    const uint32_t L = g.lines;
    const uint32_t line_raw = blockIdx.x * kCarryLines + ln;
    const bool line_ok = line_raw < L;
    const uint32_t line = line_ok ? line_raw : L - 1;
    const int M = g.M;
    const int t0 = ch * C;                                   // first owned tile (memory order)
    int nvalid = M - t0;
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);     // wave-uniform

    constexpr int KP = K < 2 ? 2 : K;
    Acc ta[MAXC][KP], tb[MAXC][KP];              // owned tiles of scan s0 / s0+1
    auto load_scan = [&](Acc (&t)[MAXC][KP], int s) {
#pragma unroll
        for (int ii = 0; ii < MAXC; ii++) {
#pragma unroll
            for (int r = 0; r < KP; r++) t[ii][r] = Acc(0);
            if (ii < nvalid) {
#pragma unroll
                for (int r = 0; r < K; r++) t[ii][r] = tails[tail_off<K>(g, s, t0 + ii, r, line)];
            }
        }
        if (g.part2 != nullptr) {          // the tile-local tails in two parts: requested with the first, added once both are there
            const Acc *p2 = static_cast<const Acc *>(g.part2);
            Acc u[MAXC][KP];
#pragma unroll
            for (int ii = 0; ii < MAXC; ii++) {
#pragma unroll
                for (int r = 0; r < KP; r++) u[ii][r] = Acc(0);
                if (ii < nvalid) {
#pragma unroll
                    for (int r = 0; r < K; r++) u[ii][r] = p2[tail_off<K>(g, s, t0 + ii, r, line)];
                }
            }
#pragma unroll
            for (int ii = 0; ii < MAXC; ii++)
#pragma unroll
                for (int r = 0; r < KP; r++) t[ii][r] = t[ii][r] + u[ii][r];
        }
    };
    load_scan(ta, s0);
    load_scan(tb, s0 + 1);

    if ((g.causal_mask >> s0) & 1u) pair_run_scan<Acc, K, KP, true, MAXC>(ta, s0, g, tails, Atab, AC, C, exits, ln, ch, n_chunks, t0, nvalid, line, line_ok, send);
    else                            pair_run_scan<Acc, K, KP, false, MAXC>(ta, s0, g, tails, Atab, AC, C, exits, ln, ch, n_chunks, t0, nvalid, line, line_ok, send);

    // ---- chaining of scan s0+1 on the completed scan s0 ----
    {
        const int q = s0, s = s0 + 1;
        const bool qc = ((g.causal_mask >> q) & 1u) != 0;
#pragma unroll
        for (int r = 0; r < K; r++) {
            edge[ch][0][ln][r] = ta[0][r];
            Acc last = ta[0][r];
#pragma unroll
            for (int ii = 1; ii < MAXC; ii++) last = (ii < nvalid) ? ta[ii][r] : last;
            edge[ch][1][ln][r] = last;
        }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < MAXC; ii++) {
            if (ii < nvalid) {
                const int tt = t0 + ii;
                const int v = ((tt == 0 && g.first_is_border) ? 1 : 0) | ((tt == M - 1 && g.last_is_border) ? 2 : 0);
                const bool q_first = qc ? (tt == 0) : (tt == M - 1);
                Acc c[K];
                if (q_first) {
#pragma unroll
                    for (int o = 0; o < K; o++) c[o] = incoming[(uint32_t)(q * K + o) * L + line];
                } else if (qc) {           // carry of a causal scan comes from the tile before
#pragma unroll
                    for (int o = 0; o < K; o++) c[o] = (ii > 0) ? ta[ii > 0 ? ii - 1 : 0][o] : edge[ch - 1][1][ln][o];
                } else {                   // ... of an anticausal scan from the tile after
#pragma unroll
                    for (int o = 0; o < K; o++)
                        c[o] = (ii + 1 < nvalid) ? ta[ii + 1 < MAXC ? ii + 1 : 0][o] : edge[ch + 1][0][ln][o];
                }
                const Acc *Wm = Wtab + (((v * g.n_scans + q) * g.n_scans + s) * K) * K;
                pair_matvec<Acc, K, KP>(Wm, c, tb[ii]);
            }
        }
    }
    __syncthreads();          // exits[] is reused by the second scan
    Acc *send2 = send != nullptr ? send + (uint32_t)K * L : nullptr;
    if ((g.causal_mask >> (s0 + 1)) & 1u) pair_run_scan<Acc, K, KP, true, MAXC>(tb, s0 + 1, g, tails, Atab, AC, C, exits, ln, ch, n_chunks, t0, nvalid, line, line_ok, send2);
    else                                  pair_run_scan<Acc, K, KP, false, MAXC>(tb, s0 + 1, g, tails, Atab, AC, C, exits, ln, ch, n_chunks, t0, nvalid, line, line_ok, send2);
}

// Row chaining for a long 1-D signal folded into NY rows of MX tiles (plan_fused.cpp, "chained rows"): the
// blocked scan above completes every row with a zero entering state; this kernel walks the rows' exit states
// (NY k-vectors -- tiny) and produces the state entering every row.  One wave; lane l owns rows
// [l*S, (l+1)*S) in scan order: local recurrence, lane states combined through LDS with AM^S, then propagated.
template <typename Acc, int K>
__global__ void __launch_bounds__(64)
row_chain_kernel(const Acc *__restrict__ exit_states, Acc *__restrict__ incoming, int NY, int causal,
                 const Acc *__restrict__ AM, const Acc *__restrict__ AMS, int S) {
    __shared__ Acc lane_exit[64][K];
    const int l = threadIdx.x;
    const int i0 = l * S;
    int i1 = i0 + S;
    i1 = i1 > NY ? NY : i1;
    Acc x[K];
#pragma unroll
    for (int r = 0; r < K; r++) x[r] = Acc(0);
    for (int i = i0; i < i1; i++) {
        const int row = causal ? i : NY - 1 - i;
        Acc nx[K];
#pragma unroll
        for (int r = 0; r < K; r++) nx[r] = exit_states[(size_t)r * NY + row];
        matvec_acc<Acc, K>(AM, x, nx);
#pragma unroll
        for (int r = 0; r < K; r++) x[r] = nx[r];
    }
#pragma unroll
    for (int r = 0; r < K; r++) lane_exit[l][r] = x[r];
    __syncthreads();
    Acc inc[K];
#pragma unroll
    for (int r = 0; r < K; r++) inc[r] = Acc(0);
    for (int c = 0; c < l; c++) {               // lanes before l own full segments of S rows
        Acc nx[K];
#pragma unroll
        for (int r = 0; r < K; r++) nx[r] = lane_exit[c][r];
        matvec_acc<Acc, K>(AMS, inc, nx);
#pragma unroll
        for (int r = 0; r < K; r++) inc[r] = nx[r];
    }
    for (int i = i0; i < i1; i++) {
        const int row = causal ? i : NY - 1 - i;
        Acc nx[K];
#pragma unroll
        for (int r = 0; r < K; r++) {
            incoming[(size_t)r * NY + row] = inc[r];
            nx[r] = exit_states[(size_t)r * NY + row];
        }
        matvec_acc<Acc, K>(AM, inc, nx);
#pragma unroll
        for (int r = 0; r < K; r++) inc[r] = nx[r];
    }
}

// The same chain and what follows it in ONE launch: every workgroup walks the rows' exit states itself (wave 0, the
// algorithm above; NY k-vectors, a few KiB), keeps the state entering every row in LDS, and then propagates it through its
// share of the rows' tails -- tail(i) += A^(i+1) * entering state, the powers tabulated on the host (GenericDimArgs::Apow),
// like carry_apply_parallel_kernel (kernels_generic.hip).  Redundant across workgroups, but a long signal's carry stage is
// launch-bound: three launches per scan become two.  The first workgroup also stores the entering states (the chaining
// terms of the next scan and the final pass read them).
constexpr int kChainApplyTiles = 8;
template <typename Acc, int K>
__global__ void __launch_bounds__(256)
chain_apply_kernel(GenericDimArgs<Acc> a, int s, const Acc *__restrict__ exit_states, Acc *__restrict__ incoming, int causal,
                   const Acc *__restrict__ AM, const Acc *__restrict__ AMS, int S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_raw[];
    Acc *entering = reinterpret_cast<Acc *>(chain_raw);        // [row][K]
    __shared__ Acc lane_exit[64][K];
    const int NY = (int)a.g.lines;
    const int l = threadIdx.x;
    const int i0 = l * S;
    int i1 = i0 + S;
    i1 = i1 > NY ? NY : i1;
    if (l < 64) {
        Acc x[K];
#pragma unroll
        for (int r = 0; r < K; r++) x[r] = Acc(0);
        for (int i = i0; i < i1; i++) {
            const int row = causal ? i : NY - 1 - i;
            Acc nx[K];
#pragma unroll
            for (int r = 0; r < K; r++) nx[r] = exit_states[(size_t)r * NY + row];
            matvec_acc<Acc, K>(AM, x, nx);
#pragma unroll
            for (int r = 0; r < K; r++) x[r] = nx[r];
        }
#pragma unroll
        for (int r = 0; r < K; r++) lane_exit[l][r] = x[r];
    }
    __syncthreads();
    if (l < 64) {
        Acc inc[K];
#pragma unroll
        for (int r = 0; r < K; r++) inc[r] = Acc(0);
        for (int c = 0; c < l; c++) {               // lanes before l own full segments of S rows
            Acc nx[K];
#pragma unroll
            for (int r = 0; r < K; r++) nx[r] = lane_exit[c][r];
            matvec_acc<Acc, K>(AMS, inc, nx);
#pragma unroll
            for (int r = 0; r < K; r++) inc[r] = nx[r];
        }
        const bool publish = blockIdx.x == 0 && blockIdx.y == 0;
        for (int i = i0; i < i1; i++) {
            const int row = causal ? i : NY - 1 - i;
            Acc nx[K];
#pragma unroll
            for (int r = 0; r < K; r++) {
                entering[row * K + r] = inc[r];
                if (publish) incoming[(size_t)r * NY + row] = inc[r];
                nx[r] = exit_states[(size_t)r * NY + row];
            }
            matvec_acc<Acc, K>(AM, inc, nx);
#pragma unroll
            for (int r = 0; r < K; r++) inc[r] = nx[r];
        }
    }
    __syncthreads();
    const int line = (int)blockIdx.x * 256 + l;
    if (line >= NY) return;
    Acc x[K];
#pragma unroll
    for (int j = 0; j < K; j++) x[j] = entering[line * K + j];
    const int t0 = (int)blockIdx.y * kChainApplyTiles;
    const size_t L = (size_t)NY;
    // all loads of the chunk first, then the stores (interleaved, every store could alias the next load)
#pragma unroll
    for (int r = 0; r < K; r++) {
        Acc cur[kChainApplyTiles];
#pragma unroll
        for (int u = 0; u < kChainApplyTiles; u++) {
            const int i = t0 + u;
            const int tt = causal ? i : a.M - 1 - i;
            cur[u] = i < a.M ? a.tails[(((size_t)s * a.M + tt) * K + r) * L + line] : Acc(0);
        }
#pragma unroll
        for (int u = 0; u < kChainApplyTiles; u++) {
            const int i = t0 + u;
            if (i < a.M) {
                const int tt = causal ? i : a.M - 1 - i;
                const Acc *Ap = a.Apow + ((size_t)s * a.M + i) * K * K;          // wave-uniform
                Acc add = Acc(0);
#pragma unroll
                for (int j = 0; j < K; j++) add = add + Ap[r * K + j] * x[j];
                a.tails[(((size_t)s * a.M + tt) * K + r) * L + line] = cur[u] + add;
            }
        }
    }
}

}  // namespace

bool chain_apply_applies(int K, int64_t NY, size_t acc_bytes) {
    static const bool off = RF_KNOB("RF_NO_CHAIN_APPLY") != nullptr;      // A/B runs: row_chain + carry_apply as two launches
    return !off && K >= 1 && K <= 3 && NY > 0 && (size_t)NY * K * acc_bytes <= 48 * 1024;
}

template <typename Acc>
int launch_chain_apply(int K, const GenericDimArgs<Acc> &a, int s, const Acc *exit_states, Acc *incoming, bool causal,
                       const Acc *AM, const Acc *AMS, int S, hipStream_t stream) {
    const int NY = (int)a.g.lines;
    if (NY <= 0 || a.M <= 0) return RF_OK;
    if (a.Apow == nullptr || a.tile_major) { set_error("chain apply: needs the tabulated powers and row-major tails"); return RF_ERR_INVALID_ARG; }
    if (S < 1 || (int64_t)S * 64 < NY) { set_error("chain apply: segment length %d does not cover %d rows", S, NY); return RF_ERR_INVALID_ARG; }
    dim3 grid((unsigned)((NY + 255) / 256), (unsigned)((a.M + kChainApplyTiles - 1) / kChainApplyTiles));
    const size_t lds = (size_t)NY * K * sizeof(Acc);
#define RF_CASE(KK) if (K == KK) { hipLaunchKernelGGL((chain_apply_kernel<Acc, KK>), grid, dim3(256), lds, stream, a, s, exit_states, incoming, causal ? 1 : 0, AM, AMS, S); RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
    RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    set_error("chain apply: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}
template int launch_chain_apply<float>(int, const GenericDimArgs<float> &, int, const float *, float *, bool, const float *, const float *, int, hipStream_t);
template int launch_chain_apply<uint32_t>(int, const GenericDimArgs<uint32_t> &, int, const uint32_t *, uint32_t *, bool, const uint32_t *, const uint32_t *, int, hipStream_t);
template int launch_chain_apply<double>(int, const GenericDimArgs<double> &, int, const double *, double *, bool, const double *, const double *, int, hipStream_t);

template <typename Acc>
int launch_row_chain(int K, const Acc *exit_states, Acc *incoming, int NY, bool causal, const Acc *AM, const Acc *AMS,
                     int S, hipStream_t stream) {
    if (NY <= 0) return RF_OK;
    if (S < 1 || (int64_t)S * 64 < NY) { set_error("row chain: segment length %d does not cover %d rows", S, NY); return RF_ERR_INVALID_ARG; }
#define RF_CASE(KK) if (K == KK) { hipLaunchKernelGGL((row_chain_kernel<Acc, KK>), dim3(1), dim3(64), 0, stream, exit_states, incoming, NY, causal ? 1 : 0, AM, AMS, S); RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
    RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    set_error("row chain: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}
template int launch_row_chain<float>(int, const float *, float *, int, bool, const float *, const float *, int, hipStream_t);
template int launch_row_chain<uint32_t>(int, const uint32_t *, uint32_t *, int, bool, const uint32_t *, const uint32_t *, int,
                                        hipStream_t);
template int launch_row_chain<double>(int, const double *, double *, int, bool, const double *, const double *, int, hipStream_t);

int carry_chunk_count(int64_t M, int64_t lines, int C, int K) {
    if (K == 3 && M > 64 && C > kCarryPair3MaxC && (int64_t)kCarryPair3Chunks * C >= M) return (int)((M + C - 1) / C);      // the order-3 pair kernel's chunking
    const int max_chunks = K <= 3 ? kCarryChunks : kCarryChunksHigh;
    const int64_t line_groups = (lines + kCarryLines - 1) / kCarryLines;
    static const int64_t target_waves_n = RF_KNOB("RF_CARRY_WAVES") ? atoll(RF_KNOB("RF_CARRY_WAVES")) : 4096;      // A/B
    int64_t want = (target_waves_n + line_groups - 1) / line_groups;
    want = want < 1 ? 1 : (want > max_chunks ? max_chunks : want);
    int64_t need = (M + C - 1) / C;          // chunks that cover the line in one block
    int64_t n = need < want ? need : want;
    // ... a line that a few more chunks cover in ONE block takes them: two scans then run in the register-chained pair kernel
    // instead of block after block in the general one (cfg4a's y carries, 128 tiles per line at 6 wanted chunks: 48 -> 41 us)
    if (need <= max_chunks && need <= 2 * want) n = need;
    return (int)(n < 1 ? 1 : n);
}

template <typename Acc>
int launch_carry_block(int K, const GenericDimArgs<Acc> &a, uint32_t causal_mask, int s_begin, int s_end, Acc *send,
                       const Acc *AC, int C, hipStream_t stream, const ChainPre<Acc> *pre) {
    if (a.g.lines <= 0 || a.M <= 0 || s_end <= s_begin) return RF_OK;
    if (C < 1 || C > (K <= 3 ? kCarryMaxC : kCarryMaxCHigh)) { set_error("carry: chunk length %d out of range", C); return RF_ERR_INVALID_ARG; }
    const uint64_t total = (uint64_t)a.n_scans * a.M * a.k * a.g.lines;
    if (total >= (1ull << 32) || a.g.lines >= (1ll << 31)) {
        set_error("carry: tails array too large for 32-bit offsets (%llu elements)", (unsigned long long)total);
        return RF_ERR_UNSUPPORTED;
    }
    CarryGeom g{};
    g.lines = (uint32_t)a.g.lines;
    g.tile_major = a.tile_major ? 1u : 0u;
    if (g.tile_major && (a.g.lines % 256 != 0)) { set_error("carry: tile-major tails need whole 256-line tiles"); return RF_ERR_INVALID_ARG; }
    g.M = a.M; g.n_scans = a.n_scans;
    g.first_is_border = a.first_is_border; g.last_is_border = a.last_is_border;
    g.causal_mask = causal_mask;
    g.part2 = a.tails_part2;
    if (g.part2 != nullptr && pre != nullptr) { set_error("carry: tails in two parts do not go with a chain prologue"); return RF_ERR_INVALID_ARG; }
    const unsigned grid = (unsigned)((a.g.lines + kCarryLines - 1) / kCarryLines);
    // one wave per chunk of C tiles; a line with few tiles gets fewer waves instead of idle ones
    int n_chunks = carry_chunk_count(a.M, a.g.lines, C, K);
    const unsigned threads = (unsigned)(kCarryLines * n_chunks);
    if (pre != nullptr) {      // chained rows: finish scan s_begin - 1 first (PRE)
        if (s_begin < 1 || s_end != s_begin + 1 || K > 3 || a.tile_major) { set_error("carry: misplaced chain prologue"); return RF_ERR_INVALID_ARG; }
        const size_t lds = (size_t)a.g.lines * K * sizeof(Acc);
#define RF_CASE(KK) if (K == KK) { hipLaunchKernelGGL((carry_block_kernel<Acc, KK, kCarryMaxC, kCarryChunks, true>), dim3(grid), dim3(threads), lds, stream, g, s_begin, s_end, a.tails, (const Acc *)a.incoming, a.W, a.A, AC, send, C, *pre); RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
        RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    }
    // two scans of order <= 2, every line's tiles in one block of chunks: the register-chained pair kernel
    // (order 3 and f64 do not fit its register budget at 16 waves and stay on the general kernel)
    if constexpr (sizeof(Acc) == 4) {
        static const bool pair_off = RF_KNOB("RF_CARRY_NO_PAIR") != nullptr;     // tuning knob: always the general kernel
        if (s_end - s_begin == 2 && K <= 2 && (int64_t)n_chunks * C >= a.M && !pair_off) {
            const size_t lds = (size_t)n_chunks * 3 * kCarryLines * K * sizeof(Acc);      // exits + edge, per chunk-wave
            if (K == 1) hipLaunchKernelGGL((carry_pair_kernel<Acc, 1, kCarryMaxC>), dim3(grid), dim3(threads), lds, stream, g, s_begin, a.tails,
                                           (const Acc *)a.incoming, a.W, a.A, AC, C, send);
            else        hipLaunchKernelGGL((carry_pair_kernel<Acc, 2, kCarryMaxC>), dim3(grid), dim3(threads), lds, stream, g, s_begin, a.tails,
                                           (const Acc *)a.incoming, a.W, a.A, AC, C, send);
            RF_HIP_CHECK(hipGetLastError());
            return RF_OK;
        }
        static const bool pair3_off = RF_KNOB("RF_CARRY_NO_PAIR3") != nullptr;
        if (s_end - s_begin == 2 && K == 3 && a.M > 64 && C > kCarryPair3MaxC && C <= kCarryMaxC && n_chunks <= kCarryPair3Chunks &&
            (int64_t)n_chunks * C >= a.M && !pair_off && !pair3_off) {
            // order 3: eight waves of up to 16 tiles (158 registers, no scratch)
            hipLaunchKernelGGL((carry_pair_kernel<Acc, 3, kCarryMaxC, kCarryPair3Chunks>), dim3(grid), dim3(threads),
                               (size_t)n_chunks * 3 * kCarryLines * 3 * sizeof(Acc), stream, g, s_begin, a.tails,
                               (const Acc *)a.incoming, a.W, a.A, AC, C, send);
            RF_HIP_CHECK(hipGetLastError());
            return RF_OK;
        }
    }
#define RF_CASE(KK, MC) if (K == KK) { hipLaunchKernelGGL((carry_block_kernel<Acc, KK, MC, (KK <= 3 ? kCarryChunks : kCarryChunksHigh)>), dim3(grid), dim3(threads), 0, stream, g, s_begin, s_end, a.tails, (const Acc *)a.incoming, a.W, a.A, AC, send, C); RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
    RF_CASE(1, kCarryMaxC) RF_CASE(2, kCarryMaxC) RF_CASE(3, kCarryMaxC)
    RF_CASE(4, kCarryMaxCHigh) RF_CASE(5, kCarryMaxCHigh) RF_CASE(6, kCarryMaxCHigh) RF_CASE(7, kCarryMaxCHigh)
    RF_CASE(8, kCarryMaxCHigh)
#undef RF_CASE
    set_error("carry: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}

int carry_chunk_length(int64_t M, int64_t lines, int K) {
    // order 3: chunks of at most 8 tiles whenever 16 of them cover the line, so that two scans of a dimension fit the
    // register-chained pair kernel
    // (long lines only: at 64 tiles per line the general kernel is as fast)
    if (K == 3 && M > 64 && M <= (int64_t)kCarryPair3Chunks * kCarryMaxC && RF_KNOB("RF_CARRY_NO_PAIR3") == nullptr &&
        RF_KNOB("RF_CARRY_NO_PAIR") == nullptr)
        return (int)((M + kCarryPair3Chunks - 1) / kCarryPair3Chunks);
    const int max_c = K <= 3 ? kCarryMaxC : kCarryMaxCHigh;
    const int max_chunks = K <= 3 ? kCarryChunks : kCarryChunksHigh;
    // Chunks (waves) per line: enough to put ~4096 waves on the chip, no more -- with many lines a single
    // wave walks all tiles of its 64 lines and the cross-chunk combine through LDS disappears.
    const int64_t line_groups = (lines + kCarryLines - 1) / kCarryLines;
    static const int64_t target_waves_l = RF_KNOB("RF_CARRY_WAVES") ? atoll(RF_KNOB("RF_CARRY_WAVES")) : 4096;      // A/B
    int64_t want = (target_waves_l + line_groups - 1) / line_groups;
    want = want < 1 ? 1 : (want > max_chunks ? max_chunks : want);
    if (want > M) want = M < 1 ? 1 : M;
    int64_t c = (M + want - 1) / want;
    return (int)(c < 1 ? 1 : (c > max_c ? max_c : c));
}

template int launch_carry_block<float>(int, const GenericDimArgs<float> &, uint32_t, int, int, float *, const float *, int,
                                       hipStream_t, const ChainPre<float> *);
template int launch_carry_block<uint32_t>(int, const GenericDimArgs<uint32_t> &, uint32_t, int, int, uint32_t *,
                                          const uint32_t *, int, hipStream_t, const ChainPre<uint32_t> *);
template int launch_carry_block<double>(int, const GenericDimArgs<double> &, uint32_t, int, int, double *, const double *, int,
                                        hipStream_t, const ChainPre<double> *);

}  // namespace rf
