// kernels_carry.hip -- the cross-tile carry recurrence as a blocked parallel scan.
//
// The reference runs this stage (create_complete_tail_term, lib/split.cpp:743-867) as one
// unrolled per-thread loop over all tiles of a line (gpu_auto_inter_schedule,
// lib/recfilter.cpp:763-785): a chain of M dependent steps, each waiting on a global load.
// On MI355X that is pure latency (16k lines = one wave per CU).  Here a workgroup owns 16 lines
// and cuts each line's M tiles into 16 chunks:
//
//   A  every thread (line, chunk) loads its <=16 tile tails at once (independent loads), adds
//      the same-dimension chaining terms (create_tail_residual_term, lib/split.cpp:912-1004)
//      and, for the y dimension of the fused path, the cross-dimension residual
//      sum_o G[x][o] * tau[o] (lib/split.cpp:1215-1633), then runs the recurrence inside its
//      chunk with a zero incoming state
//   B  chunk exit states go through LDS; every thread forms the state entering its chunk with
//      the precomputed chunk transfer matrix A^C (<= 15 k x k steps, no memory traffic)
//   C  the entering state is propagated through the chunk and the completed tails are stored
//
// All scans of the dimension run inside ONE launch (a workgroup owns its lines completely, so a
// barrier orders scan s+1's chaining loads after scan s's stores).  Results are identical to the
// serial recurrence up to f32 rounding; integer pixel types stay bit-exact (ring arithmetic).
#include "kernels.h"
#include "kernels_fused.h"

namespace rf {

namespace {

constexpr int kCarryLines = 16;    // lines per workgroup
constexpr int kCarryChunks = 16;   // chunks per line per block of tiles
constexpr int kCarryMaxC = 16;     // tiles per chunk

template <typename Acc, int K>
__device__ __forceinline__ void matvec_acc(const Acc *__restrict__ m, const Acc (&x)[K], Acc (&y)[K]) {
#pragma unroll
    for (int r = 0; r < K; r++)
#pragma unroll
        for (int j = 0; j < K; j++) y[r] = y[r] + m[r * K + j] * x[j];
}

template <typename Acc, int K>
__global__ void __launch_bounds__(256)
carry_block_kernel(GenericDimArgs<Acc> a, int s_begin, int s_end, CarryResidual<Acc> res, Acc *__restrict__ send,
                   const Acc *__restrict__ AC, int C) {
    __shared__ Acc exits[kCarryChunks][kCarryLines][K];
    __shared__ Acc carry_in[kCarryLines][K];

    const int t = threadIdx.x;
    const int ln = t & (kCarryLines - 1), ch = t >> 4;
    const int64_t line = (int64_t)blockIdx.x * kCarryLines + ln;
    const bool line_ok = line < a.g.lines;
    const int64_t L = a.g.lines;
    const int M = a.M;
    const int tiles_per_block = kCarryChunks * C;
    const int n_blocks = (M + tiles_per_block - 1) / tiles_per_block;

    // residual geometry (y dimension of the fused path): line = x + NX*z
    int tx = 0, xi = 0, vx = 0;
    int64_t z = 0;
    if (res.tau != nullptr && line_ok) {
        const int64_t x = line % res.NX;
        z = line / res.NX;
        tx = (int)(x / kFusedTX);
        xi = (int)(x % kFusedTX);
        vx = (tx == 0 ? 1 : 0) | (tx == res.MX - 1 ? 2 : 0);
    }

    for (int s = s_begin; s < s_end; s++) {
        const bool causal = a.scans[s].causal != 0;
        const Acc *Am = a.A + (int64_t)s * K * K;
        const Acc *ACm = AC + (int64_t)s * K * K;
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
            nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);
            if (!line_ok) nvalid = 0;

            // ---- A: load, add chaining and residual, chunk-local recurrence ----
            Acc cur[kCarryMaxC][K];
#pragma unroll
            for (int ii = 0; ii < kCarryMaxC; ii++) {
#pragma unroll
                for (int r = 0; r < K; r++) cur[ii][r] = Acc(0);
                if (ii < nvalid) {
                    const int i = base_i + ii;
                    const int tt = causal ? i : M - 1 - i;
#pragma unroll
                    for (int r = 0; r < K; r++) cur[ii][r] = a.tails[(((int64_t)s * M + tt) * K + r) * L + line];
                }
            }
            if (res.tau != nullptr) {
#pragma unroll
                for (int ii = 0; ii < kCarryMaxC; ii++) {
                    if (ii < nvalid) {
                        const int i = base_i + ii;
                        const int tt = causal ? i : M - 1 - i;
                        const int64_t tile = (z * M + tt) * res.MX + tx;
                        for (int q = 0; q < res.nx; q++) {
                            const Acc *g = res.G + (((int64_t)vx * res.nx + q) * kFusedTX + xi) * K;
                            const Acc *tq = res.tau + ((tile * res.nx + q) * K) * (int64_t)res.ny * K + (int64_t)s * K;
#pragma unroll
                            for (int o = 0; o < K; o++)
#pragma unroll
                                for (int r = 0; r < K; r++)
                                    cur[ii][r] = cur[ii][r] + g[o] * tq[(int64_t)o * res.ny * K + r];
                        }
                    }
                }
            }
            for (int q = 0; q < s; q++) {
                const bool qc = a.scans[q].causal != 0;
#pragma unroll
                for (int ii = 0; ii < kCarryMaxC; ii++) {
                    if (ii < nvalid) {
                        const int i = base_i + ii;
                        const int tt = causal ? i : M - 1 - i;
                        const int v = ((tt == 0 && a.first_is_border) ? 1 : 0) | ((tt == M - 1 && a.last_is_border) ? 2 : 0);
                        const bool q_first = qc ? (tt == 0) : (tt == M - 1);
                        Acc c[K];
                        if (q_first) {
#pragma unroll
                            for (int o = 0; o < K; o++) c[o] = a.incoming[((int64_t)q * K + o) * L + line];
                        } else {
                            const int tp = qc ? tt - 1 : tt + 1;
#pragma unroll
                            for (int o = 0; o < K; o++) c[o] = a.tails[(((int64_t)q * M + tp) * K + o) * L + line];
                        }
                        const Acc *Wm = a.W + ((((int64_t)v * a.n_scans + q) * a.n_scans + s) * K) * K;
                        matvec_acc<Acc, K>(Wm, c, cur[ii]);
                    }
                }
            }
            Acc xstate[K];
#pragma unroll
            for (int r = 0; r < K; r++) xstate[r] = Acc(0);
#pragma unroll
            for (int ii = 0; ii < kCarryMaxC; ii++) {
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
            for (int ii = 0; ii < kCarryMaxC; ii++) {
                if (ii < nvalid) {
                    Acc y[K];
#pragma unroll
                    for (int r = 0; r < K; r++) y[r] = Acc(0);
                    matvec_acc<Acc, K>(Am, inc, y);
                    const int i = base_i + ii;
                    const int tt = causal ? i : M - 1 - i;
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        inc[r] = y[r];
                        cur[ii][r] = cur[ii][r] + y[r];
                        a.tails[(((int64_t)s * M + tt) * K + r) * L + line] = cur[ii][r];
                        last_tail[r] = cur[ii][r];
                    }
                }
            }
            __syncthreads();   // everyone has read exits / carry_in of this block
            // the thread that owns the block's last tile publishes the state entering the next block
            const int last_i = (blk + 1) * tiles_per_block < M ? (blk + 1) * tiles_per_block - 1 : M - 1;
            if (line_ok && nvalid > 0 && base_i + nvalid - 1 == last_i) {
#pragma unroll
                for (int r = 0; r < K; r++) carry_in[ln][r] = last_tail[r];
                if (send != nullptr && last_i == M - 1) {
#pragma unroll
                    for (int r = 0; r < K; r++) send[(int64_t)r * L + line] = last_tail[r];
                }
            }
            __syncthreads();
        }
        // scan s+1 chains on the tails just stored by other threads of this workgroup
        __threadfence_block();
        __syncthreads();
    }
}

}  // namespace

template <typename Acc>
int launch_carry_block(int K, const GenericDimArgs<Acc> &a, int s_begin, int s_end, const CarryResidual<Acc> &res,
                       Acc *send, const Acc *AC, int C, hipStream_t stream) {
    if (a.g.lines <= 0 || a.M <= 0 || s_end <= s_begin) return RF_OK;
    if (C < 1 || C > kCarryMaxC) { set_error("carry: chunk length %d out of range", C); return RF_ERR_INVALID_ARG; }
    const unsigned grid = (unsigned)((a.g.lines + kCarryLines - 1) / kCarryLines);
#define RF_CASE(KK) if (K == KK) { hipLaunchKernelGGL((carry_block_kernel<Acc, KK>), dim3(grid), dim3(256), 0, stream, a, s_begin, s_end, res, send, AC, C); RF_HIP_CHECK(hipGetLastError()); return RF_OK; }
    RF_CASE(1) RF_CASE(2) RF_CASE(3)
#undef RF_CASE
    set_error("carry: unsupported order %d", K);
    return RF_ERR_UNSUPPORTED;
}

int carry_chunk_length(int64_t M) {
    int64_t c = (M + kCarryChunks - 1) / kCarryChunks;
    return (int)(c < 1 ? 1 : (c > kCarryMaxC ? kCarryMaxC : c));
}

template int launch_carry_block<float>(int, const GenericDimArgs<float> &, int, int, const CarryResidual<float> &, float *,
                                       const float *, int, hipStream_t);
template int launch_carry_block<uint32_t>(int, const GenericDimArgs<uint32_t> &, int, int, const CarryResidual<uint32_t> &,
                                          uint32_t *, const uint32_t *, int, hipStream_t);

}  // namespace rf
