// mx_stream.hip -- gate for a STREAMING final pass of the matrix path (kernels_matrix.hip): does a wave that walks its 32 lines
// sub-block by sub-block (32 samples = 128 B per line at a time, next sub-block requested before this one's MFMAs, staging in
// wave-private LDS, 16 waves per CU) move the image faster than today's pass, which stages a whole 128 x 128 block per workgroup
// (two workgroups per CU: 0.58-0.65 ms per scan of 16384^2 = its MFMA time + its memory time, one after the other)?
//
// The kernel below has the data movement and the matrix-core load of such a pass without its meaning: per sub-block it loads
// 32 lines x 128 B (16 B per lane), transposes through LDS into the B-operand layout of v_mfma_f32_32x32x2f32, runs MF dependent
// MFMAs (40 = G 16 + R 8 + the next stage's H 16 at order 12; 24 without the hand-over; 0 = movement only), transposes the
// accumulator back and stores 128 B per line.
// Build: hipcc --offload-arch=gfx950 -O3 -o mx_stream mx_stream.hip ;  mx_stream [n=16384]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kPitch = 36;      // floats per staged line (32 samples + 4: 16-byte aligned rows)

__device__ __forceinline__ int mx_row(int t, int h) { return (t >> 2) * 8 + h * 4 + (t & 3); }

// WAVES waves per workgroup, each with its own 32 lines and its own LDS; T samples per line and workgroup (a tile); MF MFMAs per
// sub-block
template <int WAVES, int MF>
__global__ void __launch_bounds__(64 * WAVES) stream_kernel(const float *__restrict__ src, float *__restrict__ dst, const float *__restrict__ frag,
                                                            int n, int T) {
    __shared__ __attribute__((aligned(16))) float lds[WAVES][2][32 * kPitch];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *in_stage = lds[w][0], *out_stage = lds[w][1];
    const int tiles = n / T;
    const size_t line0 = ((size_t)(blockIdx.x / tiles) * WAVES + w) * 32;
    const size_t x0 = (size_t)(blockIdx.x % tiles) * T;
    const int lr = lane >> 3, lc = lane & 7;
    const float *g = src + (line0 + lr) * (size_t)n + x0 + 4 * lc;
    float *o = dst + (line0 + lr) * (size_t)n + x0 + 4 * lc;
    f4 pre[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pre[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(g + (size_t)(8 * i) * n));
    f16v acc = {0};
    const int NB = T / 32;
    for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<f4 *>(in_stage + (lr + 8 * i) * kPitch + 4 * lc) = pre[i];
        if (b + 1 < NB) {
#pragma unroll
            for (int i = 0; i < 4; i++) pre[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(g + (size_t)(8 * i) * n + 32 * (b + 1)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float xb[16];
#pragma unroll
        for (int t = 0; t < 16; t++) xb[t] = in_stage[(lane & 31) * kPitch + mx_row(t, lane >> 5)];
        if (MF == 0) {
#pragma unroll
            for (int t = 0; t < 16; t++) acc[t] = xb[t] + 1.0f;
        } else {
            f16v c = {0};
#pragma unroll
            for (int t = 0; t < 16; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[t * 64 + lane], xb[t], c, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; t++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[1024 + t * 64 + lane], acc[t], c, 0, 0, 0);
            if (MF > 24) {
                f16v d = {0};
#pragma unroll
                for (int t = 0; t < MF - 24; t++) d = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[2048 + t * 64 + lane], c[t], d, 0, 0, 0);
                c[0] += d[0] * 1e-30f;
            }
            acc = c;
        }
        // the accumulator (row = sample in mx order, column = line) back to lines of 128 B
#pragma unroll
        for (int r = 0; r < 16; r++) out_stage[(lane & 31) * kPitch + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(out_stage + (lr + 8 * i) * kPitch + 4 * lc), reinterpret_cast<f4 *>(o + (size_t)(8 * i) * n + 32 * b));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int WAVES, int MF>
int run(const char *name, const float *src, float *dst, const float *frag, int n, int T, int pad_lds = 0) {
    const int blocks = (n / (32 * WAVES)) * (n / T);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((stream_kernel<WAVES, MF>), dim3(blocks), dim3(64 * WAVES), pad_lds, 0, src, dst, frag, n, T);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((stream_kernel<WAVES, MF>), dim3(blocks), dim3(64 * WAVES), pad_lds, 0, src, dst, frag, n, T);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    if (pad_lds) std::printf("(+%d KiB of LDS per workgroup) ", pad_lds / 1024);
    std::printf("%-34s T %4d  waves/wg %d  %8.3f ms  %6.2f TB/s (8 B/sample)\n", name, T, WAVES, ms, (double)n * n * 8.0 / ms / 1e9);
    return 0;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    float *src, *dst, *frag;
    CK(hipMalloc(&src, (size_t)n * n * 4)); CK(hipMalloc(&dst, (size_t)n * n * 4)); CK(hipMalloc(&frag, 3 * 1024 * 4));
    CK(hipMemset(src, 0, (size_t)n * n * 4)); CK(hipMemset(frag, 0, 3 * 1024 * 4));
    std::printf("mx_stream: %d^2 f32, a wave walks 32 lines in sub-blocks of 32 samples (128 B per line)\n", n);
    // fewer waves per CU (an unused LDS allocation bounds the residency): 36 KiB per 4-wave workgroup -> 4 per CU; + 16 KiB -> 3; + 40 KiB -> 2
    for (int pad : {16 * 1024, 40 * 1024}) {
        if (run<4, 24>("+ 24 MFMAs per sub-block", src, dst, frag, n, 128, pad)) return 1;
        if (run<4, 40>("+ 40 MFMAs per sub-block", src, dst, frag, n, 128, pad)) return 1;
    }
    for (int T : {128, 256, 1024}) {
        if (run<1, 0>("movement only", src, dst, frag, n, T)) return 1;
        if (run<1, 24>("+ 24 MFMAs per sub-block", src, dst, frag, n, T)) return 1;
        if (run<1, 40>("+ 40 MFMAs per sub-block", src, dst, frag, n, T)) return 1;
        if (run<4, 0>("movement only", src, dst, frag, n, T)) return 1;
        if (run<4, 24>("+ 24 MFMAs per sub-block", src, dst, frag, n, T)) return 1;
        if (run<4, 40>("+ 40 MFMAs per sub-block", src, dst, frag, n, T)) return 1;
    }
    return 0;
}
