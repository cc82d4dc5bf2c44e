// tile3d_pass.hip -- gate for a 3-D on-chip FINAL pass (VERDICT r4 "next" 4; /root/reference/lib/split.cpp:1647-1780 runs the
// scans of all dimensions in one Final pass over a 3-D tile): the data movement of such a pass without its arithmetic.
//
// A workgroup owns a tile of 64 x 32 x 32 samples (256 KiB: registers, with LDS as the transposition stage -- the whole tile has
// to be on chip before the anticausal z scan can finish).  It loads the tile (rows of 256 B), the carries entering it through
// its three pairs of faces (x: 4 floats per (y, z) row, y: 4 per (x, z), z: 4 per (x, y) -- two scans of order 2 per dimension:
// 80 KiB, 31 % of the tile), sends the tile twice through LDS in another thread mapping (x <-> y, y <-> z: what lets one thread
// walk a line of every dimension), mixes the carries in, and stores the tile.  HBM bytes per sample: 4 + 4 + 1.25.
// If this does not stream at >= 5.5 TB/s of its own traffic, a real kernel -- the same movement plus three pairs of
// recurrences -- cannot beat the 22.8 B/sample of the two final passes it would replace (17.3 B/sample at that rate).
// Build: hipcc --offload-arch=gfx950 -O3 -o tile3d_pass tile3d_pass.hip ;  tile3d_pass [n=1024]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TX = 64, TY = 32, TZ = 32;
constexpr int kChunks = TX * TY * TZ / 4;          // 16384 chunks of 16 bytes per tile
constexpr int kPitch = TX + 4;                     // LDS row pitch in floats (16-byte aligned, conflict-free column reads)

// THREADS: 1024 (16 chunks per thread) or 512 (32 chunks per thread, two workgroups per CU); XPOSE: LDS round trips
template <int THREADS, int XPOSE>
__global__ void __launch_bounds__(THREADS) tile3d_kernel(const float *__restrict__ src, float *__restrict__ dst, const f4 *__restrict__ xt,
                                                         const f4 *__restrict__ yt, const f4 *__restrict__ zt, int n) {
    constexpr int PER = kChunks / THREADS;          // chunks per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const size_t x0 = (size_t)blockIdx.x * TX, y0 = (size_t)blockIdx.y * TY, z0 = (size_t)blockIdx.z * TZ;
    const size_t tile = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    f4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int f = t + THREADS * i, pl = f >> 9, row = (f & 511) >> 4, c4 = f & 15;
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(src + ((z0 + pl) * n + y0 + row) * (size_t)n + x0 + 4 * c4));
    }
    // carries through the faces: 1024 + 2048 + 2048 chunks per tile
    f4 cx = f4{0, 0, 0, 0}, cy = f4{0, 0, 0, 0}, cz = f4{0, 0, 0, 0};
    for (int f = t; f < 1024; f += THREADS) cx += xt[tile * 1024 + f];
    for (int f = t; f < 2048; f += THREADS) cy += yt[tile * 2048 + f];
    for (int f = t; f < 2048; f += THREADS) cz += zt[tile * 2048 + f];
    // the tile through LDS in another thread mapping, half a tile (16 planes x 32 rows x 68 floats = 136 KiB) at a time for
    // 1024 threads; a quarter for 512 threads (two workgroups per CU)
    constexpr int HALVES = THREADS == 1024 ? 2 : 4;
    constexpr int PERH = PER / HALVES * (THREADS == 1024 ? 1 : 1);
#pragma unroll
    for (int round = 0; round < XPOSE; round++) {
#pragma unroll
        for (int hh = 0; hh < HALVES; hh++) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PERH; i++) {
                const int f = t + THREADS * i, pl = f >> 9, row = (f & 511) >> 4, c4 = f & 15;
                *reinterpret_cast<f4 *>(lds + (pl * TY + row) * kPitch + 4 * c4) = v[hh * PERH + i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PERH; i++) {
                // transposed: four rows of one column
                const int f = t + THREADS * i, pl = f >> 9, col = f & 63, r4 = (f & 511) >> 6;
                f4 w;
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = lds[(pl * TY + 4 * r4 + e) * kPitch + col];
                v[hh * PERH + i] = w * cx + cy;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int f = t + THREADS * i, pl = f >> 9, row = (f & 511) >> 4, c4 = f & 15;
        __builtin_nontemporal_store(v[i] + cz, reinterpret_cast<f4 *>(dst + ((z0 + pl) * n + y0 + row) * (size_t)n + x0 + 4 * c4));
    }
}

template <int THREADS, int XPOSE>
int run(const char *name, const float *src, float *dst, const f4 *xt, const f4 *yt, const f4 *zt, int n) {
    dim3 grid(n / TX, n / TY, n / TZ);
    const size_t lds = (size_t)(THREADS == 1024 ? 16 : 8) * TY * kPitch * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(tile3d_kernel<THREADS, XPOSE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((tile3d_kernel<THREADS, XPOSE>), grid, dim3(THREADS), lds, 0, src, dst, xt, yt, zt, n);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((tile3d_kernel<THREADS, XPOSE>), grid, dim3(THREADS), lds, 0, src, dst, xt, yt, zt, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double samples = (double)n * n * n, bytes = samples * (4 + 4 + 1.25);
    std::printf("%-44s %8.3f ms   %6.2f TB/s of its own traffic (9.25 B/sample)   %6.1f Gsamples/s\n", name, ms, bytes / ms / 1e9, samples / ms / 1e6);
    return 0;
}

// the plain copy in the same tile order, for scale
__global__ void __launch_bounds__(1024) copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int n) {
    const int t = threadIdx.x;
    const size_t x0 = (size_t)blockIdx.x * TX, y0 = (size_t)blockIdx.y * TY, z0 = (size_t)blockIdx.z * TZ;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int f = t + 1024 * i, pl = f >> 9, row = (f & 511) >> 4, c4 = f & 15;
        const size_t o = ((z0 + pl) * n + y0 + row) * (size_t)n + x0 + 4 * c4;
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f4 *>(src + o)), reinterpret_cast<f4 *>(dst + o));
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t samples = (size_t)n * n * n, tiles = samples / (TX * TY * TZ);
    float *src, *dst;
    f4 *xt, *yt, *zt;
    CK(hipMalloc(&src, samples * 4)); CK(hipMalloc(&dst, samples * 4));
    CK(hipMalloc(&xt, tiles * 1024 * 16)); CK(hipMalloc(&yt, tiles * 2048 * 16)); CK(hipMalloc(&zt, tiles * 2048 * 16));
    CK(hipMemset(src, 0, samples * 4)); CK(hipMemset(xt, 0, tiles * 1024 * 16)); CK(hipMemset(yt, 0, tiles * 2048 * 16)); CK(hipMemset(zt, 0, tiles * 2048 * 16));
    std::printf("tile3d_pass: %d^3 f32, tiles of %d x %d x %d, carries through the faces 31 %% of the volume\n", n, TX, TY, TZ);
    {
        dim3 grid(n / TX, n / TY, n / TZ);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(copy_kernel, grid, dim3(1024), 0, 0, src, dst, n);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(copy_kernel, grid, dim3(1024), 0, 0, src, dst, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        std::printf("%-44s %8.3f ms   %6.2f TB/s (8 B/sample)\n", "copy in the tile's access shape", ms, samples * 8.0 / ms / 1e9);
    }
    if (run<1024, 0>("load + carries + store, 1024 threads", src, dst, xt, yt, zt, n)) return 1;
    if (run<1024, 2>("... + two LDS transpositions, 1024 threads", src, dst, xt, yt, zt, n)) return 1;
    if (run<512, 0>("load + carries + store, 2 x 512 threads", src, dst, xt, yt, zt, n)) return 1;
    if (run<512, 2>("... + two LDS transpositions, 2 x 512 threads", src, dst, xt, yt, zt, n)) return 1;
    return 0;
}
