// walk_read.hip -- the access shape of walk_tails_kernel (kernels_tails_walk.hip) as a bare read: a workgroup of 1024 threads owns a
// patch of 256 x 32 floats and walks `tz` planes of an n x n x planes volume, D planes of its loads in flight; optionally one
// workgroup barrier per plane (the kernel's lockstep) and the kernel's stores (4 KiB of parts + 512 B of x tails per plane).
// Build: hipcc --offload-arch=gfx950 -O3 -o walk_read walk_read.hip ;  walk_read [n=2048] [planes=512]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// TALL: the patch is 128 columns x 64 rows (grid n / 128 x n / 64) instead of 256 x 32: rows of 512 B, half the combined-row parts
template <int D, bool BARRIER, int STORES, bool TALL = false>     // STORES: 0 none, 1 per patch column [column][plane][288], 2 plane-major [plane][column][288]
__global__ void __launch_bounds__(1024) walk_kernel(const float *src, float *out, f4 *parts, int n, int tz) {
    const int t = threadIdx.x, cc = TALL ? t & 31 : t & 63, rg = TALL ? t >> 5 : t >> 6;
    const int tx = blockIdx.x, py = blockIdx.y, zt = blockIdx.z;
    const size_t plane = (size_t)n * n;
    const char *spb = reinterpret_cast<const char *>(src + (size_t)zt * tz * plane + ((size_t)py * (TALL ? 64 : 32)) * n + (size_t)tx * (TALL ? 128 : 256));
    const unsigned off0 = (unsigned)rg * n * 4u + cc * 16u, off1 = off0 + (TALL ? 32u : 16u) * n * 4u;
    f4 pre[D][2];
#pragma unroll
    for (int d = 0; d < D; d++) {
        pre[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(spb + d * plane * 4 + off0));
        pre[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(spb + d * plane * 4 + off1));
        __builtin_amdgcn_sched_barrier(0);
    }
    f4 acc = f4{0, 0, 0, 0};
    f4 *pp = parts + (((size_t)zt * gridDim.y + py) * gridDim.x + tx) * (size_t)tz * 288;      // 256 + 32 chunks per plane
#pragma unroll 1
    for (int z0 = 0; z0 < tz; z0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int z = z0 + d;
            acc += pre[d][0] + pre[d][1];
            if (z + D < tz) {
                pre[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(spb + (size_t)(z + D) * plane * 4 + off0));
                pre[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(spb + (size_t)(z + D) * plane * 4 + off1));
            }
            if (STORES == 1 && t < 288) pp[(size_t)z * 288 + t] = acc;
            if (STORES == 2 && t < 288) parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t] = acc;
            if (STORES == 4 && t < 288) parts[(size_t)(blockIdx.x & 7) * 288 + t] = acc;            // the same few lines over and over: no HBM writes
            if (STORES == 5 && t < 288) parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t] = f4{1.f, 2.f, 3.f, 4.f};   // no dependence on the loads
            if (STORES == 6 && (z & 7) == 7) {          // eight planes' worth at once: 36 KiB per workgroup
                f4 *q = parts + ((((size_t)zt * (tz / 8) + (z >> 3)) * gridDim.y + py) * gridDim.x + tx) * 2304;
                for (int i = t; i < 2304; i += 1024) q[i] = acc;
            }
            if (STORES == 7 && (z & 31) == 31) {        // thirty-two planes' worth at once: 144 KiB per workgroup
                f4 *q = parts + ((((size_t)zt * (tz / 32) + (z >> 5)) * gridDim.y + py) * gridDim.x + tx) * 9216;
                for (int i = t; i < 9216; i += 1024) q[i] = acc;
            }
            if (STORES == 12 && (z & 7) == 7) {         // TALL: eight planes' worth of 3 KiB at once: 24 KiB per workgroup
                f4 *q = parts + ((((size_t)zt * (tz / 8) + (z >> 3)) * gridDim.y + py) * gridDim.x + tx) * 1536;
                for (int i = t; i < 1536; i += 1024) q[i] = acc;
            }
            if (STORES == 13 && (z & 31) == 31) {       // TALL: thirty-two planes' worth at once: 96 KiB per workgroup
                f4 *q = parts + ((((size_t)zt * (tz / 32) + (z >> 5)) * gridDim.y + py) * gridDim.x + tx) * 6144;
                for (int i = t; i < 6144; i += 1024) q[i] = acc;
            }
            if (STORES == 8 && t < 32) parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t] = acc;      // 512 B per plane
            if (STORES == 11 && t < 192) parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t] = acc;     // 3 KiB per plane
            if (STORES == 9 && t < 128) parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t] = acc;     // 2 KiB per plane
            if (STORES == 10) {                                                                                                 // a copy: 32 KiB per plane
                f4 *q = reinterpret_cast<f4 *>(out) + ((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 2048;
                q[t] = acc; q[t + 1024] = acc;
            }
            if (STORES >= 20 && t < 288) {           // the same 4.6 KiB, plane-major, under other cache policies (gfx950 sc0 / sc1 / nt bits)
                f4 *q = &parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t];
                if (STORES == 20) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(q), "v"(acc) : "memory");
                if (STORES == 21) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(acc) : "memory");
                if (STORES == 22) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(acc) : "memory");
                if (STORES == 23) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(acc) : "memory");
                if (STORES == 24) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(q), "v"(acc) : "memory");
            }
            if (STORES == 3 && t < 288) __builtin_nontemporal_store(acc, &parts[((((size_t)zt * tz + z) * gridDim.y + py) * gridDim.x + tx) * 288 + t]);
            if (BARRIER) __syncthreads();
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 1024 + t] = acc.x;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2048, planes = argc > 2 ? atoi(argv[2]) : 512, tz = 128;
    const size_t elems = (size_t)n * n * planes;
    float *src, *out; f4 *parts;
    CK(hipMalloc(&src, elems * 4)); CK(hipMalloc(&out, elems * 4 + ((size_t)1 << 24)));
    CK(hipMalloc(&parts, elems / 8192 * 288 * 16));                    // 288 chunks per patch and plane
    CK(hipMemset(src, 0, elems * 4));
    dim3 grid(n / 256, n / 32, planes / tz);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        float best = 1e30f;
        for (int i = 0; i < 5; i++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        std::printf("%-52s %8.3f ms  %6.2f TB/s read\n", name, best, elems * 4.0 / best * 1e-9);
    };
    std::printf("volume %d x %d x %d f32 (%.1f GiB), %u patch columns of 256 x 32 x %d\n", n, n, planes, elems * 4.0 / (1 << 30), grid.x * grid.y * grid.z, tz);
#define L(D, B, S) [&] { hipLaunchKernelGGL((walk_kernel<D, B, S>), grid, dim3(1024), 0, 0, src, out, parts, n, tz); }
    time("2 planes in flight", L(2, false, 0));
    time("4 planes in flight", L(4, false, 0));
    time("8 planes in flight", L(8, false, 0));
    time("2 planes in flight, barrier per plane", L(2, true, 0));
    time("4 planes in flight, barrier per plane", L(4, true, 0));
    time("2 in flight, barrier, stores per patch column", L(2, true, 1));
    time("2 in flight, barrier, stores plane-major", L(2, true, 2));
    time("2 in flight, barrier, stores plane-major nt", L(2, true, 3));
    time("2 in flight, stores plane-major", L(2, false, 2));
    time("2 in flight, barrier, stores to 8 cached blocks", L(2, true, 4));
    time("2 in flight, barrier, constant stores plane-major", L(2, true, 5));
    time("2 in flight, barrier, 512 B of stores per plane", L(2, true, 8));
    time("2 in flight, barrier, 2 KiB of stores per plane", L(2, true, 9));
    time("2 in flight, barrier, 32 KiB of stores per plane (copy)", L(2, true, 10));
    {
        dim3 gt(n / 128, n / 64, planes / tz);
#define LT(D, B, S) [&] { hipLaunchKernelGGL((walk_kernel<D, B, S, true>), gt, dim3(1024), 0, 0, src, out, parts, n, tz); }
        time("128 x 64 patches: 2 in flight, barrier", LT(2, true, 0));
        time("128 x 64 patches: 2 in flight, barrier, 3 KiB of stores per plane", LT(2, true, 11));
        time("128 x 64 patches: 2 in flight, barrier, 4.6 KiB of stores per plane", LT(2, true, 2));
        time("128 x 64 patches: 3 KiB x 8 planes stored at once", LT(2, true, 12));
        time("128 x 64 patches: 3 KiB x 32 planes stored at once", LT(2, true, 13));
        time("128 x 64 patches: 2 in flight, barrier, 3 KiB of stores per plane (again)", LT(2, true, 11));
        time("256 x 32 patches: 2 in flight, barrier, 3 KiB of stores per plane", L(2, true, 11));
        time("256 x 32 patches: 2 in flight, barrier, 4.6 KiB (again)", L(2, true, 2));
        time("256 x 32 patches: 2 in flight, barrier, no stores (again)", L(2, true, 0));
    }
    time("2 in flight, barrier, stores sc0", L(2, true, 20));
    time("2 in flight, barrier, stores sc1", L(2, true, 21));
    time("2 in flight, barrier, stores sc0 sc1", L(2, true, 22));
    time("2 in flight, barrier, stores sc0 sc1 nt", L(2, true, 23));
    time("2 in flight, barrier, stores sc1 nt", L(2, true, 24));
    time("2 in flight, barrier, stores of 8 planes at once", L(2, true, 6));
    time("2 in flight, barrier, stores of 32 planes at once", L(2, true, 7));
    return 0;
}
