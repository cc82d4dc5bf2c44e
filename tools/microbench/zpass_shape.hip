// zpass_shape.hip -- the access shape of strided_pass_kernel's final z pass (kernels_strided.hip) as a register-resident copy:
// a thread owns one (x, y) line and TZ consecutive planes of it; which launch geometry moves a 2048 x 2048 x planes volume fastest?
//   W      lanes-per-line-run knob: threads per workgroup (256 = 1 KiB of every plane per workgroup, 512 = 2 KiB, 1024 = 4 KiB)
//   OCC    workgroups per CU (an unused LDS allocation bounds the residency, as the kernel does)
//   SPLIT  1: all TZ loads requested at once; 2: the column in two halves of TZ / 2 (half the bytes in flight per wave)
//   SWZ    0: workgroup b -> run b; 1: the 8 workgroups that share an XCD under round-robin placement take 8 adjacent runs in turn
//          (b -> (b % 8) * (runs / 8) + b / 8: every XCD streams its own eighth of the plane)
//   RECUR  1: a dependent chain up and down the column (a causal and an anticausal first-order recurrence), as the filter has
// Build: hipcc --offload-arch=gfx950 -O3 -o zpass_shape zpass_shape.hip ;  zpass_shape [n=2048] [planes=512]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W, int TZ, int SPLIT, int SWZ, int RECUR, bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(W) zcopy(const float *src, float *dst, long long plane, int runs, const float *tails = nullptr,
                                           long long plane_src = 0, long long plane_dst = 0) {      // (plane pitches; 0 = dense)
    int b = blockIdx.x;
    if (SWZ == 1) b = (b & 7) * (runs >> 3) + (b >> 3);
    const long long ps = plane_src ? plane_src : plane, pd = plane_dst ? plane_dst : plane;
    const long long base = (long long)blockIdx.y * TZ * ps + (long long)b * W, based = (long long)blockIdx.y * TZ * pd + (long long)b * W;
    const unsigned lane = threadIdx.x;
    float col[TZ];
    constexpr int H = TZ / SPLIT;
    float carry = 0.f;
    float c4[4] = {0.f, 0.f, 0.f, 0.f};
    if (RECUR == 2) {        // the kernel's carries: [scan 2][tile][k 2][line], plain loads behind the column's
        const long long line = (long long)b * W + lane, M = gridDim.y;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int r = 0; r < 2; r++) c4[2 * s + r] = tails[((s * M + blockIdx.y) * 2 + r) * plane + line];
    }
#pragma unroll
    for (int h = 0; h < SPLIT; h++) {
#pragma unroll
        for (int i = 0; i < H; i++) {
            const float *p = src + (base + (long long)(h * H + i) * ps) + lane;
            col[h * H + i] = NT_LD ? __builtin_nontemporal_load(p) : *p;
        }
        if (RECUR) {
#pragma unroll
            for (int i = 0; i < H; i++) { carry = col[h * H + i] + 0.5f * carry; col[h * H + i] = carry; }
        }
    }
    if (RECUR) {
        carry = c4[2] + c4[3] + c4[0] * c4[1];
#pragma unroll
        for (int i = TZ - 1; i >= 0; i--) { carry = col[i] + 0.5f * carry; col[i] = carry; }
    }
#pragma unroll
    for (int i = 0; i < TZ; i++) {
        float *q = dst + (based + (long long)i * pd) + lane;
        if (NT_ST) __builtin_nontemporal_store(col[i], q); else *q = col[i];
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2048, planes = argc > 2 ? atoi(argv[2]) : 512;
    const size_t plane = (size_t)n * n, elems = plane * planes;
    float *src, *dst;
    CK(hipMalloc(&src, elems * 4)); CK(hipMalloc(&dst, elems * 4));
    CK(hipMemset(src, 0, elems * 4)); CK(hipMemset(dst, 0, elems * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        float best = 1e30f, sum = 0.f;
        for (int i = 0; i < 6; i++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        std::printf("%-64s %8.3f ms (mean %8.3f)  %6.2f TB/s moved\n", name, best, sum / 6, elems * 8.0 / best * 1e-9);
        std::fflush(stdout);
    };
    std::printf("volume %d x %d x %d f32 (%.1f GiB in + out)\n", n, n, planes, elems * 8.0 / (1 << 30));
#define Z(W, TZ, SPLIT, SWZ, RECUR, NL, NS, OCC) [&] { \
        const int runs = (int)(plane / W); \
        size_t lds = OCC > 0 ? ((size_t)(160 * 1024 / OCC) & ~(size_t)1023) : 0; if (lds > 64 * 1024) lds = 64 * 1024; \
        hipLaunchKernelGGL((zcopy<W, TZ, SPLIT, SWZ, RECUR, NL, NS>), dim3(runs, planes / TZ), dim3(W), lds, 0, src, dst, (long long)plane, runs); }
    time("256 thr, 128 planes, nt ld+st, 3 wg/CU (the kernel's shape)", Z(256, 128, 1, 0, 0, true, true, 3));
    time("  + recurrences", Z(256, 128, 1, 0, 1, true, true, 3));
    {
        float *tails; CK(hipMalloc(&tails, (size_t)2 * (planes / 128) * 2 * plane * 4)); CK(hipMemset(tails, 0, (size_t)2 * (planes / 128) * 2 * plane * 4));
        time("  + recurrences + the carries (4 values per line and tile)", [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 2, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, dst, (long long)plane, (int)(plane / 256), tails); });
        time("  IN PLACE + recurrences + the carries", [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 2, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, src, (long long)plane, (int)(plane / 256), tails); });
        time("  IN PLACE", [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, src, (long long)plane, (int)(plane / 256), tails); });
        time("  IN PLACE, plain stores", [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, true, false>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, src, (long long)plane, (int)(plane / 256), tails); });
        time("  IN PLACE, plain loads and stores", [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, false, false>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, src, (long long)plane, (int)(plane / 256), tails); });
        time("  IN PLACE, 2 wg/CU", [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 64 * 1024, 0, src, src, (long long)plane, (int)(plane / 256), tails); });
        time("  IN PLACE, loads in two halves", [&] { hipLaunchKernelGGL((zcopy<256, 128, 2, 0, 1, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, src, (long long)plane, (int)(plane / 256), tails); });
        CK(hipFree(tails));
    }
    {
        // does the RELATIVE placement of the two volumes matter?  Both allocations are 2 MiB aligned, so a sample's source and
        // destination share their low address bits; dst2 + delta shifts the destination by delta bytes
        float *dst2; CK(hipMalloc(&dst2, elems * 4 + ((size_t)64 << 20))); CK(hipMemset(dst2, 0, elems * 4 + ((size_t)64 << 20)));
        const size_t deltas[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 2097152 + 4096, (size_t)33 << 20};
        for (size_t d : deltas) {
            char name[96]; std::snprintf(name, sizeof name, "  out of place, destination shifted by %zu bytes", d);
            float *q = dst2 + d / 4;
            time(name, [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, src, q, (long long)plane, (int)(plane / 256), nullptr); });
        }
        CK(hipFree(dst2));
    }
    {
        // does the PLANE PITCH matter?  Planes of 2048^2 floats lie exactly 16 MiB apart: the 128 rows a wave touches may share
        // their DRAM bank bits.  A padded copy of the volume (the plan-owned one between the stages could be laid out so)
        const long long pads[] = {0, 256, 2048, 16384, 131072};          // floats added to the plane pitch
        for (long long pad : pads) {
            const long long pp = (long long)plane + pad;
            float *psrc, *pdst;
            CK(hipMalloc(&psrc, (size_t)pp * planes * 4)); CK(hipMalloc(&pdst, (size_t)pp * planes * 4));
            CK(hipMemset(psrc, 0, (size_t)pp * planes * 4)); CK(hipMemset(pdst, 0, (size_t)pp * planes * 4));
            char name[112];
            std::snprintf(name, sizeof name, "  source pitch + %lld floats, destination dense", pad);
            time(name, [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, psrc, dst, (long long)plane, (int)(plane / 256), nullptr, pp, 0LL); });
            std::snprintf(name, sizeof name, "  source and destination pitch + %lld floats", pad);
            time(name, [&] { hipLaunchKernelGGL((zcopy<256, 128, 1, 0, 0, true, true>), dim3((unsigned)(plane / 256), planes / 128), dim3(256), 53 * 1024, 0, psrc, pdst, (long long)plane, (int)(plane / 256), nullptr, pp, pp); });
            CK(hipFree(psrc)); CK(hipFree(pdst));
        }
    }
    time("  2 wg/CU", Z(256, 128, 1, 0, 0, true, true, 2));
    time("  2 wg/CU + recurrences", Z(256, 128, 1, 0, 1, true, true, 2));
    time("  (64 KiB LDS cap = 2 wg/CU) 1 wg/CU asked", Z(256, 128, 1, 0, 0, true, true, 1));
    time("  4 wg/CU (register-bound: 3)", Z(256, 128, 1, 0, 0, true, true, 4));
    time("  XCD-contiguous runs, 3 wg/CU", Z(256, 128, 1, 1, 0, true, true, 3));
    time("  XCD-contiguous runs, 3 wg/CU + recurrences", Z(256, 128, 1, 1, 1, true, true, 3));
    time("  loads in two halves, 3 wg/CU", Z(256, 128, 2, 0, 0, true, true, 3));
    time("  loads in two halves + recurrences", Z(256, 128, 2, 0, 1, true, true, 3));
    time("  plain loads, nt stores", Z(256, 128, 1, 0, 0, false, true, 3));
    time("  nt loads, plain stores", Z(256, 128, 1, 0, 0, true, false, 3));
    time("512 thr (2 KiB per plane), 128 planes, 1 wg/CU asked", Z(512, 128, 1, 0, 0, true, true, 1));
    time("512 thr, 128 planes, 2 wg/CU", Z(512, 128, 1, 0, 0, true, true, 2));
    time("512 thr, 128 planes, 2 wg/CU + recurrences", Z(512, 128, 1, 0, 1, true, true, 2));
    time("1024 thr, 64 planes, 1 wg/CU", Z(1024, 64, 1, 0, 0, true, true, 1));
    time("1024 thr, 64 planes, 2 wg/CU", Z(1024, 64, 1, 0, 0, true, true, 2));
    time("64 thr (256 B per plane), 128 planes, 8 wg/CU", Z(64, 128, 1, 0, 0, true, true, 8));
    time("64 thr, 128 planes, XCD-contiguous, 8 wg/CU", Z(64, 128, 1, 1, 0, true, true, 8));
    time("128 thr, 128 planes, 6 wg/CU", Z(128, 128, 1, 0, 0, true, true, 6));
    return 0;
}
