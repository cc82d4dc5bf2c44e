// mall_share.hip -- can two readers of the SAME bytes, launched side by side, share one trip to HBM through the Infinity
// Cache?  Question behind it (cfg5): pass 1 of the x/y stage reads a volume plane by plane (256 x 128 tiles), pass 1 of the
// z stage reads it as runs of 256 lanes over 128 planes; both read the raw input when the z operators are commuted in front
// of the x/y filter (plan_strided.h, early form).  If the workgroups of both are interleaved region by region (region = one
// 256 x 128 tile position over a z tile of 128 planes, 16 MiB: 128 tile workgroups + 128 row workgroups), the second
// reader of a line finds it on die.
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_share mall_share.hip
//   mall_share [planes=512] [mode]      volume 2048 x 2048 x planes floats
// Rows printed: each role alone, both roles on DIFFERENT buffers (no sharing possible), both on the SAME buffer interleaved,
// both on the same buffer one after the other as two launches (what the plan does today).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NX = 2048, NY = 2048, TY = 128, TZ = 128;
constexpr int MX = NX / 256, MY = NY / TY;

// role 0: the 256 x 128 tile (tx, ty) of plane p, 32-row steps
template <bool NT>
__device__ __forceinline__ float tile_role(const float *src, int tx, int ty, int64_t p) {
    const int cc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const f4 *s4 = reinterpret_cast<const f4 *>(src + p * NX * NY + ((int64_t)ty * TY + rg) * NX + tx * 256) + cc;
    float s = 0;
#pragma unroll 1
    for (int h = 0; h < TY / 32; h++) {
        f4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const f4 *q = s4 + (int64_t)(32 * h + 4 * i) * (NX / 4);
            v[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    return s;
}

// role 1: row y of the tile column tx (256 lanes), the 128 planes of z tile tz
template <bool NT>
__device__ __forceinline__ float line_role(const float *src, int tx, int y, int tz) {
    const float *q = src + ((int64_t)tz * TZ) * NX * NY + (int64_t)y * NX + tx * 256 + threadIdx.x;
    float s = 0;
#pragma unroll 1
    for (int h = 0; h < TZ / 32; h++) {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const float *qq = q + (int64_t)(32 * h + i) * NX * NY;
            v[i] = NT ? __builtin_nontemporal_load(qq) : *qq;
        }
#pragma unroll
        for (int i = 0; i < 32; i++) s += v[i];
    }
    return s;
}

// roles: 1 tiles only, 2 lines only, 3 both (interleaved inside every region); GROUP: consecutive workgroups of one role
template <bool NT, int GROUP>
__global__ void __launch_bounds__(256, 4) dual_kernel(const float *a, const float *b, float *out, int roles) {
    int blk = blockIdx.x;
    const int per_region = roles == 3 ? 256 : 128;
    const int region = blk / per_region;
    int j = blk % per_region;
    int role = roles == 3 ? (j / GROUP) & 1 : roles - 1;
    const int i = roles == 3 ? (j / (2 * GROUP)) * GROUP + j % GROUP : j;
    const int tx = region % MX, ty = (region / MX) % MY, tz = region / (MX * MY);
    float s;
    if (role == 0) s = tile_role<NT>(a, tx, ty, (int64_t)tz * TZ + i);
    else s = line_role<NT>(b, tx, ty * TY + i, tz);
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int planes = argc > 1 ? atoi(argv[1]) : 512;
    const int MZ = planes / TZ;
    const size_t n = (size_t)NX * NY * planes;
    float *a, *b, *out;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&out, (size_t)1 << 26));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    const int regions = MX * MY * MZ;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch, double bytes) {
        for (int i = 0; i < 2; i++) launch();
        float best = 1e30f, sum = 0;
        for (int i = 0; i < 6; i++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        std::printf("%-58s %8.3f ms (mean %8.3f)  %6.2f TB/s requested\n", name, best, sum / 6, bytes / best * 1e-9);
        return 0;
    };
    const double vb = (double)n * 4;
#define L(NT, G, A, B, R) [&] { hipLaunchKernelGGL((dual_kernel<NT, G>), dim3(regions * ((R) == 3 ? 256 : 128)), dim3(256), 0, 0, A, B, out, R); }
    std::printf("volume %d x %d x %d f32 = %.2f GiB, %d regions of 16 MiB\n", NX, NY, planes, vb / (1 << 30), regions);
    time("tiles alone (nt)", L(true, 1, a, a, 1), vb);
    time("tiles alone (default policy)", L(false, 1, a, a, 1), vb);
    time("lines alone (nt)", L(true, 1, a, a, 2), vb);
    time("lines alone (default policy)", L(false, 1, a, a, 2), vb);
    time("two launches, same buffer (nt)", [&] { L(true, 1, a, a, 1)(); L(true, 1, a, a, 2)(); }, 2 * vb);
    time("interleaved, different buffers (nt)", L(true, 1, a, b, 3), 2 * vb);
    time("interleaved, different buffers (default)", L(false, 1, a, b, 3), 2 * vb);
    time("interleaved, SAME buffer (nt)", L(true, 1, a, a, 3), 2 * vb);
    time("interleaved, SAME buffer (default)", L(false, 1, a, a, 3), 2 * vb);
    time("interleaved in groups of 8, SAME buffer (default)", L(false, 8, a, a, 3), 2 * vb);
    time("interleaved in groups of 32, SAME buffer (default)", L(false, 32, a, a, 3), 2 * vb);
    time("interleaved in groups of 128, SAME buffer (default)", L(false, 128, a, a, 3), 2 * vb);
    time("interleaved in groups of 128, SAME buffer (nt)", L(true, 128, a, a, 3), 2 * vb);
    return 0;
}
