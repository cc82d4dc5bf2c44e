// hbm_read_patterns.hip -- read-only streaming rates of one MI355X for the access shapes the tile kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_read_patterns hbm_read_patterns.hip ; run on a GPU box.
//   linear   : every workgroup reads one contiguous 64 KiB block (16 B per lane, 16 loads per thread)
//   tile64   : every workgroup reads a 256-float x 64-row tile of a 16384-wide image (1 KiB rows, 64 KiB apart)
//   tile32x2 : same tile as two 32-row halves, second half requested while the first is consumed (fused_tails)
//   segment  : same tile in the x-phase register layout (64 B per lane and row, four loads of 16 B)
// The loaded values are summed into one float per thread and written once (negligible traffic).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) linear_kernel(const f4 *src, float *out) {
    const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = src[base + 256 * i];
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ROWS_PER_STEP>
__global__ void __launch_bounds__(256) tile_kernel(const f4 *src, float *out, int nx4, int mx) {
    const int tx = blockIdx.x % mx, ty = blockIdx.x / mx;
    const int cc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const size_t base = ((size_t)ty * 64 + rg) * nx4 + (size_t)tx * 64 + cc;
    float s = 0;
#pragma unroll
    for (int step = 0; step < 64 / ROWS_PER_STEP; step++) {
        f4 v[ROWS_PER_STEP / 4];
#pragma unroll
        for (int i = 0; i < ROWS_PER_STEP / 4; i++) v[i] = src[base + (size_t)(step * ROWS_PER_STEP + 4 * i) * nx4];
#pragma unroll
        for (int i = 0; i < ROWS_PER_STEP / 4; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// tile rows owned by consecutive waves instead of interleaved (wave w reads rows 16w .. 16w+15)
__global__ void __launch_bounds__(256) tile_rows_per_wave_kernel(const f4 *src, float *out, int nx4, int mx) {
    const int tx = blockIdx.x % mx, ty = blockIdx.x / mx;
    const int cc = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t base = ((size_t)ty * 64 + 16 * w) * nx4 + (size_t)tx * 64 + cc;
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = src[base + (size_t)i * nx4];
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// a workgroup reads a 1024-float x 16-row tile (4 KiB rows)
__global__ void __launch_bounds__(256) wide_tile_kernel(const f4 *src, float *out, int nx4, int mx) {
    const int tx = blockIdx.x % mx, ty = blockIdx.x / mx;
    const size_t base = ((size_t)ty * 16) * nx4 + (size_t)tx * 256 + threadIdx.x;
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = src[base + (size_t)i * nx4];
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the x-phase register layout loaded straight from memory: thread = (row slot, 16-sample segment), four 16-byte loads
// per row with a 64-byte stride between lanes (every load instruction touches a quarter of each 64-byte piece)
__global__ void __launch_bounds__(256) segment_kernel(const f4 *src, float *out, int nx4, int mx) {
    const int tx = blockIdx.x % mx, ty = blockIdx.x / mx;
    const int l = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const size_t base = ((size_t)ty * 64 + slot) * nx4 + (size_t)tx * 64 + 4 * l;
    f4 v[16];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int j = 0; j < 4; j++) v[4 * n + j] = src[base + (size_t)(16 * n) * nx4 + j];
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// tile-shaped copy (what pass 2 moves): 16-byte loads of a 256 x 64 tile, 4-byte column stores like the kernel's, with
// and without the non-temporal hint on either side
// MAP: 0 = tiles in launch order (x fastest); 1 = every XCD (launch index mod 8) gets a contiguous eighth of the tiles;
// 2 = y fastest (consecutive workgroups walk down a tile column); 3 = 8 x 8 blocks of tiles per group of 64 workgroups
template <bool NT_LD, bool NT_ST, int MAP = 0>
__global__ void __launch_bounds__(256) tile_copy_kernel(const f4 *src, float *dst, int nx4, int mx) {
    __shared__ f4 tile[64 * 64];
    int b = blockIdx.x;
    const int total = gridDim.x, my = total / mx;
    int tx, ty;
    if (MAP == 1) b = (b & 7) * (total >> 3) + (b >> 3);
    if (MAP == 2) { ty = b % my; tx = b / my; }
    else if (MAP == 3) { const int g = b >> 6, i = b & 63; const int gx = g % (mx >> 3), gy = g / (mx >> 3); tx = gx * 8 + (i & 7); ty = gy * 8 + (i >> 3); }
    else { tx = b % mx; ty = b / mx; }
    const int cc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const size_t base = ((size_t)ty * 64 + rg) * nx4 + (size_t)tx * 64 + cc;
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = NT_LD ? __builtin_nontemporal_load(src + base + (size_t)(4 * i) * nx4) : src[base + (size_t)(4 * i) * nx4];
#pragma unroll
    for (int i = 0; i < 16; i++) tile[(rg + 4 * i) * 64 + (cc ^ ((rg + 4 * i) & 63))] = v[i];
    __syncthreads();
    const float *tf = reinterpret_cast<const float *>(tile);
    const int t = threadIdx.x;
    float *dp = dst + ((size_t)ty * 64) * (size_t)(4 * nx4) + (size_t)tx * 256 + t;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const float val = tf[i * 256 + ((((t >> 2) ^ (i & 63)) << 2) | (t & 3))];
        if (NT_ST) __builtin_nontemporal_store(val, dp + (size_t)i * (size_t)(4 * nx4));
        else dp[(size_t)i * (size_t)(4 * nx4)] = val;
    }
}

// plain streaming copy, 16 bytes per lane both ways
template <bool NT>
__global__ void __launch_bounds__(256) linear_copy_kernel(const f4 *src, f4 *dst) {
    const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = NT ? __builtin_nontemporal_load(src + base + 256 * i) : src[base + 256 * i];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (NT) __builtin_nontemporal_store(v[i], dst + base + 256 * i);
        else dst[base + 256 * i] = v[i];
    }
}

// `hbm_read_patterns pitch <width>`: the tile-shaped copy on an image of 16384 rows whose width (a multiple of 4) is not a
// multiple of 32 samples -- rows are 16-byte aligned but start anywhere inside a 128-byte line; whole tiles only.
int pitch_main(int width) {
    const int n = 16384;
    const int nx4 = width / 4, mx = width / 256, tiles = mx * (n / 64);
    const size_t bytes = (size_t)n * width * 4, moved = (size_t)n * mx * 256 * 4;
    f4 *src; float *dst;
    CK(hipMalloc(&src, bytes + 4096));
    CK(hipMalloc(&dst, bytes + 4096));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time2 = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; i++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        std::printf("width %5d  %-34s %.4f ms  %.2f TB/s\n", width, name, ms, 2.0 * moved / (ms * 1e-3) / 1e12);
    };
    float *out;
    CK(hipMalloc(&out, (size_t)tiles * 256 * 4));
    time2("tile read only (x2 for the rate)", [&] { hipLaunchKernelGGL(tile_kernel<64>, dim3(tiles), dim3(256), 0, 0, src, out, nx4, mx); });
    time2("tile copy", [&] { hipLaunchKernelGGL((tile_copy_kernel<false, false>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy, nt loads", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, false>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy, nt both", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    if (tiles % 8 == 0) {
        time2("tile copy nt, XCD-contiguous", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true, 1>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
        time2("tile copy nt loads, XCD-contig", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, false, 1>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    }
    return 0;
}

// `hbm_read_patterns zcopy <n>`: what the strided z pass moves, as a bare copy of an n^3 volume: thread = one (x, y) line, TZ
// samples a plane apart in registers, 256-byte runs per wave and plane -- with TZ = 64 / 128 and 4 or 8 bytes per lane.
template <int TZ, typename V, bool NT>
__global__ void __launch_bounds__(256) zcopy_kernel(const V *src, V *dst, size_t inner, size_t lines) {
    const size_t line = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= lines) return;
    const size_t base = line + (size_t)blockIdx.y * TZ * inner;
    V col[TZ];
#pragma unroll
    for (int i = 0; i < TZ; i++) col[i] = NT ? __builtin_nontemporal_load(src + base + (size_t)i * inner) : src[base + (size_t)i * inner];
#pragma unroll
    for (int i = 0; i < TZ; i++) {
        if (NT) __builtin_nontemporal_store(col[i], dst + base + (size_t)i * inner);
        else dst[base + (size_t)i * inner] = col[i];
    }
}

int zcopy_main(int n) {
    const size_t total = (size_t)n * n * n, bytes = total * 4;
    float *src, *dst;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time2 = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; i++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        std::printf("%d^3  %-40s %.3f ms  %.2f TB/s\n", n, name, ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
    };
    const size_t inner = (size_t)n * n;
    typedef float f2 __attribute__((ext_vector_type(2)));
    time2("z copy, 128 planes, 4 B per lane, nt", [&] { hipLaunchKernelGGL((zcopy_kernel<128, float, true>), dim3(inner / 256, n / 128), dim3(256), 0, 0, src, dst, inner, inner); });
    time2("z copy, 128 planes, 4 B per lane", [&] { hipLaunchKernelGGL((zcopy_kernel<128, float, false>), dim3(inner / 256, n / 128), dim3(256), 0, 0, src, dst, inner, inner); });
    time2("z copy, 64 planes, 4 B per lane, nt", [&] { hipLaunchKernelGGL((zcopy_kernel<64, float, true>), dim3(inner / 256, n / 64), dim3(256), 0, 0, src, dst, inner, inner); });
    time2("z copy, 64 planes, 8 B per lane, nt", [&] { hipLaunchKernelGGL((zcopy_kernel<64, f2, true>), dim3(inner / 512, n / 64), dim3(256), 0, 0, (const f2 *)src, (f2 *)dst, inner / 2, inner / 2); });
    time2("z copy, 32 planes, 16 B per lane, nt", [&] { hipLaunchKernelGGL((zcopy_kernel<32, f4, true>), dim3(inner / 1024, n / 32), dim3(256), 0, 0, (const f4 *)src, (f4 *)dst, inner / 4, inner / 4); });
    time2("linear copy, nt", [&] { hipLaunchKernelGGL(linear_copy_kernel<true>, dim3((unsigned)(total / 16384)), dim3(256), 0, 0, (const f4 *)src, (f4 *)dst); });
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && std::string(argv[1]) == "zcopy") return zcopy_main(atoi(argv[2]));
    if (argc >= 3 && std::string(argv[1]) == "pitch") {
        for (int i = 2; i < argc; i++)
            if (int rc = pitch_main(atoi(argv[i]))) return rc;
        return 0;
    }
    const int n = 16384;
    const size_t bytes = (size_t)n * n * 4;
    f4 *src; float *out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, (size_t)16384 * 256 * 4));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nx4 = n / 4, mx = n / 256, tiles = mx * (n / 64);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; i++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        std::printf("%-22s %.4f ms  %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    time("linear 64 KiB/WG", [&] { hipLaunchKernelGGL(linear_kernel, dim3(tiles), dim3(256), 0, 0, src, out); });
    time("tile 256x64, 16 loads", [&] { hipLaunchKernelGGL(tile_kernel<64>, dim3(tiles), dim3(256), 0, 0, src, out, nx4, mx); });
    time("tile 256x64, 2x8 loads", [&] { hipLaunchKernelGGL(tile_kernel<32>, dim3(tiles), dim3(256), 0, 0, src, out, nx4, mx); });
    time("tile, rows per wave", [&] { hipLaunchKernelGGL(tile_rows_per_wave_kernel, dim3(tiles), dim3(256), 0, 0, src, out, nx4, mx); });
    time("tile 1024x16", [&] { hipLaunchKernelGGL(wide_tile_kernel, dim3(tiles), dim3(256), 0, 0, src, out, nx4, n / 1024); });
    time("tile, segment layout", [&] { hipLaunchKernelGGL(segment_kernel, dim3(tiles), dim3(256), 0, 0, src, out, nx4, mx); });
    float *dst_base, *dst;
    CK(hipMalloc(&dst_base, bytes + (64u << 20)));
    dst = dst_base;
    std::printf("copies (rates count read + write bytes):\n");
    auto time2 = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; i++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        std::printf("%-30s %.4f ms  %.2f TB/s\n", name, ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
    };
    time2("linear copy", [&] { hipLaunchKernelGGL(linear_copy_kernel<false>, dim3(tiles), dim3(256), 0, 0, src, (f4 *)dst); });
    time2("linear copy, nt", [&] { hipLaunchKernelGGL(linear_copy_kernel<true>, dim3(tiles), dim3(256), 0, 0, src, (f4 *)dst); });
    time2("tile copy", [&] { hipLaunchKernelGGL((tile_copy_kernel<false, false>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy, nt loads", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, false>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy, nt stores", [&] { hipLaunchKernelGGL((tile_copy_kernel<false, true>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy, nt both", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy nt, XCD-contiguous", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true, 1>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy nt, y fastest", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true, 2>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    time2("tile copy nt, 8x8 groups", [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true, 3>), dim3(tiles), dim3(256), 0, 0, src, dst, nx4, mx); });
    // does the distance between the two streams matter (same channel/bank for the tile being read and written)?
    for (size_t off : {(size_t)4096, (size_t)65536, (size_t)(1u << 20) + 8192, (size_t)(16u << 20) + 4096 * 37}) {
        float *d2 = dst_base + off / 4;
        char name[64];
        std::snprintf(name, sizeof name, "tile copy nt, dst + %zu KiB", off >> 10);
        time2(name, [&] { hipLaunchKernelGGL((tile_copy_kernel<true, true>), dim3(tiles), dim3(256), 0, 0, src, d2, nx4, mx); });
    }
    (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    time2("hipMemcpyDtoD", [&] { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
