// hbm_read_patterns.hip -- read-only streaming rates of one MI355X for the access shapes the tile kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_read_patterns hbm_read_patterns.hip ; run on a GPU box.
//   linear   : every workgroup reads one contiguous 64 KiB block (16 B per lane, 16 loads per thread)
//   tile64   : every workgroup reads a 256-float x 64-row tile of a 16384-wide image (1 KiB rows, 64 KiB apart)
//   tile32x2 : same tile as two 32-row halves, second half requested while the first is consumed (fused_tails)
//   segment  : same tile in the x-phase register layout (64 B per lane and row, four loads of 16 B)
// The loaded values are summed into one float per thread and written once (negligible traffic).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

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

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
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
    return 0;
}
