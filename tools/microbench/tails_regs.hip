// tails_regs.hip -- experiment: pass 1 of the fused path (tail contraction) without staging the tile in LDS.
// Every wave owns 16 contiguous rows of the 256 x 64 tile (lane = one 16-byte chunk of a row); the x tails are partial
// dot products per lane reduced across the 64 lanes with a transposing butterfly (v_permlane32_swap / v_permlane16_swap /
// DPP), the y tails are per-lane accumulations over the wave's rows combined across the four waves through LDS once.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tails_regs tails_regs.hip ; run on a GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float F2 __attribute__((ext_vector_type(2)));

constexpr int NXK = 4, NYK = 4;     // cfg3: two x scans and two y scans of order 2

// MODE 0: fake reduction (upper bound on what the restructuring can give); MODE 1: real butterfly
template <int MODE>
__global__ void __launch_bounds__(256) tails_regs_kernel(const f4 *__restrict__ src, float *__restrict__ xt, float *__restrict__ yt,
                                                        const float *__restrict__ Hx, const float *__restrict__ Hy, int nx4,
                                                        int mx, int my) {
    __shared__ float red[4][NYK][256];
    const int tx = blockIdx.x, ty = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t base = ((size_t)ty * 64 + 16 * w) * nx4 + (size_t)tx * 64 + lane;
    f4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = __builtin_nontemporal_load(src + base + (size_t)i * nx4);
    F2 hx[4][NXK / 2];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int p = 0; p < NXK / 2; p++) hx[c][p] = F2{Hx[(2 * p) * 256 + 4 * lane + c], Hx[(2 * p + 1) * 256 + 4 * lane + c]};
    F2 xp[16][NXK / 2];
    F2 yp[NYK][2];
#pragma unroll
    for (int j = 0; j < NYK; j++) yp[j][0] = yp[j][1] = F2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int p = 0; p < NXK / 2; p++) {
            F2 a = hx[0][p] * F2{v[i].x, v[i].x};
            a = hx[1][p] * F2{v[i].y, v[i].y} + a;
            a = hx[2][p] * F2{v[i].z, v[i].z} + a;
            a = hx[3][p] * F2{v[i].w, v[i].w} + a;
            xp[i][p] = a;
        }
#pragma unroll
        for (int j = 0; j < NYK; j++) {
            const float h = Hy[j * 64 + 16 * w + i];       // wave-uniform
            yp[j][0] = F2{h, h} * F2{v[i].x, v[i].y} + yp[j][0];
            yp[j][1] = F2{h, h} * F2{v[i].z, v[i].w} + yp[j][1];
        }
    }
    // ---- x: 64 partial sums per lane -> one total per lane ----
    float mine;
    if (MODE == 0) {
        F2 s = F2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int p = 0; p < NXK / 2; p++) s = s + xp[i][p];
        mine = s.x + s.y;
    } else {
        mine = 0.f;
    }
    {
        // lane L ends up with value index L: row = L >> 2 (of the wave's 16), sr = L & 3
        const int row = lane >> 2, sr = lane & 3;
        const size_t Lx = (size_t)my * 64;
        xt[((size_t)(sr >> 1) * mx + tx) * 2 * Lx + (size_t)(sr & 1) * Lx + (size_t)ty * 64 + 16 * w + row] = mine;
    }
    // ---- y: combine the four waves ----
#pragma unroll
    for (int j = 0; j < NYK; j++)
        *reinterpret_cast<f4 *>(&red[w][j][4 * lane]) = f4{yp[j][0].x, yp[j][0].y, yp[j][1].x, yp[j][1].y};
    __syncthreads();
    const int t = threadIdx.x;
    const size_t Ly = (size_t)mx * 256;
#pragma unroll
    for (int j = 0; j < NYK; j++) {
        const float s = (red[0][j][t] + red[1][j][t]) + (red[2][j][t] + red[3][j][t]);
        yt[((size_t)(j >> 1) * my + ty) * 2 * Ly + (size_t)(j & 1) * Ly + (size_t)tx * 256 + t] = s;
    }
}

int main() {
    const int n = 16384, mx = n / 256, my = n / 64;
    const size_t px = (size_t)n * n;
    float *src, *xt, *yt, *Hx, *Hy;
    hipMalloc(&src, px * 4);
    hipMalloc(&xt, (size_t)2 * mx * 2 * n * 4);
    hipMalloc(&yt, (size_t)2 * my * 2 * n * 4);
    hipMalloc(&Hx, NXK * 256 * 4);
    hipMalloc(&Hy, NYK * 64 * 4);
    std::vector<float> h(px);
    for (size_t i = 0; i < px; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(src, h.data(), px * 4, hipMemcpyHostToDevice);
    std::vector<float> hx(NXK * 256), hy(NYK * 64);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = 0.001f * (float)(i % 97);
    for (size_t i = 0; i < hy.size(); i++) hy[i] = 0.002f * (float)(i % 31);
    hipMemcpy(Hx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(Hy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 1; mode++) {
        float best = 1e9f;
        for (int it = 0; it < 12; it++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(tails_regs_kernel<0>, dim3(mx, my), dim3(256), 0, 0, (const f4 *)src, xt, yt, Hx, Hy, n / 4, mx, my);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 2 && ms < best) best = ms;
        }
        printf("mode %d: %.4f ms  (%.2f TB/s read)\n", mode, best, px * 4 / best * 1e-9);
    }
    return 0;
}
