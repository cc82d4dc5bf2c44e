// Prints the operand/result lane maps of v_mfma_f32_4x4x1_16b_f32 on this GPU (tuning aid for kernels_stream.hip):
// D[lane][reg] = A[la] * B[lb]; run 1 (B = 1) yields la, run 2 (A = 1) yields lb.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, int mode) {
    const int l = threadIdx.x;
    const float a = mode == 0 ? (float)(l + 1) : 1.0f;
    const float b = mode == 1 ? (float)(l + 1) : 1.0f;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main() {
    float *d; hipMalloc(&d, 64 * 4 * 4);
    float h[2][256];
    for (int m = 0; m < 2; m++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, m); hipMemcpy(h[m], d, sizeof(h[m]), hipMemcpyDeviceToHost); }
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; r++) printf("  reg%d = A[%2d]*B[%2d]", r, (int)h[0][l * 4 + r] - 1, (int)h[1][l * 4 + r] - 1);
        printf("\n");
    }
    return 0;
}
