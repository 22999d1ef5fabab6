// Lane maps of v_mfma_f32_16x16x1_4B_f32 (four 16x16 blocks, K = 1 per instruction) on exact integer data:
//   D_b[i][j] = A_b[i] * B_b[j]   with A_b[i] = 1 + i + 100 b  (lane holding it: to be found)  and B_b[j] = 1 + j + 1000 b
// Each lane supplies ONE A value and ONE B value; hypothesis: lane = 16 b + i (A) and lane = 16 b + j (B), and
// D VGPR 4 b' + r of lane (q, j) = block b', row 4 q + r, column j.  Prints what the hardware says.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const int b = lane >> 4, i = lane & 15;
    const float a = 1.f + i + 100.f * b, bb = 1.f + i + 1000.f * b;
    f16v d = {};
    d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, bb, d, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[lane * 16 + v] = d[v];
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int v = 0; v < 16; ++v) {
            const int q = lane >> 4, j = lane & 15, blk = v >> 2, r = v & 3, row = 4 * q + r;
            const float want = (1.f + row + 100.f * blk) * (1.f + j + 1000.f * blk);
            if (h[lane * 16 + v] != want) { if (bad < 12) printf("lane %d v %d: got %.0f want %.0f\n", lane, v, h[lane * 16 + v], want); ++bad; }
        }
    printf("hypothesis %s (%d mismatches)\n", bad ? "WRONG" : "holds: A lane = 16 b + row, B lane = 16 b + col, D vgpr 4 b + r of lane (q, j) = block b row 4 q + r col j", bad);
    // raw dump of lane 0, 1, 16, 17
    for (int lane : {0, 1, 16, 17, 63}) { printf("lane %2d:", lane); for (int v = 0; v < 16; ++v) printf(" %.0f", h[lane * 16 + v]); printf("\n"); }
    return 0;
}
