// v_mfma_f32_4x4x1_16B_f32 (sixteen 4x4 blocks, K = 1 per instruction): lane maps on exact integer data and issue cost.
//   D_b[i][j] = A_b[i] * B_b[j],  A_b[i] = 1 + i + 10 b,  B_b[j] = 1 + j + 100 b
// Hypothesis: lane = 4 b + i supplies A_b[i], lane = 4 b + j supplies B_b[j], D VGPR r of lane 4 b + j = block b, row r, column j.
// Cycles: back-to-back issue on one SIMD with 1 / 2 / 4 independent accumulators (dependent-accumulator latency), and with
// independent v_fma_f32 between the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void lanemap(float* out) {
    const int lane = threadIdx.x, b = lane >> 2, i = lane & 3;
    f4 d = {};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f + i + 10.f * b, 1.f + i + 100.f * b, d, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[lane * 4 + v] = d[v];
}
// A-operand broadcast: cbsz = 4, abid = AB -> every block multiplies by block AB's A values (so one register holds the A rows of
// sixteen different k, selected per instruction)
template <int AB>
__global__ void lanemap_bcast(float* out) {
    const int lane = threadIdx.x, b = lane >> 2, i = lane & 3;
    f4 d = {};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f + i + 10.f * b, 1.f + i + 100.f * b, d, 4, AB, 0);
    for (int v = 0; v < 4; ++v) out[lane * 4 + v] = d[v];
}
template <int AB>
int check_bcast(float* d) {
    hipLaunchKernelGGL(lanemap_bcast<AB>, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int b = lane >> 2, j = lane & 3;
            if (h[lane * 4 + r] != (1.f + r + 10.f * AB) * (1.f + j + 100.f * b)) ++bad;
        }
    printf("cbsz = 4, abid = %2d: every block uses block %2d's A rows: %s (%d mismatches)\n", AB, AB, bad ? "NO" : "yes", bad);
    return bad;
}
template <int NACC, int NFMA>
__global__ __launch_bounds__(256) void cyc(float* out, unsigned long long* st, int iters, float a, float b) {
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + (float)j;
    float f[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NFMA; ++k) f[(j * NFMA + k) & 7] = __builtin_fmaf(f[(j * NFMA + k) & 7], a, b);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 s = acc[0];
    for (int j = 1; j < NACC; ++j) s += acc[j];
    float fs = 0; for (int k = 0; k < 8; ++k) fs += f[k];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + fs;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NACC, int NFMA>
void run(const char* name, float* d, unsigned long long* st, int wgs) {
    const int iters = 2000;
    hipLaunchKernelGGL((cyc<NACC, NFMA>), dim3(wgs), dim3(256), 0, 0, d, st, iters, 1.0001f, 0.5f);
    hipLaunchKernelGGL((cyc<NACC, NFMA>), dim3(wgs), dim3(256), 0, 0, d, st, iters, 1.0001f, 0.5f);
    unsigned long long h[4]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %6.2f cycles per MFMA (%d fma between)\n", name, (double)h[0] / (iters * 8.0 * NACC), NFMA);
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    unsigned long long* st; hipMalloc(&st, 1 << 16);
    hipLaunchKernelGGL(lanemap, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int b = lane >> 2, j = lane & 3;
            const float want = (1.f + r + 10.f * b) * (1.f + j + 100.f * b);
            if (h[lane * 4 + r] != want) { if (bad < 8) printf("lane %d r %d: got %.0f want %.0f\n", lane, r, h[lane * 4 + r], want); ++bad; }
        }
    printf("4x4x1_16B hypothesis %s (%d mismatches)\n", bad ? "WRONG" : "holds: A lane = 4 b + row, B lane = 4 b + col, D vgpr r of lane 4 b + j = block b row r col j", bad);
    for (int lane : {0, 1, 4, 5, 63}) { printf("lane %2d:", lane); for (int v = 0; v < 4; ++v) printf(" %.0f", h[lane * 4 + v]); printf("\n"); }
    check_bcast<0>(d); check_bcast<3>(d); check_bcast<15>(d);
    run<1, 0>("1 accumulator (dependent chain)", d, st, 1);
    run<2, 0>("2 accumulators", d, st, 1);
    run<4, 0>("4 accumulators", d, st, 1);
    run<8, 0>("8 accumulators", d, st, 1);
    run<4, 1>("4 accumulators + 1 v_fma each", d, st, 1);
    run<4, 2>("4 accumulators + 2 v_fma each", d, st, 1);
    run<4, 4>("4 accumulators + 4 v_fma each", d, st, 1);
    run<4, 0>("4 accumulators, 256 WGs x 4 waves", d, st, 256);
    run<4, 0>("4 accumulators, 512 WGs (2 waves / SIMD)", d, st, 512);
    return 0;
}
