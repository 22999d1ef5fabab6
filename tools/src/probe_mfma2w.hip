// Two waves per SIMD that BOTH stream MFMAs (the G256 head: eight waves per workgroup, one workgroup per CU): what does the matrix
// pipe deliver per SIMD?  One 512-thread workgroup per CU (96 KiB of LDS declared); waves 0-3 alone, then waves 0-7.  Each wave issues
// NACC independent accumulators round-robin, operands rotating over four register quads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int KIND, int NACC, bool BOTH>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ st, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + (float)j;
    f4 s[4];
    for (int j = 0; j < 4; ++j) s[j] = *reinterpret_cast<const f4*>(src + 256 * j + 4 * lane);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (BOTH || w < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, s[j & 1]), __builtin_bit_cast(b8, s[2 + ((j >> 1) & 1)]), acc[j], 0, 0, 0);
                else if (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[j & 1][j & 3], s[2 + ((j >> 1) & 1)][j & 3], acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[j & 1][j & 3], s[2 + ((j >> 1) & 1)][j & 3], acc[j], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    f4 r = acc[0];
    for (int j = 1; j < NACC; ++j) r += acc[j];
    out[blockIdx.x * 512 + threadIdx.x] = r.x + r.y + r.z + r.w + lds[threadIdx.x];
    if (lane == 0) { st[(blockIdx.x * 8 + w) * 2] = t1 - t0; st[(blockIdx.x * 8 + w) * 2 + 1] = t2 - t0; }
}
static float* g_src; static float* g_out; static unsigned long long* g_st;
template <int KIND, int NACC, bool BOTH>
void run(const char* name) {
    const int iters = 2000, wgs = 256;
    auto kk = k<KIND, NACC, BOTH>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kk, dim3(wgs), dim3(512), 98304, 0, g_src, g_out, g_st, iters);
    std::vector<unsigned long long> h(wgs * 16); hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    double wall = 0, a = 0, b = 0;
    for (int i = 0; i < wgs; ++i) {
        wall += (double)h[i * 16 + 1];
        for (int w = 0; w < 4; ++w) { a += (double)h[(i * 8 + w) * 2] / 4; b += (double)h[(i * 8 + 4 + w) * 2] / 4; }
    }
    const double n = (double)iters * NACC * (BOTH ? 2 : 1);
    printf("%-28s %d accumulators, %s: wall %7.2f cycles per MFMA of the SIMD (waves 0-3 done after %5.1f %% of the wall, waves 4-7 %5.1f %%)\n", name, NACC,
           BOTH ? "two waves per SIMD" : "one wave per SIMD ", wall / wgs / n, 100.0 * a / wall, 100.0 * b / wall);
}
int main() {
    hipMalloc(&g_src, 65536 * 4); hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_st, 256 * 16 * 8);
    std::vector<float> h(65536); for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 0.001f * (float)((i * 37) % 211);
    hipMemcpy(g_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 4, false>("v_mfma_f32_16x16x32_bf16"); run<0, 4, true>("v_mfma_f32_16x16x32_bf16");
    run<0, 8, false>("v_mfma_f32_16x16x32_bf16"); run<0, 8, true>("v_mfma_f32_16x16x32_bf16");
    run<1, 4, false>("v_mfma_f32_16x16x4_f32"); run<1, 4, true>("v_mfma_f32_16x16x4_f32");
    run<1, 8, false>("v_mfma_f32_16x16x4_f32"); run<1, 8, true>("v_mfma_f32_16x16x4_f32");
    run<2, 4, false>("v_mfma_f32_4x4x1_16b_f32"); run<2, 4, true>("v_mfma_f32_4x4x1_16b_f32");
    run<2, 8, false>("v_mfma_f32_4x4x1_16b_f32"); run<2, 8, true>("v_mfma_f32_4x4x1_16b_f32");
    return 0;
}
