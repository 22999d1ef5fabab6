// probe_fma_sgpr.hip -- issue cost of v_fma_f32 / v_pk_fma_f32 with a VGPR or an SGPR multiplicand, one wave
// per SIMD, 16 independent accumulators (no dependency stalls).  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* src, unsigned long long* out, float* sink, int iters) {
    f2 acc[16];
    const float a = src[threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f2{a + j, a - j};
    // 16 wave-uniform weights in SGPRs (loaded through a scalar pointer), 16 in VGPRs
    float ws[16], wv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { ws[j] = __builtin_amdgcn_readfirstlane(src[256 + j]); wv[j] = src[threadIdx.x + j]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) {            // v_fma_f32, VGPR weight (two per accumulator pair)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(a), "v"(wv[j]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "v"(a), "v"(wv[j]));
            } else if (MODE == 1) {     // v_fma_f32, SGPR weight
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(a), "s"(ws[j]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "v"(a), "s"(ws[j]));
            } else if (MODE == 2) {     // v_pk_fma_f32, VGPR pair weight
                f2 w2 = f2{wv[j], wv[(j + 1) & 15]};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(f2{a, a}), "v"(w2));
            } else {                    // v_pk_fma_f32, SGPR pair weight
                f2 w2 = f2{ws[j], ws[(j + 1) & 15]};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(f2{a, a}), "s"(w2));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j].x + acc[j].y;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    const int iters = 2000, blocks = 256;
    float* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 8192); hipMemset(src, 0, 8192); hipMalloc(&out, blocks * 4 * 8); hipMalloc(&sink, blocks * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x; s /= h.size();
    printf("%-40s %6.2f cycles per 2 fp32 fma lanes-op (per accumulator pair)\n", name, s / iters / 16);
    hipFree(src); hipFree(out); hipFree(sink);
}
int main() {
    run<0>("2 x v_fma_f32, VGPR weight");
    run<1>("2 x v_fma_f32, SGPR weight");
    run<2>("1 x v_pk_fma_f32, VGPR weight pair");
    run<3>("1 x v_pk_fma_f32, SGPR weight pair");
    return 0;
}
