// probe_coissue.hip -- how much VALU work issues in the shadow of v_mfma_f32_16x16x4_f32 on one SIMD?
// One wave per SIMD (256 threads per CU, 256 CUs).  Loop body: 8 independent MFMAs, each followed by
// K independent VALU ops (v_fma_f32, or v_exp_f32 when TRANS).  Reports shader cycles per MFMA.
// Development aid, not part of the library.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/probe_coissue tools/src/probe_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K, bool TRANS, int WAVES>
__global__ __launch_bounds__(256 * WAVES) void k(const float* src, unsigned long long* out, float* sink, int iters) {
    f4 acc[8];
    float v[8];
    const float a0 = src[threadIdx.x & 255], b0 = src[256 + (threadIdx.x & 255)];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j] = f4{0.f, 0.f, 0.f, 0.f} + 0.001f * j; v[j] = a0 + j; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                if (TRANS) v[(j + u) & 7] = __builtin_amdgcn_exp2f(v[(j + u) & 7]);
                else v[(j + u) & 7] = fmaf(v[(j + u) & 7], 0.999f, 0.001f);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w + v[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K, bool TRANS, int WAVES>
void run(const char* name) {
    const int iters = 2000, blocks = 256, threads = 256 * WAVES;
    float* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 4096); hipMemset(src, 0, 4096);
    hipMalloc(&out, blocks * threads / 64 * 8); hipMalloc(&sink, blocks * threads * 4);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<K, TRANS, WAVES>), dim3(blocks), dim3(threads), 0, 0, src, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x; s /= h.size();
    // per SIMD: WAVES waves each issue 8 MFMAs per iteration
    printf("%-28s waves/SIMD=%d  cycles per MFMA (per SIMD) = %6.1f   cycles per wave-iteration = %7.1f\n", name, WAVES,
           s / iters / 8 / WAVES, s / iters);
    hipFree(src); hipFree(out); hipFree(sink);
}

// Role-split: waves 0-3 of a workgroup (one per SIMD) run only MFMAs, waves 4-7 (same SIMDs) only VALU
// (v_pk_fma_f32 when PK, else v_fma_f32).  MODE 0: both roles, 1: MFMA waves only, 2: VALU waves only.
template <int MODE, bool PK>
__global__ __launch_bounds__(512) void k_roles(const float* src, unsigned long long* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    const float a0 = src[threadIdx.x & 255], b0 = src[256 + (threadIdx.x & 255)];
    float s = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (MODE != 2) {
            f4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + 0.001f * j;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[j], 0, 0, 0);
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
        }
    } else {
        if (MODE != 1) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = f2{a0 + j, b0 - j};
            const f2 m = f2{0.999f, 0.998f}, c = f2{0.001f, 0.002f};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (PK) v[j] = __builtin_elementwise_fma(v[j], m, c);
                    else { v[j].x = fmaf(v[j].x, 0.999f, 0.001f); v[j].y = fmaf(v[j].y, 0.998f, 0.002f); }
                }
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y;
        }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, bool PK>
void run_roles(const char* name) {
    const int iters = 2000, blocks = 256;
    float* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 4096); hipMemset(src, 0, 4096);
    hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, blocks * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_roles<MODE, PK>), dim3(blocks), dim3(512), 0, 0, src, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
    sm /= blocks * 4; sv /= blocks * 4;
    printf("%-34s MFMA wave: %6.1f cycles per MFMA    VALU wave: %6.2f cycles per %s (x%d per iteration)\n", name, sm / iters / 8,
           sv / iters / (PK ? 16 : 32), PK ? "v_pk_fma" : "v_fma", PK ? 16 : 32);
    hipFree(src); hipFree(out); hipFree(sink);
}

int main() {
    run_roles<1, true>("roles: MFMA waves alone");
    run_roles<2, true>("roles: pk_fma waves alone");
    run_roles<0, true>("roles: MFMA + pk_fma waves");
    run_roles<2, false>("roles: fma waves alone");
    run_roles<0, false>("roles: MFMA + fma waves");
    run<0, false, 1>("mfma only");
    run<2, false, 1>("mfma + 2 fma");
    run<4, false, 1>("mfma + 4 fma");
    run<6, false, 1>("mfma + 6 fma");
    run<8, false, 1>("mfma + 8 fma");
    run<12, false, 1>("mfma + 12 fma");
    run<1, true, 1>("mfma + 1 exp");
    run<2, true, 1>("mfma + 2 exp");
    run<4, true, 1>("mfma + 4 exp");
    run<0, false, 2>("mfma only");
    run<4, false, 2>("mfma + 4 fma");
    run<8, false, 2>("mfma + 8 fma");
    run<2, true, 2>("mfma + 2 exp");
    return 0;
}
