// Two waves per SIMD with DIFFERENT instruction streams (the owner / guest pairing of vt_blocks.h): how much of wave B's vector work
// issues beside wave A's matrix work?  One 512-thread workgroup per CU (it declares 96 KiB of LDS, so a CU holds one): waves 0-3 run
// stream A, waves 4-7 stream B (wave w and w + 4 share a SIMD).  Every combination is timed three ways -- A alone, B alone, both --
// as the workgroup's wall time in shader cycles (s_memtime around a barrier-delimited region): both ~ max(A, B) = the streams overlap,
// both ~ A + B = they share one issue port.
//   A: 0 nothing | 1 v_mfma_f32_16x16x32_bf16 back to back | 2 v_mfma_f32_16x16x4_f32 back to back | 3 K = 32 MFMAs in pairs with
//      8 independent v_fma_f32 behind each pair (a wave that interleaves its own VALU work, like fc1's GELU stages)
//   B: 0 nothing | 1 v_fma_f32 | 2 v_pk_fma_f32 (the same flops in half the instructions) | 3 v_and / v_sub / v_perm (operand split)
//      | 4 v_exp_f32 | 5 ds_read_b128 + v_pk_add (LDS traffic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int A, int B, bool RUN_A, bool RUN_B, int PRIO_B>
__global__ __launch_bounds__(512) void pair_kernel(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ st, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24576; i += 512) lds[i] = src[i & 4095];
    f4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + (float)j;
    const f4 s0 = *reinterpret_cast<const f4*>(src + 4 * lane), s1 = *reinterpret_cast<const f4*>(src + 256 + 4 * lane);
    const u4 a8 = __builtin_bit_cast(u4, s0), b8v = __builtin_bit_cast(u4, s1);
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = src[512 + 16 * lane + j];
    if (w >= 4 && PRIO_B > 0) __builtin_amdgcn_s_setprio(PRIO_B);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4 && RUN_A) {
        for (int it = 0; it < iters; ++it) {
            if (A == 1) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a8), __builtin_bit_cast(b8, b8v), acc[j], 0, 0, 0);
            } else if (A == 2) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(s0[r], s1[r], acc[j], 0, 0, 0);
            } else if (A == 3) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a8), __builtin_bit_cast(b8, b8v), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a8), __builtin_bit_cast(b8, b8v), acc[1], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.25f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (w >= 4 && RUN_B) {
        for (int it = 0; it < iters; ++it) {
            if (B == 1) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.25f);
            } else if (B == 2) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 16; j += 2) {
                        const f2 t = __builtin_elementwise_fma(f2{v[j], v[j + 1]}, f2{1.0001f, 1.0001f}, f2{0.25f, 0.25f});
                        v[j] = t.x; v[j + 1] = t.y;
                    }
            } else if (B == 3) {
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const unsigned x0 = __float_as_uint(v[j]), x1 = __float_as_uint(v[j + 1]);
                    const float r0 = v[j] - __uint_as_float(x0 & 0xffff0000u), r1 = v[j + 1] - __uint_as_float(x1 & 0xffff0000u);
                    const unsigned p = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                    v[j] = r0 + 1.0f; v[j + 1] = __uint_as_float(p | 0x3f800000u);
                }
            } else if (B == 4) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
            } else if (B == 5) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f4 t = *reinterpret_cast<const f4*>(lds + ((it * 8 + j) & 63) * 256 + 4 * lane);
                    v[2 * j] += t.x + t.z; v[2 * j + 1] += t.y + t.w;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    float sv = 0.f;
    for (int j = 0; j < 16; ++j) sv += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w + sv + lds[threadIdx.x];
    if (lane == 0) { st[(blockIdx.x * 8 + w) * 2] = t1 - t0; st[(blockIdx.x * 8 + w) * 2 + 1] = t2 - t0; }
}

static float* g_src; static float* g_out; static unsigned long long* g_st;
template <int A, int B, bool RA, bool RB, int PB>
double run(double* a_own = nullptr, double* b_own = nullptr) {
    const int iters = 1000, wgs = 256;
    auto k = pair_kernel<A, B, RA, RB, PB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 98304, 0, g_src, g_out, g_st, iters);
    std::vector<unsigned long long> h(wgs * 16); hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    double wall = 0, ao = 0, bo = 0;
    for (int b = 0; b < wgs; ++b) {
        wall += (double)h[(b * 8) * 2 + 1];
        for (int w = 0; w < 4; ++w) { ao += (double)h[(b * 8 + w) * 2] / 4; bo += (double)h[(b * 8 + 4 + w) * 2] / 4; }
    }
    if (a_own) *a_own = ao / wgs / iters;
    if (b_own) *b_own = bo / wgs / iters;
    return wall / wgs / iters;
}
template <int A, int B>
void combo(const char* an, const char* bn) {
    double ta = run<A, B, true, false, 0>(), tb = run<A, B, false, true, 0>(), a0, b0, a2, b2;
    double tab = run<A, B, true, true, 0>(&a0, &b0), tab2 = run<A, B, true, true, 2>(&a2, &b2);
    printf("A = %-44s B = %-40s alone %6.1f / %6.1f  both %6.1f (A %6.1f, B %6.1f)  B at prio 2: %6.1f (A %6.1f, B %6.1f)  [sum %6.1f]\n", an, bn, ta, tb, tab, a0, b0, tab2, a2, b2, ta + tb);
}
int main() {
    hipMalloc(&g_src, 65536 * 4); hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_st, 256 * 16 * 8);
    std::vector<float> h(65536); for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 0.001f * (float)((i * 37) % 211);
    hipMemcpy(g_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("cycles per iteration (A: 8 MFMAs; B: 32 fma | 16 pk_fma | 8 x (2 and, 2 sub, perm, add, or) | 16 exp | 8 ds_read_b128 + adds)\n");
    combo<1, 1>("8 x mfma 16x16x32 bf16", "32 v_fma_f32");
    combo<1, 2>("8 x mfma 16x16x32 bf16", "16 v_pk_fma_f32");
    combo<1, 3>("8 x mfma 16x16x32 bf16", "split-like (and / sub / perm)");
    combo<1, 4>("8 x mfma 16x16x32 bf16", "16 v_exp_f32");
    combo<1, 5>("8 x mfma 16x16x32 bf16", "8 ds_read_b128 + 16 adds");
    combo<2, 1>("8 x mfma 16x16x4 f32", "32 v_fma_f32");
    combo<2, 2>("8 x mfma 16x16x4 f32", "16 v_pk_fma_f32");
    combo<2, 3>("8 x mfma 16x16x4 f32", "split-like (and / sub / perm)");
    combo<3, 1>("4 x (2 mfma K32 + 8 v_fma)", "32 v_fma_f32");
    combo<3, 2>("4 x (2 mfma K32 + 8 v_fma)", "16 v_pk_fma_f32");
    combo<3, 3>("4 x (2 mfma K32 + 8 v_fma)", "split-like (and / sub / perm)");
    combo<0, 1>("-", "32 v_fma_f32");
    combo<0, 2>("-", "16 v_pk_fma_f32");
    return 0;
}
