// Which vector instructions issue BESIDE another wave's MFMAs on the same SIMD?  (round 4; the question behind the block kernel's
// owner / guest pairing: tools/src/probe_pair.hip showed v_fma_f32 making almost no progress beside back-to-back bf16 MFMAs while
// and / sub / perm and v_exp did.)  One 512-thread workgroup per CU (96 KiB of LDS): waves 0-3 issue stream A back to back, waves
// 4-7 (their SIMD partners) a stream of ONE vector instruction on 16 rotating registers, written in inline asm so that the operand
// kinds are exact.  Per instruction: cycles alone, and its rate while A runs as a fraction of its rate alone.
//   A: 1 = v_mfma_f32_16x16x32_bf16, 2 = v_mfma_f32_16x16x4_f32, 3 = v_mfma_f32_16x16x16_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int A, int B, bool RUN_A, bool RUN_B>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ st, int iters, float scs, unsigned msk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24576; i += 512) lds[i] = src[i & 4095];
    f4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + (float)j;
    const f4 s0 = *reinterpret_cast<const f4*>(src + 4 * lane), s1 = *reinterpret_cast<const f4*>(src + 256 + 4 * lane);
    const u4 a8 = __builtin_bit_cast(u4, s0), b8v = __builtin_bit_cast(u4, s1);
    float v[16];
    f2 p[8];
    for (int j = 0; j < 16; ++j) v[j] = src[512 + 16 * lane + j];
    for (int j = 0; j < 8; ++j) p[j] = f2{v[2 * j], v[2 * j + 1]};
    float x = src[lane], y = src[64 + lane];
    f2 xp = {x, y}, yp = {y, x};
    asm volatile("" : "+s"(msk));      // SGPR operands
    asm volatile("" : "+s"(scs));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4 && RUN_A) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (A == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a8), __builtin_bit_cast(b8, b8v), acc[j], 0, 0, 0);
                    if (A == 2) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(s0[r], s1[r], acc[j], 0, 0, 0);
                    if (A == 3) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s4{(short)a8.x, (short)a8.y, (short)a8.z, (short)a8.w}, s4{(short)b8v.x, (short)b8v.y, (short)b8v.z, (short)b8v.w}, acc[j], 0, 0, 0);
                }
        }
    }
    if (w >= 4 && RUN_B) {
        for (int it = 0; it < iters; ++it) {
#define I0(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
#define I1(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "s"(scs));
#define I2(j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "s"(scs));
#define I3(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(x));
#define I4(j) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[j]) : "s"(scs));
#define I5(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(x));
#define I6(j) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[j]) : "s"(scs));
#define I7(j) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[j]));
#define I8(j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "s"(msk));
#define I9(j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(x));
#define I10(j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
#define I11(j) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(x));
#define I12(j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
#define I13(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(xp), "v"(yp));
#define I14(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(xp));
#define I15(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(xp));
#define I16(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
#define I17(j) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[j]) : "v"(x));
#define I18(j) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[j]));
#define I19(j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(x));
#define I20(j) asm volatile("v_fmamk_f32 %0, %0, 0x3f800100, %1" : "+v"(v[j]) : "v"(x));
#define I21(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
#define I22(j) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[j]) : "s"(msk), "v"(x));
            if (B == 0) { REP16(I0) REP16(I0) }
            if (B == 1) { REP16(I1) REP16(I1) }
            if (B == 2) { REP16(I2) REP16(I2) }
            if (B == 3) { REP16(I3) REP16(I3) }
            if (B == 4) { REP16(I4) REP16(I4) }
            if (B == 5) { REP16(I5) REP16(I5) }
            if (B == 6) { REP16(I6) REP16(I6) }
            if (B == 7) { REP16(I7) REP16(I7) }
            if (B == 8) { REP16(I8) REP16(I8) }
            if (B == 9) { REP16(I9) REP16(I9) }
            if (B == 10) { REP16(I10) REP16(I10) }
            if (B == 11) { REP16(I11) REP16(I11) }
            if (B == 12) { REP16(I12) REP16(I12) }
            if (B == 13) { REP16(I13) REP16(I13) }
            if (B == 14) { REP16(I14) REP16(I14) }
            if (B == 15) { REP16(I15) REP16(I15) }
            if (B == 16) { REP16(I16) REP16(I16) }
            if (B == 17) { REP16(I17) REP16(I17) }
            if (B == 18) { REP16(I18) REP16(I18) }
            if (B == 19) { REP16(I19) REP16(I19) }
            if (B == 20) { REP16(I20) REP16(I20) }
            if (B == 21) { REP16(I21) REP16(I21) }
            if (B == 22) { REP16(I22) REP16(I22) }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    float sv = 0.f;
    for (int j = 0; j < 16; ++j) sv += v[j];
    for (int j = 0; j < 8; ++j) sv += p[j].x + p[j].y;
    out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w + sv + lds[threadIdx.x];
    if (lane == 0) st[blockIdx.x * 8 + w] = t1 - t0;
}

// ONE wave per SIMD, its own stream: 8 x { 1 MFMA K = 32, N vector instructions } -- what a wave's own fillers cost
template <int KIND, int N>
__global__ __launch_bounds__(256) void self_kernel(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ st, int iters, float scs) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + (float)j;
    const f4 s0 = *reinterpret_cast<const f4*>(src + 4 * lane), s1 = *reinterpret_cast<const f4*>(src + 256 + 4 * lane);
    const u4 a8 = __builtin_bit_cast(u4, s0), b8v = __builtin_bit_cast(u4, s1);
    float v[16];
    f2 p[8];
    for (int j = 0; j < 16; ++j) v[j] = src[512 + 16 * lane + j];
    for (int j = 0; j < 8; ++j) p[j] = f2{v[2 * j], v[2 * j + 1]};
    float x = src[lane], y = src[64 + lane];
    f2 xp = {x, y}, yp = {y, x};
    asm volatile("" : "+s"(scs));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a8), __builtin_bit_cast(b8, b8v), acc[g & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int j = (g * N + n) & 15;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(xp), "v"(yp));
                if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[j]));
                if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(xp));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    float sv = 0.f;
    for (int j = 0; j < 16; ++j) sv += v[j];
    for (int j = 0; j < 8; ++j) sv += p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + sv;
    if (lane == 0) st[blockIdx.x * 4 + w] = t1 - t0;
}
static float* g_src; static float* g_out; static unsigned long long* g_st;
template <int A, int B, bool RA, bool RB>
void run(double* a_own, double* b_own) {
    const int iters = 1000, wgs = 256;
    auto kk = k<A, B, RA, RB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kk, dim3(wgs), dim3(512), 98304, 0, g_src, g_out, g_st, iters, 1.0001f, 0x07060302u);
    std::vector<unsigned long long> h(wgs * 8); hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    double ao = 0, bo = 0;
    for (int b = 0; b < wgs; ++b)
        for (int w = 0; w < 4; ++w) { ao += (double)h[b * 8 + w] / 4; bo += (double)h[b * 8 + 4 + w] / 4; }
    *a_own = ao / wgs / iters; *b_own = bo / wgs / iters;
}
template <int A, int B>
void combo(const char* bn) {
    double ta, tb, a2, b2, d;
    run<A, B, true, false>(&ta, &d); run<A, B, false, true>(&d, &tb); run<A, B, true, true>(&a2, &b2);
    // B ran X of its 1000 iterations while A was running (ta * 1000 cycles... a2 if B slowed A), the rest alone
    const double x = 1000.0 - (b2 - a2) * 1000.0 / tb;        // iterations of B done by the time A finished
    const double frac = (x * tb) / (a2 * 1000.0);              // B's rate beside A / its rate alone
    printf("  %-46s alone %5.2f cyc/instr | beside A: A %6.1f -> %6.1f per 8 MFMAs, B progresses at %4.0f %% of its own rate = %5.2f instr per MFMA\n",
           bn, tb / 32.0, ta, a2, 100.0 * frac, x * 32.0 / 8000.0);
}
template <int A>
void all(const char* an) {
    printf("A = %s\n", an);
    combo<A, 0>("v_fma_f32 v, v, v, v");
    combo<A, 16>("v_fmac_f32 v, v, v");
    combo<A, 1>("v_fma_f32 v, v, v, s");
    combo<A, 20>("v_fmamk_f32 v, v, K, v");
    combo<A, 2>("v_fma_f32 v, v, s, s");
    combo<A, 3>("v_mul_f32 v, v, v");
    combo<A, 4>("v_mul_f32 v, s, v");
    combo<A, 5>("v_add_f32 v, v, v");
    combo<A, 17>("v_sub_f32 v, v, v");
    combo<A, 6>("v_add_f32 v, s, v");
    combo<A, 7>("v_and_b32 v, K, v");
    combo<A, 18>("v_lshlrev_b32 v, 16, v");
    combo<A, 8>("v_perm_b32 v, v, v, s");
    combo<A, 21>("v_and_or_b32 v, v, v, v");
    combo<A, 22>("v_bfi_b32 v, s, v, v");
    combo<A, 9>("v_max_f32 v, v, v");
    combo<A, 10>("v_max3_f32 v, v, v, v");
    combo<A, 11>("v_mov_b32 v, v");
    combo<A, 12>("v_exp_f32 v, v");
    combo<A, 19>("v_cvt_pk_bf16_f32 v, v, v");
    combo<A, 13>("v_pk_fma_f32 v2, v2, v2, v2");
    combo<A, 14>("v_pk_mul_f32 v2, v2, v2");
    combo<A, 15>("v_pk_add_f32 v2, v2, v2");
}
template <int KIND, int N>
double self_run() {
    const int iters = 1000, wgs = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((self_kernel<KIND, N>), dim3(wgs), dim3(256), 0, 0, g_src, g_out, g_st, iters, 1.0001f);
    std::vector<unsigned long long> h(wgs * 4); hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    return s / h.size() / iters / 8.0;
}
template <int KIND>
void self_all(const char* name) {
    printf("  own stream { 1 mfma K32 + n x %-14s}: cycles per group  n=0 %5.1f  n=1 %5.1f  n=2 %5.1f  n=3 %5.1f  n=4 %5.1f  n=6 %5.1f  n=8 %5.1f\n", name,
           self_run<KIND, 0>(), self_run<KIND, 1>(), self_run<KIND, 2>(), self_run<KIND, 3>(), self_run<KIND, 4>(), self_run<KIND, 6>(), self_run<KIND, 8>());
}
int main() {
    hipMalloc(&g_src, 65536 * 4); hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_st, 256 * 8 * 8);
    std::vector<float> h(65536); for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 0.001f * (float)((i * 37) % 211);
    hipMemcpy(g_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("one wave per SIMD, fillers in the wave's own stream:\n");
    self_all<0>("v_fma_f32"); self_all<1>("v_pk_fma_f32"); self_all<2>("v_and_b32"); self_all<3>("v_exp_f32"); self_all<4>("v_pk_add_f32");
    all<1>("v_mfma_f32_16x16x32_bf16 back to back");
    all<3>("v_mfma_f32_16x16x16_bf16 back to back");
    all<2>("v_mfma_f32_16x16x4_f32 back to back");
    return 0;
}
