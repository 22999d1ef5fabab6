// probe_overlap.hip -- what breaks MFMA / VALU overlap between two waves of one SIMD?
// Workgroup = 8 waves: waves 0-3 (older, one per SIMD) run stream X, waves 4-7 (younger) stream Y.
//   M : MFMA stream as the conv / GEMM kernels issue it: per 4 MFMAs one ds_read_b128 operand (+ waitcnt),
//       optionally MV VALU ops (address / activation work) per 4 MFMAs
//   V : VALU stream: v_pk_fma_f32 with optional ds_write_b128 and v_exp_f32 sprinkled in, optional s_setprio 3
// Prints, per configuration, the cycles each role needs alone and together for the same work.
// Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Cfg { int mfma_first; int mv; int prio; int lds_w; int trans; int run_m; int run_v; int no_barrier; };

template <int MV, int LDSW, int TRANS, int MSTYLE>
__device__ __forceinline__ void role(bool is_m, int prio, int iters, const f4* lds, f4* ldsw, int lane, float a, float& sink) {
    if (is_m) {
        f4 acc[4] = {f4{0, 0, 0, 0}, f4{1, 1, 1, 1}, f4{2, 2, 2, 2}, f4{3, 3, 3, 3}};
        float va = a;
        f4 b = lds[lane];
        f4 junk = f4{0, 0, 0, 0};
        if (MSTYLE == 4) {          // constant operands, 8 independent accumulators (dependency distance 256 cycles)
            f4 acc8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] = f4{0, 0, 0, 0} + (float)j;
            for (int it = 0; it < iters; it += 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc8[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sink += acc8[j].x;
        } else if (MSTYLE == 3) {          // LDS-fed, unrolled by two with two operand registers: no copies, the wait sits after the MFMAs
            f4 b1;
            for (int it = 0; it < iters; it += 2) {
                b1 = lds[((it + 1) & 7) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a, acc[j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < MV; ++u) va = fmaf(va, 0.999f, 0.001f);
                __builtin_amdgcn_sched_barrier(0);
                b = lds[((it + 2) & 7) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j], a, acc[j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < MV; ++u) va = fmaf(va, 0.999f, 0.001f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                f4 nb = b;
                if (MSTYLE != 2) nb = lds[((it + 1) & 7) * 64 + lane];       // next operand, read while this one is used
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(MSTYLE == 0 ? b[j] : a, a, acc[j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < MV; ++u) va = fmaf(va, 0.999f, 0.001f);
                if (MSTYLE == 1) junk = junk + nb; else b = nb;
            }
        }
        sink += acc[0].x + acc[1].y + acc[2].z + acc[3].w + va + junk.x + b.x;
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        f2 v[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) v[j] = f2{a + j, a - j};
        const f2 m = f2{0.999f, 0.998f}, c = f2{0.001f, 0.002f};
        float e = a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = __builtin_elementwise_fma(v[j], m, c);
            if (TRANS) { e = __builtin_amdgcn_exp2f(e); e = __builtin_amdgcn_rcpf(e + 2.f); }
            if (LDSW) ldsw[(it & 3) * 64 + lane] = f4{v[0].x, v[1].x, v[2].x, e};
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) sink += v[j].x + v[j].y;
        sink += e;
    }
}

// Minimal form without LDS / barrier: waves 0-3 run 8-accumulator constant-operand MFMAs, waves 4-7 NV chains of v_pk_fma_f32.
template <int NV, int MODE, int LDSF4 = 0, int MTYPE = 0>
__global__ __launch_bounds__(512) void kmin(const float* src, unsigned long long* out, float* sinkp, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ f4 pad[LDSF4 > 0 ? LDSF4 : 1];
    if (LDSF4 > 0) pad[threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f};
    const float a = src[threadIdx.x & 255];
    float sink = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (MODE != 1) {
            f4 acc8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] = f4{0, 0, 0, 0} + (float)j;
            t0 = __builtin_amdgcn_s_memtime();
            typedef short s4 __attribute__((ext_vector_type(4)));
            const s4 ab = s4{(short)lane, 1, 2, 3};
            for (int it = 0; it < iters; it += 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (MTYPE == 0) acc8[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc8[j], 0, 0, 0);
                    else acc8[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, ab, acc8[j], 0, 0, 0);
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < 8; ++j) sink += acc8[j].x;
        }
    } else {
        if (MODE != 0) {
            f2 v[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = f2{a + j, a - j};
            const f2 m = f2{0.999f, 0.998f}, c = f2{0.001f, 0.002f};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = __builtin_elementwise_fma(v[j], m, c);
            }
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < NV; ++j) sink += v[j].x + v[j].y;
        }
    }
    if (LDSF4 > 0) sink += pad[(threadIdx.x + 1) & 511].x;
    sinkp[blockIdx.x * 512 + threadIdx.x] = sink;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NV, int LDSF4 = 0, int MTYPE = 0>
void run_min(const char* name, int iters) {
    const int blocks = 256;
    float* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 4096); hipMemset(src, 0, 4096); hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, blocks * 512 * 4);
    double r[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL((kmin<NV, 0, LDSF4, MTYPE>), dim3(blocks), dim3(512), 0, 0, src, out, sink, iters);
            if (mode == 1) hipLaunchKernelGGL((kmin<NV, 1, LDSF4, MTYPE>), dim3(blocks), dim3(512), 0, 0, src, out, sink, iters);
            if (mode == 2) hipLaunchKernelGGL((kmin<NV, 2, LDSF4, MTYPE>), dim3(blocks), dim3(512), 0, 0, src, out, sink, iters);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double sm = 0, sv = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
        r[mode][0] = sm / (blocks * 4); r[mode][1] = sv / (blocks * 4);
    }
    printf("%-40s iters %5d: M alone %8.0f  V alone %8.0f | together: M %8.0f  V %8.0f  total cycles\n", name, iters, r[0][0], r[1][1], r[2][0], r[2][1]);
    hipFree(src); hipFree(out); hipFree(sink);
}

template <int MV, int LDSW, int TRANS, int MSTYLE>
__global__ __launch_bounds__(512) void k(Cfg cfg, const float* src, unsigned long long* out, float* sinkp, int iters) {
    __shared__ f4 lds[8 * 64];
    __shared__ f4 ldsw[8 * 4 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 512) lds[threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f};
    if (!cfg.no_barrier) __syncthreads();
    const bool older = wave < 4;
    const bool is_m = older == (cfg.mfma_first != 0);
    const float a = src[threadIdx.x & 255];
    float sink = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if ((is_m && cfg.run_m) || (!is_m && cfg.run_v)) {
        t0 = __builtin_amdgcn_s_memtime();
        role<MV, LDSW, TRANS, MSTYLE>(is_m, cfg.prio, iters, lds, ldsw + wave * 4 * 64, lane, a, sink);
        t1 = __builtin_amdgcn_s_memtime();
    }
    sinkp[blockIdx.x * 512 + threadIdx.x] = sink;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MV, int LDSW, int TRANS, int MSTYLE = 0>
void run(const char* name, int mfma_first, int prio, int no_barrier = 0) {
    const int iters = 1500, blocks = 256;
    float* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 4096); hipMemset(src, 0, 4096); hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, blocks * 512 * 4);
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {     // 0: M alone, 1: V alone, 2: both
        Cfg cfg{mfma_first, MV, prio, LDSW, TRANS, mode != 1, mode != 0, no_barrier};
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MV, LDSW, TRANS, MSTYLE>), dim3(blocks), dim3(512), 0, 0, cfg, src, out, sink, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double sm = 0, sv = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) {
            const bool is_m = (w < 4) == (mfma_first != 0);
            (is_m ? sm : sv) += (double)h[b * 8 + w];
        }
        res[mode][0] = sm / (blocks * 4) / iters; res[mode][1] = sv / (blocks * 4) / iters;
    }
    printf("%-46s M alone %6.1f  V alone %6.1f | together: M %6.1f  V %6.1f  (cycles per iteration: 4 MFMAs / 12 pk_fma)\n", name,
           res[0][0], res[1][1], res[2][0], res[2][1]);
    hipFree(src); hipFree(out); hipFree(sink);
}

int main() {
    run_min<12>("min: 12 chains", 1500);
    run_min<16>("min: 16 chains", 1500);
    run_min<16>("min: 16 chains", 4000);
    run_min<12>("min: 12 chains", 500);
    run_min<12, 512>("min: 12 chains + 8 KB LDS", 1500);
    run_min<12, 2560>("min: 12 chains + 40 KB LDS", 1500);
    run_min<12, 5120>("min: 12 chains + 80 KB LDS", 1500);
    run_min<12, 2560, 1>("min: bf16 MFMA 16x16x16, 12 chains + 40 KB LDS", 1500);
    run_min<12, 0, 1>("min: bf16 MFMA 16x16x16, 12 chains, no LDS", 1500);
    run<0, 0, 0, 4>("NO BARRIER: M(const, 8 acc) older, V plain", 1, 0, 1);
    run<0, 0, 0, 4>("M(const, 8 accumulators) older, V plain", 1, 0);
    run<0, 0, 0, 4>("V plain older, M(const, 8 accumulators)", 0, 0);
    run<0, 0, 0, 2>("M(const operand, no lds) older, V plain", 1, 0);
    run<0, 0, 0, 2>("V plain older, M(const operand, no lds)", 0, 0);
    run<0, 0, 0, 2>("V prio 3 younger, M(const operand) older", 1, 1);
    run<0, 0, 0, 1>("M(const operand, lds read unused) older, V", 1, 0);
    run<0, 0, 0, 3>("M(lds-fed, unrolled x2, pinned) older, V", 1, 0);
    run<0, 0, 0, 3>("V older, M(lds-fed, unrolled x2, pinned)", 0, 0);
    run<6, 0, 0, 3>("M(lds-fed x2 pinned)+6 valu older, V plain", 1, 0);
    run<6, 1, 1, 3>("M(lds-fed x2 pinned)+6 valu older, V+ds_w+exp", 1, 0);
    run<0, 0, 0>("M(lds) older, V plain", 1, 0);
    run<0, 0, 0>("V plain older, M(lds)", 0, 0);
    run<0, 0, 0>("M(lds) older, V prio 3", 1, 1);
    run<0, 0, 0>("V prio 3 older, M(lds)", 0, 1);
    run<6, 0, 0>("M(lds)+6 valu older, V plain", 1, 0);
    run<6, 0, 0>("M(lds)+6 valu older, V prio 3", 1, 1);
    run<6, 0, 0>("V prio 3 older, M(lds)+6 valu", 0, 1);
    run<0, 1, 1>("M(lds) older, V +ds_write +exp/rcp", 1, 0);
    run<6, 1, 1>("M(lds)+6 valu older, V +ds_write +exp/rcp", 1, 0);
    run<6, 1, 1>("V +ds_write +exp/rcp prio 3 older, M+6", 0, 1);
    return 0;
}
