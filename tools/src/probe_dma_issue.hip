// What does one LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) cost the wave that issues it?  Each wave issues 16 of
// them on L2-warm weight-like data (every workgroup reads the same 16 KiB, as the block kernel's staging does) and stamps
// s_memtime after the last issue (before any wait) and after vmcnt(0):
//   mode 0  back to back, a new M0 (LDS base) per instruction
//   mode 1  back to back, ONE M0 per four instructions, the tile selected by the instruction's immediate offset (0 / 1024 / 2048 / 3072:
//           the offset applies to the global AND the LDS address)
//   mode 2  as mode 0 with ~300 cycles of independent VALU work between two instructions
//   mode 3  16 global_load_dwordx4 into registers back to back, then 16 ds_write_b128 (the same bytes through the register file)
// L2 rows: every repetition reads a different 16 KiB per wave (a 4 MiB buffer, L2-resident, never in the 32 KiB vector L1); L1 rows: the same
// 16 KiB every time.
// for 1 / 4 / 8 waves per workgroup, one workgroup per CU on 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OFF>
__device__ __forceinline__ void glds(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, OFF, 0);
}
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src0, unsigned long long* __restrict__ out, float* __restrict__ sink, int reps, int l2) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* dst = smem + w * 16384;
    const char* s = src0 + lane * 16;
    float v = (float)lane;
    unsigned long long ti = 0, td = 0;
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        if (l2) s = src0 + lane * 16 + (size_t)((r * 8 + w) % 256) * 16384;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if constexpr (MODE == 3) {
            u4 rg[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) rg[j] = *reinterpret_cast<const u4*>(s + j * 1024);
            asm volatile("" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < 16; ++j) *reinterpret_cast<u4*>(dst + j * 1024 + lane * 16) = rg[j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            if (r > 0) { ti += t1 - t0; td += t2 - t0; }
            continue;
        }
        if constexpr (MODE == 4) {       // the block kernel's staging loop: 54 tiles of one 54 KiB image over the workgroup's waves, tile t -> wave t % nw
            const int nw = blockDim.x >> 6;
            const char* img = src0 + (l2 ? (size_t)(r % 64) * 55296 : 0) + lane * 16;
            for (int t = w; t < 54; t += nw)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (size_t)t * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + t * 1024), 16, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                glds<0>(s + j * 1024, dst + j * 1024);
                glds<1024>(s + j * 1024, dst + j * 1024);
                glds<2048>(s + j * 1024, dst + j * 1024);
                glds<3072>(s + j * 1024, dst + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                glds<0>(s + j * 1024, dst + j * 1024);
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 64; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
                    asm volatile("" : "+v"(v));
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (r > 0) { ti += t1 - t0; td += t2 - t0; }
    }
    if (lane == 0) { out[(blockIdx.x * 8 + w) * 2] = ti / (reps - 1); out[(blockIdx.x * 8 + w) * 2 + 1] = td / (reps - 1); }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = v + *reinterpret_cast<float*>(smem + threadIdx.x * 4);
}
template <int MODE>
void run(const char* name, const char* src, unsigned long long* out, float* sink, int waves, int wgs, int l2) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipMemset(out, 0, 256 * 8 * 2 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(waves * 64), 8 * 16384, 0, src, out, sink, 50, l2);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8 * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double ti = 0, td = 0;
    int n = 0;
    for (int b = 0; b < wgs; ++b)
        for (int w = 0; w < waves; ++w) { ti += h[(b * 8 + w) * 2]; td += h[(b * 8 + w) * 2 + 1]; ++n; }
    printf("%s %-34s %d wave(s)/WG x %3d WGs: issue of 16 pieces %7.0f ticks (%5.1f per piece), landed %7.0f\n", l2 ? "L2" : "L1", name, waves, wgs, ti / n, ti / n / 16, td / n);
}
int main() {
    char* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 5 << 20); hipMemset(src, 1, 5 << 20);
    hipMalloc(&out, 256 * 8 * 2 * 8); hipMalloc(&sink, 256 * 512 * 4);
    printf("s_memtime ticks (100 MHz constant clock on gfx950? compare columns, not absolute)\n");
    for (int l2 : {0, 1})
        for (int wgs : {1, 256})
            for (int waves : {1, 4, 8}) {
                run<0>("back to back, M0 per piece", src, out, sink, waves, wgs, l2);
                run<1>("back to back, M0 per 4 pieces", src, out, sink, waves, wgs, l2);
                if (!l2 && wgs == 1) run<2>("64 dependent fmas between pieces", src, out, sink, waves, wgs, l2);
                run<3>("global_load x16, then ds_write x16", src, out, sink, waves, wgs, l2);
                run<4>("54-tile image over the WG's waves", src, out, sink, waves, wgs, l2);
            }
    return 0;
}
