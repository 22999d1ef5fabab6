// L2 -> LDS fill rate of one CU under the ViT-Base GEMM's access pattern: 512 threads per workgroup (one workgroup per CU) fetch
// 64 KiB "k-tiles" -- 256 rows x 128 B of X plus 256 rows x 128 B of W, a row = 8 lanes x 16 B -- tile after tile as the persistent
// GEMM does (M = 81920, N = 3072, K = 768), by
//   A  LDS-DMA (global_load_lds_dwordx4), DEPTH k-tiles in flight, counted vmcnt
//   B  global_load_dwordx4 into registers + ds_write_b128, one k-tile in flight behind the one being written
//   C  global_load_dwordx4 only (no LDS), one k-tile in flight
// Prints GB/s per CU and cycles per 1 KiB wave-instruction for 256 / 128 / 64 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int K = 768, KB = K * 2, M = 81920, N = 3072, NKT = K / 64;
__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ X, const char* __restrict__ W, unsigned* __restrict__ sink, int ntiles,
                                            unsigned long long* __restrict__ cyc) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, row = tid >> 3, ch = tid & 7;      // 64 rows per pass, 8 passes per k-tile: 4 of X, 4 of W
    u4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int issued = 0, total = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) total += NKT;
    // flat list of (tile, kt) steps for this workgroup
    auto src = [&](int step, int pass) -> const char* {
        const int t = blockIdx.x + (step / NKT) * gridDim.x, kt = step % NKT;
        const int tn = t % (N / 256), tm = t / (N / 256);
        const int r = (pass & 3) * 64 + row;
        return (pass < 4 ? X + (size_t)(tm * 256 + r) * KB : W + (size_t)(tn * 256 + r) * KB) + kt * 128 + ch * 16;
    };
    if constexpr (MODE == 0) {
        auto issue = [&](int step) {
            char* buf = smem + (step % (DEPTH + 1)) * 65536;
#pragma unroll
            for (int p = 0; p < 8; ++p) glds16(src(step, p), buf + p * 8192 + tid * 16);
        };
        for (; issued < DEPTH && issued < total; ++issued) issue(issued);
        for (int s = 0; s < total; ++s) {
            if (issued < total) { issue(issued); ++issued; wait_vm<8 * DEPTH>(); } else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (s == total - 1) acc.x += *reinterpret_cast<unsigned*>(smem + (s % (DEPTH + 1)) * 65536 + tid * 4);
            __builtin_amdgcn_s_barrier();      // the GEMM re-stages a buffer only after every wave has read it
        }
    } else {
        u4 cur[8], nxt[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) cur[p] = *reinterpret_cast<const u4*>(src(0, p));
        for (int s = 0; s < total; ++s) {
            if (s + 1 < total) {
#pragma unroll
                for (int p = 0; p < 8; ++p) nxt[p] = *reinterpret_cast<const u4*>(src(s + 1, p));
            }
            if constexpr (MODE == 1) {
                char* buf = smem + (s & 1) * 65536;
#pragma unroll
                for (int p = 0; p < 8; ++p) *reinterpret_cast<u4*>(buf + p * 8192 + tid * 16) = cur[p];
                __builtin_amdgcn_s_barrier();
                if (s == total - 1) acc.x += *reinterpret_cast<unsigned*>(buf + ((tid * 4 + 64) & 65535));
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) acc ^= cur[p];
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) cur[p] = nxt[p];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* X, const char* W, unsigned* sink, unsigned long long* cyc, int wgs) {
    const int ntiles = (M / 256) * (N / 256) * wgs / 256;      // the same work per workgroup whatever the grid
    const int lds = MODE == 0 ? (DEPTH + 1) * 65536 : (MODE == 1 ? 2 * 65536 : 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<MODE, DEPTH>), dim3(wgs), dim3(512), lds, 0, X, W, sink, ntiles, cyc);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill<MODE, DEPTH>), dim3(wgs), dim3(512), lds, 0, X, W, sink, ntiles, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double ktiles = (double)ntiles / wgs * NKT, us_per_kt = ms * 1e3 / ktiles;
    std::vector<unsigned long long> h(wgs); hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (auto v : h) cs += (double)v; cs /= wgs;
    printf("%-58s wgs %3d: %7.1f us, %5.2f us per 64 KiB k-tile = %5.1f GB/s per CU, %5.1f clk per KiB\n", name, wgs, ms * 1e3, us_per_kt,
           65536.0 / us_per_kt / 1e3, cs / ktiles / 64.0);
}
// ---- store side: each workgroup writes "tiles" of 256 rows x 512 B (a 256 x 256 bf16 output tile, row pitch = pitch bytes), as the
// GEMM epilogue does: one wave-instruction = 8 rows x 128 B (PATTERN 0) or 2 rows x 512 B (PATTERN 1); NT = non-temporal hint.
template <int PATTERN, int NT>
__global__ __launch_bounds__(512) void spill(char* __restrict__ out, int tiles_per_wg, int tiles_n, size_t pitch) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u4 v = {(unsigned)tid, 1u, 2u, 3u};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x + t * gridDim.x, tn = tile % tiles_n, tm = tile / tiles_n;
        char* base = out + (size_t)tm * 256 * pitch + (size_t)tn * 512;
#pragma unroll
        for (int i = 0; i < 16; ++i) {        // 16 wave-instructions per wave = 16 KiB; 8 waves = 128 KiB
            char* p;
            if (PATTERN == 0) {               // wave w owns column slice (w & 3) * 128 B of rows (w >> 2) * 128 + i * 8 + lane / 8
                const int r = (w >> 2) * 128 + i * 8 + (lane >> 3);
                p = base + (size_t)r * pitch + (w & 3) * 128 + (lane & 7) * 16;
            } else {                          // wave w owns rows w * 32 + i * 2 + lane / 32, 512 B each
                const int r = w * 32 + i * 2 + (lane >> 5);
                p = base + (size_t)r * pitch + (lane & 31) * 16;
            }
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p));
            else *reinterpret_cast<u4*>(p) = v;
            v.x += 1;
        }
    }
}
template <int PATTERN, int NT>
void run_store(const char* name, char* out, int wgs) {
    const int tiles_n = 6, tiles_m = 320, per = tiles_n * tiles_m / 256;       // qk: N = 1536 -> 6 column tiles, pitch 3072 B; 7 tiles per workgroup
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((spill<PATTERN, NT>), dim3(wgs), dim3(512), 0, 0, out, per, tiles_n, (size_t)3072);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((spill<PATTERN, NT>), dim3(wgs), dim3(512), 0, 0, out, per, tiles_n, (size_t)3072);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-58s wgs %3d: %7.1f us, %5.2f us per 128 KiB tile = %5.1f GB/s per CU, %5.2f TB/s\n", name, wgs, ms * 1e3, ms * 1e3 / per,
           131072.0 / (ms * 1e3 / per) / 1e3, 131072.0 * per * wgs / (ms * 1e-3) / 1e12);
}
int main() {
    char *X, *W; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&X, (size_t)M * KB); hipMalloc(&W, (size_t)N * KB); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(X, 1, (size_t)M * KB); hipMemset(W, 2, (size_t)N * KB);
    for (int wgs : {256, 128, 64}) {
        run<0, 1>("A LDS-DMA, 1 k-tile in flight", X, W, sink, cyc, wgs);
        run<0, 1>("A LDS-DMA, 1 k-tile in flight (again)", X, W, sink, cyc, wgs);
        run<1, 1>("B global_load_dwordx4 + ds_write_b128, 1 k-tile in flight", X, W, sink, cyc, wgs);
        run<2, 1>("C global_load_dwordx4 only, 1 k-tile in flight", X, W, sink, cyc, wgs);
    }
    run<0, 1>("A LDS-DMA, 1 k-tile in flight", X, W, sink, cyc, 256);
    char* out; hipMalloc(&out, (size_t)320 * 256 * 3072);
    for (int wgs : {256, 128, 64}) {
        run_store<0, 0>("D stores, 8 rows x 128 B per instruction", out, wgs);
        run_store<0, 1>("E the same, non-temporal", out, wgs);
        run_store<1, 0>("F stores, 2 rows x 512 B per instruction", out, wgs);
        run_store<1, 1>("G the same, non-temporal", out, wgs);
    }
    return 0;
}
