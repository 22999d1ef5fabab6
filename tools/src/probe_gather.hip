// probe_gather.hip -- what a per-lane gather costs the texture-address path: wave64 buffer loads of 4 / 8 / 12 / 16 bytes per lane at a
// lane stride of `stride` bytes, aligned down to 1 (as is), 4, 8 or 16 bytes, from a 256 KiB window (L2-resident), 8 waves per SIMD.
// Prints cycles per wave-instruction per CU.   hipcc -O3 --offload-arch=gfx950 -o probe_gather probe_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u3v __attribute__((ext_vector_type(3)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int BYTES, int ALIGN>
__global__ __launch_bounds__(256) void k(const unsigned char* buf, int stride, int iters, unsigned* sink) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(buf), 0, 1 << 20, 0x00020000);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned acc = 0;
    unsigned base = (blockIdx.x * 4 + w) * 977u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned off = (base + (unsigned)(it * 8 + u) * 1531u) % (192u << 10);
            off += (unsigned)(lane * stride);
            off &= ~(unsigned)(ALIGN - 1);
            if constexpr (BYTES == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
            if constexpr (BYTES == 8) { u2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0); acc += v.x ^ v.y; }
            if constexpr (BYTES == 12) { u3v v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)off, 0, 0); acc += v.x ^ v.y ^ v.z; }
            if constexpr (BYTES == 16) { u4v v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int BYTES, int ALIGN>
void run(const unsigned char* buf, unsigned* sink, int stride) {
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<BYTES, ALIGN>), dim3(blocks), dim3(256), 0, 0, buf, stride, 4, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<BYTES, ALIGN>), dim3(blocks), dim3(256), 0, 0, buf, stride, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double winst_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
    printf("bytes %2d align %2d stride %3d: %7.1f us, %6.1f ns per wave-load per CU (%5.1f cycles at 2.4 GHz)\n", BYTES, ALIGN, stride, ms * 1e3,
           ms * 1e6 / winst_per_cu, ms * 1e6 / winst_per_cu * 2.4);
}
int main() {
    unsigned char* buf; unsigned* sink;
    hipMalloc(&buf, 1 << 20); hipMalloc(&sink, 256 * 8 * 256 * 4);
    hipMemset(buf, 1, 1 << 20);
    for (int stride : {3, 13, 28}) {
        run<4, 1>(buf, sink, stride); run<4, 4>(buf, sink, stride);
        run<8, 1>(buf, sink, stride); run<8, 4>(buf, sink, stride); run<8, 8>(buf, sink, stride);
        run<12, 4>(buf, sink, stride);
        run<16, 1>(buf, sink, stride); run<16, 4>(buf, sink, stride); run<16, 16>(buf, sink, stride);
    }
    return 0;
}
