// Do 32-bit global / buffer loads at byte-unaligned addresses return the right bytes on this box (unaligned access mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* p, unsigned* out, int nbytes) {
    const int o = threadIdx.x;     // byte offset
    unsigned a;
    const unsigned char* q = p + o;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(q) : "memory");
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), 0, nbytes, 0x00020000);
    const unsigned b = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0);
    out[2 * o] = a; out[2 * o + 1] = b;
}
int main() {
    const int n = 256;
    std::vector<unsigned char> h(n); for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i * 7 + 3);
    unsigned char* d; unsigned* o; hipMalloc(&d, n); hipMalloc(&o, 2 * 64 * 4);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n - 190);      // buffer covers 66 bytes: the loads at offsets 63.. run past it
    std::vector<unsigned> r(128); hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
    int bad_g = 0, bad_b = 0;
    for (int i = 0; i < 64; ++i) {
        const unsigned want = h[i] | (h[i + 1] << 8) | (h[i + 2] << 16) | ((unsigned)h[i + 3] << 24);
        if (r[2 * i] != want) ++bad_g;
        if (i + 4 <= 66 && r[2 * i + 1] != want) ++bad_b;
        if (i >= 60) printf("offset %d: global %08x buffer %08x want %08x\n", i, r[2 * i], r[2 * i + 1], want);
    }
    printf("unaligned dword loads: global %s (%d wrong), buffer in-range %s (%d wrong)\n", bad_g ? "WRONG" : "ok", bad_g, bad_b ? "WRONG" : "ok", bad_b);
    return 0;
}
