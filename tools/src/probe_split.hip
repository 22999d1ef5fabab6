// probe_split -- vt3::split3 as the kernels run it (vt_bf3.h, VT_SPLIT_DOT2: residuals by v_dot2c_f32_bf16) against the v_and / v_sub
// form it replaced, bit for bit, on the hardware: random mantissas at EVERY fp32 exponent (denormals, zeros, both signs), plus the
// adversarial mantissas (all ones below a piece boundary).  Also times the two forms on a VALU-only loop.
//   hipcc -O3 --offload-arch=gfx950 -I vittracker_amd/csrc -o tools/bin/probe_split tools/src/probe_split.hip && ./tools/bin/probe_split
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../vittracker_amd/csrc/vt_bf3.h"

__device__ __forceinline__ void split3_ref(f4 x, vt3::u32x2& h, vt3::u32x2& m, vt3::u32x2& l) {
    unsigned xb[4], r1b[4], r2b[4];
    for (int i = 0; i < 4; ++i) {
        xb[i] = __float_as_uint(x[i]);
        const float r1 = x[i] - __uint_as_float(xb[i] & 0xffff0000u);
        r1b[i] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(r1b[i] & 0xffff0000u);
        r2b[i] = __float_as_uint(r2);
    }
    h = vt3::u32x2{__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
    m = vt3::u32x2{__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u)};
    l = vt3::u32x2{__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u)};
}

__global__ void check_kernel(const f4* x, unsigned* out, int n) {      // out[6 i ..]: h, m, l of the kernel form; then the reference form
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vt3::u32x2 h, m, l, hr, mr, lr;
    vt3::split3(x[i], h, m, l);
    split3_ref(x[i], hr, mr, lr);
    unsigned* o = out + 12 * (size_t)i;
    o[0] = h.x; o[1] = h.y; o[2] = m.x; o[3] = m.y; o[4] = l.x; o[5] = l.y;
    o[6] = hr.x; o[7] = hr.y; o[8] = mr.x; o[9] = mr.y; o[10] = lr.x; o[11] = lr.y;
}

template <bool REF>
__global__ void time_kernel(f4* x, int iters) {
    f4 v = x[threadIdx.x];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        vt3::u32x2 h, m, l;
        if (REF) split3_ref(v, h, m, l);
        else vt3::split3(v, h, m, l);
        acc ^= h.x ^ h.y ^ m.x ^ m.y ^ l.x ^ l.y;
        v.x = __uint_as_float((__float_as_uint(v.x) ^ (acc & 0x7fffu)));      // a dependency, so the loop is not hoisted
        asm volatile("" : "+v"(v));
    }
    if (acc == 0x12345u) x[threadIdx.x] = v;
}

int main() {
    std::vector<float> hx;
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (unsigned e = 0; e < 255; ++e)            // every finite exponent field, 0 = zeros / denormals
        for (int k = 0; k < 512; ++k) {
            unsigned mant = rnd() & 0x7fffffu;
            if (k == 0) mant = 0;
            if (k == 1) mant = 0x7fffffu;
            if (k == 2) mant = 0x00ffffu;        // leading piece 1.0000000, all ones below it
            if (k == 3) mant = 0x0000ffu;
            if (k == 4) mant = 0x7f0000u;
            if (k == 5) mant = 0x00ff00u;
            const unsigned bits = ((rnd() & 1u) << 31) | (e << 23) | mant;
            float f;
            __builtin_memcpy(&f, &bits, 4);
            hx.push_back(f);
        }
    while (hx.size() % 4) hx.push_back(0.f);
    const int n4 = (int)hx.size() / 4;
    f4* dx;
    unsigned* dout;
    hipMalloc(reinterpret_cast<void**>(&dx), hx.size() * 4);
    hipMalloc(reinterpret_cast<void**>(&dout), (size_t)n4 * 12 * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3((n4 + 255) / 256), dim3(256), 0, nullptr, dx, dout, n4);
    std::vector<unsigned> out((size_t)n4 * 12);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    long bad_normal = 0, bad_tiny = 0, inexact = 0;
    for (int i = 0; i < n4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int w = j >> 1, sh = (j & 1) ? 16 : 0;
            unsigned p[3], q[3];
            for (int pc = 0; pc < 3; ++pc) {
                p[pc] = (out[12 * (size_t)i + 2 * pc + w] >> sh) & 0xffffu;
                q[pc] = (out[12 * (size_t)i + 6 + 2 * pc + w] >> sh) & 0xffffu;
            }
            unsigned xb;
            __builtin_memcpy(&xb, &hx[4 * (size_t)i + j], 4);
            const unsigned e = (xb >> 23) & 0xffu;
            const bool same = p[0] == q[0] && p[1] == q[1] && p[2] == q[2];
            if (!same) (e >= 40 ? bad_normal : bad_tiny)++;
            if (!same && e >= 40 && bad_normal <= 5) printf("  x %08x: kernel form %04x %04x %04x, reference %04x %04x %04x (element %d)\n", xb, p[0], p[1], p[2], q[0], q[1], q[2], j);      // below 2^-87 a residual can be denormal (flushed by either form's mode)
            // exactness of the kernel form's pieces: h + m + l == x (in double), for the exponents the nets live in
            if (e >= 40) {
                double sum = 0;
                for (int pc = 0; pc < 3; ++pc) { const unsigned b = p[pc] << 16; float f; __builtin_memcpy(&f, &b, 4); sum += f; }
                if (sum != (double)hx[4 * (size_t)i + j]) ++inexact;
            }
        }
    fflush(stdout);
    printf("values %zu: pieces differing from the v_and / v_sub form: %ld at exponents >= 2^-87, %ld below; h + m + l != x: %ld\n", hx.size(), bad_normal, bad_tiny, inexact);
    for (int ref = 0; ref < 2; ++ref) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (ref) hipLaunchKernelGGL(time_kernel<true>, dim3(256 * 4), dim3(256), 0, nullptr, dx, iters);
            else hipLaunchKernelGGL(time_kernel<false>, dim3(256 * 4), dim3(256), 0, nullptr, dx, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s form: %.3f ms for %d splits per lane, 4 waves per SIMD\n", ref ? "v_and / v_sub" : "vt3::split3 ", ms, iters);
    }
    return bad_normal == 0 && inexact == 0 ? 0 : 1;
}
