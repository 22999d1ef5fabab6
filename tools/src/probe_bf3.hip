// A three-piece bf16 split of fp32 operands (x = h + m + l exactly, 8 mantissa bits each) on the bf16 matrix pipe instead of
// v_mfma_f32_16x16x4_f32: six products (hh, hm, mh, hl, lh, mm) reproduce an fp32 contraction to ~2^-23.
//   * issue cost per SIMD of v_mfma_f32_16x16x16_bf16 (legacy K) and v_mfma_f32_16x16x32_bf16 against v_mfma_f32_16x16x4_f32
//   * the same with the split of one operand (f4 -> 3 x bf16x4, ~22 VALU) interleaved: does the VALU work hide under bf16 MFMAs?
//   * accuracy of the six-term product against fp64 on random data, next to the fp32 MFMA's
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(f4 x, s4& h, s4& m, s4& l) {      // truncating split: x = h + m + l exactly
    unsigned xb[4], hb[4], r1b[4], mb[4], r2b[4];
    float r1[4], r2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xb[i] = __float_as_uint(x[i]);
        hb[i] = xb[i] & 0xffff0000u;
        r1[i] = x[i] - __uint_as_float(hb[i]);
        r1b[i] = __float_as_uint(r1[i]);
        mb[i] = r1b[i] & 0xffff0000u;
        r2[i] = r1[i] - __uint_as_float(mb[i]);
        r2b[i] = __float_as_uint(r2[i]);
    }
    u2 hp = {__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
    u2 mp = {__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u)};
    u2 lp = {__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u)};
    h = __builtin_bit_cast(s4, hp); m = __builtin_bit_cast(s4, mp); l = __builtin_bit_cast(s4, lp);
}

// MODE 0: 16 x f32 16x16x4 per iteration (mfma4 on 4 output tiles); 1: 24 x bf16 16x16x16 (6 terms x 4 tiles), operands pre-split;
// 2: as 1 + the split of the shared operand each iteration; 3: the split alone; 4: 12 x bf16 16x16x32 (6 terms x 4 tiles, two chunks
// per instruction -> per-chunk cost = half); 5: as 4 + two splits per iteration
template <int MODE>
__global__ __launch_bounds__(256) void cyc(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ st, int iters) {
    const int lane = threadIdx.x & 63;
    f4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f} + (float)j;
    f4 bx = *reinterpret_cast<const f4*>(src + 4 * lane), ax[4];
    for (int j = 0; j < 4; ++j) ax[j] = *reinterpret_cast<const f4*>(src + 256 + 256 * j + 4 * lane);
    s4 ah[4], am[4], al[4], bh, bm, bl;
    for (int j = 0; j < 4; ++j) split3(ax[j], ah[j], am[j], al[j]);
    split3(bx, bh, bm, bl);
    b8 ah8[4], am8[4], al8[4], bh8, bm8, bl8;
    for (int j = 0; j < 4; ++j) {
        ah8[j] = __builtin_bit_cast(b8, __builtin_shufflevector(ah[j], ah[j], 0, 1, 2, 3, 4, 5, 6, 7));
        am8[j] = __builtin_bit_cast(b8, __builtin_shufflevector(am[j], am[j], 0, 1, 2, 3, 4, 5, 6, 7));
        al8[j] = __builtin_bit_cast(b8, __builtin_shufflevector(al[j], al[j], 0, 1, 2, 3, 4, 5, 6, 7));
    }
    bh8 = __builtin_bit_cast(b8, __builtin_shufflevector(bh, bh, 0, 1, 2, 3, 4, 5, 6, 7));
    bm8 = __builtin_bit_cast(b8, __builtin_shufflevector(bm, bm, 0, 1, 2, 3, 4, 5, 6, 7));
    bl8 = __builtin_bit_cast(b8, __builtin_shufflevector(bl, bl, 0, 1, 2, 3, 4, 5, 6, 7));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][r], bx[r], acc[j], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3 || MODE == 5) {
            bx = bx * 1.0001f;                      // a new operand every iteration
            split3(bx, bh, bm, bl);
            if (MODE == 5) { s4 h2, m2, l2; split3(bx + 1.f, h2, m2, l2); bh8 = __builtin_bit_cast(b8, __builtin_shufflevector(bh, h2, 0, 1, 2, 3, 4, 5, 6, 7));
                bm8 = __builtin_bit_cast(b8, __builtin_shufflevector(bm, m2, 0, 1, 2, 3, 4, 5, 6, 7)); bl8 = __builtin_bit_cast(b8, __builtin_shufflevector(bl, l2, 0, 1, 2, 3, 4, 5, 6, 7)); }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[j], bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[j], bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am[j], bm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am[j], bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[j], bm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[j], bh, acc[j], 0, 0, 0);
            }
        }
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al8[j], bh8, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah8[j], bl8, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am8[j], bm8, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am8[j], bh8, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah8[j], bm8, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah8[j], bh8, acc[j], 0, 0, 0);
            }
        }
        if (MODE == 3) { acc[0].x += __builtin_bit_cast(float, u2{(unsigned)bh[0], 0}[0]) + (float)bm[1] + (float)bl[2]; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + bx.x;
    if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// accuracy: D = A (16 x 16) * B (16 x 16) for one chunk of 16 k, f32 MFMA vs six-term bf16, against fp64
__global__ void acc_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D32, float* __restrict__ D6) {
    const int lane = threadIdx.x, rc = lane & 15, q = lane >> 4;
    f4 a, b;
    for (int r = 0; r < 4; ++r) { a[r] = A[rc * 16 + 4 * q + r]; b[r] = B[(4 * q + r) * 16 + rc]; }
    f4 d = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) {       // f32: k-step r of lane (rc, q) must be k = q within each 16x16x4 -> rebuild operands per k-step
        f4 z = {0, 0, 0, 0};
        (void)z;
    }
    // f32 path with the kernels' operand image convention: element r of lane (rc, q) is k = 4 r' ... use four MFMAs over k = 4 q + r by
    // feeding a[r], b[r]: MFMA r contracts k in {4*0 + r, 4*1 + r, 4*2 + r, 4*3 + r} (lane q supplies k = 4 q + r)
    for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r], d, 0, 0, 0);
    s4 ah, am, al, bh, bm, bl;
    split3(a, ah, am, al); split3(b, bh, bm, bl);
    f4 e = {0, 0, 0, 0};
    e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, e, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, e, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bm, e, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bh, e, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bm, e, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, e, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { D32[(4 * q + r) * 16 + rc] = d[r]; D6[(4 * q + r) * 16 + rc] = e[r]; }
}

template <int MODE>
void run(const char* name, const float* src, float* out, unsigned long long* st, int wps) {
    const int iters = 2000, wgs = 256 * wps;
    hipLaunchKernelGGL((cyc<MODE>), dim3(wgs), dim3(256), 0, 0, src, out, st, iters);
    hipLaunchKernelGGL((cyc<MODE>), dim3(wgs), dim3(256), 0, 0, src, out, st, iters);
    std::vector<unsigned long long> h(wgs * 4); hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= h.size();
    printf("%-78s %d wave(s)/SIMD: %7.1f cycles per iteration per wave, %7.1f per SIMD\n", name, wps, s / iters, s / iters / 1.0 * 1.0 / 1.0 * (1.0));
}
int main() {
    float *src, *out; unsigned long long* st;
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 4 * 8);
    std::vector<float> h(8192); std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = nd(rng);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 3}) {
        run<0>("f32: 16 x v_mfma_f32_16x16x4_f32 (one 16-k chunk x 4 output tiles)", src, out, st, wps);
        run<1>("bf16 x 3: 24 x v_mfma_f32_16x16x16_bf16 (the same chunk, six terms), operands pre-split", src, out, st, wps);
        run<2>("   + the shared operand split every iteration (f4 -> 3 x bf16x4)", src, out, st, wps);
        run<3>("   the split alone", src, out, st, wps);
        run<4>("bf16 x 3: 24 x v_mfma_f32_16x16x32_bf16 (TWO chunks, six terms)", src, out, st, wps);
        run<5>("   + two splits every iteration", src, out, st, wps);
    }
    // accuracy
    float *A, *B, *D32, *D6; hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&D32, 1024); hipMalloc(&D6, 1024);
    double e32 = 0, e6 = 0, ref_max = 0;
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<float> a(256), b(256), d32(256), d6(256);
        for (auto& v : a) v = nd(rng) * std::exp(nd(rng));
        for (auto& v : b) v = nd(rng) * std::exp(nd(rng));
        hipMemcpy(A, a.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(B, b.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, A, B, D32, D6);
        hipMemcpy(d32.data(), D32, 1024, hipMemcpyDeviceToHost); hipMemcpy(d6.data(), D6, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double r = 0, mag = 0; for (int k = 0; k < 16; ++k) { r += (double)a[i * 16 + k] * b[k * 16 + j]; mag += std::fabs((double)a[i * 16 + k] * b[k * 16 + j]); }
            // D[k-row layout]: D32[(row)*16 + col] with row = output i? the kernel writes D[(4q+r)*16 + rc]: row index = 4q+r, col = rc
            e32 = std::fmax(e32, std::fabs(d32[i * 16 + j] - r) / mag); e6 = std::fmax(e6, std::fabs(d6[i * 16 + j] - r) / mag); ref_max = std::fmax(ref_max, mag);
        }
    }
    printf("max |error| / sum|a b| over 200 random 16 x 16 x 16 products: f32 MFMA %.3g   six-term bf16 %.3g   (2^-24 = %.3g)\n", e32, e6, std::ldexp(1.0, -24));
    return 0;
}
