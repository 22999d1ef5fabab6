// vt_stem.h -- LeViT-style patch embedding: 4 x (conv3x3 stride 2 pad 1 + folded BN), Hardswish
// after the first three, tokenise, add pos-embed, write [z ; x] rows of the token matrix.
//
// Replaces Conv2d_BN / b16 / LevitPatchEmbedding.forward and the pos-embed add + cat of
// OstrackDist.forward (lib/models/vit_dist/vit_dist.py:10-54,78-84).
//
// v1 kernels: direct convolution, one thread per output pixel and output-channel group.  The
// folded weights of a group are wave-uniform, laid out [group][tap][cin][OCG] so the compiler
// fetches them with scalar loads and feeds them to v_fma_f32 as SGPR operands; activations
// between layers are channels-last (NHWC) so a thread's CIN inputs per tap are one or two
// 16-byte loads.  The first layer reads the NCHW crops the boundary hands over.
#pragma once
#include "vt_common.h"

namespace vts {

template <int CIN, bool NCHW>
__device__ __forceinline__ void load_pixel(const float* __restrict__ in, int b, int iy, int ix, int H, int W,
                                           float (&v)[CIN]) {
    if constexpr (NCHW) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[c] = in[(((size_t)b * CIN + c) * H + iy) * W + ix];
    } else {
        const float* p = in + (((size_t)b * H + iy) * W + ix) * CIN;
        if constexpr (CIN % 4 == 0) {
#pragma unroll
            for (int c = 0; c < CIN; c += 4) {
                f4 t = ld4(p + c);
                v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CIN; c += 2) {
                float2 t = *reinterpret_cast<const float2*>(p + c);
                v[c] = t.x; v[c + 1] = t.y;
            }
        }
    }
}

// out: NHWC (B, H/2, W/2, COUT), or with TOKENS the token matrix rows [tok_off, tok_off + HoWo)
// of (B, L, COUT) with pos (HoWo, COUT) added.
template <int CIN, int COUT, int OCG, bool NCHW, bool HSWISH, bool TOKENS>
__global__ __launch_bounds__(256) void conv_s2_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                      const float* __restrict__ bp, float* __restrict__ out,
                                                      int B, int H, int W, const float* __restrict__ pos,
                                                      int tok_off, int L) {
    static_assert(COUT % OCG == 0, "group size");
    const int Ho = H >> 1, Wo = W >> 1;
    const int g = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * Ho * Wo) return;
    const int qx = idx % Wo, py = (idx / Wo) % Ho, b = idx / (Wo * Ho);

    float acc[OCG];
#pragma unroll
    for (int j = 0; j < OCG; ++j) acc[j] = bp[g * OCG + j];
    const float* __restrict__ wg = wp + (size_t)g * 9 * CIN * OCG;

#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = 2 * py + r - 1;
        if (iy < 0 || iy >= H) continue;      // zero padding: skipped taps add exactly 0
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int ix = 2 * qx + s - 1;
            if (ix < 0 || ix >= W) continue;
            float v[CIN];
            load_pixel<CIN, NCHW>(in, b, iy, ix, H, W, v);
            const float* __restrict__ wt = wg + (r * 3 + s) * CIN * OCG;
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int j = 0; j < OCG; ++j) acc[j] = fmaf(v[c], wt[c * OCG + j], acc[j]);
        }
    }
    if constexpr (HSWISH) {
#pragma unroll
        for (int j = 0; j < OCG; ++j) acc[j] = hardswish(acc[j]);
    }
    float* dst;
    if constexpr (TOKENS) {
        const int t = py * Wo + qx;
        const float* pp = pos + (size_t)t * COUT + g * OCG;
#pragma unroll
        for (int j = 0; j < OCG; ++j) acc[j] += pp[j];
        dst = out + ((size_t)b * L + tok_off + t) * COUT + g * OCG;
    } else {
        dst = out + (size_t)idx * COUT + g * OCG;
    }
    if constexpr (OCG % 4 == 0) {
#pragma unroll
        for (int j = 0; j < OCG; j += 4) st4(dst + j, f4{acc[j], acc[j + 1], acc[j + 2], acc[j + 3]});
    } else {
#pragma unroll
        for (int j = 0; j < OCG; j += 2) *reinterpret_cast<float2*>(dst + j) = float2{acc[j], acc[j + 1]};
    }
}

}  // namespace vts
