// vb_misc.h -- the HBM-bound kernels around the GEMMs of the ViT-Base path: patch gathering, LayerNorm, head tail.
#pragma once
#include "vb_gemm.h"

namespace vbm {

using vbg::bf16;
using vbg::bf16x4;
using vbg::bf16x8;

// ---------------------------------------------------------------------------------------------- patches
// PatchEmbed's Conv2d(3, C, 16, stride 16) (lib/models/layers/patch_embed.py:20-32) as a GEMM: this kernel lays the
// crops out as its left operand P[m = frame * L + token][k = c * 256 + r * 16 + s] in bf16 (k order = the conv weight's
// own [3][16][16] order), template tokens first (combine_tokens 'direct', lib/models/ostrack/utils.py:13-14).
// One thread = 8 consecutive s of one (m, c, r): 32 B in, 16 B out, writes fully coalesced.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                       bf16* __restrict__ P, int B, int Tz, int Tx) {
    const int gz = Tz / 16, gx = Tx / 16, Lz = gz * gz, L = Lz + gx * gx;
    const size_t total = (size_t)B * L * 96;                    // 96 = 3 * 16 * 2 chunks of 8 per patch
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i % 96);
        const size_t m = i / 96;
        const int t = (int)(m % L), f = (int)(m / L);
        const int c = ch >> 5, r = (ch >> 1) & 15, s0 = (ch & 1) * 8;
        const float* src;
        if (t < Lz) {
            const int py = t / gz, px = t - py * gz;
            src = z + (((size_t)f * 3 + c) * Tz + py * 16 + r) * Tz + px * 16 + s0;
        } else {
            const int tt = t - Lz, py = tt / gx, px = tt - py * gx;
            src = x + (((size_t)f * 3 + c) * Tx + py * 16 + r) * Tx + px * 16 + s0;
        }
        const bf16x4 lo = vbg::to_bf16x4(ld4(src)), hi = vbg::to_bf16x4(ld4(src + 4));
        *reinterpret_cast<bf16x8*>(P + m * 768 + ch * 8) = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
}

// -------------------------------------------------------------------------------------------- LayerNorm
// nn.LayerNorm(C, eps) over the f32 residual stream (lib/models/ostrack/vit.py:78,82,130; eps 1e-6): biased variance,
// two passes over registers.  One wave per token row, C = 768 = 64 lanes x 3 float4.
//   xn   (optional)  bf16 [M][C]                                   -> the next GEMM's left operand
//   map  (optional)  bf16 zero-bordered NHWC [B][F+2][F+2][C]: search rows only (forward_head's (B,C,F,F) view,
//                    lib/models/ostrack/ostrack.py:126-129)         -> the head's implicit-GEMM input
//   feat (optional)  f32 [B][Lx][C]: search rows only               -> stage API / tests
//   xb + rstd (optional, together)  the LayerNorm folded into the next GEMM (vb_gemm.h): bf16 copy of the RAW row and its
//                    1 / sqrt(var + eps) -- what the residual-writing GEMM epilogues + ln_finalize_kernel produce, for a
//                    residual stream that arrived from outside (vt_blocks on caller-supplied tokens)
template <int C>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ resid, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int M, int L, int Lz, int F,
                                                        bf16* __restrict__ xn, bf16* __restrict__ map, float* __restrict__ feat,
                                                        bf16* __restrict__ xb, float* __restrict__ rstd_out, float* __restrict__ mean_out) {
    static_assert(C % 256 == 0, "C = 64 lanes x float4 x n");
    constexpr int NV = C / 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* p = resid + (size_t)row * C;
    f4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = ld4(p + i * 256 + lane * 4);
        s += hsum4(v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = v[i] - splat4(mean);
        ss += hsum4(v[i] * v[i]);
        // the raw rows' bf16 copy for the LayerNorm-folded GEMMs, CENTRED (vb_gemm.h Args::cm): the folded weights ignore a per-row constant
        if (xb) *reinterpret_cast<bf16x4*>(xb + (size_t)row * C + i * 256 + lane * 4) = vbg::to_bf16x4(v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / C) + eps);
    if (rstd_out) {
        if (lane == 0) { rstd_out[row] = rstd; if (mean_out) mean_out[row] = mean; }
        if (!xn && !map && !feat) return;
    }
    const int f = row / L, t = row - f * L;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        const f4 y = v[i] * splat4(rstd) * ld4(gamma + c) + ld4(beta + c);
        if (xn) *reinterpret_cast<bf16x4*>(xn + (size_t)row * C + c) = vbg::to_bf16x4(y);
        if (t >= Lz) {
            const int tt = t - Lz;
            if (map) {
                const int yy = tt / F, xx = tt - yy * F, P = F + 2;
                *reinterpret_cast<bf16x4*>(map + ((size_t)(f * P + yy + 1) * P + xx + 1) * C + c) = vbg::to_bf16x4(y);
            }
            if (feat) st4(feat + ((size_t)f * (L - Lz) + tt) * C + c, y);
        }
    }
}

// Per-row rstd from the (sum, centred sum of squares) pairs the residual-writing GEMM epilogues leave per 64-column wave slice
// (vb_gemm.h): Chan's pairwise update, exact mean first.  stats: [P][ld] float2, P = C / 64.  One thread per row.
template <int P>
__global__ __launch_bounds__(256) void ln_finalize_kernel(const vbg::f2* __restrict__ stats, int ld, int M, float eps, float* __restrict__ rstd,
                                                          float* __restrict__ mean_out) {      // mean_out: the next residual-writing GEMM centres its bf16 copy on it
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    vbg::f2 v[P];           // all P pairs requested at once: one memory round trip per row
#pragma unroll
    for (int p = 0; p < P; ++p) v[p] = stats[(size_t)p * ld + m];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) s += v[p].x;
    const float inv_c = 1.0f / (64.0f * P), mean = s * inv_c;
    float m2 = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float d = v[p].x * (1.0f / 64.0f) - mean;
        m2 += v[p].y + 64.0f * d * d;
    }
    rstd[m] = 1.0f / sqrtf(m2 * inv_c + eps);
    if (mean_out) mean_out[m] = mean;
}

// f32 (B, Lx, C) tokens -> the zero-bordered bf16 map (stage API: vt_head on caller-supplied features)
__global__ __launch_bounds__(256) void feat_to_map_kernel(const float* __restrict__ feat, bf16* __restrict__ map, int B, int F, int C) {
    const size_t total = (size_t)B * F * F * (C / 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (C / 4)) * 4;
        const size_t px = i / (C / 4);
        const int x = (int)(px % F), y = (int)((px / F) % F), b = (int)(px / ((size_t)F * F)), P = F + 2;
        *reinterpret_cast<bf16x4*>(map + ((size_t)(b * P + y + 1) * P + x + 1) * C + c) = vbg::to_bf16x4(ld4(feat + px * C + c));
    }
}

// ------------------------------------------------------------------------------------------- head tail
// conv5_{ctr,offset,size}: 1x1 conv W/8 -> 1 / 2 / 2 (+bias), then sigmoid + clamp on ctr and size
// (lib/models/layers/head.py:175-201).  t4: [3 towers][B * F * F][CW] bf16 (towers in the order ctr, offset, size);
// w5: [5][CW] f32 rows = ctr, offset0, offset1, size0, size1;  b5: [5].  One thread per pixel.
template <int CW>
__global__ __launch_bounds__(256) void conv5_kernel(const bf16* __restrict__ t4, const float* __restrict__ w5,
                                                    const float* __restrict__ b5, int npix_total, int FF, size_t tower_stride,
                                                    float* __restrict__ score, float* __restrict__ size, float* __restrict__ offset) {
    __shared__ float sw[5 * CW + 5];
    for (int i = threadIdx.x; i < 5 * CW + 5; i += 256) sw[i] = i < 5 * CW ? w5[i] : b5[i - 5 * CW];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npix_total) return;
    float o[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) o[j] = sw[5 * CW + j];
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) {
        const bf16* src = t4 + (size_t)tw * tower_stride + (size_t)p * CW;    // tower_stride: elements between towers
#pragma unroll
        for (int c8 = 0; c8 < CW / 8; ++c8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (float)v[e];
                if (tw == 0) o[0] = fmaf(a, sw[0 * CW + c8 * 8 + e], o[0]);
                if (tw == 1) { o[1] = fmaf(a, sw[1 * CW + c8 * 8 + e], o[1]); o[2] = fmaf(a, sw[2 * CW + c8 * 8 + e], o[2]); }
                if (tw == 2) { o[3] = fmaf(a, sw[3 * CW + c8 * 8 + e], o[3]); o[4] = fmaf(a, sw[4 * CW + c8 * 8 + e], o[4]); }
            }
        }
    }
    const int b = p / FF, px = p - b * FF;
    score[p] = sigmoid_clamped(o[0]);
    offset[((size_t)b * 2 + 0) * FF + px] = o[1];
    offset[((size_t)b * 2 + 1) * FF + px] = o[2];
    size[((size_t)b * 2 + 0) * FF + px] = sigmoid_clamped(o[3]);
    size[((size_t)b * 2 + 1) * FF + px] = sigmoid_clamped(o[4]);
}

}  // namespace vbm
