// vt_blocks_tile.h -- the transformer stack for SMALL batches: one wave per 16-token tile, tiles spread over the chip.
//
// blocks_kernel (vt_blocks.h) gives a frame one workgroup on one CU: its latency is a CU's worth of MFMA issue per frame
// (G256: 243 us) however few frames there are -- a B = 1 client (the reference harness's tracker plugin) leaves 255 CUs idle.
// Here a block is two launches over (tile, frame) workgroups of ONE wave each:
//     tile_qkv_kernel       LayerNorm-1 + qkv of the tile        -> q, K image, V^T image in a global workspace
//     tile_attn_mlp_kernel  softmax(q K^T) V + proj + residual, LayerNorm-2 + MLP + residual of the tile (all keys from the
//                           workspace, L2-resident), the final LayerNorm after the last block
// The kernel boundary is the only synchronisation (K / V of every tile must exist before any tile attends).  Same arithmetic,
// same operand images and the same accumulation order as the tiles of blocks_kernel's plain path; weights and the small
// parameters come straight from L2 (one wave per CU: nothing to stage for).  Used when frames x tiles is far below the number
// of SIMDs (vittrack.hip::run_blocks), e.g. G256 B = 1: 243 -> ~70 us.
#pragma once
#include "vt_blocks.h"

namespace vtb {

// zc: exact template cache of block 0 (vt_set_template): 0 off, 1 compute and store, 2 load instead of computing
template <int NT>
__global__ __launch_bounds__(64) void tile_qkv_kernel(const float* __restrict__ x,        // (B, L, C) residual stream
                                                    const float* __restrict__ P,        // this block's parameters
                                                    f4* __restrict__ qb, f4* __restrict__ kb, f4* __restrict__ vb,
                                                    float* __restrict__ zcache, int zc, int len_z) {
    constexpr int L = NT * 16;
    const int T = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, tok = lane & 15, q = lane >> 4;
    f4* const qo = qb + ((size_t)b * NT + T) * NC * 64 + lane;          // [b][T][ot][lane]
    f4* const ko = kb + ((size_t)b * NT + T) * NC * 64 + lane;          // [b][T][ot][lane]
    f4* const vo = vb + (size_t)b * NC * NT * 64 + (size_t)T * 64 + lane;   // [b][ot][T][lane]: + ot * NT * 64
    const bool z_tile = zc != 0 && 16 * T < len_z;
    f4* const zt = reinterpret_cast<f4*>(zcache) + (((size_t)b * (len_z >> 4) + T) * 3 * NC) * 64 + lane;
    if (z_tile && zc == 2) {
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) {
            qo[ot * 64] = zt[ot * 64];
            ko[ot * 64] = zt[(NC + ot) * 64];
            vo[(size_t)ot * NT * 64] = zt[(2 * NC + ot) * 64];
        }
        return;
    }
    f4 xr[NC], h[NC];
    {
        const float* src = x + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
        for (int c = 0; c < NC; ++c) xr[c] = ld4(src + 16 * c);
    }
    layer_norm_img(xr, h, P + O_LN1G, P + O_LN1B, q);
    {   // q and k: rows = features, cols = tokens (B = h shared)
        f4 acc[2 * NC];
#pragma unroll
        for (int ot = 0; ot < 2 * NC; ++ot) acc[ot] = ld4(P + O_BQKV + 16 * ot + 4 * q);
        gemm_stage<NC, 2 * NC, true, false>(
            [&](int c, f4 (&a)[2 * NC]) {
#pragma unroll
                for (int ot = 0; ot < 2 * NC; ++ot) a[ot] = wimg(P + O_WQKV, ot * NC + c, lane);
            },
            [&](int c) { return h[c]; }, acc);
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) {
            qo[ot * 64] = acc[ot];
            ko[ot * 64] = acc[NC + ot];
            if (z_tile) { zt[ot * 64] = acc[ot]; zt[(NC + ot) * 64] = acc[NC + ot]; }
        }
    }
    {   // v, operands swapped: rows = tokens, cols = v features (A = h shared) -> V^T image
        f4 acc[NC];
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) acc[ot] = splat4(P[O_BQKV + 2 * C + 16 * ot + tok]);
        gemm_stage<NC, NC, false, false>(
            [&](int c, f4 (&bw)[NC]) {
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) bw[ot] = wimg(P + O_WQKV, (2 * NC + ot) * NC + c, lane);
            },
            [&](int c) { return h[c]; }, acc);
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) {
            vo[(size_t)ot * NT * 64] = acc[ot];
            if (z_tile) zt[(2 * NC + ot) * 64] = acc[ot];
        }
    }
}

// xin: residual stream read (the caller's tokens in block 0), xout: residual stream written (the model's workspace; may equal
// xin: a workgroup touches only its own tile).  normP != null: this is the last executed block -- feat receives
// LayerNorm(x) of the search tiles and resid (optional) the residual stream.  skip_z: the template tiles stop here (their
// attention / MLP output never reaches the head, vit_dist.py:126).
template <int NT>
__global__ __launch_bounds__(64) void tile_attn_mlp_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                         const float* __restrict__ P, const f4* __restrict__ qb,
                                                         const f4* __restrict__ kb, const f4* __restrict__ vb,
                                                         const float* __restrict__ normP, float* __restrict__ feat,
                                                         float* __restrict__ resid, int len_z, int skip_z) {
    constexpr int L = NT * 16;
    constexpr float SCALE_LOG2E = 0.14433756729740643f * 1.4426950408889634f;   // 48^-0.5 (attn.py:15) * log2 e
    const int T = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, tok = lane & 15, q = lane >> 4;
    if (skip_z && 16 * T < len_z) return;
    f4 x[NC], qr[NC];
    {
        const float* src = xin + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
        const f4* qs = qb + ((size_t)b * NT + T) * NC * 64 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) { x[c] = ld4(src + 16 * c); qr[c] = qs[c * 64]; }
    }
    const f4* const kf = kb + (size_t)b * NT * NC * 64 + lane;          // [J][c]
    const f4* const vf = vb + (size_t)b * NC * NT * 64 + lane;          // [t][J]
    // ---- attention (the softmax arithmetic of blocks_kernel: raw scores, one packed fma per two of them)
    f4 s[NT];
    float m0 = -3.0e38f, m1 = -3.0e38f;
    constexpr int JG = 5;
    static_assert(NT % JG == 0, "NT must be a multiple of 5");
#pragma unroll
    for (int j0 = 0; j0 < NT; j0 += JG) {
        f4 acc[JG];
#pragma unroll
        for (int j = 0; j < JG; ++j) acc[j] = splat4(0.f);
        gemm_stage<NC, JG, true, false>(
            [&](int c, f4 (&a)[JG]) {
#pragma unroll
                for (int j = 0; j < JG; ++j) a[j] = kf[((j0 + j) * NC + c) * 64];
            },
            [&](int c) { return qr[c]; }, acc);
#pragma unroll
        for (int j = 0; j < JG; ++j) {
            s[j0 + j] = acc[j];
            m0 = fmaxf(fmaxf(m0, acc[j].x), acc[j].y);
            m1 = fmaxf(fmaxf(m1, acc[j].z), acc[j].w);
        }
    }
    const float m = quad_max(fmaxf(m0, m1));
    const f2 k2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-m * SCALE_LOG2E, -m * SCALE_LOG2E};
    f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
    for (int J = 0; J < NT; ++J) {
        const f2 a = __builtin_elementwise_fma(f2{s[J].x, s[J].y}, k2, nm2);
        const f2 c = __builtin_elementwise_fma(f2{s[J].z, s[J].w}, k2, nm2);
        const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
        s[J] = f4{ea.x, ea.y, ec.x, ec.y};
        d0 += ea;
        d1 += ec;
    }
    const f2 dd = d0 + d1;
    const float rden = __builtin_amdgcn_rcpf(quad_sum(dd.x + dd.y));
    f4 o[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
    gemm_stage<NT, NC, true, false>(
        [&](int J, f4 (&a)[NC]) {
#pragma unroll
            for (int t = 0; t < NC; ++t) a[t] = vf[(t * NT + J) * 64];
        },
        [&](int J) { return s[J]; }, o);
#pragma unroll
    for (int t = 0; t < NC; ++t) o[t] = o[t] * splat4(rden);
    // ---- proj + residual
#pragma unroll
    for (int ot = 0; ot < NC; ++ot) x[ot] = x[ot] + ld4(P + O_BPROJ + 16 * ot + 4 * q);
    gemm_stage<NC, NC, true, false>(
        [&](int c, f4 (&a)[NC]) {
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) a[ot] = wimg(P + O_WPROJ, ot * NC + c, lane);
        },
        [&](int c) { return o[c]; }, x);
    // ---- LayerNorm-2 + MLP + residual (two groups of 6 hidden tiles, as blocks_kernel's plain path)
    {
        f4 h[NC];
        layer_norm_img(x, h, P + O_LN2G, P + O_LN2B, q);
        f4 hd[NH];
        constexpr int G6 = 6;
#pragma unroll
        for (int g = 0; g < NH; g += G6) {
            f4 acc[G6];
#pragma unroll
            for (int j = 0; j < G6; ++j) acc[j] = ld4(P + O_B1 + 16 * (g + j) + 4 * q);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f4 a[G6];
#pragma unroll
                for (int j = 0; j < G6; ++j) a[j] = wimg(P + O_W1, (g + j) * NC + c, lane);
                mfma4_shared_b(a, h[c], acc);
            }
#pragma unroll
            for (int j = 0; j < G6; ++j)
                hd[g + j] = f4{gelu_erf(acc[j].x), gelu_erf(acc[j].y), gelu_erf(acc[j].z), gelu_erf(acc[j].w)};
        }
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) x[ot] = x[ot] + ld4(P + O_B2 + 16 * ot + 4 * q);
#pragma unroll
        for (int c = 0; c < NH; ++c) {
            f4 a[NC];
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) a[ot] = wimg(P + O_W2, ot * NH + c, lane);
            mfma4_shared_b(a, hd[c], x);
        }
    }
    {
        float* dst = xout + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
        for (int c = 0; c < NC; ++c) st4(dst + 16 * c, x[c]);
    }
    if (normP != nullptr) {
        if (resid != nullptr) {
            float* dst = resid + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
            for (int c = 0; c < NC; ++c) st4(dst + 16 * c, x[c]);
        }
        if (16 * T >= len_z) {
            f4 h[NC];
            layer_norm_img(x, h, normP, normP + C, q);
            float* dst = feat + ((size_t)b * (L - len_z) + (16 * T - len_z) + tok) * C + 4 * q;
#pragma unroll
            for (int c = 0; c < NC; ++c) st4(dst + 16 * c, h[c]);
        }
    }
}

}  // namespace vtb
