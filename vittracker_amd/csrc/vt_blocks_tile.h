// vt_blocks_tile.h -- the transformer stack for SMALL batches: one wave per 16-token tile, tiles spread over the chip.
//
// blocks_kernel (vt_blocks.h) gives a frame one workgroup on one CU: its latency is a CU's worth of MFMA issue per frame
// (G256: 243 us) however few frames there are -- a B = 1 client (the reference harness's tracker plugin) leaves 255 CUs idle.
// Here the stack is 1 + depth launches over (tile, frame) workgroups:
//     tile_qkv_kernel       (1 wave)  block 0's LayerNorm-1 + qkv of the tile -> q, K image, V^T image in a global workspace
//     tile_attn_mlp_kernel  (4 waves) softmax(q K^T) V + proj + residual, LayerNorm-2 + MLP + residual of the tile (all keys
//                           from the workspace, L2-resident); then the NEXT block's LayerNorm-1 + qkv of the tile into the
//                           other workspace set (per-token work), or the final LayerNorm after the last block
// The kernel boundary is the only synchronisation (K / V of every tile must exist before any tile attends).  Same arithmetic and
// operand images as blocks_kernel (the four-way split of a tile is its guest waves' split); weights and the small parameters
// come straight from L2.  Used when frames x tiles is well below what fills the chip (vittrack.hip::run_blocks): G256 B = 1
// 281 -> 92 us per step, B = 32 300 -> 146, B = 128 371 -> 323; G128 B = 1 81 -> 67, B = 32 83 -> 72.
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
    layer_norm_plain(xr, h);
    {   // q and k: rows = features, cols = tokens (B = h shared)
        f4 acc[2 * NC];
#pragma unroll
        for (int ot = 0; ot < 2 * NC; ++ot) acc[ot] = ld4(P + O_BQKV + 16 * ot + 4 * q);
        gemm_stage<NC, 2 * NC, true, false, f4>(
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
        gemm_stage<NC, NC, false, false, f4>(
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
//
// FOUR waves per tile: one wave's single instruction stream (a latency-bound chain of ~800 MFMAs with operands from L2) measured
// 25.7 us per launch at G256, the four-way split 9.  The split is the one blocks_kernel's guest waves use: keys four ways with
// a flash-style merge of the partial softmaxes, proj by output tile, the MLP by hidden tiles with fc2 as four partial sums
// added in a fixed order; LayerNorms and the residual stream are kept redundantly.
template <int NT>
__global__ __launch_bounds__(256) void tile_attn_mlp_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                          const float* __restrict__ P, const f4* __restrict__ qb,
                                                          const f4* __restrict__ kb, const f4* __restrict__ vb,
                                                          const float* __restrict__ normP, float* __restrict__ feat,
                                                          float* __restrict__ resid, int len_z, int skip_z,
                                                          const float* __restrict__ Pn,      // next block's parameters, or null
                                                          f4* __restrict__ qn, f4* __restrict__ kn, f4* __restrict__ vn) {
    constexpr int L = NT * 16;
    constexpr float SCALE_LOG2E = 0.14433756729740643f * 1.4426950408889634f;
    __shared__ f4 Pg[4 * NC * 64];        // partial P.V / partial fc2 outputs of the four waves
    __shared__ f4 Dg[NC * 64];            // proj output tiles
    __shared__ float Mg[4 * 2 * 64];      // partial softmax maxima (raw units) and sums
    const int T = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, tok = lane & 15, q = lane >> 4;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (skip_z && 16 * T < len_z) return;
    f4 x[NC], qr[NC];
    {
        const float* src = xin + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
        const f4* qs = qb + ((size_t)b * NT + T) * NC * 64 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) { x[c] = ld4(src + 16 * c); qr[c] = qs[c * 64]; }
    }
    const f4* const kf = kb + (size_t)b * NT * NC * 64 + lane;
    const f4* const vf = vb + (size_t)b * NC * NT * 64 + lane;
    // ---- attention over this wave's keys -> partial (max, sum, P.V)
    auto attn_part = [&](auto njc, int J0) {
        constexpr int NJ = decltype(njc)::value;
        f4 sc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) sc[j] = splat4(0.f);
        gemm_stage<NC, NJ, true, false, f4>(
            [&](int c, f4 (&a)[NJ]) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) a[j] = kf[((J0 + j) * NC + c) * 64];
            },
            [&](int c) { return qr[c]; }, sc);
        float m0 = -3.0e38f, m1 = -3.0e38f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            m0 = fmaxf(fmaxf(m0, sc[j].x), sc[j].y);
            m1 = fmaxf(fmaxf(m1, sc[j].z), sc[j].w);
        }
        const float mraw = quad_max(fmaxf(m0, m1));
        const f2 k2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-mraw * SCALE_LOG2E, -mraw * SCALE_LOG2E};
        f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f2 a = __builtin_elementwise_fma(f2{sc[j].x, sc[j].y}, k2, nm2);
            const f2 c = __builtin_elementwise_fma(f2{sc[j].z, sc[j].w}, k2, nm2);
            const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
            const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
            sc[j] = f4{ea.x, ea.y, ec.x, ec.y};
            d0 += ea;
            d1 += ec;
        }
        const f2 dd = d0 + d1;
        const float den = quad_sum(dd.x + dd.y);
        f4 o[NC];
#pragma unroll
        for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
        gemm_stage<NJ, NC, true, false, f4>(
            [&](int J, f4 (&a)[NC]) {
#pragma unroll
                for (int t = 0; t < NC; ++t) a[t] = vf[(t * NT + J0 + J) * 64];
            },
            [&](int J) { return sc[J]; }, o);
#pragma unroll
        for (int t = 0; t < NC; ++t) Pg[(g * NC + t) * 64 + lane] = o[t];
        Mg[(g * 2 + 0) * 64 + lane] = mraw;
        Mg[(g * 2 + 1) * 64 + lane] = den;
    };
    constexpr int NJB = NT / 4, NJL = NT - 3 * NJB;      // keys per wave: NT / 4, the remainder on the last wave
    if (g == 3) attn_part(std::integral_constant<int, NJL>{}, 3 * NJB);
    else attn_part(std::integral_constant<int, NJB>{}, g * NJB);
    __syncthreads();
    f4 o[NC];
    {   // merge the four partial softmaxes (every wave, redundantly)
        float mk[4], lk[4], M = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mk[k] = Mg[(k * 2 + 0) * 64 + lane];
            lk[k] = Mg[(k * 2 + 1) * 64 + lane];
            M = fmaxf(M, mk[k]);
        }
        float Lsum = 0.f;
#pragma unroll
        for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = __builtin_amdgcn_exp2f((mk[k] - M) * SCALE_LOG2E);
            Lsum = fmaf(f, lk[k], Lsum);
#pragma unroll
            for (int t = 0; t < NC; ++t) o[t] = o[t] + splat4(f) * Pg[(k * NC + t) * 64 + lane];
        }
        const float rl = __builtin_amdgcn_rcpf(Lsum);
#pragma unroll
        for (int t = 0; t < NC; ++t) o[t] = o[t] * splat4(rl);
    }
    // ---- proj: output tile g (waves 0..2) -> LDS; everybody adds all three
    if (g < NC) {
        f4 acc[1] = {splat4(0.f)};
        gemm_stage<NC, 1, true, false, f4>([&](int c, f4 (&a)[1]) { a[0] = wimg(P + O_WPROJ, g * NC + c, lane); },
                                       [&](int c) { return o[c]; }, acc);
        Dg[g * 64 + lane] = acc[0];
    }
    __syncthreads();          // also: every wave has read Pg / Mg (the fc2 partials reuse Pg)
#pragma unroll
    for (int ot = 0; ot < NC; ++ot) x[ot] = x[ot] + ld4(P + O_BPROJ + 16 * ot + 4 * q) + Dg[ot * 64 + lane];
    // ---- LayerNorm-2 (redundant), fc1 + GELU of hidden tiles 3g..3g+2, fc2 as four partial sums
    {
        f4 h[NC];
        layer_norm_plain(x, h);
        constexpr int HPW = NH / 4;                      // 3 hidden tiles per wave
        f4 hd[HPW];
#pragma unroll
        for (int j = 0; j < HPW; ++j) hd[j] = ld4(P + O_B1 + 16 * (HPW * g + j) + 4 * q);
        gemm_stage<NC, HPW, true, false, f4>(
            [&](int c, f4 (&a)[HPW]) {
#pragma unroll
                for (int j = 0; j < HPW; ++j) a[j] = wimg(P + O_W1, (HPW * g + j) * NC + c, lane);
            },
            [&](int c) { return h[c]; }, hd);
#pragma unroll
        for (int j = 0; j < HPW; ++j) hd[j] = gelu4(hd[j]);
        f4 part[NC];
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) part[ot] = splat4(0.f);
        gemm_stage<HPW, NC, true, false, f4>(
            [&](int cc, f4 (&a)[NC]) {
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) a[ot] = wimg(P + O_W2, ot * NH + HPW * g + cc, lane);
            },
            [&](int cc) { return hd[cc]; }, part);
#pragma unroll
        for (int ot = 0; ot < NC; ++ot) Pg[(g * NC + ot) * 64 + lane] = part[ot];
    }
    __syncthreads();
#pragma unroll
    for (int ot = 0; ot < NC; ++ot) {                    // every wave: the tile's residual stream after this block
        x[ot] = x[ot] + ld4(P + O_B2 + 16 * ot + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[ot] = x[ot] + Pg[(k * NC + ot) * 64 + lane];
    }
    if (g == 0) {
        float* dst = xout + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
        for (int c = 0; c < NC; ++c) st4(dst + 16 * c, x[c]);
        if (normP != nullptr) {
            if (resid != nullptr) {
                float* rd = resid + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(rd + 16 * c, x[c]);
            }
            if (16 * T >= len_z) {
                f4 h[NC];
                layer_norm_img(x, h, normP, normP + C, q);
                float* fd = feat + ((size_t)b * (L - len_z) + (16 * T - len_z) + tok) * C + 4 * q;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(fd + 16 * c, h[c]);
            }
        }
    }
    // ---- the NEXT block's LayerNorm-1 + qkv of this tile (per-token work: it needs nothing from the other tiles), into the other
    // workspace set -- the tile_qkv launch of blocks 1.. disappears.  Wave 0: q, wave 1: k, wave 2: v^T, the chains of tile_qkv_kernel.
    if (Pn != nullptr && g < 3) {
        f4 h[NC];
        layer_norm_plain(x, h);
        f4 acc[NC];
        if (g < 2) {
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) acc[ot] = ld4(Pn + O_BQKV + 16 * (g * NC + ot) + 4 * q);
            gemm_stage<NC, NC, true, false, f4>(
                [&](int c, f4 (&a)[NC]) {
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) a[ot] = wimg(Pn + O_WQKV, (g * NC + ot) * NC + c, lane);
                },
                [&](int c) { return h[c]; }, acc);
            f4* dst = (g == 0 ? qn : kn) + ((size_t)b * NT + T) * NC * 64 + lane;
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) dst[ot * 64] = acc[ot];
        } else {
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) acc[ot] = splat4(Pn[O_BQKV + 2 * C + 16 * ot + tok]);
            gemm_stage<NC, NC, false, false, f4>(
                [&](int c, f4 (&bw)[NC]) {
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) bw[ot] = wimg(Pn + O_WQKV, (2 * NC + ot) * NC + c, lane);
                },
                [&](int c) { return h[c]; }, acc);
            f4* dst = vn + (size_t)b * NC * NT * 64 + (size_t)T * 64 + lane;
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) dst[(size_t)ot * NT * 64] = acc[ot];
        }
    }
}

}  // namespace vtb
