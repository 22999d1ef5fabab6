// vb_qkvattn.h -- the qkv projection FUSED into the attention of one (frame, head), bf16 MFMA, gfx950 (round 5).
//
// lib/models/ostrack/vit.py:51-66 (Attention.forward) with norm1 folded in (vb_gemm.h): per frame and head
//     [q | k | v] = rstd * (x W'^T) + b'      x: the frame's L = 320 raw residual rows (bf16), W': the head's 3 x 64 folded weight rows
//     out         = softmax(q k^T) v          (q pre-scaled by 64^-0.5 at load time)
// The separate kernels (qk GEMM, v GEMM, attention) write q / k / v^T for all heads to HBM (378 MB at B = 256) and read them back
// (378 MB): 1.49 GB of fabric traffic per block with their operand re-fetches, in a step whose phases are memory-system-bound
// (NOTES R5-2).  Here one workgroup computes the 320 x 192 projection of ITS (frame, head) into registers, writes q, K and V^T as the
// attention's operand images into LDS and attends on chip: what reaches the fabric is x (the 12 heads of a frame run on one XCD at
// about the same time: one fetch + L2 hits), the head's 295 KB of weights (L2-resident), and the 40 KB output.
//
// Projection: the GEMM of vb_gemm.h at a 320 x 192 tile -- the same 64 KiB of operands per 64-deep k-tile as the 256 x 256 tile
// (X 40 pieces + W 24 pieces of 8 rows x 128 B, chunk ^ row swizzle, LDS-DMA), 8 waves as 4 (token groups of 5 tiles) x 2 (feature
// groups of 6 tiles: q0-1 k0-1 v0-1 | q2-3 k2-3 v2-3), 30 accumulator tiles per wave.  The v tiles swap the MFMA operands (tokens as
// rows), so a lane holds four consecutive TOKENS of one feature: V^T is written transposed, as the P.V product wants it.  Every wave
// has the same tile pattern (which tiles swap is a compile-time property of the tile index): with a wave-uniform branch between two
// operand orders hipcc spilled 300 registers around the k-loop.
// Attention: vb_attn.h's arithmetic (S^T = K q^T with keys on the MFMA rows, softmax rows in registers, O^T = V^T P^T, two query tiles
// per pass), on 8 waves, q from LDS like K.  The three images alias the projection's staging buffers.
#pragma once
#include "vb_gemm.h"

#ifndef VB_QA_SWP
#define VB_QA_SWP 1         // 1: software-pipelined k-loop (DMA of k-tile t + 2 under the MFMAs of t); 0: the simple two-buffer loop
#endif

namespace vbq {

using vbg::bf16;
using vbg::bf16x4;
using vbg::bf16x8;
using vbg::glds16;
using vbg::swz_byte;

constexpr int L = 320, HD = 64, DM = 768, NT = L / 16, NC = L / 32, KS = HD / 32, DT = HD / 16;
constexpr int XP = L / 8, WP = 3 * HD / 8;                  // 1 KiB DMA pieces per k-tile: 40 + 24
constexpr int STAGE_BYTES = (XP + WP) * 1024;               // 64 KiB
constexpr int IMG_BYTES = NT * KS * 1024;                   // q, K, V^T images: 40 KiB each
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
static_assert(3 * IMG_BYTES <= LDS_BYTES && (XP + WP) % 8 == 0, "images alias the stages; pieces divide over 8 waves");

struct Args {
    const bf16* X;        // [B L][DM] raw residual rows (bf16 copy written by the residual-writing GEMMs)
    const bf16* W;        // [3 DM][DM] folded qkv weights (q rows pre-scaled)
    const float* bias;    // [3 DM]
    const float* rstd;    // [B L] or null
    bf16* out;            // [B L][DM] attention output (head h at columns h HD ..)
    int B, heads;
};

__global__ __launch_bounds__(512) void qkv_attn_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane_k = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    char* const Qimg = smem;
    char* const Kimg = smem + IMG_BYTES;
    char* const Vimg = smem + 2 * IMG_BYTES;

    // ---- items: XCD x (blocks with blockIdx % 8 == x share an L2) owns frames x, x + 8, ...; its workgroups walk (frame, head) pairs
    // frame-major, so the heads of a frame run at about the same time on one XCD and share the fetch of its rows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int nfx = (a.B - xcd + 7) / 8;                     // frames of this XCD
    const int nitems = nfx * a.heads;

    // ---- DMA sources of this wave's 8 pieces per k-tile: per-lane BYTE offsets from the item's (wave-uniform) X / W bases
    unsigned soff[8];
    {
        const int drow = lane_k >> 3, dk = ((lane_k & 7) ^ drow) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = w + 8 * i;
            if (p < XP) soff[i] = (unsigned)((p * 8 + drow) * DM + dk) * 2u;
            else {
                const int pw = p - XP, blk = pw >> 3, rr = (pw & 7) * 8 + drow;
                soff[i] = (unsigned)((blk * DM + rr) * DM + dk) * 2u;            // + the head's h HD rows, in the base
            }
        }
    }

    for (int li = slot; li < nitems; li += per) {
        const int lf = li / a.heads, h = li - lf * a.heads, f = xcd + 8 * lf;
        // Every per-lane address of an item comes from a FRESH opaque copy of the lane index: as invariants of the item loop hipcc
        // computes dozens of them up front and spills them around the k-loop (the first build: 464 B of scratch per lane)
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int q = lane >> 4;
        const unsigned long long xb64 = reinterpret_cast<unsigned long long>(a.X + (size_t)f * L * DM);
        const unsigned long long wb64 = reinterpret_cast<unsigned long long>(a.W + (size_t)h * HD * DM);
        auto issue = [&](int kt, int stg) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = w + 8 * i;
                const unsigned long long b64 = (p < XP ? xb64 : wb64) + (unsigned long long)kt * 128ull;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b64), hi = __builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
                const char* base = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
                unsigned off = soff[i];
                asm volatile("" : "+v"(off));
                glds16(base + (size_t)off, smem + stg * STAGE_BYTES + p * 1024);
            }
        };
        // fragment addresses: one opaque base register per (operand, stage, k-step) + an immediate per tile
        int fxa[2][2], fwa[2][2];
        {
            const int r7 = lane & 7, fbase = ((lane & 15) >> 3) * 1024 + r7 * 128;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int fkk = fbase + (((4 * kk + q) ^ r7) << 4);
                    fxa[st][kk] = st * STAGE_BYTES + wm * 5 * 2048 + fkk;
                    fwa[st][kk] = st * STAGE_BYTES + XP * 1024 + wn * 2 * 2048 + fkk;
                    asm volatile("" : "+v"(fxa[st][kk]), "+v"(fwa[st][kk]));
                }
        }
        f4 acc[6][5];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = splat4(0.f);
        // this wave's feature tile i: kind i >> 1 (0 q, 1 k, 2 v), tile (i >> 1) 4 + wn 2 + (i & 1) of the head's 12
        auto kstep = [&](auto stg, int kk) {
            constexpr int ST = decltype(stg)::value;
            bf16x8 fx[5], fw[6];
            const char* xp = smem + fxa[ST][kk];
            const char* wp = smem + fwa[ST][kk];
#pragma unroll
            for (int j = 0; j < 5; ++j) fx[j] = *reinterpret_cast<const bf16x8*>(xp + j * 2048);
#pragma unroll
            for (int i = 0; i < 6; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(wp + ((i >> 1) * 4 + (i & 1)) * 2048);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if (i >= 4) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[j], fw[i], acc[i][j], 0, 0, 0);     // v: tokens on rows
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
                }
        };
        constexpr int NK = DM / 64;
        static_assert(NK % 2 == 0, "k-tiles come in stage pairs");
        const std::integral_constant<int, 0> S0{};
        const std::integral_constant<int, 1> S1{};
        auto ktile = [&](auto stg) {
            kstep(stg, 0);
            kstep(stg, 1);
        };
        // ---- projection k-loop: two stages, the DMA of k-tile t + 1 under the MFMAs of t
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < NK; kt += 2) {
            issue(kt + 1, 1);
            ktile(S0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 2 < NK) issue(kt + 2, 0);
            ktile(S1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- epilogue: y = rstd * acc + bias, as bf16 into the attention's operand images (the stages are free: every wave has
        // passed the last k-tile's barrier)
        {
            int le = lane_k;
            asm volatile("" : "+v"(le));
            const int l15 = le & 15, q = le >> 4;
            float rs[5];            // q / k tiles: token tj 16 + l15
            f4 rs4[5];              // v tiles: tokens tj 16 + 4 q + {0..3}
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int tj = wm * 5 + j;
                rs[j] = a.rstd ? a.rstd[(size_t)f * L + tj * 16 + l15] : 1.f;
                rs4[j] = a.rstd ? ld4(a.rstd + (size_t)f * L + tj * 16 + 4 * q) : splat4(1.f);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int ft = (i >> 1) * 4 + wn * 2 + (i & 1);      // 0-3 q, 4-7 k, 8-11 v
                if (ft < 8) {
                    const int nh = (ft & 3) * 16 + 4 * q;    // feature inside the head
                    const f4 bv = ld4(a.bias + (ft >> 2) * DM + h * HD + nh);
                    const int ks = nh >> 5, kb = (nh & 31) * 2;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int tj = wm * 5 + j, m = tj * 16 + l15;
                        const bf16x4 y = vbg::to_bf16x4(vbg::fma4(acc[i][j], splat4(rs[j]), bv));
                        if (ft < 4) {
                            *reinterpret_cast<bf16x4*>(Qimg + (tj * KS + ks) * 1024 + swz_byte(l15 * 64 + kb)) = y;
                        } else {           // K rows permuted: token 32 c + 8 a + b -> tile 2 c + (b >> 2), row 4 a + (b & 3)   (vb_attn.h)
                            const int c = m >> 5, aa = (m & 31) >> 3, b = m & 7, t = 2 * c + (b >> 2), row = 4 * aa + (b & 3);
                            *reinterpret_cast<bf16x4*>(Kimg + (t * KS + ks) * 1024 + swz_byte(row * 64 + kb)) = y;
                        }
                    }
                } else {
                    const int dt = ft - 8;                   // feature dt 16 + l15 of the head
                    const float bs = a.bias[2 * DM + h * HD + dt * 16 + l15];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int m = (wm * 5 + j) * 16 + 4 * q, c = m >> 5;
                        const bf16x4 y = vbg::to_bf16x4(vbg::fma4(acc[i][j], rs4[j], splat4(bs)));
                        *reinterpret_cast<bf16x4*>(Vimg + (dt * NC + c) * 1024 + swz_byte(l15 * 64 + (m & 31) * 2)) = y;
                    }
                }
            }
        }
        __syncthreads();
        // ---- attention of the frame's 20 query tiles: wave w takes tiles w, w + 8 (and w + 16 for w < 4); two tiles per pass
        constexpr float LOG2E = 1.4426950408889634f;
        auto pass = [&](auto nq_tag, int qt0, int qt1) {
            constexpr int NQ = decltype(nq_tag)::value;
            const int qts[2] = {qt0, qt1};
            int la = lane_k;
            asm volatile("" : "+v"(la));        // a fresh lane index per pass: keeps the K / V^T fragment reads (and the output addresses) inside it
            const int l15 = la & 15, q = la >> 4;
            const int frq = swz_byte(l15 * 64 + q * 16);          // fragment byte inside a 16 x 32 sub-tile of the images
            bf16x8 qf[NQ][KS];
#pragma unroll
            for (int u = 0; u < NQ; ++u)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[u][ks] = *reinterpret_cast<const bf16x8*>(Qimg + (qts[u] * KS + ks) * 1024 + frq);
            f4 S[NQ][NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int u = 0; u < NQ; ++u) S[u][t] = splat4(0.f);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf8 = *reinterpret_cast<const bf16x8*>(Kimg + (t * KS + ks) * 1024 + frq);
#pragma unroll
                    for (int u = 0; u < NQ; ++u) S[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf8, qf[u][ks], S[u][t], 0, 0, 0);
                }
            }
            float inv[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                float m0 = fmaxf(S[u][0].x, S[u][0].y), m1 = fmaxf(S[u][0].z, S[u][0].w);
#pragma unroll
                for (int t = 1; t < NT; ++t) {
                    m0 = fmaxf(fmaxf(m0, S[u][t].x), S[u][t].y);
                    m1 = fmaxf(fmaxf(m1, S[u][t].z), S[u][t].w);
                }
                const float mx = quad_max(fmaxf(m0, m1));
                const vbg::f2 l2 = {LOG2E, LOG2E}, nmb = {-mx * LOG2E, -mx * LOG2E};
                vbg::f2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const vbg::f2 x0 = __builtin_elementwise_fma(vbg::f2{S[u][t].x, S[u][t].y}, l2, nmb);
                    const vbg::f2 x1 = __builtin_elementwise_fma(vbg::f2{S[u][t].z, S[u][t].w}, l2, nmb);
                    const vbg::f2 pa = {__builtin_amdgcn_exp2f(x0.x), __builtin_amdgcn_exp2f(x0.y)};
                    const vbg::f2 pb = {__builtin_amdgcn_exp2f(x1.x), __builtin_amdgcn_exp2f(x1.y)};
                    S[u][t] = f4{pa.x, pa.y, pb.x, pb.y};
                    s0 += pa;
                    s1 += pb;
                }
                const vbg::f2 st = s0 + s1;
                inv[u] = 1.0f / quad_sum(st.x + st.y);
            }
            f4 O[NQ][DT];
#pragma unroll
            for (int u = 0; u < NQ; ++u)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) O[u][dt] = splat4(0.f);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                bf16x8 p[NQ];
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const bf16x4 lo = vbg::to_bf16x4(S[u][2 * c]), hi = vbg::to_bf16x4(S[u][2 * c + 1]);
                    p[u] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf8 = *reinterpret_cast<const bf16x8*>(Vimg + (dt * NC + c) * 1024 + frq);
#pragma unroll
                    for (int u = 0; u < NQ; ++u) O[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf8, p[u], O[u][dt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                bf16* o = a.out + (size_t)(f * L + qts[u] * 16 + l15) * DM + h * HD + q * 4;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<bf16x4*>(o + dt * 16) = vbg::to_bf16x4(O[u][dt] * splat4(inv[u]));
            }
        };
        pass(std::integral_constant<int, 2>{}, w, w + 8);
        if (w < 4) pass(std::integral_constant<int, 1>{}, w + 16, 0);
        __syncthreads();             // the images are dead: the next item's staging may overwrite them
    }
}

}  // namespace vbq
