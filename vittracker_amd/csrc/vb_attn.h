// vb_attn.h -- multi-head self-attention over the joint (template + search) token set, bf16 MFMA, gfx950.
//
// lib/models/ostrack/vit.py:51-66: softmax(q k^T * d^-0.5) v per head (12 heads x 64 at ViT-Base), all L = 320 keys,
// no mask.  q arrives pre-scaled (the scale, a power of two, is folded into W_q / b_q at load time).
//
// One workgroup = one (frame, head).  K (L x 64) and V^T (64 x L, written transposed by the qkv GEMM) are staged once
// into LDS by LDS-DMA as 16-row x 32-k sub-tiles (the GEMM's st_16x32 image, conflict-free ds_read_b128); the four waves
// then walk the L / 16 query tiles.  Per query tile:
//     S^T = K q^T          keys on the MFMA rows: a softmax row is 4 * L/16 registers of one lane + two permlane swaps
//     P   = exp2((S - max) * log2 e), row sum in f32
//     O^T = V^T P^T        P^T's B-operand image is S^T's own registers (rounded to bf16): no LDS round trip
// The L x L score matrix never exists in memory.  Key order inside a 32-key chunk is permuted identically on both
// sides (rows of the K image are stored as key = 32c + 8(i >> 2) + (i & 3) [+ 4 for the odd tile]), so that the eight
// P values a lane owns after the two S^T tiles of a chunk are exactly keys 32c + 8q + {0..7}: V^T is read in natural order.
#pragma once
#include "vb_gemm.h"

#ifndef VB_ATTN_DBG
#define VB_ATTN_DBG 0       // timing experiments only (wrong results): 1 = no q.k / softmax / P.V (staging, q loads and stores only),
#endif                      // 2 = no K / V^T staging (the arithmetic on whatever the LDS holds)

namespace vba {

using vbg::bf16;
using vbg::bf16x4;
using vbg::bf16x8;

template <int L, int HD>
struct Geo {
    static constexpr int NT = L / 16;              // key / query tiles
    static constexpr int NC = L / 32;              // 32-key chunks
    static constexpr int KS = HD / 32;             // k-steps of q.k (d = 64 -> 2)
    static constexpr int DT = HD / 16;             // output d tiles
    static constexpr int K_SUB = NT * KS;          // 1 KiB sub-tiles of the K image
    static constexpr int V_SUB = DT * NC;          // ... of the V^T image
    static constexpr int LDS_BYTES = (K_SUB + V_SUB) * 1024;
};

// qk: [M][2 C] bf16 (q | k), vt: [B][C][L] bf16, out: [M][C] bf16;  C = heads * HD.  grid = B * heads, 256 threads.
template <int L, int HD>
__global__ __launch_bounds__(256, 2) void attn_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt,
                                                   bf16* __restrict__ out, int heads) {
    using G = Geo<L, HD>;
    static_assert(L % 32 == 0 && HD % 32 == 0, "tile shapes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kimg = smem;
    char* Vimg = smem + G::K_SUB * 1024;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = blockIdx.x / heads, h = blockIdx.x - f * heads, C = heads * HD;
    const bf16* qf = qk + (size_t)f * L * 2 * C + h * HD;          // q rows of this frame / head
    const bf16* kf = qf + C;
    const bf16* vf = vt + ((size_t)f * C + h * HD) * L;

    const int l15 = lane & 15, q = lane >> 4;
    // Q fragments live across passes: the first pass's are requested before the K / V^T staging (oldest in vmcnt order: the
    // K-only wait below covers them), the NEXT pass's into the same registers as soon as a pass's q.k products are done
    // (they are dead from there on), so their global-load latency is covered by softmax + P.V
    bf16x8 qfrag[2][G::KS];
    auto load_q = [&](int qt, bf16x8 (&dst)[G::KS]) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks)
            dst[ks] = *reinterpret_cast<const bf16x8*>(qf + (size_t)(qt * 16 + l15) * 2 * C + ks * 32 + q * 8);
    };
    load_q(w, qfrag[0]);
    load_q(w + 4, qfrag[1]);
    // ---- stage K (rows permuted) and V^T
    const int pl = vbg::swz_byte(lane * 16), prow = pl >> 6, pk = (pl & 63) >> 1;
    for (int s = w; s < (VB_ATTN_DBG == 2 ? 0 : G::K_SUB); s += 4) {
        const int t = s / G::KS, ks = s - t * G::KS;
        const int key = 32 * (t >> 1) + 8 * (prow >> 2) + (prow & 3) + 4 * (t & 1);
        vbg::glds16(kf + (size_t)key * 2 * C + ks * 32 + pk, Kimg + s * 1024 + lane * 16);
    }
    for (int s = w; s < (VB_ATTN_DBG == 2 ? 0 : G::V_SUB); s += 4) {
        const int dt = s / G::NC, c = s - dt * G::NC;
        vbg::glds16(vf + (size_t)(dt * 16 + prow) * L + c * 32 + pk, Vimg + s * 1024 + lane * 16);
    }
    // K is needed first: wait for this wave's K pieces only (the V^T pieces, issued after them, stay in flight through the
    // first q.k pass) -- vmcnt counts in issue order
    constexpr int VPW = (G::V_SUB + 3) / 4;                 // V^T pieces per wave (upper bound: waves with fewer wait longer)
    static_assert(G::V_SUB % 4 == 0 && G::K_SUB % 4 == 0, "pieces divide over the four waves");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VPW) : "memory");
    __syncthreads();
    bool v_ready = false;

    const int fr = vbg::swz_byte(l15 * 64 + q * 16);
    constexpr float LOG2E = 1.4426950408889634f;
    // NQ query tiles at once: every K / V^T fragment read from LDS feeds NQ MFMAs.  With one tile per pass the kernel issues one
    // ds_read_b128 per MFMA -- 256 B/clk/CU for four SIMDs' worth of 16-cycle MFMAs is exactly the LDS bandwidth, so it was
    // LDS-bound; two tiles halve the reads (a wave's five tiles go as 2 + 2 + 1).
    auto pass = [&](auto nq_tag, int qt0, int qt1, auto nxt_tag, int nx0, int nx1) {
        constexpr int NQ = decltype(nq_tag)::value, NXT = decltype(nxt_tag)::value;
        const int qts[2] = {qt0, qt1};
        // the K / V^T fragments do not depend on the query tile: without this the compiler hoists all 80 ds_reads
        // (320 VGPRs) out of the loop and spills them
        int frq = fr;
        asm volatile("" : "+v"(frq));
        f4 S[NQ][G::NT];
        if constexpr (VB_ATTN_DBG == 1) {
            if constexpr (NXT >= 1) load_q(nx0, qfrag[0]);
            if constexpr (NXT >= 2) load_q(nx1, qfrag[1]);
            if (!v_ready) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); v_ready = true; }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                bf16* o = out + (size_t)(f * L + qts[u] * 16 + l15) * C + h * HD + q * 4;
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt) *reinterpret_cast<bf16x4*>(o + dt * 16) = bf16x4{qfrag[u][0][0], qfrag[u][0][1], qfrag[u][1][0], qfrag[u][1][1]};
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < G::NT; ++t) {
#pragma unroll
            for (int u = 0; u < NQ; ++u) S[u][t] = splat4(0.f);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const bf16x8 kf8 = *reinterpret_cast<const bf16x8*>(Kimg + (t * G::KS + ks) * 1024 + frq);
#pragma unroll
                for (int u = 0; u < NQ; ++u) S[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf8, qfrag[u][ks], S[u][t], 0, 0, 0);
            }
        }
        if constexpr (NXT >= 1) load_q(nx0, qfrag[0]);
        if constexpr (NXT >= 2) load_q(nx1, qfrag[1]);
        float inv[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            // row maximum on v_max3_f32 (two values per instruction, two chains); exponent argument and row sum on packed
            // f32 math (v_pk_fma_f32 / v_pk_add_f32): per query tile 80 exp + ~140 other VALU instead of ~310
            float m0 = fmaxf(S[u][0].x, S[u][0].y), m1 = fmaxf(S[u][0].z, S[u][0].w);
#pragma unroll
            for (int t = 1; t < G::NT; ++t) {
                m0 = fmaxf(fmaxf(m0, S[u][t].x), S[u][t].y);
                m1 = fmaxf(fmaxf(m1, S[u][t].z), S[u][t].w);
            }
            const float mx = quad_max(fmaxf(m0, m1));
            const vbg::f2 l2 = {LOG2E, LOG2E}, nmb = {-mx * LOG2E, -mx * LOG2E};
            vbg::f2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < G::NT; ++t) {
                const vbg::f2 a = __builtin_elementwise_fma(vbg::f2{S[u][t].x, S[u][t].y}, l2, nmb);
                const vbg::f2 b = __builtin_elementwise_fma(vbg::f2{S[u][t].z, S[u][t].w}, l2, nmb);
                const vbg::f2 pa = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                const vbg::f2 pb = {__builtin_amdgcn_exp2f(b.x), __builtin_amdgcn_exp2f(b.y)};
                S[u][t] = f4{pa.x, pa.y, pb.x, pb.y};
                s0 += pa;
                s1 += pb;
            }
            const vbg::f2 st = s0 + s1;
            inv[u] = 1.0f / quad_sum(st.x + st.y);
        }
        f4 O[NQ][G::DT];
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) O[u][dt] = splat4(0.f);
        if (!v_ready) {                      // first pass only (wave-uniform): V^T has had the whole q.k + softmax phase to land
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            v_ready = true;
        }
#pragma unroll
        for (int c = 0; c < G::NC; ++c) {
            bf16x8 p[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const bf16x4 lo = vbg::to_bf16x4(S[u][2 * c]), hi = vbg::to_bf16x4(S[u][2 * c + 1]);
                p[u] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                const bf16x8 vf8 = *reinterpret_cast<const bf16x8*>(Vimg + (dt * G::NC + c) * 1024 + frq);
#pragma unroll
                for (int u = 0; u < NQ; ++u) O[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf8, p[u], O[u][dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            bf16* o = out + (size_t)(f * L + qts[u] * 16 + l15) * C + h * HD + q * 4;
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) *reinterpret_cast<bf16x4*>(o + dt * 16) = vbg::to_bf16x4(O[u][dt] * splat4(inv[u]));
        }
    };
    // a wave's query tiles w, w + 4, ... go two per pass (plus a last single one when their number is odd)
    constexpr int TPW = G::NT / 4;                      // query tiles per wave
    static_assert(G::NT % 4 == 0 && TPW >= 2, "query tiles divide over the four waves");
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int i = 0; i + 1 < TPW; i += 2) {
        const int qt = w + 4 * i, left = TPW - (i + 2);                      // tiles left after this pass
        if (left >= 2) pass(I2{}, qt, qt + 4, I2{}, qt + 8, qt + 12);
        else if (left == 1) pass(I2{}, qt, qt + 4, I1{}, qt + 8, qt + 8);
        else pass(I2{}, qt, qt + 4, I0{}, 0, 0);
    }
    if constexpr (TPW % 2 == 1) pass(I1{}, w + 4 * (TPW - 1), 0, I0{}, 0, 0);
}

}  // namespace vba
