// vt_blocks.h -- the joint template+search transformer stack, one workgroup per frame.
//
// Replaces, for C = 48 / 1 head / mlp_ratio 4:
//   timm Block.forward x depth + final LayerNorm   (lib/models/vit_dist/vit_dist.py:88-94;
//   block arithmetic restated in-tree at lib/models/layers/attn_blocks.py:130-133 and
//   lib/models/layers/attn.py:33-59).
//
// Design (MI355X): every contraction runs on v_mfma_f32_16x16x4_f32 (exact f32).  Each wave owns
// 16-token tiles and keeps their residual stream in registers as three "operand images"
// (vt_common.h): lane (tok = lane & 15, q = lane >> 4) holds features 16c + 4q + {0..3} of its
// token for chunk c.  With the weights as the A operand (rows = output features) and the
// activations as the B operand (columns = tokens), an MFMA result is again an operand image of
// the next layer, so LN -> QKV -> softmax -> PV -> proj -> LN -> fc1 -> GELU -> fc2 chains
// entirely in registers.  Only K and V cross waves, through LDS, as lane-linear 1 KiB tiles
// (conflict-free ds_write_b128 / ds_read_b128):
//   Kimg[J][c]   = k features of key tile J          (from W_k as A, h as B)
//   Vimg[t][J]   = V^T: feature tile t x key tile J  (from h as A, W_v as B: swapped operands,
//                  which lands V already transposed for the P.V product)
// S^T = K q^T is computed with keys as rows, so a softmax row lives in one lane's registers plus
// the 3 partner lanes (two xor-shuffles), never in LDS.
#pragma once
#include "vt_common.h"

namespace vtb {

constexpr int C = 48;          // embed dim
constexpr int NC = C / 16;     // 3 feature chunks
constexpr int HID = 4 * C;     // 192
constexpr int NH = HID / 16;   // 12 hidden chunks
constexpr float LN_EPS = 1e-5f;

// per-block packed parameter offsets (floats); images are [out_tile][k_chunk][lane][4]
constexpr int O_LN1G = 0;
constexpr int O_LN1B = O_LN1G + C;
constexpr int O_WQKV = O_LN1B + C;                 // 9 x 3 tiles
constexpr int O_BQKV = O_WQKV + 9 * NC * 256;
constexpr int O_WPROJ = O_BQKV + 3 * C;            // 3 x 3 tiles
constexpr int O_BPROJ = O_WPROJ + NC * NC * 256;
constexpr int O_LN2G = O_BPROJ + C;
constexpr int O_LN2B = O_LN2G + C;
constexpr int O_W1 = O_LN2B + C;                   // 12 x 3 tiles
constexpr int O_B1 = O_W1 + NH * NC * 256;
constexpr int O_W2 = O_B1 + HID;                   // 3 x 12 tiles
constexpr int O_B2 = O_W2 + NC * NH * 256;
constexpr int BLOCK_STRIDE = O_B2 + C;             // 28272 floats
static_assert(BLOCK_STRIDE % 4 == 0 && O_WQKV % 4 == 0 && O_W1 % 4 == 0 && O_W2 % 4 == 0, "16B alignment");

__device__ __forceinline__ f4 wimg(const float* __restrict__ base, int tile, int lane) {
    return ld4(base + (size_t)tile * 256 + lane * 4);
}

// LayerNorm of one token held as NC operand-image chunks; g/b point at gamma/beta (C floats).
__device__ __forceinline__ void layer_norm_img(const f4 (&x)[NC], f4 (&h)[NC], const float* __restrict__ g,
                                               const float* __restrict__ b, int q) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s += hsum4(x[c]);
    const float mean = quad_sum(s) * (1.0f / C);
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        f4 d = x[c] - splat4(mean);
        v += hsum4(d * d);
    }
    const float inv = 1.0f / sqrtf(quad_sum(v) * (1.0f / C) + LN_EPS);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f4 gg = ld4(g + 16 * c + 4 * q), bb = ld4(b + 16 * c + 4 * q);
        h[c] = (x[c] - splat4(mean)) * splat4(inv) * gg + bb;
    }
}

// NT = token tiles per frame (L / 16), NW = waves per workgroup, TPW = tiles per wave.
template <int NT, int NW, int TPW>
__global__ __launch_bounds__(NW * 64) void blocks_kernel(const float* __restrict__ tokens,   // (B, L, C)
                                                         const float* __restrict__ params,   // packed, see O_*
                                                         float* __restrict__ feat,           // (B, Lx, C)
                                                         float* __restrict__ resid,          // (B, L, C) or null
                                                         int len_z, int depth_total, int nblocks) {
    static_assert(NW * TPW >= NT, "tiles must be covered");
    constexpr int L = NT * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* Kimg = reinterpret_cast<f4*>(lds);                 // [NT][NC][64]
    f4* Vimg = Kimg + NT * NC * 64;                        // [NC][NT][64]

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok = lane & 15, q = lane >> 4;
    const float scale = 0.14433756729740643f;  // 48^-0.5  (head_dim ** -0.5, attn.py:15)

    f4 x[TPW][NC];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int T = w + NW * i;
        if (T < NT) {
            const float* src = tokens + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
            for (int c = 0; c < NC; ++c) x[i][c] = ld4(src + 16 * c);
        }
    }

    for (int blk = 0; blk < nblocks; ++blk) {
        const float* __restrict__ P = params + (size_t)blk * BLOCK_STRIDE;
        f4 qr[TPW][NC];
        // ---- LN1 + QKV; publish K / V^T images ------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int T = w + NW * i;
            if (T < NT) {
                f4 h[NC];
                layer_norm_img(x[i], h, P + O_LN1G, P + O_LN1B, q);
                {   // q and k: 6 independent chains, rows = features, cols = tokens (B = h shared)
                    f4 acc[2 * NC];
#pragma unroll
                    for (int ot = 0; ot < 2 * NC; ++ot) acc[ot] = ld4(P + O_BQKV + 16 * ot + 4 * q);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        f4 a[2 * NC];
#pragma unroll
                        for (int ot = 0; ot < 2 * NC; ++ot) a[ot] = wimg(P + O_WQKV, ot * NC + c, lane);
                        mfma4_shared_b(a, h[c], acc);
                    }
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) {
                        qr[i][ot] = acc[ot];
                        Kimg[(T * NC + ot) * 64 + lane] = acc[NC + ot];
                    }
                }
                {   // v, operands swapped: rows = tokens, cols = v features (A = h shared) -> V^T image
                    f4 acc[NC];
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) acc[ot] = splat4(P[O_BQKV + 2 * C + 16 * ot + tok]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        f4 bw[NC];
#pragma unroll
                        for (int ot = 0; ot < NC; ++ot) bw[ot] = wimg(P + O_WQKV, (2 * NC + ot) * NC + c, lane);
                        mfma4_shared_a(h[c], bw, acc);
                    }
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) Vimg[(ot * NT + T) * 64 + lane] = acc[ot];
                }
            }
        }
        __syncthreads();
        // In the last block the template rows only matter as keys / values: their attention
        // output, proj and MLP never reach the head (vit_dist.py:126 keeps the search rows only),
        // so those tiles stop after publishing K / V -- unless the caller asked for the residual.
        const bool last_skip_z = (blk == depth_total - 1) && (resid == nullptr);
        // ---- attention + proj (residual add) --------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int T = w + NW * i;
            if (T < NT && !(last_skip_z && 16 * T < len_z)) {
                f4 s[NT];
                float m = -3.0e38f;
                constexpr int JG = 5;                    // key tiles per group = independent chains
                static_assert(NT % JG == 0, "NT must be a multiple of 5");
#pragma unroll
                for (int j0 = 0; j0 < NT; j0 += JG) {    // S^T tiles: rows = keys, cols = queries (B = q shared)
                    f4 acc[JG];
#pragma unroll
                    for (int j = 0; j < JG; ++j) acc[j] = splat4(0.f);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        f4 a[JG];
#pragma unroll
                        for (int j = 0; j < JG; ++j) a[j] = Kimg[((j0 + j) * NC + c) * 64 + lane];
                        mfma4_shared_b(a, qr[i][c], acc);
                    }
#pragma unroll
                    for (int j = 0; j < JG; ++j) {
                        acc[j] = acc[j] * splat4(scale);     // (q @ k^T) * scale, attn.py:40
                        s[j0 + j] = acc[j];
                        m = fmaxf(m, hmax4(acc[j]));
                    }
                }
                m = quad_max(m);
                float den = 0.f;
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    f4 e;
                    e.x = __expf(s[J].x - m); e.y = __expf(s[J].y - m);
                    e.z = __expf(s[J].z - m); e.w = __expf(s[J].w - m);
                    s[J] = e;
                    den += hsum4(e);
                }
                const float rden = 1.0f / quad_sum(den);
                f4 o[NC];
#pragma unroll
                for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
#pragma unroll
                for (int J = 0; J < NT; ++J) {           // O^T = V^T P^T: 3 feature-tile chains share B = P_J
                    f4 a[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) a[t] = Vimg[(t * NT + J) * 64 + lane];
                    mfma4_shared_b(a, s[J], o);
                }
#pragma unroll
                for (int t = 0; t < NC; ++t) o[t] = o[t] * splat4(rden);
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(P + O_BPROJ + 16 * ot + 4 * q);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    f4 a[NC];
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) a[ot] = wimg(P + O_WPROJ, ot * NC + c, lane);
                    mfma4_shared_b(a, o[c], x[i]);
                }
            }
        }
        __syncthreads();   // every wave is done reading K/V before the next block overwrites them
        // ---- LN2 + MLP (residual add) ---------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int T = w + NW * i;
            if (T < NT && !(last_skip_z && 16 * T < len_z)) {
                f4 h[NC];
                layer_norm_img(x[i], h, P + O_LN2G, P + O_LN2B, q);
                f4 hid[NH];
                constexpr int HG = 6;                    // hidden tiles per group = independent chains
#pragma unroll
                for (int g = 0; g < NH; g += HG) {
                    f4 acc[HG];
#pragma unroll
                    for (int j = 0; j < HG; ++j) acc[j] = ld4(P + O_B1 + 16 * (g + j) + 4 * q);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        f4 a[HG];
#pragma unroll
                        for (int j = 0; j < HG; ++j) a[j] = wimg(P + O_W1, (g + j) * NC + c, lane);
                        mfma4_shared_b(a, h[c], acc);
                    }
#pragma unroll
                    for (int j = 0; j < HG; ++j)
                        hid[g + j] = f4{gelu_erf(acc[j].x), gelu_erf(acc[j].y), gelu_erf(acc[j].z), gelu_erf(acc[j].w)};
                }
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(P + O_B2 + 16 * ot + 4 * q);
#pragma unroll
                for (int c = 0; c < NH; ++c) {
                    f4 a[NC];
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) a[ot] = wimg(P + O_W2, ot * NH + c, lane);
                    mfma4_shared_b(a, hid[c], x[i]);
                }
            }
        }
    }

    // ---- epilogue: optional residual dump; final LayerNorm on the search tokens ----------------
    const float* __restrict__ PF = params + (size_t)depth_total * BLOCK_STRIDE;   // norm.weight, norm.bias
    const int Lx = L - len_z;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int T = w + NW * i;
        if (T < NT) {
            if (resid != nullptr) {
                float* dst = resid + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(dst + 16 * c, x[i][c]);
            }
            if (16 * T >= len_z) {   // len_z is a multiple of 16 for both supported geometries
                f4 h[NC];
                layer_norm_img(x[i], h, PF, PF + C, q);
                float* dst = feat + ((size_t)b * Lx + (16 * T - len_z) + tok) * C + 4 * q;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(dst + 16 * c, h[c]);
            }
        }
    }
}

}  // namespace vtb
