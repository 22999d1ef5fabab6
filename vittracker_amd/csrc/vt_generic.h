// vt_generic.h -- the vit_dist forward for ANY stride-16 (template, search) geometry and (round 6) ANY widths of the vit_dist config surface.
//
// build_ostrack_dist (lib/models/vit_dist/vit_dist.py:159-198) builds the model from whatever DATA.TEMPLATE.SIZE / DATA.SEARCH.SIZE
// the YAML names (lib/utils/ce_utils.py:22-32 lists template feature sizes 8 / 12 / 7 / 14); the tuned kernels of this library are
// written for the two geometries the repository's configs use ((64, 128) and (128, 256): tile counts and map sides are template
// parameters there, which is where their register blocking comes from).  Every other geometry runs HERE: plain fp32 kernels, one
// thread per output value, fp32 FMA accumulation, no MFMA, no LDS tiling -- correct for any sizes that are multiples of 16 (token
// counts need not be multiples of 16: no tile padding, no key masking), at reference-implementation speed.  Same folded weights
// (BatchNorm into the convs, LayerNorm-1 / -2 affine into qkv / fc1, in fp64 at vt_load_weights), same outputs, same C ABI.
//
//   stem     4 x conv3x3 stride 2 pad 1 (+ folded BN), Hardswish after the first three (vit_dist.py:36-54); the last layer writes
//            token rows (+ pos_embed) straight into the (B, L, C) token matrix, template rows first (vit_dist.py:81-84)
//   blocks   LayerNorm -> qkv; softmax(q k^T / sqrt(C)) v, one thread per query with an online softmax over the keys (exact: the
//            running maximum only rescales); proj + residual; LayerNorm -> fc1 -> GELU(erf); fc2 + residual
//            (lib/models/layers/attn_blocks.py:117-133, attn.py:33-59); final LayerNorm of the search rows (vit_dist.py:94,126)
//   head     3 towers x 4 x (conv3x3 + folded BN + ReLU), 1x1 conv, sigmoid + clamp (lib/models/layers/head.py:98-201); the
//            decode is vth::decode_kernel (shape-generic already)
#pragma once
#include "vt_common.h"

namespace vtg {

constexpr float LN_EPS = 1e-5f;

// Round 6: the widths are run-time values too -- build_ostrack_dist takes embed_dim = MODEL.BACKBONE.CHANNELS, num_heads = MODEL.BACKBONE.HEADS
// and the head's MODEL.HEAD.NUM_CHANNELS from the YAML (lib/models/vit_dist/vit_dist.py:159-164; lib/models/layers/head.py:352-359), and
// every combination other than the shipped (48, 1, 32) runs here, at any stride-16 geometry.
struct Dims {
    int C;        // embed_dim: stem channels C/8, C/4, C/2, C (vit_dist.py:36-54: b16); MLP hidden 4 C
    int heads;    // attention heads, head_dim = C / heads
    int W;        // head tower widths W, W/2, W/4, W/8 (head.py:104-128)
    __host__ __device__ int hid() const { return 4 * C; }
    __host__ __device__ int hd() const { return C / heads; }
    // per-block parameter offsets (floats): plain row-major nn.Linear weights [out][in], LayerNorm folded
    __host__ __device__ int o_wqkv() const { return 0; }
    __host__ __device__ int o_bqkv() const { return 3 * C * C; }
    __host__ __device__ int o_wproj() const { return o_bqkv() + 3 * C; }
    __host__ __device__ int o_bproj() const { return o_wproj() + C * C; }
    __host__ __device__ int o_w1() const { return o_bproj() + C; }
    __host__ __device__ int o_b1() const { return o_w1() + 4 * C * C; }
    __host__ __device__ int o_w2() const { return o_b1() + 4 * C; }
    __host__ __device__ int o_b2() const { return o_w2() + 4 * C * C; }
    __host__ __device__ int block_stride() const { return o_b2() + C; }          // the final norm's gamma, beta (2 C) follow the last block
    // per-tower head parameters: conv i as [cout][cin][9] (BN folded) + bias, conv5 [<= 2][W / 8] + bias
    __host__ __device__ int hch(int i) const { return i == 0 ? C : W >> (i - 1); }
    __host__ __device__ int ho_w(int i) const { int o = 0; for (int k = 0; k < i; ++k) o += hch(k + 1) * hch(k) * 9 + hch(k + 1); return o; }
    __host__ __device__ int ho_b(int i) const { return ho_w(i) + hch(i + 1) * hch(i) * 9; }
    __host__ __device__ int ho_w5() const { return ho_w(4); }
    __host__ __device__ int ho_b5() const { return ho_w5() + 2 * hch(4); }
    __host__ __device__ int tower_stride() const { return ho_b5() + 4; }
};

__device__ __forceinline__ float hardswish(float v) { return v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f; }
__device__ __forceinline__ float gelu_erf(float u) { return u * 0.5f * (1.0f + erff(u * 0.70710678118654752f)); }

// conv3x3, stride 2, pad 1.  in (B, Cin, S, S) NCHW; w [Cout][Cin][9]; out (B, Cout, S/2, S/2) NCHW with Hardswish -- or, tokens != null
// (the last layer): tokens[(b L + row0 + oy So + ox) C + oc] = conv + pos[(oy So + ox) C + oc]
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                        int B, int Cin, int Cout, int S, float* __restrict__ out, float* __restrict__ tokens,
                                                        const float* __restrict__ pos, int L, int row0) {
    const int So = S / 2;
    const size_t total = (size_t)B * Cout * So * So;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int b, oc, oy, ox;
    if (tokens) {
        oc = (int)(idx % Cout);
        const size_t r = idx / Cout;
        const int pix = (int)(r % ((size_t)So * So));
        b = (int)(r / ((size_t)So * So));
        oy = pix / So; ox = pix - oy * So;
    } else {
        ox = (int)(idx % So);
        size_t r = idx / So;
        oy = (int)(r % So); r /= So;
        oc = (int)(r % Cout);
        b = (int)(r / Cout);
    }
    float acc = bias[oc];
    for (int ic = 0; ic < Cin; ++ic) {
        const float* src = in + ((size_t)b * Cin + ic) * S * S;
        const float* wk = w + ((size_t)oc * Cin + ic) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy + r - 1;
            if (iy < 0 || iy >= S) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = 2 * ox + s - 1;
                if (ix < 0 || ix >= S) continue;
                acc = fmaf(src[(size_t)iy * S + ix], wk[r * 3 + s], acc);
            }
        }
    }
    if (tokens) tokens[((size_t)b * L + row0 + oy * So + ox) * Cout + oc] = acc + pos[((size_t)oy * So + ox) * Cout + oc];      // the last layer: Cout = embed_dim
    else out[idx] = hardswish(acc);
}

// Layer 1 of the stem from the tracker step's uint8 patch (round 6): in = (B, S, S, 3) uint8 HWC, sample_target's return value.  Each tap is
// normalised as Preprocessor.process does it (lib/test/tracker/data_utils.py:14: three separately rounded fp32 operations, `/ 255.0` as the
// multiplication by the float reciprocal a GPU torch runs) and meets the SAME weights as stem_conv_kernel: the result is bit-identical to
// vt_crop + stem_conv_kernel.  One thread per output value, as everything in this file.
__global__ __launch_bounds__(256) void stem_conv_u8_kernel(const unsigned char* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                           int B, int Cout, int S, float m0, float m1, float m2, float s0, float s1, float s2,
                                                           float* __restrict__ out) {
    const int So = S / 2;
    const size_t total = (size_t)B * Cout * So * So;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % So);
    size_t r_ = idx / So;
    const int oy = (int)(r_ % So);
    r_ /= So;
    const int oc = (int)(r_ % Cout), b = (int)(r_ / Cout);
    const float meanv[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    float acc = bias[oc];
    for (int ic = 0; ic < 3; ++ic) {
        const float* wk = w + ((size_t)oc * 3 + ic) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy + r - 1;
            if (iy < 0 || iy >= S) continue;
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int ix = 2 * ox + sx - 1;
                if (ix < 0 || ix >= S) continue;
                float v = (float)in[(((size_t)b * S + iy) * S + ix) * 3 + ic] * (1.0f / 255.0f);
                asm volatile("" : "+v"(v));          // three roundings, as three torch kernels: no contraction into an fma
                v = v - meanv[ic];
                asm volatile("" : "+v"(v));
                v = v / stdv[ic];
                acc = fmaf(v, wk[r * 3 + sx], acc);
            }
        }
    }
    out[idx] = hardswish(acc);
}

// out[row][o] = act(LN_plain(x[row]) . W[o] + bias[o]); the LayerNorm's affine part is folded into W / bias.  ACT: 0 none, 1 GELU(erf)
// (the row is read three times -- mean, variance, product -- from L1: no per-thread array of run-time length)
template <int ACT>
__global__ __launch_bounds__(256) void ln_linear_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                        size_t rows, int C, int OUT, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * OUT) return;
    const size_t row = idx / OUT;
    const int o = (int)(idx - row * OUT);
    const float* xr = x + row * C;
    float mean = 0.f;
    for (int i = 0; i < C; ++i) mean += xr[i];
    mean *= 1.0f / C;
    float var = 0.f;
    for (int i = 0; i < C; ++i) { const float d = xr[i] - mean; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(var * (1.0f / C) + LN_EPS);
    float acc = 0.f;
    const float* wr = W + (size_t)o * C;
    for (int i = 0; i < C; ++i) acc = fmaf(xr[i] - mean, wr[i], acc);
    acc = fmaf(acc, rstd, bias[o]);
    out[idx] = ACT == 1 ? gelu_erf(acc) : acc;
}

// one thread per (query, head): softmax((q . k_j) * head_dim^-0.5) over ALL L keys of the frame, online (attn.py:33-59).  qkv rows =
// [q | k | v] (3 C floats), head h = columns h HD .. of each.  HD = head_dim as a template value (registers); HD = 0: any head_dim
// up to MAXHD through local arrays.
constexpr int MAXHD = 256;
template <int HDT>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, int B, int L, int C, int heads, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * L * heads) return;
    const int h = (int)(idx % heads);
    const size_t tokrow = idx / heads;                 // b L + query
    const int b = (int)(tokrow / L);
    const int HD = HDT > 0 ? HDT : C / heads;
    const float scale = 1.0f / sqrtf((float)HD);       // head_dim ** -0.5 (attn.py:15)
    float q[HDT > 0 ? HDT : MAXHD], o[HDT > 0 ? HDT : MAXHD];
    for (int i = 0; i < HD; ++i) { q[i] = qkv[tokrow * 3 * C + h * HD + i]; o[i] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    const float* kv = qkv + (size_t)b * L * 3 * C + h * HD;
    for (int j = 0; j < L; ++j) {
        const float* kj = kv + (size_t)j * 3 * C + C;
        float s = 0.f;
        for (int i = 0; i < HD; ++i) s = fmaf(q[i], kj[i], s);
        s *= scale;
        const float mn = fmaxf(m, s), corr = expf(m - mn), p = expf(s - mn);
        l = fmaf(l, corr, p);
        for (int i = 0; i < HD; ++i) o[i] = fmaf(o[i], corr, p * kj[C + i]);
        m = mn;
    }
    const float rl = 1.0f / l;
    for (int i = 0; i < HD; ++i) out[tokrow * C + h * HD + i] = o[i] * rl;
}

// x[row][o] += in[row] . W[o] + bias[o]   (proj: K = C; fc2: K = 4 C)
__global__ __launch_bounds__(256) void linear_resid_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                           size_t rows, int C, int K, float* __restrict__ x) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const size_t row = idx / C;
    const int o = (int)(idx - row * C);
    const float* ir = in + row * K;
    const float* wr = W + (size_t)o * K;
    float acc = bias[o];
    for (int k = 0; k < K; ++k) acc = fmaf(ir[k], wr[k], acc);
    x[idx] += acc;
}

// feat[(b Lx + t)][c] = LayerNorm(x[b L + len_z + t]) (affine): the search rows only (vit_dist.py:94,126)
__global__ __launch_bounds__(256) void final_norm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ be,
                                                         int B, int L, int len_z, int C, float* __restrict__ feat) {
    const int Lx = L - len_z;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * Lx * C) return;
    const int c = (int)(idx % C);
    const size_t r = idx / C;
    const int t = (int)(r % Lx), b = (int)(r / Lx);
    const float* xr = x + ((size_t)b * L + len_z + t) * C;
    float mean = 0.f;
    for (int i = 0; i < C; ++i) mean += xr[i];
    mean *= 1.0f / C;
    float var = 0.f;
    for (int i = 0; i < C; ++i) { const float d = xr[i] - mean; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(var * (1.0f / C) + LN_EPS);
    feat[idx] = (xr[c] - mean) * rstd * g[c] + be[c];
}

// conv3x3 stride 1 pad 1 + folded BN + ReLU over pixel-major maps: in [tower][B][F F][Cin] (in_tower_stride = 0: one shared input, the
// normalised search tokens), out [tower][B][F F][Cout]; hw = the towers' parameter blocks (tower_stride apart)
__global__ __launch_bounds__(256) void head_conv_kernel(const float* __restrict__ in, size_t in_tower_stride, const float* __restrict__ hw,
                                                        int tower_stride, int w_off, int b_off, int B, int F, int Cin, int Cout,
                                                        float* __restrict__ out) {
    const size_t per_tower = (size_t)B * F * F * Cout;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * per_tower) return;
    const int t = (int)(idx / per_tower);
    size_t r = idx - (size_t)t * per_tower;
    const int oc = (int)(r % Cout); r /= Cout;
    const int pix = (int)(r % ((size_t)F * F)), b = (int)(r / ((size_t)F * F));
    const int y = pix / F, x = pix - y * F;
    const float* P = hw + (size_t)t * tower_stride;
    const float* src = in + (size_t)t * in_tower_stride + (size_t)b * F * F * Cin;
    float acc = P[b_off + oc];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const int iy = y + rr - 1;
        if (iy < 0 || iy >= F) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int ix = x + s - 1;
            if (ix < 0 || ix >= F) continue;
            const float* px = src + ((size_t)iy * F + ix) * Cin;
            const float* wk = P + w_off + (size_t)oc * Cin * 9 + rr * 3 + s;
            for (int ic = 0; ic < Cin; ++ic) acc = fmaf(px[ic], wk[(size_t)ic * 9], acc);
        }
    }
    out[idx] = fmaxf(acc, 0.f);
}

// conv5 (1x1) of the three towers + sigmoid / clamp on ctr and size (head.py:175-201).  t4: [tower][B][F F][C4], C4 = W / 8
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ t4, const float* __restrict__ hw, int tower_stride, int o_w5,
                                                       int o_b5, int C4, int B, int F, float* __restrict__ score, float* __restrict__ size,
                                                       float* __restrict__ offset) {
    const int n = F * F;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * n) return;
    const int b = (int)(idx / n), pix = (int)(idx - (size_t)b * n);
    const size_t per_tower = (size_t)B * n * C4;
    float o[5];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float* P = hw + (size_t)t * tower_stride;
        const float* v = t4 + (size_t)t * per_tower + idx * C4;
        const int nout = t == 0 ? 1 : 2, o0 = t == 0 ? 0 : (t == 1 ? 1 : 3);
        for (int k = 0; k < nout; ++k) {
            float acc = P[o_b5 + k];
            for (int c = 0; c < C4; ++c) acc = fmaf(v[c], P[o_w5 + k * C4 + c], acc);
            o[o0 + k] = acc;
        }
    }
    score[idx] = sigmoid_clamped(o[0]);
    offset[((size_t)b * 2 + 0) * n + pix] = o[1];
    offset[((size_t)b * 2 + 1) * n + pix] = o[2];
    size[((size_t)b * 2 + 0) * n + pix] = sigmoid_clamped(o[3]);
    size[((size_t)b * 2 + 1) * n + pix] = sigmoid_clamped(o[4]);
}

}  // namespace vtg
