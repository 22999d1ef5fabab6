// vt_head.h -- CENTER box head and bbox decode.
//
// Replaces OstrackDist.forward_head's token -> map reshape (lib/models/vit_dist/vit_dist.py:126-129),
// CenterPredictor.get_score_map / forward / cal_bbox (lib/models/layers/head.py:130-201) and the
// tracker's Hann-windowed second decode (lib/test/tracker/vit_dist.py:103-105).
//
// v1 towers kernel: one workgroup per (frame, tower).  The F x F x 48 feature map and every
// intermediate live in LDS as zero-bordered channel planes, so a 3x3 tap is a constant address
// offset and needs no bounds test.  Wave w owns output-channel group w of each layer; its folded
// conv+BN weights are wave-uniform ([group][tap][cin][OCG]) and arrive as scalar operands.
#pragma once
#include "vt_common.h"

namespace vth {

constexpr int C = 48;   // head input channels
constexpr int W1 = 32;  // MODEL.HEAD.NUM_CHANNELS

// packed per-tower offsets (floats): 4 folded 3x3 layers then the 1x1
constexpr int O_W1 = 0;                          // [4][9][48][8]
constexpr int O_B1 = O_W1 + 9 * C * W1;          // 32
constexpr int O_W2 = O_B1 + W1;                  // [4][9][32][4]
constexpr int O_B2 = O_W2 + 9 * W1 * 16;         // 16
constexpr int O_W3 = O_B2 + 16;                  // [4][9][16][2]
constexpr int O_B3 = O_W3 + 9 * 16 * 8;          // 8
constexpr int O_W4 = O_B3 + 8;                   // [4][9][8][1]
constexpr int O_B4 = O_W4 + 9 * 8 * 4;           // 4
constexpr int O_W5 = O_B4 + 4;                   // [2][4] (ctr uses row 0)
constexpr int O_B5 = O_W5 + 8;                   // 2
constexpr int TOWER_STRIDE = ((O_B5 + 2 + 3) / 4) * 4;

// One 3x3 stride-1 layer + ReLU on LDS planes.  NPW = 64-pixel groups per map (F*F/64).
template <int CIN, int COUT, int F, int NPS>
__device__ __forceinline__ void conv_relu_layer(const float* in_s, float* out_s, const float* __restrict__ w,
                                                const float* __restrict__ bias, int wave, int lane) {
    constexpr int OCG = COUT / 4;
    constexpr int NPW = (F * F) / 64;
    constexpr int P = F + 2;
    int pc[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int pix = k * 64 + lane;
        pc[k] = (pix / F + 1) * P + (pix % F) + 1;
    }
    float acc[NPW][OCG];
#pragma unroll
    for (int k = 0; k < NPW; ++k)
#pragma unroll
        for (int j = 0; j < OCG; ++j) acc[k][j] = bias[wave * OCG + j];
    const float* __restrict__ wg = w + (size_t)wave * 9 * CIN * OCG;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3 - 1) * P + (tap % 3 - 1);
#pragma unroll 4
        for (int ic = 0; ic < CIN; ++ic) {
            float v[NPW];
#pragma unroll
            for (int k = 0; k < NPW; ++k) v[k] = in_s[ic * NPS + pc[k] + toff];
#pragma unroll
            for (int j = 0; j < OCG; ++j) {
                const float ww = wg[(tap * CIN + ic) * OCG + j];
#pragma unroll
                for (int k = 0; k < NPW; ++k) acc[k][j] = fmaf(v[k], ww, acc[k][j]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPW; ++k)
#pragma unroll
        for (int j = 0; j < OCG; ++j) out_s[(wave * OCG + j) * NPS + pc[k]] = fmaxf(acc[k][j], 0.f);
}

template <int F>
struct HeadLds {
    static constexpr int P = F + 2;
    static constexpr int NPS = ((P * P + 3) / 4) * 4;        // plane stride (floats)
    static constexpr int FLOATS = (C + W1 + 16) * NPS;       // in(48) + ping(32) + pong(16)
};

// grid (B, 3): tower 0 = ctr, 1 = offset, 2 = size.   feat: (B, F*F, 48) normalised search tokens.
template <int F>
__global__ __launch_bounds__(256) void head_towers_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ hw,
                                                          float* __restrict__ score, float* __restrict__ size,
                                                          float* __restrict__ offset) {
    constexpr int NPS = HeadLds<F>::NPS, P = F + 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* in_s = sm;
    float* a_s = in_s + C * NPS;
    float* b_s = a_s + W1 * NPS;
    const int b = blockIdx.x, t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ tw = hw + (size_t)t * TOWER_STRIDE;

    for (int i = threadIdx.x; i < HeadLds<F>::FLOATS; i += 256) sm[i] = 0.f;
    __syncthreads();
    // (B,HW,C) -> (C,F,F) planes: f[c][p][q] = feat[b][p*F+q][c]   (vit_dist.py:126-129)
    for (int i = threadIdx.x; i < F * F * (C / 4); i += 256) {
        const int pix = i / (C / 4), c4 = i % (C / 4);
        const f4 v = ld4(feat + ((size_t)b * F * F + pix) * C + 4 * c4);
        const int pc = (pix / F + 1) * P + (pix % F) + 1;
        in_s[(4 * c4 + 0) * NPS + pc] = v.x;
        in_s[(4 * c4 + 1) * NPS + pc] = v.y;
        in_s[(4 * c4 + 2) * NPS + pc] = v.z;
        in_s[(4 * c4 + 3) * NPS + pc] = v.w;
    }
    __syncthreads();
    conv_relu_layer<C, W1, F, NPS>(in_s, a_s, tw + O_W1, tw + O_B1, wave, lane);
    __syncthreads();
    conv_relu_layer<W1, 16, F, NPS>(a_s, b_s, tw + O_W2, tw + O_B2, wave, lane);
    __syncthreads();
    conv_relu_layer<16, 8, F, NPS>(b_s, a_s, tw + O_W3, tw + O_B3, wave, lane);
    __syncthreads();
    conv_relu_layer<8, 4, F, NPS>(a_s, b_s, tw + O_W4, tw + O_B4, wave, lane);
    __syncthreads();
    // 1x1 conv + activation (head.py:187,194,200-201)
    for (int pix = threadIdx.x; pix < F * F; pix += 256) {
        const int pc = (pix / F + 1) * P + (pix % F) + 1;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = b_s[c * NPS + pc];
        const int nout = (t == 0) ? 1 : 2;
        for (int o = 0; o < nout; ++o) {
            float y = tw[O_B5 + o];
#pragma unroll
            for (int c = 0; c < 4; ++c) y = fmaf(v[c], tw[O_W5 + o * 4 + c], y);
            if (t == 0) score[(size_t)b * F * F + pix] = sigmoid_clamped(y);
            else if (t == 2) size[((size_t)b * 2 + o) * F * F + pix] = sigmoid_clamped(y);
            else offset[((size_t)b * 2 + o) * F * F + pix] = y;
        }
    }
}

// (value, index) argmax with torch.max's tie rule on CPU: the first (lowest) index wins.
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

// cal_bbox on the raw score and on window * score in one pass; one wave per frame.
// Any of pred / hann / conf / maxscore may be null.  window may be null (then hann is skipped).
__global__ __launch_bounds__(64) void decode_kernel(const float* __restrict__ score, const float* __restrict__ size,
                                                    const float* __restrict__ offset,
                                                    const float* __restrict__ window, int F,
                                                    float* __restrict__ pred, float* __restrict__ hann,
                                                    float* __restrict__ conf) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = F * F;
    float v0 = -3.0e38f, v1 = -3.0e38f;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float s = score[(size_t)b * n + i];
        argmax_merge(v0, i0, s, i);
        if (window != nullptr) argmax_merge(v1, i1, window[i] * s, i);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        argmax_merge(v0, i0, __shfl_xor(v0, off, 64), __shfl_xor(i0, off, 64));
        argmax_merge(v1, i1, __shfl_xor(v1, off, 64), __shfl_xor(i1, off, 64));
    }
    if (lane == 0) {
        const float fF = (float)F;
        const float* sz = size + (size_t)b * 2 * n;
        const float* of = offset + (size_t)b * 2 * n;
        if (pred != nullptr) {
            // head.py:154-156: [(idx_x + off_x)/F, (idx_y + off_y)/F, w, h]
            pred[b * 4 + 0] = ((float)(i0 % F) + of[i0]) / fF;
            pred[b * 4 + 1] = ((float)(i0 / F) + of[n + i0]) / fF;
            pred[b * 4 + 2] = sz[i0];
            pred[b * 4 + 3] = sz[n + i0];
        }
        if (hann != nullptr && window != nullptr) {
            hann[b * 4 + 0] = ((float)(i1 % F) + of[i1]) / fF;
            hann[b * 4 + 1] = ((float)(i1 / F) + of[n + i1]) / fF;
            hann[b * 4 + 2] = sz[i1];
            hann[b * 4 + 3] = sz[n + i1];
        }
        if (conf != nullptr) conf[b] = v0;
    }
}

}  // namespace vth
