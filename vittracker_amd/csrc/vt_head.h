// vt_head.h -- CENTER box head and bbox decode.
//
// Replaces OstrackDist.forward_head's token -> map reshape (lib/models/vit_dist/vit_dist.py:126-129),
// CenterPredictor.get_score_map / forward / cal_bbox (lib/models/layers/head.py:130-201) and the
// tracker's Hann-windowed second decode (lib/test/tracker/vit_dist.py:103-105).
//
// Towers kernel: one workgroup per (frame, tower); every 3x3 conv (+ folded BN + ReLU) is an
// implicit GEMM on v_mfma_f32_16x16x4_f32 with the folded weights as the A operand (rows = output
// channels) and the activations as the B operand (columns = 16 output pixels), so a result tile is
// already "4 consecutive channels of one pixel per lane" = the storage unit of the next layer.
//
// LDS maps are QUAD-PLANAR: float4 map[icq][pix] = channels 4*icq..4*icq+3 of padded pixel pix,
// pix = row * P + col over a zero-bordered (F+2) x (F+2) grid, plane size NPIX a multiple of 16.
// A 3x3 tap is a constant pixel offset (no bounds tests), and the B operand of k-chunk c for lane
// (px = lane & 15, q = lane >> 4) is ONE ds_read_b128 at quad Q = 4c + q -> (tap, icq) =
// divmod(Q, CIN/4): 16 consecutive pixels per quarter-wave, conflict-free at F = 16
// (tools/lds_conflicts.py).
#pragma once
#include <type_traits>

#include "vt_common.h"
#include "vt_conv.h"

namespace vth {

constexpr int C = 48;   // head input channels
constexpr int W1 = 32;  // MODEL.HEAD.NUM_CHANNELS

constexpr int nchunks(int cin) { return (9 * (cin / 4) + 3) / 4; }
constexpr int ntiles(int cout) { return (cout + 15) / 16; }

// packed per-tower parameter offsets (floats). Images: [oc_tile][chunk][64 lanes][4];
// biases padded to 16 * tiles.
constexpr int O_W1 = 0;
constexpr int O_B1 = O_W1 + ntiles(32) * nchunks(48) * 256;
constexpr int O_W2 = O_B1 + 32;
constexpr int O_B2 = O_W2 + ntiles(16) * nchunks(32) * 256;
constexpr int O_W3 = O_B2 + 16;
constexpr int O_B3 = O_W3 + ntiles(8) * nchunks(16) * 256;
constexpr int O_W4 = O_B3 + 16;
constexpr int O_B4 = O_W4 + ntiles(4) * nchunks(8) * 256;
constexpr int O_W5 = O_B4 + 16;   // [2][4] (ctr uses row 0)
constexpr int O_B5 = O_W5 + 8;    // 2 (+2 pad)
// conv3 / conv4 once more for the 4 x 4-block MFMA form of vt_head3.h (SeqConvQ): [oc group of 4][k = (tap, channel), padded to 16][oc 4]
constexpr int kpad16(int cin) { return (9 * cin + 15) / 16 * 16; }
constexpr int O_W3Q = O_B5 + 4;
constexpr int O_W4Q = O_W3Q + (8 / 4) * kpad16(16) * 4;
constexpr int TOWER_STRIDE = O_W4Q + (4 / 4) * kpad16(8) * 4;
static_assert(TOWER_STRIDE % 4 == 0 && O_W2 % 4 == 0 && O_W3 % 4 == 0 && O_W4 % 4 == 0, "16B alignment");

template <int F>
struct Geo {
    // F = 16: a zero-bordered (F+2) x (F+2) grid; a quarter-wave reads 16 consecutive pixels of ONE row: conflict-free.
    // F = 8: a quarter-wave reads 8 pixels of each of TWO rows; with a 10-pixel pitch the two 8-wide windows overlap in two bank
    // groups (a ds_read_b128 lane covers 4 banks, so 16 float4 = one pass over the 64 banks): 43 % of the LDS-active cycles
    // were conflict cycles.  NOHALO layout: the halo COLUMNS are not stored -- row pitch F = 8, so consecutive rows sit exactly
    // half a bank cycle apart -- and a lane whose tap falls into the left / right halo reads a zero entry from the plane's zero
    // tail instead, placed in the bank groups the other lanes leave free (left: 7 / 15, right: 0 / 8; + dy * 8 keeps that).
    static constexpr bool NOHALO = F == 8;
    static constexpr int P = NOHALO ? F : F + 2;                               // row pitch (pixels)
    static constexpr int ROWS = F + 2;                                          // zero row above and below
    static constexpr int NPIX = NOHALO ? ROWS * P + 32 : ((P * P + 15) / 16) * 16;   // plane size (pixels); NOHALO: 32 zero entries behind the rows
    static constexpr int ZR = ROWS * P, ZL = ROWS * P + 7;                      // NOHALO: zero entries for the right / left halo (+ 8 dy)
    static_assert(!NOHALO || (P == 8 && ZL + 3 * P < NPIX && NPIX % 16 == 0), "zero tail: ZL / ZR + 8 (odd quarter-wave) + 8 dy");
    static constexpr int NT = F * F / 16;                    // 16-pixel output tiles
    static constexpr int NPT = NT / 4;                       // tiles per wave (4 waves)
    static constexpr int QUADS = C / 4 + W1 / 4 + 4;         // in(12) + ping(8) + pong(4)
    static constexpr int LDS_BYTES = QUADS * NPIX * 16;
    // plane index of map pixel (y, x), 0 <= y, x < F
    __host__ __device__ static constexpr int interior(int y, int x) { return NOHALO ? (y + 1) * P + x : (y + 1) * P + x + 1; }
    // per-lane plane indices of the three taps dx = 0, 1, 2 in kernel row dy = 0 for this lane's output pixel of tile t
    // (add dy * P for the other kernel rows)
    __device__ static __forceinline__ void tap_cols(int t, int lane, int (&cb)[3]) {
        const int px = lane & 15;
        static_assert(NOHALO, "the zero-bordered layout adds dx to one base instead");
        // A ds_read_b128 is served in four groups of 16 lanes that mix two quarter-waves -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
        // and the same + 32 (MI355X_MICROARCH.md, LDS) -- i.e. two channel-quad planes (a multiple of 16 entries apart: same banks)
        // per group; the 14 interior lanes of a group cover 14 of the 16 bank groups, and its two halo lanes -- one from each
        // quarter-wave -- take the two free ones: the odd quarter-wave reads its zero entry 8 entries further on.
        const int x = px & 7, m = (2 * t + (px >> 3)) * P + x, zq = 8 * ((lane >> 4) & 1);
        cb[1] = m;
        cb[0] = x > 0 ? m - 1 : ZL + zq;
        cb[2] = x < F - 1 ? m + 1 : ZR + zq;
    }
};

// One 3x3 stride-1 conv + bias + ReLU between two LDS maps on MFMA (core: vt_conv.h).
// Work split over the 4 waves: a 2-output-tile layer (conv1) gives each wave ONE output tile and
// half of the pixel tiles, so the waves stream different weights; 1-output-tile layers split the
// pixel tiles four ways.  Weights move in passes of <= 9 chunks, double-buffered: prefetch() issues
// pass 0 (callable a whole layer early -- weights do not depend on activations) and run() requests
// pass p+1 before the MFMAs of pass p.
// NW = waves working on one tower (4, or 8 where LDS allows only one workgroup per CU: two waves per SIMD).
template <int CIN, int COUT, int F, int NW = 4>
struct HeadConv {
    using G = Geo<F>;
    static constexpr int NQ = CIN / 4, NCH = nchunks(CIN), NOT = ntiles(COUT);
    static constexpr bool SPLIT_OT = NOT == 2;
    static constexpr int TSTEP = SPLIT_OT ? NW / 2 : NW;
    static constexpr int NPT = G::NT / TSTEP;
    static_assert(G::NT % TSTEP == 0 && NPT >= 1, "pixel tiles must divide over the waves");
    static constexpr int MAXC = NCH < 9 ? NCH : 9;
    static constexpr int NPASS = (NCH + MAXC - 1) / MAXC;
    static_assert(NOT <= 2, "layers here have at most 2 output tiles");
    opnd a[2][MAXC][1];      // stored operands (vt_conv.h load_weights)

    __device__ __forceinline__ const float* wbase(const float* __restrict__ wimg, int wave) const {
        return wimg + (SPLIT_OT ? (size_t)(wave & 1) * NCH * 256 : 0);
    }
    __device__ __forceinline__ void prefetch(const float* __restrict__ wimg, int wave, int lane) {
        vtc::load_weights<1, MAXC, NCH>(wbase(wimg, wave), 0, MAXC, lane, a[0]);
    }
    __device__ __forceinline__ void run(const f4* in_map, f4* out_map, const float* __restrict__ wimg,
                                        const float* __restrict__ bias, int wave, int lane) {
        const int q = lane >> 4;
        const int ot = SPLIT_OT ? (wave & 1) : 0, tfirst = SPLIT_OT ? (wave >> 1) : wave;
        f4 acc[NPT][1];
        const f4 bv = ld4(bias + 16 * ot + 4 * q);
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i][0] = bv;
        const float* __restrict__ wb = wbase(wimg, wave);
        int centre[NPT];       // plane index of this lane's output pixel of tile i
        auto passes = [&](auto pass) {
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                constexpr int LASTN = NCH - (NPASS - 1) * MAXC;
                if (p + 1 < NPASS)
                    vtc::load_weights<1, MAXC, NCH>(wb, (p + 1) * MAXC, p + 2 < NPASS ? MAXC : LASTN, lane, a[(p + 1) & 1]);
                if (p + 1 < NPASS) pass(std::integral_constant<int, MAXC>{}, p);
                else pass(std::integral_constant<int, LASTN>{}, p);
            }
        };
        if constexpr (G::NOHALO) {
            int cb[NPT][3];
#pragma unroll
            for (int i = 0; i < NPT; ++i) { G::tap_cols(tfirst + TSTEP * i, lane, cb[i]); centre[i] = cb[i][1] + G::P; }
            // map offset of this lane's quad of chunk c for pixel tile i.  With a multiple of 4 channel-quads per tap a chunk
            // never straddles two taps, so the tap (and with it the column base) is a compile-time property of the unrolled chunk.
            auto at = [&](int c, int i) {
                int tap, icq;
                if constexpr (NQ % 4 == 0) { tap = (4 * c) / NQ; icq = 4 * c - tap * NQ + q; }
                else vtc::decode_quad<NQ>(4 * c + q, tap, icq);
                const int dy = tap / 3, dx = tap - 3 * dy;
                return icq * G::NPIX + dy * G::P + (dx == 0 ? cb[i][0] : (dx == 1 ? cb[i][1] : cb[i][2]));
            };
            passes([&](auto n, int p) { vtc::mma_pass_at<1, NPT, MAXC, decltype(n)::value, true>(in_map, a[p & 1], p * MAXC, at, acc); });
        } else {
            int base[NPT];     // tap (0,0): one row up, one column left of the output pixel in the zero-bordered grid
#pragma unroll
            for (int i = 0; i < NPT; ++i) { base[i] = (tfirst + TSTEP * i) * G::P + (lane & 15); centre[i] = base[i] + G::P + 1; }
            auto off = [&](int c) {
                int tap, icq;
                vtc::decode_quad<NQ>(4 * c + q, tap, icq);
                const int dy = tap / 3, dx = tap - 3 * dy;
                return icq * G::NPIX + dy * G::P + dx;
            };
            passes([&](auto n, int p) { vtc::mma_pass<1, NPT, MAXC, decltype(n)::value, true>(in_map, base, a[p & 1], p * MAXC, off, acc); });
        }
        if (16 * ot + 4 * q < COUT) {       // skip the zero-padded output channels
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                f4 v = acc[i][0];
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                out_map[(4 * ot + q) * G::NPIX + centre[i]] = v;
            }
        }
    }
};

// grid (B, 3): tower 0 = ctr, 1 = offset, 2 = size.   feat: (B, F*F, 48) normalised search tokens.
// DIAG: diagnostic build (VT_SKIP_HEAD); production instantiations compile `skip` out.
// FROM_M1: conv1's output map comes from global memory (head_conv1_kernel ran it over several workgroups); `feat` then points at
// that buffer, [B][3 towers][8 planes][NPIX] float4 with zero borders.
template <int F, int NW = 4, bool DIAG = false, bool FROM_M1 = false>
__global__ __launch_bounds__(NW * 64) void head_towers_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ hw,
                                                          float* __restrict__ score, float* __restrict__ size,
                                                          float* __restrict__ offset, int skip_arg) {   // skip: diagnostic
    using G = Geo<F>;
    const int skip = DIAG ? skip_arg : 0;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    f4* in_map = reinterpret_cast<f4*>(sm);            // 12 quads
    f4* m1 = in_map + (C / 4) * G::NPIX;               // 8 quads
    f4* m2 = m1 + (W1 / 4) * G::NPIX;                  // 4 quads
    const int b = blockIdx.x, t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ tw = hw + (size_t)t * TOWER_STRIDE;

    HeadConv<C, W1, F, NW> c1;
    HeadConv<W1, 16, F, NW> c2;
    HeadConv<16, 8, F, NW> c3;
    HeadConv<8, 4, F, NW> c4;
    if constexpr (FROM_M1) {
        c2.prefetch(tw + O_W2, wave, lane);
        // conv1's output (borders included: they are zero in the buffer) -> m1; clear m2 (its borders are read by conv3)
        const f4* src = reinterpret_cast<const f4*>(feat) + ((size_t)b * 3 + t) * (W1 / 4) * G::NPIX;
        for (int i = threadIdx.x; i < (W1 / 4) * G::NPIX; i += NW * 64) m1[i] = src[i];
        for (int i = threadIdx.x; i < 4 * G::NPIX; i += NW * 64) m2[i] = splat4(0.f);
        c3.prefetch(tw + O_W3, wave, lane);
        __syncthreads();
    } else {
    c1.prefetch(tw + O_W1, wave, lane);        // first weight burst flies during the map set-up
    if (!(skip & 1))
        for (int i = threadIdx.x; i < G::QUADS * G::NPIX; i += NW * 64) in_map[i] = splat4(0.f);
    __syncthreads();
    // (B,HW,C) tokens -> quad planes: map[c/4][p][q] = feat[b][p*F+q][c..c+3]   (vit_dist.py:126-129)
    if (!(skip & 2))
    for (int i = threadIdx.x; i < F * F * (C / 4); i += NW * 64) {
        const int icq = i / (F * F), pix = i % (F * F);
        in_map[icq * G::NPIX + G::interior(pix / F, pix % F)] = ld4(feat + ((size_t)b * F * F + pix) * C + 4 * icq);
    }
    __syncthreads();
    c2.prefetch(tw + O_W2, wave, lane);        // each layer's first burst is requested a layer early
    if (!(skip & 4)) c1.run(in_map, m1, tw + O_W1, tw + O_B1, wave, lane);
    c3.prefetch(tw + O_W3, wave, lane);
    __syncthreads();
    }
    if (!(skip & 8)) c2.run(m1, m2, tw + O_W2, tw + O_B2, wave, lane);
    c4.prefetch(tw + O_W4, wave, lane);
    __syncthreads();
    if (!(skip & 16)) c3.run(m2, m1, tw + O_W3, tw + O_B3, wave, lane);
    __syncthreads();
    if (!(skip & 16)) c4.run(m1, m2, tw + O_W4, tw + O_B4, wave, lane);
    __syncthreads();
    // 1x1 conv + activation (head.py:187,194,200-201)
    for (int pix = threadIdx.x; pix < F * F; pix += NW * 64) {
        const f4 v = m2[G::interior(pix / F, pix % F)];
        const int nout = (t == 0) ? 1 : 2;
        for (int o = 0; o < nout; ++o) {
            const f4 w5 = ld4(tw + O_W5 + 4 * o);
            float y = tw[O_B5 + o];
            y = fmaf(v.x, w5.x, y); y = fmaf(v.y, w5.y, y); y = fmaf(v.z, w5.z, y); y = fmaf(v.w, w5.w, y);
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
// Any of pred / hann / conf may be null.  window may be null (then hann is skipped).
__global__ __launch_bounds__(64) void decode_kernel(const float* __restrict__ score, const float* __restrict__ size,
                                                    const float* __restrict__ offset,
                                                    const float* __restrict__ window, int F,
                                                    float* __restrict__ pred, float* __restrict__ hann,
                                                    float* __restrict__ conf, TrackTail tail, int has_tail) {
    // has_tail: lane 0 also runs the tracker's tail on its frame (vt_track_step: map back, clip, state update, record) --
    // one launch less per step on the small-batch path, where every kernel boundary is a latency
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
    // a score map of NaNs never beats the initial value: keep the gathers in bounds (the box then inherits the NaNs of the
    // maps at index 0 instead of reading far outside them)
    if (i0 >= n) i0 = 0;
    if (i1 >= n) i1 = 0;
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
        if (window != nullptr && (hann != nullptr || has_tail)) {
            const float hb[4] = {((float)(i1 % F) + of[i1]) / fF, ((float)(i1 / F) + of[n + i1]) / fF, sz[i1], sz[n + i1]};
            if (hann != nullptr) {
                hann[b * 4 + 0] = hb[0];
                hann[b * 4 + 1] = hb[1];
                hann[b * 4 + 2] = hb[2];
                hann[b * 4 + 3] = hb[3];
            }
            if (has_tail) update_state_one(b, hb, v0, tail);
        }
        if (conf != nullptr) conf[b] = v0;
    }
}

// ---------------------------------------------------------------------------------------- fused head
// All three towers of a frame in one workgroup of 12 waves (tower = wave / 4), then the two argmax
// decodes from LDS: the token -> map staging is done once instead of three times, and the separate
// decode launch (and its read-back of the maps) disappears.  LDS: one shared input map (12 quad
// planes) + per tower 8 + 4 planes; fits for F = 8 (86 KB), not for F = 16.
// The maps are still written to global memory (they are outputs of the boundary); any of
// pred / hann / conf may be null, window may be null (then hann is skipped).
template <int F>
struct FusedHeadGeo {
    using G = Geo<F>;
    static constexpr int TOWER_F4 = (W1 / 4 + 4) * G::NPIX;                   // m1 (8 planes) + m2 (4 planes)
    static constexpr int OUT_FLOATS = 5 * F * F;                               // score, size x2, offset x2
    static constexpr int LDS_BYTES = ((C / 4) * G::NPIX + 3 * TOWER_F4) * 16 + OUT_FLOATS * 4;
};

template <int F, bool DIAG = false>
__global__ __launch_bounds__(768) void head_fused_kernel(const float* __restrict__ feat, const float* __restrict__ hw,
                                                         const float* __restrict__ window, float* __restrict__ score,
                                                         float* __restrict__ size, float* __restrict__ offset,
                                                         float* __restrict__ pred, float* __restrict__ hann,
                                                         float* __restrict__ conf, int skip_arg,      // skip: diagnostic
                                                         TrackTail tail, int has_tail) {   // has_tail: the decoding lane also runs the tracker's tail (vt_track_step)
    using G = Geo<F>;
    const int skip = DIAG ? skip_arg : 0;
    using FG = FusedHeadGeo<F>;
    static_assert(F * F == 64, "the in-kernel decode assumes one map pixel per lane of a wave");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    f4* in_map = reinterpret_cast<f4*>(sm);                                   // 12 quads, shared by the towers
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = wave >> 2, wv = wave & 3, tid = threadIdx.x - 256 * t;      // tower, wave / thread within the tower
    f4* m1 = in_map + (C / 4) * G::NPIX + t * FG::TOWER_F4;                   // 8 quads
    f4* m2 = m1 + (W1 / 4) * G::NPIX;                                         // 4 quads
    float* outs = reinterpret_cast<float*>(in_map + (C / 4) * G::NPIX + 3 * FG::TOWER_F4);   // [5][F*F]
    const float* __restrict__ tw = hw + (size_t)t * TOWER_STRIDE;

    HeadConv<C, W1, F> c1;
    HeadConv<W1, 16, F> c2;
    HeadConv<16, 8, F> c3;
    HeadConv<8, 4, F> c4;
    c1.prefetch(tw + O_W1, wv, lane);
    // (B,HW,C) tokens -> quad planes, once for all towers (vit_dist.py:126-129): exactly one float4 per thread, requested
    // before the LDS is cleared so that its L2 round trip runs under the clear and the barrier; the decode's window value too
    static_assert(F * F * (C / 4) == 768, "one staged element per thread");
    const int icq_s = threadIdx.x / (F * F), pix_s = threadIdx.x % (F * F);
    f4 fv = splat4(0.f);
    if (!(skip & 2)) fv = ld4(feat + ((size_t)b * F * F + pix_s) * C + 4 * icq_s);
    float win = 0.f;
    if (wave == 0 && window != nullptr) win = window[lane];
    if (!(skip & 1))
        for (int i = threadIdx.x; i < (C / 4) * G::NPIX + 3 * FG::TOWER_F4; i += 768) in_map[i] = splat4(0.f);
    __syncthreads();
    if (!(skip & 2)) in_map[icq_s * G::NPIX + G::interior(pix_s / F, pix_s % F)] = fv;
    __syncthreads();
    c2.prefetch(tw + O_W2, wv, lane);
    if (!(skip & 4)) c1.run(in_map, m1, tw + O_W1, tw + O_B1, wv, lane);
    c3.prefetch(tw + O_W3, wv, lane);
    __syncthreads();
    if (!(skip & 8)) c2.run(m1, m2, tw + O_W2, tw + O_B2, wv, lane);
    c4.prefetch(tw + O_W4, wv, lane);
    __syncthreads();
    if (!(skip & 16)) c3.run(m2, m1, tw + O_W3, tw + O_B3, wv, lane);
    __syncthreads();
    if (!(skip & 16)) c4.run(m1, m2, tw + O_W4, tw + O_B4, wv, lane);
    __syncthreads();
    // 1x1 conv + activation (head.py:187,194,200-201) -> global maps and the LDS copy for the decode
    if (tid < F * F) {
        const int pix = tid;
        const f4 v = m2[G::interior(pix / F, pix % F)];
        const int nout = (t == 0) ? 1 : 2;
        for (int o = 0; o < nout; ++o) {
            const f4 w5 = ld4(tw + O_W5 + 4 * o);
            float y = tw[O_B5 + o];
            y = fmaf(v.x, w5.x, y); y = fmaf(v.y, w5.y, y); y = fmaf(v.z, w5.z, y); y = fmaf(v.w, w5.w, y);
            if (t == 0) { y = sigmoid_clamped(y); score[(size_t)b * F * F + pix] = y; outs[pix] = y; }
            else if (t == 2) { y = sigmoid_clamped(y); size[((size_t)b * 2 + o) * F * F + pix] = y; outs[(1 + o) * F * F + pix] = y; }
            else { offset[((size_t)b * 2 + o) * F * F + pix] = y; outs[(3 + o) * F * F + pix] = y; }
        }
    }
    __syncthreads();
    // cal_bbox on the raw score and on window * score (head.py:142-160; lib/test/tracker/vit_dist.py:103-105)
    if (wave == 0) {
        const float sc = outs[lane];
        float v0 = sc, v1 = window != nullptr ? win * sc : -3.0e38f;
        int i0 = lane, i1 = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            argmax_merge(v0, i0, __shfl_xor(v0, off, 64), __shfl_xor(i0, off, 64));
            argmax_merge(v1, i1, __shfl_xor(v1, off, 64), __shfl_xor(i1, off, 64));
        }
        if (lane == 0) {
            constexpr int n = F * F;
            const float fF = (float)F;
            const float* sz = outs + n;
            const float* of = outs + 3 * n;
            if (pred != nullptr) {
                pred[b * 4 + 0] = ((float)(i0 % F) + of[i0]) / fF;
                pred[b * 4 + 1] = ((float)(i0 / F) + of[n + i0]) / fF;
                pred[b * 4 + 2] = sz[i0];
                pred[b * 4 + 3] = sz[n + i0];
            }
            if (window != nullptr && (hann != nullptr || has_tail)) {
                const float hb[4] = {((float)(i1 % F) + of[i1]) / fF, ((float)(i1 / F) + of[n + i1]) / fF, sz[i1], sz[n + i1]};
                if (hann != nullptr) {
                    hann[b * 4 + 0] = hb[0];
                    hann[b * 4 + 1] = hb[1];
                    hann[b * 4 + 2] = hb[2];
                    hann[b * 4 + 3] = hb[3];
                }
                if (has_tail) update_state_one(b, hb, v0, tail);
            }
            if (conf != nullptr) conf[b] = v0;
        }
    }
}

// ------------------------------------------------------------------------------------------ head_conv1 (small batches, F = 16)
// With few frames a tower's conv1 (69 % of its MACs) on ONE workgroup is one CU's worth of MFMA issue (11.5 us) while the chip
// idles.  Here it is a launch of its own over (4-row strip, tower, frame) workgroups of 8 waves -- wave = (output tile, row of
// the strip), 108 MFMAs each -- on a 6-row slice of the input map; the ReLU'd result goes to a global buffer with zero borders
// ([B][3][8 planes][NPIX] float4) that head_towers_kernel<..., FROM_M1> stages instead of running conv1.  Same chains as
// HeadConv<C, W1, 16>::run: same results.
template <int F>
__global__ __launch_bounds__(512) void head_conv1_kernel(const float* __restrict__ feat, const float* __restrict__ hw,
                                                         float* __restrict__ m1g) {
    using G = Geo<F>;
    static_assert(F == 16 && !G::NOHALO, "written for the zero-bordered 16 x 16 layout");
    constexpr int ROWS = 4, SR = ROWS + 2, NPL = ((SR * G::P + 15) / 16) * 16;       // strip: 4 rows + halo rows, 112 entries per plane
    constexpr int NQ = C / 4, NCH = nchunks(C), MAXC = 9, NPASS = NCH / MAXC;
    static_assert(NCH % MAXC == 0, "27 chunks in 3 passes");
    __shared__ f4 strip[NQ * NPL];
    const int sg = blockIdx.x, t = blockIdx.y, b = blockIdx.z;      // strip (rows 4 sg .. 4 sg + 3), tower, frame
    const int lane = threadIdx.x & 63, q = lane >> 4, px = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ot = wave & 1, lrow = wave >> 1;                       // output tile, row inside the strip
    const float* __restrict__ tw = hw + (size_t)t * TOWER_STRIDE;
    const float* __restrict__ wb = tw + O_W1 + (size_t)ot * NCH * 256;
    opnd a[2][MAXC][1];      // stored operands (vt_conv.h load_weights)
    vtc::load_weights<1, MAXC, NCH>(wb, 0, MAXC, lane, a[0]);
    for (int i = threadIdx.x; i < NQ * NPL; i += 512) strip[i] = splat4(0.f);
    __syncthreads();
    // rows 4 sg - 1 .. 4 sg + 4 of the frame's map (those inside the image), local row = row - (4 sg - 1)
    for (int i = threadIdx.x; i < SR * F * NQ; i += 512) {
        const int icq = i / (SR * F), r = (i / F) % SR, col = i % F, row = ROWS * sg - 1 + r;
        if (row >= 0 && row < F)
            strip[icq * NPL + r * G::P + col + 1] = ld4(feat + ((size_t)b * F * F + row * F + col) * C + 4 * icq);
    }
    __syncthreads();
    int base[1] = {lrow * G::P + px};                                // tap (0,0) of this wave's row: one row up, one column left
    f4 acc[1][1] = {{ld4(tw + O_B1 + 16 * ot + 4 * q)}};
    auto off = [&](int c) {
        int tap, icq;
        vtc::decode_quad<NQ>(4 * c + q, tap, icq);
        const int dy = tap / 3, dx = tap - 3 * dy;
        return icq * NPL + dy * G::P + dx;
    };
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (p + 1 < NPASS) vtc::load_weights<1, MAXC, NCH>(wb, (p + 1) * MAXC, MAXC, lane, a[(p + 1) & 1]);
        vtc::mma_pass<1, 1, MAXC, MAXC, true>(strip, base, a[p & 1], p * MAXC, off, acc);
    }
    f4 v = acc[0][0];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    f4* dst = reinterpret_cast<f4*>(m1g) + (((size_t)b * 3 + t) * (W1 / 4) + 4 * ot + q) * G::NPIX;
    dst[G::interior(ROWS * sg + lrow, px)] = v;
}

// cal_bbox on the raw score and on window * score (head.py:142-160; lib/test/tracker/vit_dist.py:103-105); one wave.
// outs: the score plane [F * F] (LDS); sz, of: the size / offset planes [2][F * F], gathered at the two winning pixels only (LDS
// copies, or the global maps this workgroup wrote before its last barrier)
template <int F>
__device__ __forceinline__ void seq_decode(const float* outs, const float* sz, const float* of, const float* __restrict__ window,
                                           int b, int lane, float* __restrict__ pred, float* __restrict__ hann,
                                           float* __restrict__ conf, const TrackTail& tail, int has_tail) {
    constexpr int n = F * F;
    float v0 = -3.0e38f, v1 = -3.0e38f;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float sc = outs[i];
        argmax_merge(v0, i0, sc, i);
        if (window != nullptr) argmax_merge(v1, i1, window[i] * sc, i);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        argmax_merge(v0, i0, __shfl_xor(v0, off, 64), __shfl_xor(i0, off, 64));
        argmax_merge(v1, i1, __shfl_xor(v1, off, 64), __shfl_xor(i1, off, 64));
    }
    if (i0 >= n) i0 = 0;      // a map of NaNs never beats the initial value: keep the gathers in bounds (as decode_kernel)
    if (i1 >= n) i1 = 0;
    if (lane == 0) {
        const float fF = (float)F;
        // size / offset may be the global maps other waves of this workgroup have just written: read them at agent scope (from L2,
        // never from a line this CU's vector cache might hold), whatever the barrier in front of this call already guarantees
        auto g = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        if (pred != nullptr) {
            pred[b * 4 + 0] = ((float)(i0 % F) + g(of + i0)) / fF;
            pred[b * 4 + 1] = ((float)(i0 / F) + g(of + n + i0)) / fF;
            pred[b * 4 + 2] = g(sz + i0);
            pred[b * 4 + 3] = g(sz + n + i0);
        }
        if (window != nullptr && (hann != nullptr || has_tail)) {
            const float hb[4] = {((float)(i1 % F) + g(of + i1)) / fF, ((float)(i1 / F) + g(of + n + i1)) / fF, g(sz + i1), g(sz + n + i1)};
            if (hann != nullptr) {
                hann[b * 4 + 0] = hb[0];
                hann[b * 4 + 1] = hb[1];
                hann[b * 4 + 2] = hb[2];
                hann[b * 4 + 3] = hb[3];
            }
            if (has_tail) update_state_one(b, hb, v0, tail);
        }
        if (conf != nullptr) conf[b] = v0;
    }
}

// ------------------------------------------------------------------------------------------ head_seq
// F = 16 (129 KB of maps: one workgroup per CU whatever the form): one workgroup per FRAME runs the three towers one after
// the other on ONE staged input map and decodes from LDS -- instead of three workgroups per frame (head_towers) that each clear
// and stage the same 48-channel map, plus a decode launch.  Same arithmetic in the same order as head_towers + decode.
template <int F>
struct SeqHeadGeo {
    using G = Geo<F>;
    static constexpr int OUT_FLOATS = 5 * F * F;                               // score, size x2, offset x2
    static constexpr int LDS_BYTES = G::LDS_BYTES + OUT_FLOATS * 4;
};

template <int F, int NW, bool DIAG = false>
__global__ __launch_bounds__(NW * 64) void head_seq_kernel(const float* __restrict__ feat, const float* __restrict__ hw,
                                                       const float* __restrict__ window, float* __restrict__ score,
                                                       float* __restrict__ size, float* __restrict__ offset,
                                                       float* __restrict__ pred, float* __restrict__ hann,
                                                       float* __restrict__ conf, int skip_arg,      // skip: diagnostic
                                                         TrackTail tail, int has_tail) {   // has_tail: the decoding lane also runs the tracker's tail (vt_track_step)
    using G = Geo<F>;
    const int skip = DIAG ? skip_arg : 0;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    f4* in_map = reinterpret_cast<f4*>(sm);            // 12 quads
    f4* m1 = in_map + (C / 4) * G::NPIX;               // 8 quads
    f4* m2 = m1 + (W1 / 4) * G::NPIX;                  // 4 quads
    float* outs = reinterpret_cast<float*>(m2 + 4 * G::NPIX);   // [5][F*F]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int n = F * F;

    HeadConv<C, W1, F, NW> c1;
    HeadConv<W1, 16, F, NW> c2;
    HeadConv<16, 8, F, NW> c3;
    HeadConv<8, 4, F, NW> c4;
    c1.prefetch(hw + O_W1, wave, lane);        // first weight burst flies during the map set-up
    if (!(skip & 1))
        for (int i = threadIdx.x; i < G::QUADS * G::NPIX; i += NW * 64) in_map[i] = splat4(0.f);
    __syncthreads();
    // (B,HW,C) tokens -> quad planes, once for the three towers   (vit_dist.py:126-129)
    // (requesting these before the clear, as head_fused does, measured 1 % slower here: six float4 per thread held over the clear)
    if (!(skip & 2))
        for (int i = threadIdx.x; i < n * (C / 4); i += NW * 64) {
            const int icq = i / n, pix = i % n;
            in_map[icq * G::NPIX + G::interior(pix / F, pix % F)] = ld4(feat + ((size_t)b * n + pix) * C + 4 * icq);
        }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {              // tower 0 = ctr, 1 = offset, 2 = size
        const float* __restrict__ tw = hw + (size_t)t * TOWER_STRIDE;
        c2.prefetch(tw + O_W2, wave, lane);    // each layer's first burst is requested a layer early
        if (!(skip & 4)) c1.run(in_map, m1, tw + O_W1, tw + O_B1, wave, lane);
        c3.prefetch(tw + O_W3, wave, lane);
        __syncthreads();
        if (!(skip & 8)) c2.run(m1, m2, tw + O_W2, tw + O_B2, wave, lane);
        c4.prefetch(tw + O_W4, wave, lane);
        __syncthreads();
        if (!(skip & 16)) c3.run(m2, m1, tw + O_W3, tw + O_B3, wave, lane);
        __syncthreads();
        if (!(skip & 16)) c4.run(m1, m2, tw + O_W4, tw + O_B4, wave, lane);
        if (t < 2) c1.prefetch(tw + TOWER_STRIDE + O_W1, wave, lane);   // the next tower's first burst
        __syncthreads();
        // 1x1 conv + activation (head.py:187,194,200-201) -> global maps and the LDS copy for the decode
        for (int pix = threadIdx.x; pix < n; pix += NW * 64) {
            const f4 v = m2[G::interior(pix / F, pix % F)];
            const int nout = (t == 0) ? 1 : 2;
            for (int o = 0; o < nout; ++o) {
                const f4 w5 = ld4(tw + O_W5 + 4 * o);
                float y = tw[O_B5 + o];
                y = fmaf(v.x, w5.x, y); y = fmaf(v.y, w5.y, y); y = fmaf(v.z, w5.z, y); y = fmaf(v.w, w5.w, y);
                if (t == 0) { y = sigmoid_clamped(y); score[(size_t)b * n + pix] = y; outs[pix] = y; }
                else if (t == 2) { y = sigmoid_clamped(y); size[((size_t)b * 2 + o) * n + pix] = y; outs[(1 + o) * n + pix] = y; }
                else { offset[((size_t)b * 2 + o) * n + pix] = y; outs[(3 + o) * n + pix] = y; }
            }
        }
        // the next tower's conv1 writes m1 (last read by c4, before the barrier above) and its conv2 writes m2 only after
        // the barrier that follows conv1: the 1x1 reads of m2 above need no barrier of their own
    }
    __syncthreads();
    if (wave == 0) seq_decode<F>(outs, outs + n, outs + 3 * n, window, b, lane, pred, hann, conf, tail, has_tail);
}

}  // namespace vth
