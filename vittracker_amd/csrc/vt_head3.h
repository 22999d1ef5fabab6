// vt_head3.h -- the CENTER head's towers (F = 8) on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is split EXACTLY
// into three bf16 pieces, x = h + m + l (8 mantissa bits each, by truncation), and a product a b is the six terms
// hh + hm + mh + hl + lh + mm accumulated in fp32 by v_mfma_f32_16x16x32_bf16 -- what is dropped (ml, lm, ll) is about one fp32
// rounding per product (2^-24.6 of |a b| on average, < 2^-21 always: vt_bf3.h) and vanishes in the accumulator's own (tools/src/probe_bf3.hip: max error / sum |a b| over random 16 x 16 x 16
// products 2.6e-7 against 3.0e-7 for v_mfma_f32_16x16x4_f32).  Same results contract as vt_head.h (reference:
// lib/models/layers/head.py:130-201), same maps, same weights -- folded BatchNorm, fp32 bias, ReLU, conv5 and the decode are
// untouched fp32 code.
//
// Why: an fp32 MFMA occupies a SIMD's matrix pipe for 32 cycles per 16 x 16 x 4 block and shares its issue with the VALU; six
// 16 x 16 x 32 bf16 MFMAs cover EIGHT times the K in 6 x 16 cycles -- 2.6 x less pipe time per fp32-equivalent MAC -- and VALU work
// of other waves issues beside them.  The towers are 63 % MFMA issue at fp32 (vt_head.h).
//
// What changes against vt_head.h:
//   * LDS maps hold the three pieces as planes of 8-byte entries, uint2 map[piece][channel quad][pixel] (4 bf16 = the four
//     channels of a quad); a layer's epilogue splits its ReLU'd fp32 results once, readers never convert.  Geometry as Geo<8>
//     (no halo columns, zero tail).
//   * weights are split on the host (vt_load_weights) into images [output tile][chunk PAIR][piece][64 lanes][8 bf16]: a lane's
//     8 values are its quad of chunk 2 p and its quad of chunk 2 p + 1 (the K order inside an MFMA is free as long as both
//     operands agree), so the fp32 kernels' 16-deep chunks, quad decoding and tap offsets carry over unchanged.
//   * an odd last chunk pairs with zero weights (the activation read for it clamps to a valid quad, as pad quads always did).
#pragma once
#include "vt_bf3.h"
#include "vt_head.h"

#ifndef VT_F16
#ifndef VT_SEQ3_C2_MAXP
#define VT_SEQ3_C2_MAXP 5     // head_seq3's conv2: chunk pairs (= taps) per weight pass -- 5 + 4: 0 B of scratch; all 9 resident: 44 B (NOTES R5-9)
#endif
#ifndef VT_SEQ3_C2_ACT
#define VT_SEQ3_C2_ACT 8      // waves that work on it; 4 (half the weight stream, four tiles per wave) compiles to 40-136 B of scratch: not used
#endif
#ifndef VT_H3_SKIP
#define VT_H3_SKIP 0       // timing experiments only (wrong results): 1 = no conv1, 2 = no conv2-4 MFMAs (head_seq3: no conv2), 4 = no weight loads in conv1's steady state,
                           // 8 = head_seq3: no conv3 / conv4, 16 = head_seq3: no activation reads in conv1's steady state
#endif
namespace vth3 {

using vth::C;
using vth::W1;
using vth::nchunks;
using vth::ntiles;
using vt3::bf16x8;
using vt3::u32x2;
using vt3::u32x4;
using vt3::split3;
using vt3::join3;
using G = vth::Geo<8>;

constexpr int npairs(int cin) { return (nchunks(cin) + 1) / 2; }
// weight images per tower, in 16-byte units: [ot][pair][piece][64]
constexpr int img16(int cin, int cout) { return ntiles(cout) * npairs(cin) * 3 * 64; }
constexpr int O3_W1 = 0;
constexpr int O3_W2 = O3_W1 + img16(48, 32);
constexpr int O3_W3 = O3_W2 + img16(32, 16);
constexpr int O3_W4 = O3_W3 + img16(16, 8);
constexpr int TOWER3_STRIDE = O3_W4 + img16(8, 4);      // 16-byte units

// One 3x3 stride-1 conv + bias + ReLU between two piece-planar LDS maps.  Work split over the NW waves of a tower as in
// vth::HeadConv: a 2-output-tile layer gives each wave one output tile and half of the pixel tiles, 1-tile layers split the
// pixel tiles.  Weights move in passes of <= MAXP chunk pairs, double-buffered in registers.
// ACT: waves of the tower that work on a 1-output-tile layer (the others skip it).  Every active wave streams the layer's whole
// weight image from L2, and with the MFMAs 2.6 x cheaper that stream -- not the matrix pipe -- bounds the towers (1.1 MB per frame
// with four waves per layer against 405 KB of distinct weights, at ~72 GB/s per CU); two waves with two pixel tiles each halve
// it for conv2-4 but double those waves' chains: measured 17.3 against 16.6 us, so all four work.
template <int CIN, int COUT, int NW = 4, int ACT = NW, int MAXP_ = 2>
struct HeadConv3 {
    static constexpr int NQ = CIN / 4, NQO = COUT / 4, NCH = nchunks(CIN), NCP = npairs(CIN), NOT = ntiles(COUT);
    static constexpr bool SPLIT_OT = NOT == 2;
    static constexpr int TSTEP = SPLIT_OT ? NW / 2 : ACT;
    static constexpr int NPT = G::NT / TSTEP;
    static_assert(G::NT % TSTEP == 0 && NPT >= 1 && NOT <= 2, "work split");
    static constexpr int MAXP = NCP < MAXP_ ? NCP : MAXP_;      // pairs per register pass: 2 x 3 pieces x 4 registers, double-buffered = 48 registers (768 threads: 168 per lane)
    static constexpr int NPASS = (NCP + MAXP - 1) / MAXP;
    static constexpr int PS_IN = NQ * G::NPIX, PS_OUT = NQO * G::NPIX;       // piece strides (entries)
    u32x4 a[2][MAXP][3];

    __device__ __forceinline__ const u32x4* wbase(const u32x4* __restrict__ wimg, int wave) const {
        return wimg + (SPLIT_OT ? (size_t)(wave & 1) * NCP * 192 : 0);
    }
    __device__ __forceinline__ void load_pass(const u32x4* __restrict__ wb, int p0, int n, int lane, u32x4 (&dst)[MAXP][3]) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if (k < n)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) dst[k][pc] = wb[((size_t)(p0 + k) * 3 + pc) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wimg, int wave, int lane) {
        load_pass(wbase(wimg, wave), 0, MAXP, lane, a[0]);
    }
    // FIRST: run() requests the first pass itself (no prefetch() a layer earlier: the fused kernel's 168 registers per lane do not
    // hold the next layer's first pass beside this layer's working set)
    template <bool FIRST = false>
    __device__ __forceinline__ void run(const u32x2* in_map, u32x2* out_map, const u32x4* __restrict__ wimg,
                                        const float* __restrict__ bias, int wave, int lane) {
        if (!SPLIT_OT && wave >= ACT) return;          // `wave` = the wave's slot inside its tower (the caller rotates slots per tower)
        if constexpr (FIRST) prefetch(wimg, wave, lane);
        const int q = lane >> 4;
        const int ot = SPLIT_OT ? (wave & 1) : 0, tfirst = SPLIT_OT ? (wave >> 1) : wave;
        f4 acc[NPT];
        const f4 bv = ld4(bias + 16 * ot + 4 * q);
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = bv;
        const u32x4* __restrict__ wb = wbase(wimg, wave);
        int cb[NPT][3], centre[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) { G::tap_cols(tfirst + TSTEP * i, lane, cb[i]); centre[i] = cb[i][1] + G::P; }
        // entry offset (inside a piece) of this lane's quad of 16-deep chunk c for pixel tile i
        auto at = [&](int c, int i) {
            int tap, icq;
            if constexpr (NQ % 4 == 0) { const int cc = c < NCH ? c : NCH - 1; tap = (4 * cc) / NQ; icq = 4 * cc - tap * NQ + q; }
            else vtc::decode_quad<NQ>(4 * c + q, tap, icq);
            const int dy = tap / 3, dx = tap - 3 * dy;
            return icq * G::NPIX + dy * G::P + (dx == 0 ? cb[i][0] : (dx == 1 ? cb[i][1] : cb[i][2]));
        };
        auto read_b = [&](int cp, u32x4 (&b)[NPT][3]) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int o0 = at(2 * cp, i), o1 = at(2 * cp + 1, i);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    const u32x2 lo = in_map[pc * PS_IN + o0], hi = in_map[pc * PS_IN + o1];
                    b[i][pc] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
        };
        u32x4 b[2][NPT][3];
        read_b(0, b[0]);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            constexpr int LASTN = NCP - (NPASS - 1) * MAXP;
            const int n = p + 1 < NPASS ? MAXP : LASTN;
            if (p + 1 < NPASS) load_pass(wb, (p + 1) * MAXP, p + 2 < NPASS ? MAXP : LASTN, lane, a[(p + 1) & 1]);
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                if (k >= n) break;
                const int cp = p * MAXP + k;
                if (cp + 1 < NCP) {
                    read_b(cp + 1, b[(cp + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);        // keep the next pair's reads ahead of this pair's MFMAs
                }
                const u32x4 (&w)[3] = a[p & 1][k];
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const u32x4 (&x)[3] = b[cp & 1][i];
                    auto mm = [&](int wp, int xp) {
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[wp]), __builtin_bit_cast(bf16x8, x[xp]), acc[i], 0, 0, 0);
                    };
                    mm(2, 0); mm(0, 2); mm(1, 1); mm(1, 0); mm(0, 1); mm(0, 0);      // smallest terms first
                }
            }
        }
        if (16 * ot + 4 * q < COUT) {       // skip the zero-padded output channels
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                f4 v = acc[i];
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                u32x2 h, m, l;
                split3(v, h, m, l);
                const int e = (4 * ot + q) * G::NPIX + centre[i];
                out_map[e] = h; out_map[PS_OUT + e] = m; out_map[2 * PS_OUT + e] = l;
            }
        }
    }
    // ---- the same layer with its weights shared by the tower's four waves through LDS (1-output-tile layers, one pixel tile per
    // wave): every wave of a tower needs the WHOLE image, so streamed per wave it crosses the L2 -> CU path four times.  The images
    // of conv2-4 move as one sequence of staged passes of <= SP chunk pairs (WeightPipe below): pass j is fetched by the tower's 256
    // threads at the start of pass j - 2, parked in staging buffer j & 1 at the end of pass j - 1 and read by all four waves in
    // pass j; one workgroup barrier per pass, which is also the layer's barrier after its last pass.  (Fetched only one pass ahead
    // every pass waited for its own L2 round trip: 6 passes x ~1 us.)
    static constexpr int SP = 3, NSP = (NCP + SP - 1) / SP, SBUF16 = SP * 192;       // staging buffer: 9 KiB
    static constexpr int pass_n16(int p) { return (p + 1 < NSP ? SP : NCP - (NSP - 1) * SP) * 192; }
    template <typename Pipe>
    __device__ __forceinline__ void run_staged(const u32x2* in_map, u32x2* out_map, const float* __restrict__ bias, int wave, int lane,
                                               int g0, Pipe& pipe) {
        static_assert(NOT == 1 && NPT == 1 && ACT == NW, "staged form: one output tile, one pixel tile per wave");
        const int q = lane >> 4;
        f4 acc = ld4(bias + 4 * q);
        int cb[3];
        G::tap_cols(wave, lane, cb);
        const int centre = cb[1] + G::P;
        auto at = [&](int c) {
            int tap, icq;
            if constexpr (NQ % 4 == 0) { const int cc = c < NCH ? c : NCH - 1; tap = (4 * cc) / NQ; icq = 4 * cc - tap * NQ + q; }
            else vtc::decode_quad<NQ>(4 * c + q, tap, icq);
            const int dy = tap / 3, dx = tap - 3 * dy;
            return icq * G::NPIX + dy * G::P + (dx == 0 ? cb[0] : (dx == 1 ? cb[1] : cb[2]));
        };
        auto read_b = [&](int cp, u32x4 (&b)[3]) {
            const int o0 = at(2 * cp), o1 = at(2 * cp + 1);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const u32x2 lo = in_map[pc * PS_IN + o0], hi = in_map[pc * PS_IN + o1];
                b[pc] = u32x4{lo.x, lo.y, hi.x, hi.y};
            }
        };
        u32x4 b[2][3], w[2][3];
        read_b(0, b[0]);
#pragma unroll
        for (int p = 0; p < NSP; ++p) {
            const int n = pass_n16(p) / 192, g = g0 + p;
            pipe.fetch(g + 2);
            const u32x4* sb = pipe.buf(g);
            auto read_w = [&](int k, u32x4 (&ww)[3]) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) ww[pc] = sb[(k * 3 + pc) * 64 + lane];
            };
            read_w(0, w[0]);
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                if (k >= n) break;
                const int cp = p * SP + k;
                if (k + 1 < n) read_w(k + 1, w[(k + 1) & 1]);
                if (cp + 1 < NCP) read_b(cp + 1, b[(cp + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&x)[3] = b[cp & 1];
                const u32x4 (&ww)[3] = w[k & 1];
                auto mm = [&](int wp, int xp) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ww[wp]), __builtin_bit_cast(bf16x8, x[xp]), acc, 0, 0, 0);
                };
                if (!(VT_H3_SKIP & 2)) { mm(2, 0); mm(0, 2); mm(1, 1); mm(1, 0); mm(0, 1); mm(0, 0); }
            }
            if (p == NSP - 1 && 4 * q < COUT) {
                f4 v = acc;
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                u32x2 h, m, l;
                split3(v, h, m, l);
                const int e = q * G::NPIX + centre;
                out_map[e] = h; out_map[PS_OUT + e] = m; out_map[2 * PS_OUT + e] = l;
            }
            pipe.park(g + 1);
            __syncthreads();
        }
    }
};

// LDS entries (uint2) of the maps: input 3 x 12 planes, per tower m1 3 x 8 and m2 3 x 4 planes
constexpr int IN_E = 3 * (C / 4) * G::NPIX, M1_E = 3 * (W1 / 4) * G::NPIX, M2_E = 3 * 4 * G::NPIX;
constexpr int TOWERS3_LDS_BYTES = (IN_E + M1_E + M2_E) * 8;
constexpr int SBUF_BYTES = 3 * 192 * 16;                        // one staged pass of weights (HeadConv3::run_staged): 9 KiB
static_assert(IN_E * 8 / 3 >= SBUF_BYTES, "a tower's second staging buffer is its third of the input map (dead after conv1)");
constexpr int FUSED3_LDS_BYTES = (IN_E + 3 * (M1_E + M2_E)) * 8 + 5 * 64 * 4 + 3 * SBUF_BYTES;
static_assert(FUSED3_LDS_BYTES <= 160 * 1024, "LDS");

// The staged passes of conv2, conv3, conv4 of one tower, in order (3 + 2 + 1 passes), two register slots and two LDS buffers.
struct WeightPipe {
    static constexpr int N2 = HeadConv3<W1, 16>::NSP, N3 = HeadConv3<16, 8>::NSP, N4 = HeadConv3<8, 4>::NSP, NP = N2 + N3 + N4;
    const u32x4* tw3;
    u32x4 *buf0, *buf1;
    int tid;
    u32x4 r[2][3];
    // 16-byte offset (inside the tower's images) and size of global pass j
    static constexpr int off16(int j) {
        return j < N2 ? O3_W2 + j * HeadConv3<W1, 16>::SBUF16 : j < N2 + N3 ? O3_W3 + (j - N2) * HeadConv3<16, 8>::SBUF16 : O3_W4 + (j - N2 - N3) * HeadConv3<8, 4>::SBUF16;
    }
    static constexpr int n16(int j) {
        return j < N2 ? HeadConv3<W1, 16>::pass_n16(j) : j < N2 + N3 ? HeadConv3<16, 8>::pass_n16(j - N2) : HeadConv3<8, 4>::pass_n16(j - N2 - N3);
    }
    __device__ __forceinline__ const u32x4* buf(int j) const { return (j & 1) ? buf1 : buf0; }
    __device__ __forceinline__ void fetch(int j) {           // j compile-time at every call site (the pass loops are unrolled)
        if (j >= NP) return;
        const u32x4* __restrict__ src = tw3 + off16(j);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + 256 * k < n16(j)) r[j & 1][k] = src[tid + 256 * k];
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void park(int j) {
        if (j >= NP) return;
        u32x4* dst = (j & 1) ? buf1 : buf0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + 256 * k < n16(j)) dst[tid + 256 * k] = r[j & 1][k];
    }
};

// tokens (B,HW,C) fp32 -> the piece planes of the input map (vit_dist.py:126-129): item = (quad, pixel)
__device__ __forceinline__ void stage_tokens(u32x2* in_map, const float* __restrict__ feat, int b, int item) {
    const int icq = item >> 6, pix = item & 63;
    u32x2 h, m, l;
    split3(ld4(feat + ((size_t)b * 64 + pix) * C + 4 * icq), h, m, l);
    const int e = icq * G::NPIX + G::interior(pix >> 3, pix & 7);
    in_map[e] = h; in_map[(C / 4) * G::NPIX + e] = m; in_map[2 * (C / 4) * G::NPIX + e] = l;
}

// grid (B, 3), 256 threads: the per-tower form (small batches).  hw: the fp32 parameter block of vt_head.h (biases, conv5); hw3: images.
__global__ __launch_bounds__(256) void head_towers3_kernel(const float* __restrict__ feat, const float* __restrict__ hw,
                                                          const u32x4* __restrict__ hw3, float* __restrict__ score,
                                                          float* __restrict__ size, float* __restrict__ offset) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    u32x2* in_map = reinterpret_cast<u32x2*>(sm);
    u32x2* m1 = in_map + IN_E;
    u32x2* m2 = m1 + M1_E;
    const int b = blockIdx.x, t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ tw = hw + (size_t)t * vth::TOWER_STRIDE;
    const u32x4* __restrict__ tw3 = hw3 + (size_t)t * TOWER3_STRIDE;
    HeadConv3<C, W1> c1;
    HeadConv3<W1, 16> c2;
    HeadConv3<16, 8> c3;
    HeadConv3<8, 4> c4;
    c1.prefetch(tw3 + O3_W1, wave, lane);
    for (int i = threadIdx.x; i < (IN_E + M1_E + M2_E) / 2; i += 256) reinterpret_cast<u32x4*>(in_map)[i] = u32x4{0, 0, 0, 0};
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * (C / 4); i += 256) stage_tokens(in_map, feat, b, i);
    __syncthreads();
    c2.prefetch(tw3 + O3_W2, wave, lane);      // each layer's first pass is requested a layer early
    c1.run(in_map, m1, tw3 + O3_W1, tw + vth::O_B1, wave, lane);
    c3.prefetch(tw3 + O3_W3, wave, lane);
    __syncthreads();
    c2.run(m1, m2, tw3 + O3_W2, tw + vth::O_B2, wave, lane);
    c4.prefetch(tw3 + O3_W4, wave, lane);
    __syncthreads();
    c3.run(m2, m1, tw3 + O3_W3, tw + vth::O_B3, wave, lane);
    __syncthreads();
    c4.run(m1, m2, tw3 + O3_W4, tw + vth::O_B4, wave, lane);
    __syncthreads();
    if (threadIdx.x < 64) {      // 1x1 conv + activation (head.py:187,194,200-201)
        const int pix = threadIdx.x, e = G::interior(pix >> 3, pix & 7);
        const f4 v = join3(m2[e], m2[G::NPIX + e], m2[2 * G::NPIX + e]);
        const int nout = (t == 0) ? 1 : 2;
        for (int o = 0; o < nout; ++o) {
            const f4 w5 = ld4(tw + vth::O_W5 + 4 * o);
            float y = tw[vth::O_B5 + o];
            y = fmaf(v.x, w5.x, y); y = fmaf(v.y, w5.y, y); y = fmaf(v.z, w5.z, y); y = fmaf(v.w, w5.w, y);
            if (t == 0) score[(size_t)b * 64 + pix] = sigmoid_clamped(y);
            else if (t == 2) size[((size_t)b * 2 + o) * 64 + pix] = sigmoid_clamped(y);
            else offset[((size_t)b * 2 + o) * 64 + pix] = y;
        }
    }
}

// One workgroup of 12 waves per frame: the three towers + both decodes (the structure of vth::head_fused_kernel<8>).
__global__ __launch_bounds__(768) void head_fused3_kernel(const float* __restrict__ feat, const float* __restrict__ hw,
                                                          const u32x4* __restrict__ hw3, const float* __restrict__ window,
                                                          float* __restrict__ score, float* __restrict__ size,
                                                          float* __restrict__ offset, float* __restrict__ pred,
                                                          float* __restrict__ hann, float* __restrict__ conf, TrackTail tail,
                                                          int has_tail) {      // has_tail: the decoding lane also runs the tracker's tail (vt_track_step)
    constexpr int F = 8;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    u32x2* in_map = reinterpret_cast<u32x2*>(sm);
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = wave >> 2, wv = wave & 3, tid = threadIdx.x - 256 * t;
    u32x2* m1 = in_map + IN_E + t * (M1_E + M2_E);
    u32x2* m2 = m1 + M1_E;
    float* outs = reinterpret_cast<float*>(in_map + IN_E + 3 * (M1_E + M2_E));     // [5][64]
    const float* __restrict__ tw = hw + (size_t)t * vth::TOWER_STRIDE;
    const u32x4* __restrict__ tw3 = hw3 + (size_t)t * TOWER3_STRIDE;
    HeadConv3<C, W1, 4, 4, 1> c1;          // one pair per register pass: the staged layers' two fetch slots are live across conv1
    HeadConv3<W1, 16> c2;
    HeadConv3<16, 8> c3;
    HeadConv3<8, 4> c4;
    c1.prefetch(tw3 + O3_W1, wv, lane);        // flies during the map set-up
    static_assert(64 * (C / 4) == 768, "one staged element per thread");
    const f4 fv = ld4(feat + ((size_t)b * 64 + (threadIdx.x & 63)) * C + 4 * (threadIdx.x >> 6));     // requested before the LDS is cleared
    float win = 0.f;
    if (wave == 0 && window != nullptr) win = window[lane];
    // zero what is read without ever being written: per plane the border rows (entries [0, P) and [(F + 1) P, (F + 2) P)) and the zero tail
    // behind them -- 48 of a plane's 112 entries; the interiors are written (staging, layer epilogues) before they are read.  No barrier
    // between this and the staging: they touch different entries.
    {
        constexpr int NPL = (IN_E + 3 * (M1_E + M2_E)) / G::NPIX, ZU = (G::NPIX - F * G::P) / 2, TOPU = G::P / 2;     // planes; 16-byte units to zero per plane; of them the top row
        static_assert(G::P % 2 == 0 && G::NPIX % 2 == 0 && ((F + 1) * G::P) % 2 == 0, "16-byte units");
        for (int j = threadIdx.x; j < NPL * ZU; j += 768) {
            const int pl = j / ZU, u = j - pl * ZU;
            reinterpret_cast<u32x4*>(in_map + pl * G::NPIX)[u < TOPU ? u : (F + 1) * G::P / 2 + (u - TOPU)] = u32x4{0, 0, 0, 0};
        }
    }
    {
        const int icq = threadIdx.x >> 6, pix = threadIdx.x & 63;
        u32x2 h, m, l;
        split3(fv, h, m, l);
        const int e = icq * G::NPIX + G::interior(pix >> 3, pix & 7);
        in_map[e] = h; in_map[(C / 4) * G::NPIX + e] = m; in_map[2 * (C / 4) * G::NPIX + e] = l;
    }
    __syncthreads();
    // conv2-4: weights through the tower's two LDS staging buffers (run_staged): buffer 0 behind the output rows, buffer 1 the
    // tower's third of the input map once conv1 is done with it.  conv2's first pass is fetched before conv1 and parked after it.
    u32x4* const sbuf0 = reinterpret_cast<u32x4*>(outs + 5 * 64) + t * (SBUF_BYTES / 16);
    u32x4* const sbuf1 = reinterpret_cast<u32x4*>(in_map) + t * (IN_E / 6);          // IN_E / 3 entries of 8 bytes = IN_E / 6 x 16 bytes
    WeightPipe pipe{tw3, sbuf0, sbuf1, tid, {}};
    pipe.fetch(0);
    pipe.fetch(1);
    if (!(VT_H3_SKIP & 1)) c1.run(in_map, m1, tw3 + O3_W1, tw + vth::O_B1, wv, lane);
    pipe.park(0);
    __syncthreads();
    c2.run_staged(m1, m2, tw + vth::O_B2, wv, lane, 0, pipe);
    c3.run_staged(m2, m1, tw + vth::O_B3, wv, lane, WeightPipe::N2, pipe);
    c4.run_staged(m1, m2, tw + vth::O_B4, wv, lane, WeightPipe::N2 + WeightPipe::N3, pipe);
    if (tid < F * F) {
        const int pix = tid, e = G::interior(pix >> 3, pix & 7);
        const f4 v = join3(m2[e], m2[G::NPIX + e], m2[2 * G::NPIX + e]);
        const int nout = (t == 0) ? 1 : 2;
        for (int o = 0; o < nout; ++o) {
            const f4 w5 = ld4(tw + vth::O_W5 + 4 * o);
            float y = tw[vth::O_B5 + o];
            y = fmaf(v.x, w5.x, y); y = fmaf(v.y, w5.y, y); y = fmaf(v.z, w5.z, y); y = fmaf(v.w, w5.w, y);
            if (t == 0) { y = sigmoid_clamped(y); score[(size_t)b * F * F + pix] = y; outs[pix] = y; }
            else if (t == 2) { y = sigmoid_clamped(y); size[((size_t)b * 2 + o) * F * F + pix] = y; outs[(1 + o) * F * F + pix] = y; }
            else { offset[((size_t)b * 2 + o) * F * F + pix] = y; outs[(3 + o) * F * F + pix] = y; }
        }
    }
    __syncthreads();
    if (wave == 0) {      // cal_bbox on the raw score and on window * score (head.py:142-160; lib/test/tracker/vit_dist.py:103-105)
        const float sc = outs[lane];
        float v0 = sc, v1 = window != nullptr ? win * sc : -3.0e38f;
        int i0 = lane, i1 = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            vth::argmax_merge(v0, i0, __shfl_xor(v0, off, 64), __shfl_xor(i0, off, 64));
            vth::argmax_merge(v1, i1, __shfl_xor(v1, off, 64), __shfl_xor(i1, off, 64));
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

// ------------------------------------------------------------------------------------------ head_seq3 (F = 16)
// The sequential head of vt_head.h (one workgroup per frame runs the three towers in turn on ONE staged input map) with conv1 --
// 70 % of a tower's MACs -- as three-piece bf16 products.  Only the INPUT map is held as pieces (tokens are split once, at staging,
// and read by three towers): with conv1's output as pieces too the maps would need 190 KB.  conv1 writes fp32, conv2 is vt_head.h's
// fp32-MFMA layer unchanged, conv3 / conv4 (8 and 4 output channels) run on the 4 x 4-block fp32 MFMA, which pads nothing (SeqConvQ).  LDS: input pieces 94.5 KB + m1 42 KB + m2 21 KB + the score plane 1 KB = 158.5 KB; the
// decode gathers size / offset at its two winning pixels from the global maps the workgroup has just written.
using G16 = vth::Geo<16>;
constexpr int SEQ3_IN_E = 3 * (C / 4) * G16::NPIX;                                         // uint2 entries
constexpr int SEQ3_LDS_BYTES = SEQ3_IN_E * 8 + (W1 / 4 + 4) * G16::NPIX * 16 + 256 * 4;
static_assert(SEQ3_LDS_BYTES <= 160 * 1024, "LDS");
// a quarter-wave reads 16 consecutive 8-byte entries (128 B); the two quarter-waves a ds_read_b64 pass serves together read planes
// icq and icq + 1: NPIX * 8 = 128 (mod 256) puts them in complementary bank halves
static_assert((G16::NPIX * 8) % 256 == 128, "piece planes: conflict-free ds_read_b64");

// One 3 x 3 conv + bias + ReLU of the sequential head as three-piece products: conv1 (48 -> 32) from the input map's pieces, conv2
// (32 -> 16) from conv1's output, which conv1's epilogue writes as pieces.  compute() leaves the ReLU'd results in registers; the
// caller stores them (as pieces / as fp32) behind the barrier that protects the region they go to.  Work split: a 2-output-tile
// layer gives wave -> (output tile wave & 1, pixel tiles (wave >> 1) + NW / 2 * i), a 1-tile layer pixel tiles wave + NW * i (a pixel
// tile = one row of the 16 x 16 map).  Weights: the [ot][pair][piece][lane] images of head_fused3 (a chunk pair of conv2 = one tap's
// 32 channels), in register passes of MAXP chunk pairs, double-buffered.
// ACT: waves that work on a 1-output-tile layer (each then covers 16 / ACT pixel tiles per weight fragment it loads: the layer's weight
// image is streamed once per ACTIVE wave, and that stream -- L1 -> registers at 64 B per cycle and CU -- is what bounds these layers).
template <int CIN, int COUT, int NW, int MAXP_, int ACT = NW>
struct SeqConvB {
    static constexpr int NQ = CIN / 4, NCH = nchunks(CIN), NCP = npairs(CIN), NOT = ntiles(COUT);
    static constexpr int TSTEP = NOT == 2 ? NW / 2 : ACT, NPT = G16::NT / TSTEP;
    static constexpr int MAXP = NCP < MAXP_ ? NCP : MAXP_;
    static constexpr int NPASS = (NCP + MAXP - 1) / MAXP, LASTN = NCP - (NPASS - 1) * MAXP;
    static constexpr int PS_IN = NQ * G16::NPIX;
    static_assert(NQ % 4 == 0 && G16::NT % TSTEP == 0 && NOT >= 1 && NOT <= 2, "a chunk never straddles two taps; pixel tiles divide over the waves");
    u32x4 a[2][MAXP][3];

    __device__ static __forceinline__ bool active(int wave) { return NOT == 2 || ACT == NW || wave < ACT; }
    __device__ static __forceinline__ int ot_of(int wave) { return NOT == 2 ? (wave & 1) : 0; }
    __device__ static __forceinline__ int tile_of(int wave, int i) { return (NOT == 2 ? (wave >> 1) : wave) + TSTEP * i; }
    __device__ __forceinline__ void load_pass(const u32x4* __restrict__ wb, int p0, int n, int lane, u32x4 (&dst)[MAXP][3]) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if (k < n)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) dst[k][pc] = wb[((size_t)(p0 + k) * 3 + pc) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wimg, int wave, int lane) {
        if (active(wave)) load_pass(wimg + (size_t)ot_of(wave) * NCP * 192, 0, MAXP, lane, a[0]);
    }
    __device__ __forceinline__ void compute(const u32x2* in_map, const u32x4* __restrict__ wimg, const float* __restrict__ bias, int wave,
                                            int lane, bool skip, f4 (&acc)[NPT]) {
        const int q = lane >> 4, ot = ot_of(wave);
        const u32x4* __restrict__ wb = wimg + (size_t)ot * NCP * 192;
        const f4 bv = ld4(bias + 16 * ot + 4 * q);
        int base[NPT];      // tap (0,0) of this lane's pixel of tile i in plane q: one row up, one column left in the zero-bordered grid
#pragma unroll
        for (int i = 0; i < NPT; ++i) { acc[i] = bv; base[i] = q * G16::NPIX + tile_of(wave, i) * G16::P + (lane & 15); }
        if (skip || !active(wave)) return;
        // entry offset (compile-time part) of chunk c: its tap and its first channel quad
        auto off = [&](int c) {
            const int cc = c < NCH ? c : NCH - 1, tap = (4 * cc) / NQ, icq0 = 4 * cc - tap * NQ, dy = tap / 3, dx = tap - 3 * dy;
            return icq0 * G16::NPIX + dy * G16::P + dx;
        };
        // one opaque LDS base per (pixel tile, piece): every chunk offset (< 30 KB) then sits in the instruction's offset field
        lds_cptr<u32x2> pb[NPT][3];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) pb[i][pc] = lds_lane_base<u32x2>(in_map + pc * PS_IN, base[i] * 8);
        auto read_b = [&](int cp, u32x4 (&b)[NPT][3]) {
            const int o0 = off(2 * cp), o1 = off(2 * cp + 1);
#pragma unroll
            for (int i = 0; i < NPT; ++i)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    const u32x2 lo = pb[i][pc][o0], hi = pb[i][pc][o1];
                    b[i][pc] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
        };
        u32x4 b[2][NPT][3];
        read_b(0, b[0]);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int n = p + 1 < NPASS ? MAXP : LASTN;
            if (p + 1 < NPASS && !(VT_H3_SKIP & 4)) load_pass(wb, (p + 1) * MAXP, p + 2 < NPASS ? MAXP : LASTN, lane, a[(p + 1) & 1]);
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                if (k >= n) break;
                const int cp = p * MAXP + k;
                if (cp + 1 < NCP && !(VT_H3_SKIP & 16)) {
                    read_b(cp + 1, b[(cp + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);        // keep the next pair's reads ahead of this pair's MFMAs
                }
                const u32x4 (&w)[3] = a[p & 1][k];
                // term by term over the pixel tiles: consecutive MFMAs never share an accumulator
                auto mm = [&](int wp, int xp) {
#pragma unroll
                    for (int i = 0; i < NPT; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[wp]), __builtin_bit_cast(bf16x8, b[cp & 1][i][xp]), acc[i], 0, 0, 0);
                };
                mm(2, 0); mm(0, 2); mm(1, 1); mm(1, 0); mm(0, 1); mm(0, 0);      // smallest terms first
            }
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            f4 v = acc[i];
            acc[i] = f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        }
    }
    // this lane's results of compute(): channels 4 (4 ot + q) .. + 3 of pixel (row tile_of(wave, i), column lane & 15)
    __device__ static __forceinline__ int out_entry(int wave, int lane, int i) {       // entry inside ONE plane set: quad * NPIX + pixel
        return (4 * ot_of(wave) + (lane >> 4)) * G16::NPIX + G16::interior(tile_of(wave, i), lane & 15);
    }
};
// the k-th entry of a plane that is not an interior pixel: the border rows and columns of the (F + 2) x (F + 2) grid and the plane's tail
constexpr int SEQ_HALO = G16::NPIX - 16 * 16;
__device__ __forceinline__ int seq_halo_pix(int k) {
    constexpr int P = G16::P;
    return k < P ? k : (k < 2 * P ? (P - 1) * P + (k - P) : (k < 2 * P + 32 ? (1 + ((k - 2 * P) >> 1)) * P + ((k - 2 * P) & 1) * (P - 1) : P * P + (k - 2 * P - 32)));
}

// conv3 (16 -> 8) and conv4 (8 -> 4) of the sequential head on v_mfma_f32_4x4x1_16B_f32 (sixteen independent 4 x 4 blocks, K = 1):
// on the 16 x 16 MFMA these layers fill 8 and 4 of the 16 output rows; here a block's rows are ONE group of four output channels
// and its columns four pixels, so nothing is padded: a wave covers 64 pixels (four map rows) x 4 output channels per instruction
// (tools/src/probe_mfma4x4.hip: A from lane 4 b + row, B from lane 4 b + column, D register r of lane 4 b + j = row r, column j;
// 8.8 cycles per instruction for one wave, and two waves of a SIMD both keep that rate).  Lane l supplies its own pixel's
// activations (one ds_read_b128 = four k) and receives its pixel's four output channels = one float4 of the quad-planar map.
// Weights use the instruction's A broadcast (cbsz = 4: every block multiplies by block `abid`'s rows): register kg of a lane holds
// output channel l & 3 at k = 16 kg + (l >> 2), so a layer's whole [k][oc] image is 9 (conv3) / 5 (conv4) registers, loaded
// a layer early by one coalesced dword load each.  Jobs = pixel groups (4) x output-channel groups: conv3 keeps all eight waves
// busy, conv4 four (one per SIMD).  Four accumulators (k mod 4) keep dependent MFMAs 35 cycles apart (14.6 needed).
template <int AB>
__device__ __forceinline__ f4 mfma4x4_bcast(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, AB, 0); }
__device__ __forceinline__ f4 mfma4x4_bcast(float a, float b, f4 c, int ab) {      // ab is a constant after unrolling
    switch (ab) {
        case 0: return mfma4x4_bcast<0>(a, b, c);   case 1: return mfma4x4_bcast<1>(a, b, c);
        case 2: return mfma4x4_bcast<2>(a, b, c);   case 3: return mfma4x4_bcast<3>(a, b, c);
        case 4: return mfma4x4_bcast<4>(a, b, c);   case 5: return mfma4x4_bcast<5>(a, b, c);
        case 6: return mfma4x4_bcast<6>(a, b, c);   case 7: return mfma4x4_bcast<7>(a, b, c);
        case 8: return mfma4x4_bcast<8>(a, b, c);   case 9: return mfma4x4_bcast<9>(a, b, c);
        case 10: return mfma4x4_bcast<10>(a, b, c); case 11: return mfma4x4_bcast<11>(a, b, c);
        case 12: return mfma4x4_bcast<12>(a, b, c); case 13: return mfma4x4_bcast<13>(a, b, c);
        case 14: return mfma4x4_bcast<14>(a, b, c); default: return mfma4x4_bcast<15>(a, b, c);
    }
}
template <int CIN, int COUT, int NW>
struct SeqConvQ {
    static constexpr int NQ = CIN / 4, NG = COUT / 4, NPG = 16 * 16 / 64, JOBS = NG * NPG, KP = vth::kpad16(CIN), NKR = KP / 16;
    static_assert(CIN % 4 == 0 && COUT % 4 == 0 && JOBS <= NW, "shapes");
    float wr[NKR];
    __device__ __forceinline__ void prefetch(const float* __restrict__ wq, int wave, int lane) {
        if (wave >= JOBS) return;
#pragma unroll
        for (int kg = 0; kg < NKR; ++kg) wr[kg] = wq[(size_t)(wave % NG) * KP * 4 + kg * 64 + lane];
    }
    // Pixel of (pixel group pg, lane): column lane & 15 of row 2 pg + (lane >> 5) + 8 ((lane >> 4) & 1) -- the four 16-lane rows of a wave are
    // map rows r, r + 8, r + 1, r + 9.  A ds_read_b128 is served in groups that mix lanes 0-3 / 12-15 with lanes 20-27 (and 32-35 / 44-47 with
    // 52-59): with those on ADJACENT rows (row pitch 18 x 16 B = 72 dwords: bank offset 8) columns 12-13 of one row and 10-11 of the next
    // share banks and every read and write of the fp32 maps took 8 LDS cycles instead of 4 (1.16 M conflict cycles per launch, all of the
    // kernel's); eight rows apart the offset is 576 dwords = 0 mod 64 (tools/lds_conflicts.py).
    __device__ static __forceinline__ int pixel_of(int pg, int lane) { return 16 * (2 * pg + (lane >> 5) + 8 * ((lane >> 4) & 1)) + (lane & 15); }
    // the ReLU'd output channels 4 g .. 4 g + 3 (g = wave % NG) of pixel pixel_of(wave / NG, lane); waves >= JOBS have no job
    __device__ __forceinline__ f4 compute(const f4* in_map, const float* __restrict__ bias, int wave, int lane) {
        const int g = wave % NG, pg = wave / NG;
        const int p = pixel_of(pg, lane), y = p >> 4, x = p & 15;
        const f4* in0 = in_map + y * G16::P + x;                                 // tap (0, 0) of this lane's pixel
        f4 acc[4] = {ld4(bias + 4 * g), splat4(0.f), splat4(0.f), splat4(0.f)};
        f4 bv[2][NQ];
        auto fetch = [&](int tap, f4 (&bd)[NQ]) {
            const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int icq = 0; icq < NQ; ++icq) bd[icq] = in0[icq * G16::NPIX + dy * G16::P + dx];
            __builtin_amdgcn_sched_barrier(0);
        };
        fetch(0, bv[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) fetch(tap + 1, bv[(tap + 1) & 1]);
#pragma unroll
            for (int icq = 0; icq < NQ; ++icq)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = (tap * NQ + icq) * 4 + c;
                    acc[c] = mfma4x4_bcast(wr[k >> 4], bv[tap & 1][icq][c], acc[c], k & 15);
                }
        }
        f4 v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        return v;
    }
    __device__ __forceinline__ void run(const f4* in_map, f4* out_map, const float* __restrict__ bias, int wave, int lane) {
        if (wave >= JOBS) return;
        const int p = pixel_of(wave / NG, lane);
        out_map[(wave % NG) * G16::NPIX + G16::interior(p >> 4, p & 15)] = compute(in_map, bias, wave, lane);
    }
};

// grid B, NW * 64 threads.  Arguments as vth::head_seq_kernel + hw3 (the piece images; only conv1's are read).
// DIAG: s_memtime stamps per wave and phase (VT_DBG_STAMPS, tools/head_stamps.py); compiled out of the production instantiation.
template <int NW, int MAXP, bool DIAG = false>
__global__ __launch_bounds__(NW * 64) void head_seq3_kernel(const float* __restrict__ feat, const float* __restrict__ hw,
                                                        const u32x4* __restrict__ hw3, const float* __restrict__ window,
                                                        float* __restrict__ score, float* __restrict__ size,
                                                        float* __restrict__ offset, float* __restrict__ pred,
                                                        float* __restrict__ hann, float* __restrict__ conf, TrackTail tail,
                                                        int has_tail, unsigned long long* __restrict__ stamps) {
    constexpr int F = 16, n = F * F;
    static_assert(NW * 64 >= n, "one thread per pixel in the 1 x 1 stage");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // region R (the first 63 KiB; offsets inside the DS instructions' 16-bit field): conv1's output as PIECES, 3 x 8 quads of 8-byte
    // entries -- and, once conv2 has read them, the fp32 maps of conv2 (4 quads) and conv3 (2) in the same bytes
    u32x2* m1p = reinterpret_cast<u32x2*>(sm);
    f4* m2 = reinterpret_cast<f4*>(sm);
    f4* m3 = m2 + 4 * G16::NPIX;
    constexpr int R_F4 = (W1 / 4 + 4) * G16::NPIX;
    static_assert(R_F4 * 16 == 3 * (W1 / 4) * G16::NPIX * 8 && 6 * G16::NPIX <= R_F4, "conv1's pieces fill the region the fp32 maps shared");
    float* sc = reinterpret_cast<float*>(m2 + R_F4);             // the score plane, for the decode's argmax
    u32x2* in_map = reinterpret_cast<u32x2*>(sc + 256);          // 3 pieces x 12 quads
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    SeqConvB<C, W1, NW, MAXP> c1;
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (DIAG) {
            if (stamps != nullptr && lane == 0) stamps[((size_t)b * NW + wave) * 64 + nstamp] = __builtin_amdgcn_s_memtime();
            ++nstamp;
        }
    };
    stamp();
    SeqConvB<W1, 16, NW, VT_SEQ3_C2_MAXP, VT_SEQ3_C2_ACT> c2;
    SeqConvQ<16, 8, NW> c3;
    SeqConvQ<8, 4, NW> c4;
    c1.prefetch(hw3 + O3_W1, wave, lane);      // first weight pass flies during the map set-up
    // (B,HW,C) tokens -> piece planes, once for the three towers (vit_dist.py:126-129): requested before the LDS is cleared so that
    // their L2 / HBM round trip runs under the clear and its barrier (stamps: 7.4 k cycles when requested after it, item by item)
    constexpr int NST = n * (C / 4) / (NW * 64);
    static_assert(n * (C / 4) % (NW * 64) == 0, "staged items divide over the threads");
    f4 tok[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        // item i = ((pixel group pg, quad group r), lane = (c4, px)): pixel 16 pg + px, channel quad 4 r + c4.  A wave-load reads 16 rows x
        // 64 B (four lanes per row), and a 16-lane pass of the LDS writes below is 16 consecutive entries of ONE plane: conflict-free.
        // (Lanes along the pixels of one quad read 64 rows at a 192-byte stride; lanes along a row's quads write 12 planes, 6-way conflicts.)
        const int i = threadIdx.x + k * NW * 64, g = i >> 6, pix = 16 * (g / 3) + (i & 15), icq = 4 * (g % 3) + ((i >> 4) & 3);
        tok[k] = ld4(feat + ((size_t)b * n + pix) * C + 4 * icq);
    }
#ifdef VT_SEQ3_START_STAMPS
    stamp();
#endif
    // zero borders: only what is READ without having been written -- the border entries of the input's 36 piece planes and of piece 2 of
    // conv1's output (pieces 0 / 1 and the fp32 maps get theirs per tower, below).  Clearing all 160 KB cost 6.9 k cycles per frame.
    constexpr int PS1_ = (W1 / 4) * G16::NPIX;
    // thread -> (plane of the pass, border entry): HPP planes per pass; the entry's pixel is computed once per call, a pass costs an add
    constexpr int HPP = NW * 64 / SEQ_HALO;
    auto zero_borders = [&](auto* planes, int nplanes, auto zero) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int pl = tid / SEQ_HALO, hp = seq_halo_pix(tid - pl * SEQ_HALO);
        if (pl < HPP)
            for (int p0 = 0; p0 < nplanes; p0 += HPP)
                if (p0 + pl < nplanes) planes[(p0 + pl) * G16::NPIX + hp] = zero;
    };
    zero_borders(in_map, 3 * (C / 4), u32x2{0u, 0u});
    zero_borders(m1p + 2 * PS1_, W1 / 4, u32x2{0u, 0u});
    stamp();
#ifdef VT_SEQ3_START_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
#endif
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = threadIdx.x + k * NW * 64, g = i >> 6, pix = 16 * (g / 3) + (i & 15), icq = 4 * (g % 3) + ((i >> 4) & 3);
        u32x2 h, m, l;
        split3(tok[k], h, m, l);
        const int e = icq * G16::NPIX + G16::interior(pix / F, pix % F);
        in_map[e] = h; in_map[(C / 4) * G16::NPIX + e] = m; in_map[2 * (C / 4) * G16::NPIX + e] = l;
    }
#ifdef VT_SEQ3_START_STAMPS
    stamp();
#endif
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {              // tower 0 = ctr, 1 = offset, 2 = size
        const float* __restrict__ tw = hw + (size_t)t * vth::TOWER_STRIDE;
        const u32x4* __restrict__ tw3 = hw3 + (size_t)t * TOWER3_STRIDE;
        stamp();
        constexpr int PS1 = (W1 / 4) * G16::NPIX;                 // entries per piece of conv1's output
        {
            f4 r1[decltype(c1)::NPT];
            c1.compute(in_map, tw3 + O3_W1, tw + vth::O_B1, wave, lane, (VT_H3_SKIP & 1) != 0, r1);
            c2.prefetch(tw3 + O3_W2, wave, lane);    // (not before conv1: its working set is 200 registers)
            __syncthreads();                         // the previous tower's conv4 has read m3, which lies inside R
#pragma unroll
            for (int i = 0; i < decltype(c1)::NPT; ++i) {
                u32x2 h, m, l;
                split3(r1[i], h, m, l);
                const int e = decltype(c1)::out_entry(wave, lane, i);
                m1p[e] = h; m1p[PS1 + e] = m; m1p[2 * PS1 + e] = l;
            }
            // the borders of pieces 0 and 1 held interior values of the previous tower's fp32 maps (piece 2's bytes are never reused)
            zero_borders(m1p, 2 * (W1 / 4), u32x2{0u, 0u});
        }
        stamp();
        __syncthreads();
        stamp();
        {
            f4 r2[decltype(c2)::NPT];
            c2.compute(m1p, tw3 + O3_W2, tw + vth::O_B2, wave, lane, (VT_H3_SKIP & 2) != 0, r2);
            c3.prefetch(tw + vth::O_W3Q, wave, lane);
            __syncthreads();                         // every wave has read conv1's pieces: conv2's fp32 map may overwrite them
#pragma unroll
            for (int i = 0; i < decltype(c2)::NPT; ++i)
                if (decltype(c2)::active(wave)) m2[decltype(c2)::out_entry(wave, lane, i)] = r2[i];
            zero_borders(m2, 6, splat4(0.f));        // borders of m2 (4 planes) and of m3 (2 planes, contiguous with them)
        }
        c4.prefetch(tw + vth::O_W4Q, wave, lane);
        stamp();
        __syncthreads();
        stamp();
        if (!(VT_H3_SKIP & 8)) c3.run(m2, m3, tw + vth::O_B3, wave, lane);
        stamp();
        __syncthreads();
        stamp();
        // conv4 and the 1 x 1 conv + activation (head.py:187,194,200-201) in one go: a lane of waves 0-3 ends conv4 with its pixel's four
        // channels in registers, which is all the 1 x 1 stage reads -> global maps (+ the score plane in LDS); no map, no barrier
        if (wave < decltype(c4)::JOBS) {
            const f4 v = (VT_H3_SKIP & 8) ? splat4(0.f) : c4.compute(m3, tw + vth::O_B4, wave, lane);
            int pix = decltype(c4)::pixel_of(wave, lane);
            asm volatile("" : "+v"(pix));      // addresses are rebuilt here per tower: hoisted out of the tower loop they were spilled
            const int nout = (t == 0) ? 1 : 2;
            for (int o = 0; o < nout; ++o) {
                const f4 w5 = ld4(tw + vth::O_W5 + 4 * o);
                float y = tw[vth::O_B5 + o];
                y = fmaf(v.x, w5.x, y); y = fmaf(v.y, w5.y, y); y = fmaf(v.z, w5.z, y); y = fmaf(v.w, w5.w, y);
                if (t == 0) { y = sigmoid_clamped(y); score[(size_t)b * n + pix] = y; sc[pix] = y; }
                else if (t == 2) size[((size_t)b * 2 + o) * n + pix] = sigmoid_clamped(y);
                else offset[((size_t)b * 2 + o) * n + pix] = y;
            }
        }
        if (t < 2) c1.prefetch(tw3 + TOWER3_STRIDE + O3_W1, wave, lane);   // the next tower's first pass
        stamp();
        stamp();
        // the next tower's conv1 writes R only behind the barrier in front of its epilogue: conv4's reads of m3 need none of their own
    }
    // size / offset of this frame were written by this workgroup's own threads: the barrier (workgroup-scope release / acquire)
    // makes them visible to the decoding wave
    stamp();
    __syncthreads();
    stamp();
    if (wave == 0) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));          // (the decode's lane and addresses are built here, not held across the tower loop)
        const int dl = tx & 63;
        vth::seq_decode<F>(sc, size + (size_t)b * 2 * n, offset + (size_t)b * 2 * n, window, b, dl, pred, hann, conf, tail, has_tail);
    }
}

}  // namespace vth3
#endif  // !VT_F16
