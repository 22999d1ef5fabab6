// vt_stem.h -- LeViT-style patch embedding: 4 x (conv3x3 stride 2 pad 1 + folded BN), Hardswish
// after the first three, tokenise, add pos-embed, write [z ; x] rows of the token matrix.
//
// Replaces Conv2d_BN / b16 / LevitPatchEmbedding.forward and the pos-embed add + cat of
// OstrackDist.forward (lib/models/vit_dist/vit_dist.py:10-54,78-84).
//
// The two-kernel form (both crops in each launch).  The default paths are stem_fused_kernel (G128: all four
// layers of a frame in one workgroup) and stem_pipe_kernel + stem_b (G256), both in vt_stem_fused.h; stem_a /
// stem_a2 stay selectable (VT_STEM_FUSED / VT_STEM_PIPE = 0) and parity-tested.
//
//  stem_a  layers 1+2, the HBM-streaming half.  One workgroup per (frame, crop, band of R2 layer-2
//          rows).  Layer 1 (3 -> 6, VALU: an MFMA tile would be 64 % padding) reads the NCHW crop straight
//          from HBM -- each wave-instruction covers whole 512/1024-byte image rows with float4 loads, two
//          output pixels per thread, folded weights as wave-uniform SGPR operands -- and leaves its 2*R2+1
//          output rows in LDS as a column-parity-split quad-planar map.  Layer 2 (6 -> 12, channels padded
//          to 8) is an implicit GEMM on MFMA over that map and writes the result as three channel-quad
//          planes (B, 3, S2, S2, 4).  stem_a2: the same with band k of both crops in one workgroup.
//
//  stem_b  layers 3+4 on v_mfma_f32_16x16x4_f32 as implicit GEMMs (weights = A operand, 16 output
//          pixels = B operand columns), one workgroup per (frame, crop, band of R4 token rows).
//          Inputs live in LDS as parity-split QUAD-PLANAR maps: float4 map[icq][row][even|odd col],
//          so the stride-2 taps of 16 consecutive output pixels are 16 consecutive float4 and the B
//          operand of a k-chunk is one ds_read_b128.  Layer 3's result tile is written straight
//          into layer 4's input map; layer 4 adds the pos-embed and stores token rows.
#pragma once
#include <type_traits>

#include "vt_common.h"
#include "vt_conv.h"

namespace vts {

// Makes a wave-uniform pointer opaque at this program point so loads through it cannot be hoisted
// above it (hipcc otherwise hoists every loop-invariant scalar weight load out of the pixel loops,
// runs out of SGPRs and spills them to VGPR lanes: one v_readlane per weight use).
// The result points into the constant address space, so the loads stay scalar (s_load).
typedef __attribute__((address_space(4))) const float* sptr;
__device__ __forceinline__ sptr opaque(const float* p) {
    unsigned long long v = reinterpret_cast<unsigned long long>(p);
    asm volatile("" : "+s"(v));
    return (sptr)v;
}

constexpr int round16(int v) { return (v + 15) & ~15; }
// Stride-2 3x3 conv over a parity-split quad-planar LDS map (core: vt_conv.h).
//   in_map[icq * npix_in + row * pitch_in + (col odd ? half_in + 1 + col/2 : col/2)], local row 0
//   = the row above the band's first needed row (zero row at the image top), entry half_in = col -1.
// Per-lane map offset of this lane's quad of chunk c for a layer with NQ channel-quads per tap.
template <int NQ>
__device__ __forceinline__ int s2_chunk_off(int c, int q, int npix_in, int pitch_in, int half_in) {
    int tap, icq;
    vtc::decode_quad<NQ>(4 * c + q, tap, icq);
    const int dy = tap / 3, dx = tap - 3 * dy;
    return icq * npix_in + dy * pitch_in + (dx == 1 ? 0 : (dx == 0 ? half_in : half_in + 1));
}

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

struct CropA {            // one crop for stem_a
    const float* in;      // (B, 3, T, T) NCHW
    float* out;           // (B, 3, T/4, T/4, 4): layer-2 map as three channel-quad planes
    int T;                // crop side
    int r2;               // layer-2 rows per band
    int bands;            // (T/4) / r2
};

// LDS floats stem_a needs for a crop of side T with r2 rows per band
// (two quad planes: layer-1 channels 0-3 and 4-5 + zero padding; plane = rows x (even | -1 | odd) columns)
__host__ __device__ constexpr int stem_a_npix1(int T, int r2) { return round16((2 * r2 + 1) * (T / 2 + 1)); }
__host__ __device__ constexpr int stem_a_lds_floats(int T, int r2) { return 2 * stem_a_npix1(T, r2) * 4; }

// Scalar weight sections.  Weights are packed [r][c][s][cout] so the 3*COUT weights a thread needs
// for one (kernel row r, input channel c) are contiguous; a section is fetched with a few wide
// s_loads into SGPRs, and the NEXT section is requested before the current one is consumed.
template <int N>
__device__ __forceinline__ void load_section(float (&w)[N], const float* base, int sec) {
    const sptr p = opaque(base + sec * N);
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = p[i];
}

#ifndef VT_STEM_A_WAVES_PER_SIMD
#define VT_STEM_A_WAVES_PER_SIMD 4
#endif

// value of the previous lane (lane 0: 0) in VALU latency: v_mov_b32_dpp wave_shr:1
// (v_mov_b32_dpp with no `old` operand: lanes without a source -- lane 0 -- read 0; with update_dpp's `old` hipcc initialises the
// destination with a v_mov in front of every call)
__device__ __forceinline__ float lane_left(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));
}

// ---- layer 1's two input forms ------------------------------------------------------------------------------------------------
// A layer-1 thread consumes input rows 2 p1 - 1 .. 2 p1 + 1 x columns 4 qp .. 4 qp + 3 x 3 channels.
//   fp32 form (U8 = false): the normalised NCHW crop Preprocessor.process produces (lib/test/tracker/data_utils.py:11-17) -- nine
//     float4, one per (row, channel).
//   uint8 form (U8 = true; round 6): the (T, T, 3) uint8 patch exactly as sample_target returns it
//     (lib/train/data/processing_utils.py:68-79; vt_crop_u8 writes it).  A row's four pixels are 12 consecutive bytes = ONE
//     buffer_load_dwordx3 per row: 3 loads and 9 registers per thread instead of 9 and 36, a quarter of the bytes.
//     Preprocessor.process is affine per channel, x = a_c u + b_c with a_c = 1 / (255 std_c), b_c = -mean_c / std_c, and layer 1
//     is linear in x, so it is folded into the layer's weights (fp64, vt_load_weights / vt_set_normalization):
//         w1u = [r][c][s][6] sections of w a_c  |  6 biases b1 + sum_taps w b_c  |  3 pad values 255 mean_c
//     The conv's zero padding is zero in the NORMALISED domain, i.e. the byte value u where a_c u + b_c = 0: taps left of / above
//     the crop substitute 255 mean_c.  (The crop's own zero padding -- the part of the window outside the frame -- is ordinary
//     uint8 zeros inside the patch, as in the reference.)  Against the three separately rounded fp32 operations of the reference
//     this changes a layer-1 sum by a few 1e-7 (tests: maps within 1e-5 of vt_crop + vt_forward and of the reference vectors).
typedef unsigned u3v __attribute__((ext_vector_type(3)));
template <bool U8> struct L1In { f4 v[3][3]; };
template <> struct L1In<true> { u3v v[3]; };
constexpr int W1U_BIAS = 162, W1U_PAD = 168, W1U_FLOATS = 176;      // offsets inside w1u (floats)
// channel c of the row's four pixels: byte 3 k + c of the 12-byte group; each conversion is one v_cvt_f32_ubyteN
__device__ __forceinline__ f4 l1_channel(const u3v& d, int c) {
    auto ub = [&](int i) { return (float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu); };
    return f4{ub(c), ub(3 + c), ub(6 + c), ub(9 + c)};
}

// w1g: [r][c][s][6] scalar sections (layer 1, VALU); w2img: [1][5][64][4] MFMA image of layer 2 with
// the input channels padded 6 -> 8; b2: 16 (12 used).
// U8: the crops are uint8 (T, T, 3) patches, w1g / b1 point into the folded image w1u (see L1In above).
template <bool U8 = false>
__global__ __launch_bounds__(256, VT_STEM_A_WAVES_PER_SIMD) void stem_a_kernel(
    CropA cx, CropA cz, const float* __restrict__ w1g, const float* __restrict__ b1,
    const float* __restrict__ w2img, const float* __restrict__ b2, int skip) {   // skip: phase-timing diagnostic, 0 in production
    extern __shared__ __attribute__((aligned(16))) float lds_a[];
    f4* map1 = reinterpret_cast<f4*>(lds_a);     // [2 quads][NR1 rows][PITCH]: layer-1 output
    const int per = cx.bands + cz.bands;
    const int b = blockIdx.x / per;
    int k = blockIdx.x - b * per;
    const bool is_z = k >= cx.bands;
    if (is_z) k -= cx.bands;
    const float* __restrict__ in = is_z ? cz.in : cx.in;
    float* __restrict__ out = is_z ? cz.out : cx.out;
    const int T = is_z ? cz.T : cx.T;
    const int R2 = is_z ? cz.r2 : cx.r2;
    const int W1 = T >> 1, HALF = T >> 2, PITCH = W1 + 1, NR1 = 2 * R2 + 1, W2 = T >> 2;
    const int p0 = k * R2;                       // first layer-2 row of this band
    const int npix1 = stem_a_npix1(T, R2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- layer 1 (3 -> 6, VALU): rows 2*p0-1 .. 2*p0+2*R2-1 of the layer-1 map, two pixels per
    // thread, software-pipelined over passes of 256 pixel pairs: the nine 16-byte row fetches of the
    // next pass are in flight while the current pass is computed.
    const int npairs = NR1 * HALF;
    auto fetch = [&](int i, L1In<U8>& v) {
        i = i < npairs ? i : npairs - 1;            // clamped lanes recompute the last pair
        const int lr = i / HALF, qp = i - lr * HALF;
        const int p1 = 2 * p0 - 1 + lr;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int iy = 2 * p1 + r - 1;                // < 0 only for r = 0 at the image top (and for
            const float keep = iy >= 0 ? 1.f : 0.f; // the padding row p1 = -1): branch-free zeroing
            iy = iy >= 0 ? iy : 0;
            if constexpr (U8) {                     // the row's 12 bytes; a padding row is replaced in compute()
                typedef unsigned u3a __attribute__((ext_vector_type(3), aligned(4)));
                const u3a t = *reinterpret_cast<const u3a*>(reinterpret_cast<const unsigned char*>(in) + (((size_t)b * T + iy) * T + 4 * qp) * 3);
                v.v[r] = u3v{t.x, t.y, t.z};
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v.v[r][c] = ld4(in + (((size_t)b * 3 + c) * T + iy) * T + 4 * qp) * splat4(keep);
            }
        }
    };
    auto compute = [&](int i, const L1In<U8>& v) {
        const bool in_range = i < npairs;           // clamped lanes compute (keeps shuffles whole) but do not store
        i = in_range ? i : npairs - 1;
        const int lr = i / HALF, qp = i - lr * HALF;
        const int p1 = 2 * p0 - 1 + lr;
        float a0[6], a1[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) a0[j] = a1[j] = b1[j];
        float wa[18], wb[18];
        load_section(wa, w1g, 0);
#pragma unroll
        for (int sec = 0; sec < 9; ++sec) {
            float (&cur)[18] = (sec & 1) ? wb : wa;
            float (&nxt)[18] = (sec & 1) ? wa : wb;
            if (sec + 1 < 9) load_section(nxt, w1g, sec + 1);
            const int r = sec / 3, c = sec % 3;
            f4 vv;
            float padv = 0.f;                       // what a tap outside the crop reads (fp32 form: the zero padding itself)
            if constexpr (U8) {
                padv = b1[W1U_PAD - W1U_BIAS + c];
                vv = l1_channel(v.v[r], c);
                if (2 * p1 + r - 1 < 0) vv = splat4(padv);
            } else {
                vv = v.v[r][c];
            }
            // column 4*qp-1 is the previous lane's .w (same image row); the padding left of the image
            const float left = __shfl_up(vv.w, 1, 64);
            const float t0[3] = {qp > 0 ? left : padv, vv.x, vv.y}, t1[3] = {vv.y, vv.z, vv.w};
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    a0[j] = fmaf(t0[s], cur[s * 6 + j], a0[j]);
                    a1[j] = fmaf(t1[s], cur[s * 6 + j], a1[j]);
                }
        }
        const float live = p1 < 0 ? 0.f : 1.f;      // row -1 of the layer-1 map = layer 2's zero padding
#pragma unroll
        for (int j = 0; j < 6; ++j) { a0[j] = hardswish(a0[j]) * live; a1[j] = hardswish(a1[j]) * live; }
        if (in_range) {
            f4* dst = map1 + lr * PITCH;
            dst[qp] = f4{a0[0], a0[1], a0[2], a0[3]};                   // even column 2*qp, channels 0-3
            dst[npix1 + qp] = f4{a0[4], a0[5], 0.f, 0.f};               //                   channels 4-5 (+ padding)
            dst[HALF + 1 + qp] = f4{a1[0], a1[1], a1[2], a1[3]};        // odd column 2*qp+1
            dst[npix1 + HALF + 1 + qp] = f4{a1[4], a1[5], 0.f, 0.f};
        }
    };
    if (!(skip & 1)) {
        L1In<U8> va, vb;
        fetch(threadIdx.x, va);
        for (int base = 0; base + (int)(threadIdx.x & ~63) < npairs; base += 256) {   // whole waves drop out
            const int i = base + threadIdx.x;
            if (base + 256 + (int)(threadIdx.x & ~63) < npairs) fetch(i + 256, vb);   // next pass in flight
            compute(i, va);
            va = vb;
        }
    }
    f4 w2a[5][1];                                // layer-2 weights: in flight across the barrier below
    vtc::load_weights_f4<1, 5, 5>(w2img, 0, 5, lane, w2a);
    for (int i = threadIdx.x; i < 2 * NR1; i += 256)                     // column -1 of every row
        map1[(i / NR1) * npix1 + (i % NR1) * PITCH + HALF] = splat4(0.f);
    __syncthreads();

    // ---- layer 2 (6 -> 12, MFMA implicit GEMM) -> quad planes in global memory --------------------
    const int w2_log2 = ilog2(W2);
    const int q = lane >> 4, px = lane & 15;
    auto store2 = [&](int t, int ot, f4 v) {
        (void)ot;
        if (q < 3) {
            const int op = 16 * t + px;
            const int y = op >> w2_log2, x = op - (y << w2_log2);
            v.x = hardswish(v.x); v.y = hardswish(v.y); v.z = hardswish(v.z); v.w = hardswish(v.w);
            st4(out + ((((size_t)b * 3 + q) * W2 + p0 + y) * W2 + x) * 4, v);     // quad plane q: 16 lanes = 256 contiguous bytes
        }
    };
    if (!(skip & 2)) {
        const int ntiles = (R2 * W2) >> 4;
        auto off2 = [&](int c) { return s2_chunk_off<2>(c, q, npix1, PITCH, HALF); };
        for (int t0 = wave; t0 < ntiles; t0 += 16) {      // ntiles is a multiple of 16 (host check)
            int base[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int op = 16 * (t0 + 4 * i) + px;
                const int y = op >> w2_log2, x = op - (y << w2_log2);
                base[i] = 2 * y * PITCH + x;
            }
            f4 acc[4][1];
            const f4 bv = ld4(b2 + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][0] = bv;
            vtc::mma_pass<1, 4, 5, 5>(map1, base, w2a, 0, off2, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) store2(t0 + 4 * i, 0, acc[i][0]);
        }
    }
}

// ---------------------------------------------------------------------------------- stem_a, fused crops
// Same arithmetic as stem_a_kernel, but one workgroup handles band k of the search crop AND band k
// of the template crop (the crops must have the same number of bands).  At G128 that turns 5
// workgroups per frame into 4: 1024 for a batch of 256, i.e. exactly one resident round on 256 CUs
// x 4 workgroups (the 5-per-frame form needs a second round that is only a quarter full).
// Layer-1 work is dealt to the waves in wave-uniform units of 64 pixel pairs (x units first, then z),
// so every per-crop quantity is a scalar; all sizes are powers of two, so index arithmetic is shifts,
// and a load address is a scalar plane pointer plus one 32-bit per-thread offset.
struct JobA {
    const float* in;      // this frame's crop, (3, T, T)
    float* out;           // this frame's layer-2 map, (3, T/4, T/4, 4)
    int map_off;          // f4 offset of this crop's layer-1 map in LDS
    int T, lgT, HALF, lgHALF, PITCH, NR1, p0, npairs, npix1, R2, lgW2;
};
__device__ __forceinline__ JobA make_job_a(const CropA& c, int b, int k, int map_off) {
    JobA j;
    j.T = c.T; j.lgT = ilog2(c.T); j.HALF = c.T >> 2; j.lgHALF = j.lgT - 2; j.PITCH = (c.T >> 1) + 1;
    j.R2 = c.r2; j.NR1 = 2 * c.r2 + 1; j.p0 = k * c.r2; j.npairs = j.NR1 * j.HALF; j.npix1 = stem_a_npix1(c.T, c.r2);
    j.lgW2 = j.lgT - 2; j.map_off = map_off;
    j.in = c.in + (size_t)b * 3 * c.T * c.T;
    j.out = c.out + (size_t)b * (c.T >> 2) * (c.T >> 2) * 12;
    return j;
}
__global__ __launch_bounds__(256, VT_STEM_A_WAVES_PER_SIMD) void stem_a2_kernel(
    CropA cx, CropA cz, const float* __restrict__ w1g, const float* __restrict__ b1,
    const float* __restrict__ w2img, const float* __restrict__ b2, int skip) {
    extern __shared__ __attribute__((aligned(16))) float lds_a[];
    f4* maps = reinterpret_cast<f4*>(lds_a);
    const int bands = cx.bands;                   // == cz.bands (host check)
    const int b = blockIdx.x / bands, k = blockIdx.x - b * bands;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const JobA jx = make_job_a(cx, b, k, 0);
    const JobA jz = make_job_a(cz, b, k, 2 * jx.npix1);
    const int ux = (jx.npairs + 63) >> 6, nu = ux + ((jz.npairs + 63) >> 6);

    // ---- layer 1 (3 -> 6, VALU) ---------------------------------------------------------------------
    auto fetch = [&](const JobA& J, int i, f4 (&v)[3][3]) {
        i = i < J.npairs ? i : J.npairs - 1;        // clamped lanes recompute the last pair
        const int lr = i >> J.lgHALF, qp = i & (J.HALF - 1);
        const int p1 = 2 * J.p0 - 1 + lr;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int iy = 2 * p1 + r - 1;                // < 0 only at the image top: zeroed, branch-free
            const float keep = iy >= 0 ? 1.f : 0.f;
            iy = iy >= 0 ? iy : 0;
            const unsigned off = ((unsigned)iy << J.lgT) + 4u * (unsigned)qp;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[r][c] = ld4(J.in + ((size_t)c << (2 * J.lgT)) + off) * splat4(keep);
        }
    };
    auto compute = [&](const JobA& J, int i, const f4 (&v)[3][3]) {
        const bool in_range = i < J.npairs;
        i = in_range ? i : J.npairs - 1;
        const int lr = i >> J.lgHALF, qp = i & (J.HALF - 1);
        const int p1 = 2 * J.p0 - 1 + lr;
        float a0[6], a1[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) a0[j] = a1[j] = b1[j];
        float wa[18], wb[18];
        load_section(wa, w1g, 0);
#pragma unroll
        for (int sec = 0; sec < 9; ++sec) {
            float (&cur)[18] = (sec & 1) ? wb : wa;
            float (&nxt)[18] = (sec & 1) ? wa : wb;
            if (sec + 1 < 9) load_section(nxt, w1g, sec + 1);
            const int r = sec / 3, c = sec % 3;
            // column 4*qp-1 is the previous lane's .w (a unit starts at a row start, so lane 0 has qp = 0)
            const float left = lane_left(v[r][c].w);
            const float t0[3] = {qp > 0 ? left : 0.f, v[r][c].x, v[r][c].y}, t1[3] = {v[r][c].y, v[r][c].z, v[r][c].w};
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    a0[j] = fmaf(t0[s], cur[s * 6 + j], a0[j]);
                    a1[j] = fmaf(t1[s], cur[s * 6 + j], a1[j]);
                }
        }
        const float live = p1 < 0 ? 0.f : 1.f;      // row -1 of the layer-1 map = layer 2's zero padding
#pragma unroll
        for (int j = 0; j < 6; ++j) { a0[j] = hardswish(a0[j]) * live; a1[j] = hardswish(a1[j]) * live; }
        if (in_range) {
            f4* dst = maps + J.map_off + lr * J.PITCH;
            dst[qp] = f4{a0[0], a0[1], a0[2], a0[3]};
            dst[J.npix1 + qp] = f4{a0[4], a0[5], 0.f, 0.f};
            dst[J.HALF + 1 + qp] = f4{a1[0], a1[1], a1[2], a1[3]};
            dst[J.npix1 + J.HALF + 1 + qp] = f4{a1[4], a1[5], 0.f, 0.f};
        }
    };
    if (!(skip & 1)) {
        f4 va[3][3], vb[3][3];
        if (wave < nu) fetch(wave < ux ? jx : jz, ((wave < ux ? wave : wave - ux) << 6) + lane, va);
        for (int u = wave; u < nu; u += 4) {
            const int un = u + 4;
            if (un < nu) fetch(un < ux ? jx : jz, ((un < ux ? un : un - ux) << 6) + lane, vb);   // next unit in flight
            compute(u < ux ? jx : jz, ((u < ux ? u : u - ux) << 6) + lane, va);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) va[r][c] = vb[r][c];
        }
    }
    f4 w2a[5][1];                                // layer-2 weights: in flight across the barrier below
    vtc::load_weights_f4<1, 5, 5>(w2img, 0, 5, lane, w2a);
    for (int i = threadIdx.x; i < 2 * (jx.NR1 + jz.NR1); i += 256) {     // column -1 of every row of both maps
        const bool zz = i >= 2 * jx.NR1;
        const int ii = zz ? i - 2 * jx.NR1 : i;
        const int nr = zz ? jz.NR1 : jx.NR1, npx = zz ? jz.npix1 : jx.npix1, pt = zz ? jz.PITCH : jx.PITCH;
        const int plane = ii >= nr ? 1 : 0, row = ii - plane * nr;
        maps[(zz ? jz.map_off : 0) + plane * npx + row * pt + (zz ? jz.HALF : jx.HALF)] = splat4(0.f);
    }
    __syncthreads();

    // ---- layer 2 (6 -> 12, MFMA implicit GEMM) -> quad planes in global memory --------------------
    if (!(skip & 2)) {
        const int q = lane >> 4, px = lane & 15;
        const f4 bv = ld4(b2 + 4 * q);
        auto run2 = [&](const JobA& J, auto npt_c, int t0, int tstride) {
            constexpr int NPT = decltype(npt_c)::value;
            int base[NPT];
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int op = 16 * (t0 + tstride * i) + px;
                const int y = op >> J.lgW2, x = op & ((1 << J.lgW2) - 1);
                base[i] = 2 * y * J.PITCH + x;
            }
            f4 acc[NPT][1];
#pragma unroll
            for (int i = 0; i < NPT; ++i) acc[i][0] = bv;
            auto off2 = [&](int c) { return s2_chunk_off<2>(c, q, J.npix1, J.PITCH, J.HALF); };
            if (!(skip & 8)) vtc::mma_pass<1, NPT, 5, 5>(maps + J.map_off, base, w2a, 0, off2, acc);
            if (q < 3 && !(skip & 4)) {
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const int op = 16 * (t0 + tstride * i) + px;
                    const int y = op >> J.lgW2, x = op & ((1 << J.lgW2) - 1);
                    f4 v = acc[i][0];
                    v.x = hardswish(v.x); v.y = hardswish(v.y); v.z = hardswish(v.z); v.w = hardswish(v.w);
                    st4(J.out + ((((size_t)q << (2 * J.lgW2)) + (((size_t)J.p0 + y) << J.lgW2) + x) << 2), v);   // quad plane q
                }
            }
        };
        const int ntx = (jx.R2 << jx.lgW2) >> 4;             // multiple of 16 (host check)
        for (int t0 = wave; t0 < ntx; t0 += 16) run2(jx, std::integral_constant<int, 4>{}, t0, 4);
        run2(jz, std::integral_constant<int, 1>{}, wave, 0);  // the template band: 4 tiles, one per wave (host check)
    }
}

// ------------------------------------------------------------------------------------------ stem_b
struct CropB {
    const float* in;      // (B, 3, S2, S2, 4) quad planes, S2 = T/4
    const float* pos;     // (S4*S4, 48)
    int S2;               // layer-2 map side
    int r4;               // token rows per band
    int bands;            // (S2/4) / r4
    int tok_off;          // first token row of this crop in the (B, L, 48) matrix
};

// plane sizes (pixels) of the two LDS maps for a band of r4 token rows of a crop with S2
__host__ __device__ constexpr int stem_b_npix2(int S2, int r4) { return round16((4 * r4 + 3) * (S2 + 1)); }
__host__ __device__ constexpr int stem_b_npix3(int S2, int r4) { return round16((2 * r4 + 1) * (S2 / 2 + 1)); }
__host__ __device__ constexpr int stem_b_lds_bytes(int S2, int r4) {
    return (3 * stem_b_npix2(S2, r4) + 6 * stem_b_npix3(S2, r4)) * 16;
}

// w3img: [2][7][64][4] (24 -> 32 padded output channels), b3: 32; w4img: [3][14][64][4], b4: 48.
// DIAG: the diagnostic build (VT_SKIP_STEM_B); the production instantiation compiles `skip` out.
template <bool DIAG>
__global__ __launch_bounds__(256) void stem_b_kernel(CropB cx, CropB cz, const float* __restrict__ w3img,
                                                     const float* __restrict__ b3, const float* __restrict__ w4img,
                                                     const float* __restrict__ b4, float* __restrict__ tokens, int L,
                                                     int skip_arg) {   // skip: phase-timing diagnostic, 0 in production
    const int skip = DIAG ? skip_arg : 0;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int per = cx.bands + cz.bands;
    const int b = blockIdx.x / per;
    int k = blockIdx.x - b * per;
    const bool is_z = k >= cx.bands;
    if (is_z) k -= cx.bands;
    const float* __restrict__ in = is_z ? cz.in : cx.in;
    const float* __restrict__ pos = is_z ? cz.pos : cx.pos;
    const int S2 = is_z ? cz.S2 : cx.S2;
    const int R4 = is_z ? cz.r4 : cx.r4;
    const int tok_off = is_z ? cz.tok_off : cx.tok_off;
    const int S3 = S2 >> 1, S4 = S2 >> 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, px = lane & 15;

    // band geometry: token rows [y4_0, y4_0+R4) <- layer-3 rows [2*y4_0-1, ..] (2R4+1 rows)
    //                                         <- layer-2 rows [4*y4_0-3, ..] (4R4+3 rows)
    const int y4_0 = k * R4;
    const int r3_0 = 2 * y4_0 - 1, NR3 = 2 * R4 + 1;
    const int r2_0 = 2 * r3_0 - 1, NR2 = 2 * NR3 + 1;
    const int pitch2 = S2 + 1, half2 = S2 >> 1, npix2 = stem_b_npix2(S2, R4);
    const int pitch3 = S3 + 1, half3 = S3 >> 1, npix3 = stem_b_npix3(S2, R4);
    f4* map2 = reinterpret_cast<f4*>(sm);          // 3 quads
    f4* map3 = map2 + 3 * npix2;                   // 6 quads

    // Work split that keeps the waves on DIFFERENT weights (each weight tile crosses the 64 B/clk
    // L1 fill path twice per workgroup instead of four times):
    //   layer 3: wave = (output tile ot3 = wave & 1, pixel tiles of parity wave >> 1), all 7 chunks;
    //   layer 4: wave = k-quarter of the 14 chunks for both pixel tiles and all 3 output tiles,
    //            partial sums reduced through LDS.
    // Both weight bursts are requested before the data they multiply exists.
    constexpr int NCH3 = 7, NCH4 = 14;
    const int ot3 = wave & 1, th3 = wave >> 1;
    opnd w3a[NCH3][1];
    vtc::load_weights<1, NCH3, NCH3>(w3img + (size_t)ot3 * NCH3 * 256, 0, NCH3, lane, w3a);

    if (!(skip & 1))
        for (int i = threadIdx.x; i < 3 * npix2 + 6 * npix3; i += 256) map2[i] = splat4(0.f);
    __syncthreads();
    // layer-2 activations of the band -> map2 (rows outside the image stay zero).  A thread keeps one column and walks rows
    // (S2 is a power of two: 256 threads = 256 / S2 whole rows per step), so the index arithmetic is shifts and adds -- as a flat
    // loop over (plane, row, column) with run-time divisors it was ~40 VALU instructions per element, more issue time than the
    // band's MFMAs -- and a plane's loads are all in flight before the first store.
    if (!(skip & 2)) {
        const int lg2 = ilog2(S2);
        const int col = threadIdx.x & (S2 - 1), rsub = threadIdx.x >> lg2, rstep = 256 >> lg2;
        const int cpos = (col & 1) ? half2 + 1 + (col >> 1) : (col >> 1);
        constexpr int MAXR = 5;                                   // rows per thread and plane: ceil(19 / 4), ceil(19 / 8)
#pragma unroll 1                                                  // one plane's 5 loads in flight: 3 x 5 cost a workgroup per CU in VGPRs
        for (int icq = 0; icq < 3; ++icq) {
            const float* src = in + (((size_t)b * 3 + icq) << (2 * lg2)) * 4 + col * 4;
            f4 v[MAXR];
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                const int lr = rsub + j * rstep, r2 = r2_0 + lr;
                if (lr < NR2 && r2 >= 0 && r2 < S2) v[j] = ld4(src + ((size_t)r2 << lg2) * 4);
            }
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                const int lr = rsub + j * rstep, r2 = r2_0 + lr;
                if (lr < NR2 && r2 >= 0 && r2 < S2) map2[icq * npix2 + lr * pitch2 + cpos] = v[j];
            }
        }
    }
    const int c4_0 = (NCH4 * wave) >> 2, c4_n = ((NCH4 * (wave + 1)) >> 2) - c4_0;     // 3, 4, 3, 4 chunks
    opnd w4a[4][3];
    vtc::load_weights<3, 4, NCH4>(w4img, c4_0, c4_n, lane, w4a);
    __syncthreads();

    // ---- layer 3 (12 -> 24, Hardswish): band rows r3_0 .. r3_0+NR3-1; rows outside the image stay 0
    if (!(skip & 4)) {
        const int lr_first = r3_0 < 0 ? 1 : 0;                 // local layer-3 row 0 is the padding row at the top
        const int nrows = NR3 - lr_first;
        const int w3_log2 = ilog2(S3);
        const int ntiles = (nrows * S3) >> 4;                  // <= 10 (checked on the host)
        const f4* in3 = map2 + 2 * lr_first * pitch2;          // first computed row reads layer-2 local rows 2*lr_first..
        auto off3 = [&](int c) { return s2_chunk_off<3>(c, q, npix2, pitch2, half2); };
        // tiles th3, th3+2, ... of this wave, in passes of exactly NPT tiles (compile-time: no branch
        // between MFMAs); counts that occur: 2 (z crop), 4, 5 = 3 + 2
        auto run3 = [&](auto npt_c, int tb) {
            constexpr int NPT = decltype(npt_c)::value;
            int base[NPT];
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int op = 16 * (tb + 2 * i) + px;
                const int y = op >> w3_log2, x = op - (y << w3_log2);
                base[i] = 2 * y * pitch2 + x;
            }
            f4 acc[NPT][1];
            const f4 bv = ld4(b3 + 16 * ot3 + 4 * q);
#pragma unroll
            for (int i = 0; i < NPT; ++i) acc[i][0] = bv;
            vtc::mma_pass<1, NPT, NCH3, NCH3>(in3, base, w3a, 0, off3, acc);
            if (16 * ot3 + 4 * q < 24) {
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const int op = 16 * (tb + 2 * i) + px;
                    const int y = op >> w3_log2, x = op - (y << w3_log2);
                    f4 v = acc[i][0];
                    v.x = hardswish(v.x); v.y = hardswish(v.y); v.z = hardswish(v.z); v.w = hardswish(v.w);
                    map3[(4 * ot3 + q) * npix3 + (y + lr_first) * pitch3 + ((x & 1) ? half3 + 1 + (x >> 1) : (x >> 1))] = v;
                }
            }
        };
        int left = (ntiles - th3 + 1) >> 1, tb = th3;
        while (left >= 4) { run3(std::integral_constant<int, 4>{}, tb); tb += 8; left -= 4; }
        if (left == 3) run3(std::integral_constant<int, 3>{}, tb);
        else if (left == 2) run3(std::integral_constant<int, 2>{}, tb);
        else if (left == 1) run3(std::integral_constant<int, 1>{}, tb);
    }
    __syncthreads();
    // ---- layer 4 (24 -> 48) + pos-embed -> token rows -------------------------------------------
    if (!(skip & 8)) {
        const int w4_log2 = ilog2(S4);
        const int npx = R4 * S4;                            // 16 or 32 output pixels (checked on the host)
        const int ntiles = (npx + 15) >> 4;
        auto off4 = [&](int c) { return s2_chunk_off<6>(c, q, npix3, pitch3, half3); };
        f4 acc[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ot = 0; ot < 3; ++ot) acc[i][ot] = splat4(0.f);
        auto run4 = [&](auto npt_c, auto n_c) {
            constexpr int NPT = decltype(npt_c)::value, N = decltype(n_c)::value;
            int base[NPT];
            f4 a4[NPT][3];
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int op = 16 * i + px;
                const int y = op >> w4_log2, x = op - (y << w4_log2);
                base[i] = 2 * y * pitch3 + x;
#pragma unroll
                for (int ot = 0; ot < 3; ++ot) a4[i][ot] = splat4(0.f);
            }
            vtc::mma_pass<3, NPT, 4, N>(map3, base, w4a, c4_0, off4, a4);
#pragma unroll
            for (int i = 0; i < NPT; ++i)
#pragma unroll
                for (int ot = 0; ot < 3; ++ot) acc[i][ot] = a4[i][ot];
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        if (ntiles == 2) { if (c4_n == 4) run4(I2{}, I4{}); else run4(I2{}, I3{}); }
        else             { if (c4_n == 4) run4(I1{}, I4{}); else run4(I1{}, I3{}); }
        // partial sums -> [wave][tile < ntiles][ot][lane] in map2's space (dead since layer 3's
        // barrier; 4 * ntiles * 3 KiB <= 3 * npix2 * 16 B for every supported band, host-checked)
        f4* part = map2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (i < ntiles)
#pragma unroll
                for (int ot = 0; ot < 3; ++ot) part[((wave * ntiles + i) * 3 + ot) * 64 + lane] = acc[i][ot];
        __syncthreads();
        // six (tile, ot) results; wave w finalises items w and w + 4, summing the four k-quarters in order
        for (int item = wave; item < 2 * 3; item += 4) {
            const int i = item / 3, ot = item - 3 * i;
            const int op = 16 * i + px;
            if (i < ntiles && op < npx) {
                f4 v = ld4(b4 + 16 * ot + 4 * q);
#pragma unroll
                for (int w = 0; w < 4; ++w) v = v + part[((w * ntiles + i) * 3 + ot) * 64 + lane];
                const int tk = y4_0 * S4 + op;              // token index inside this crop
                const f4 pe = ld4(pos + (size_t)tk * 48 + 16 * ot + 4 * q);
                st4(tokens + ((size_t)b * L + tok_off + tk) * 48 + 16 * ot + 4 * q, v + pe);
            }
        }
    }
}

}  // namespace vts
