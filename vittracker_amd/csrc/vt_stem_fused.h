// vt_stem_fused.h -- the whole patch embedding of one frame in one workgroup (G128: 128 / 64 px crops).
//
// Same arithmetic as stem_a + stem_b (vt_stem.h; reference: Conv2d_BN / b16 / LevitPatchEmbedding and
// the pos-embed add + cat of OstrackDist.forward, lib/models/vit_dist/vit_dist.py:10-54,78-84), but
// nothing except the token rows leaves the CU: the layer-2 and layer-3 maps of both crops stay in LDS
// (139 KB), so the 15.7 MB intermediate, its end-of-kernel write-back, the second launch and its
// re-read disappear.
//
// Why the wave groups: a wave never overlaps its own VALU work with its own MFMAs
// (tools/src/probe_coissue.hip), and workgroups launched together run their phases in lock step, so
// in the two-kernel form the VALU layer and the MFMA layer simply added up.  Here the 16 waves form
// two groups of 8 (two waves per SIMD each) that work half a period apart:
//
//   interval      0        1        2        3        4        5
//   group A    L1(z)    L2(z)    L1(x1)   L2(x1)   L1(x3)   L2(x3)
//   group B   (fetch)   L1(x0)   L2(x0)   L1(x2)   L2(x2)     -
//
// L1 = layer 1 (3 -> 6, VALU, reads the crop from HBM; 512 pixel pairs = one per thread) into the
// group's own LDS ring; L2 = layer 2 (6 -> 12, MFMA) from that ring into the frame's layer-2 map.
// The two groups do not co-issue (f32 MFMA and VALU share the SIMD's issue, tools/src/probe_overlap.hip); what the
// stagger buys is that one group's waits -- HBM latency, scalar weight loads, LDS round trips -- are filled by the
// other group's work, and that only token rows ever leave the CU.  A band's input is fetched one interval ahead (during the group's L2), so HBM latency is
// exposed once per frame.  A band's top halo row is the previous band's last row, copied from the
// other group's ring (stable while that group is in its L2 interval).  Then all 16 waves run layer 3
// (12 -> 24) and layer 4 (24 -> 48) + pos-embed as in stem_b, on whole maps.
#pragma once
#include <type_traits>

#include "vt_common.h"
#include "vt_bf3.h"
#include "vt_conv.h"
#include "vt_stem.h"

namespace vts {

struct FusedGeo {                       // TX = 128, TZ = 64
    static constexpr int TX = 128, TZ = 64;
    static constexpr int R2X = 8, NBX = (TX / 4) / R2X;            // 4 search bands of 8 layer-2 rows
    static constexpr int R2Z = TZ / 4;                              // the template crop is one band of 16 rows
    // layer-1 ring (one per group): [2 planes][2 R2 + 1 rows][T/2 + 1]
    static constexpr int NPIX1X = round16((2 * R2X + 1) * (TX / 2 + 1));   // 1120
    static constexpr int NPIX1Z = round16((2 * R2Z + 1) * (TZ / 2 + 1));   // 1104
    static constexpr int RING = 2 * (NPIX1X > NPIX1Z ? NPIX1X : NPIX1Z);   // f4 per ring
    // layer-2 maps: [3 planes][S2 + 1 rows][S2 + 1], parity-split columns (stem_b's map2 layout)
    static constexpr int NPIX2X = round16((TX / 4 + 1) * (TX / 4 + 1));    // 1104
    static constexpr int NPIX2Z = round16((TZ / 4 + 1) * (TZ / 4 + 1));    // 304
    // layer-3 maps: [6 planes][S3 + 1 rows][S3 + 1] (stem_b's map3 layout); they reuse the rings
    static constexpr int NPIX3X = round16((TX / 8 + 1) * (TX / 8 + 1));    // 304
    static constexpr int NPIX3Z = round16((TZ / 8 + 1) * (TZ / 8 + 1));    // 96
    // small constants, copied once: layer-2 weight images (5 x 64 f4), b2 (4 f4), b3 (8 f4), b4 (12 f4)
    static constexpr int CONST_F4 = 5 * 64 + 4 + 8 + 12;
    // per-chunk map offsets of layers 3 / 4 as tables [crop][quad q][chunk] (ints): s2_chunk_off costs ~15 VALU instructions per
    // chunk and lane -- 220 of them around layer 3's 84 MFMAs, on the pipe the MFMAs need -- a table entry one LDS read
    static constexpr int OFF3 = 8, OFF4 = 16;                              // chunks per table row (7 and 14 used)
    static constexpr int OFFTAB_F4 = (2 * 4 * OFF3 + 2 * 4 * OFF4) / 4;    // 48
    static constexpr int LDS_F4 = 2 * RING + 3 * NPIX2X + 3 * NPIX2Z + CONST_F4 + OFFTAB_F4;
    static constexpr int LDS_BYTES = LDS_F4 * 16;                          // 144,768
    static_assert(6 * NPIX3X + 6 * NPIX3Z <= 2 * RING, "layer-3 maps must fit in the rings");
    // fp32 build: layer 3's three-piece weight image (24 KiB), staged behind the layer-3 maps in group B's ring during the pipeline's
    // last interval (that ring dies an interval early and its group idles)
    static constexpr int W3L_OFF = 6 * NPIX3X + 6 * NPIX3Z, W3L_TILES = 2 * 4 * 3;
    static_assert(W3L_OFF >= RING + 0 && W3L_OFF + W3L_TILES * 64 <= 2 * RING, "layer-3 weight pieces must lie inside group B's ring");
    // L3BF3 (round 4): layer 3 writes its output as PIECES (uint2 map[piece][plane][pixel]: 1.5 x the bytes) so that layer 4 runs as
    // three-piece bf16 products too.  Search map: the rings' first 3 * 6 * NPIX3X uint2 = 2736 f4 (layer 3's weight pieces move up
    // behind it, still inside group B's ring); template map: behind the offset tables, in LDS the fp32 form does not use.
    static constexpr int M3XP_F4 = 3 * 6 * NPIX3X / 2, M3ZP_F4 = 3 * 6 * NPIX3Z / 2;                     // 2736, 864
    static constexpr int W3P_OFF = M3XP_F4;
    static_assert(W3P_OFF >= RING && W3P_OFF + W3L_TILES * 64 <= 2 * RING, "layer-3 weight pieces (piece-map form) must lie inside group B's ring");
    static constexpr int M3ZP_OFF = LDS_F4;                                                              // f4 units from the LDS base
    static constexpr int LDS_BYTES_P = (LDS_F4 + M3ZP_F4) * 16;                                          // 158,592
    static_assert(LDS_BYTES_P <= 160 * 1024, "LDS");
    static constexpr int W4P_UNITS = 3 * 7 * 3 * 64;                                                     // layer-4 weight pieces: [out tile 3][pair 7][piece 3][64 lanes] x 16 B
    static_assert((2 * R2X) * (TX / 4) == 512 && (2 * R2Z) * (TZ / 4) == 512, "one pixel pair per thread of a group");
};

// One band of one crop.  Everything but the crop pointer is a compile-time constant of (crop, band index): left as run-time
// fields (selected per group and step) they cost ~100 VALU instructions of address set-up per layer-2 call.
template <bool IS_Z, int XB>
struct BandF {
    using G = FusedGeo;
    static constexpr bool is_z = IS_Z;
    static constexpr int T = IS_Z ? G::TZ : G::TX;
    static constexpr int lgT = IS_Z ? 6 : 7, HALF = T >> 2, lgHALF = lgT - 2, PITCH = (T >> 1) + 1;
    static constexpr int npix1 = IS_Z ? G::NPIX1Z : G::NPIX1X;
    static constexpr int R2 = IS_Z ? G::R2Z : G::R2X, p0 = IS_Z ? 0 : XB * G::R2X, lgW2 = lgT - 2;
    static constexpr int m2_off = IS_Z ? 3 * G::NPIX2X : 0, pitch2 = (T >> 2) + 1, half2 = T >> 3;
    static constexpr int npix2 = IS_Z ? G::NPIX2Z : G::NPIX2X;
    static constexpr bool halo = !IS_Z && XB > 0;      // top halo row comes from the other group's ring (else: image top, zeros)
    const float* in;                                    // this frame's crop (3, T, T)
};

// ZMODE 0: both crops; 1: search crop only (the template's token rows are cached in `tokens`); 2: template only.
// DIAG: the diagnostic build (VT_SKIP_* / VT_DBG_STAMPS); production instantiations compile `skip` and the stamps out -- as
// run-time conditions they put every band's prefetched registers through a copy at each conditional call.
// L3B (fp32 build; VT_STEM_BF3, default 1): layer 3 as exact three-piece bf16 products; false = on fp32 MFMAs, the form the f16 build
// always runs (on its own MFMA).
// U8 (round 6; ZMODE 1 only -- the tracker step, whose template is cached): `xin` is the uint8 (B, 128, 128, 3) patch vt_crop_u8 wrote
// (sample_target's own output), w1g / b1 point into the folded layer-1 image w1u (vt_stem.h: L1In).  A band's fetch is 3 loads of
// 12 bytes per thread instead of 9 of 16: a quarter of the bytes the first interval waits for.
template <int ZMODE, bool DIAG, bool L3B = true, bool U8 = false>
__global__ __launch_bounds__(1024) void stem_fused_kernel(
    const float* __restrict__ zin, const float* __restrict__ xin,                       // (B,3,64,64), (B,3,128,128)
    const float* __restrict__ w1g, const float* __restrict__ b1, const float* __restrict__ w2img, const float* __restrict__ b2,
    const float* __restrict__ w3img, const float* __restrict__ b3, const float* __restrict__ w4img, const float* __restrict__ b4,
    const float* __restrict__ pos_z, const float* __restrict__ pos_x, float* __restrict__ tokens, int L, int len_z, int skip_arg,
    unsigned long long* __restrict__ stamps,       // diagnostic (VT_DBG_STAMPS), null in production: [B][16][32]
    const float* __restrict__ w2k,                 // layer-2 weights as [tap][input channels 0-3 | 4-5 + padding][16 output channels][4] (f32 build)
    const float* __restrict__ w3b,                 // layer-3 weights as three-piece bf16 images [out tile 2][chunk pair 4][piece 3][64 lanes][8 bf16] (f32 build)
    const float* __restrict__ w4b) {               // layer-4 weights, the same way: [out tile 3][chunk pair 7][piece 3][64 lanes][8 bf16]
    using G = FusedGeo;
    static_assert(!U8 || ZMODE == 1, "the uint8 patch form is the search-only (cached template) step");
    constexpr bool L3BF3 = L3B && !VT_IS_F16;
    constexpr bool do_z = ZMODE != 1, do_x = ZMODE != 2;
    const int skip = DIAG ? skip_arg : 0;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    f4* const lds = reinterpret_cast<f4*>(lds_f);
    f4* const ring0 = lds;                                   // group A's layer-1 ring
    f4* const m2 = lds + 2 * G::RING;                        // layer-2 maps: x then z
    constexpr int M2Z_OFF = 3 * G::NPIX2X;
    f4* const m3x = lds;                                     // layer-3 maps reuse the rings (dead by then)
    f4* const m3z = lds + 6 * G::NPIX3X;
    // L3BF3: the layer-3 maps as pieces, uint2 map[piece][plane][pixel] (FusedGeo): layer 4 runs on the bf16 pipe as well
    vt3::u32x2* const m3xp = reinterpret_cast<vt3::u32x2*>(lds);
    vt3::u32x2* const m3zp = reinterpret_cast<vt3::u32x2*>(lds + G::M3ZP_OFF);
    f4* const cw2 = lds + 2 * G::RING + 3 * G::NPIX2X + 3 * G::NPIX2Z;   // [5][64] layer-2 weight images
    const float* const cb2 = reinterpret_cast<const float*>(cw2 + 5 * 64);   // 16 floats
    const float* const cb3 = cb2 + 16;                                       // 32
    const float* const cb4 = cb3 + 32;                                       // 48
    int* const otab = reinterpret_cast<int*>(cw2 + G::CONST_F4);             // [x | z][4][OFF3] then [x | z][4][OFF4]

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 3, gw = wave & 7;
    const int q = lane >> 4, px = lane & 15;
    f4* const ring = ring0 + grp * G::RING;
    const f4* const other_ring = ring0 + (1 - grp) * G::RING;

    // bands of this frame: group A = {z, x1, x3}, group B = {x0, x2}
    const float* const zin_b = zin + (size_t)b * 3 * G::TZ * G::TZ;
    const float* const xin_b = U8 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(xin) + (size_t)b * 3 * G::TX * G::TX)
                                  : xin + (size_t)b * 3 * G::TX * G::TX;
    const BandF<true, 0> bz{zin_b};
    const BandF<false, 0> bx0{xin_b};
    const BandF<false, 1> bx1{xin_b};
    const BandF<false, 2> bx2{xin_b};
    const BandF<false, 3> bx3{xin_b};

    // ---- layer 1 pieces ------------------------------------------------------------------------------
    const int pair_ = gw * 64 + lane;                        // this thread's pixel pair of a band (0..511)
    // Every phase derives its indices and LDS addresses from a FRESH copy of the thread's index: left to itself hipcc shares
    // those sub-expressions between the five bands' phases, keeps them all live from the first use on and runs out of registers
    // (36 B/lane of scratch, with the spill's wait in front of the first band's loads).
    auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // Raw loads only: nothing here may depend on the loaded data, so the requests stay in flight across
    // the layer-2 work and the barrier that follow (the top-of-image zeroing is applied in layer1).
    // Buffer loads (scalar descriptor + one 32-bit offset VGPR; kernel rows 1 and 2 are immediate offsets of the same register,
    // the channel plane is the scalar offset): a fetch holds 2 address registers instead of 9 64-bit pairs.
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zin_b), 0, 3 * G::TZ * G::TZ * 4, 0x00020000);
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin_b), 0, 3 * G::TX * G::TX * (U8 ? 1 : 4), 0x00020000);
    auto fetch = [&](const auto& J, L1In<U8>& vin) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const int pair = fresh(pair_);
        const int lr = 1 + (pair >> J.lgHALF), qp = pair & (J.HALF - 1);
        const int p1 = 2 * J.p0 - 1 + lr;                    // layer-1 row (>= 0)
        if constexpr (U8) {     // the uint8 patch: row y, pixels 4 qp .. 4 qp + 3 = bytes 12 (y T / 4 + qp) .. + 11, one load per kernel row
            const unsigned o1 = 12u * ((((unsigned)(2 * p1)) << (J.lgT - 2)) + (unsigned)qp);
            const unsigned o0 = p1 > 0 ? o1 - (3u << J.lgT) : o1;                              // the image top reads row 0 (replaced in layer1)
            vin.v[0] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o0, 0, 0);
            vin.v[1] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o1, 0, 0);
            vin.v[2] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o1 + (3u << J.lgT), 0, 0);
            return;
        } else {
        auto& v = vin.v;
        const unsigned off1 = ((((unsigned)(2 * p1)) << J.lgT) + 4u * (unsigned)qp) << 2;      // input row 2 p1 (kernel row 1), bytes
        const unsigned off0 = p1 > 0 ? off1 - (4u << J.lgT) : off1;                            // row 2 p1 - 1; the image top reads row 0 (zeroed in layer1)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const u4 t = __builtin_amdgcn_raw_buffer_load_b128(J.is_z ? rsrc_z : rsrc_x, r == 0 ? off0 : off1 + (r == 2 ? (4u << J.lgT) : 0u),
                                                                   c << (2 * J.lgT + 2), 0);
                v[r][c] = __builtin_bit_cast(f4, t);
            }
        }
    };
    auto layer1 = [&](const auto& J, const L1In<U8>& vin) {
        // layer 1 is the long pole of an interval (VALU-bound); without this the issue arbiter favours the
        // older group whatever it is doing, and the younger group's layer 1 takes three times as long
        __builtin_amdgcn_s_setprio(3);
        const int pair = fresh(pair_);
        const int lr = 1 + (pair >> J.lgHALF), qp = pair & (J.HALF - 1);
        const float keep0 = (2 * J.p0 - 1 + lr) > 0 ? 1.f : 0.f;   // kernel row 0 of layer-1 row 0 is the zero padding
        // housekeeping by a few threads: column -1 of every ring row, and the halo row
        const int nrow = 2 * J.R2 + 1;
        if (pair < 2 * nrow) {
            const int plane = pair >= nrow ? 1 : 0;
            ring[plane * J.npix1 + (pair - plane * nrow) * J.PITCH + J.HALF] = splat4(0.f);
        }
        if (J.halo && pair >= 128 && pair < 128 + 2 * J.PITCH) {
            const int e = pair - 128, plane = e >= J.PITCH ? 1 : 0, col = e - plane * J.PITCH;
            ring[plane * J.npix1 + col] = other_ring[plane * J.npix1 + 2 * J.R2 * J.PITCH + col];
        }
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
            float padv = 0.f;                                 // what a tap outside the crop reads (fp32 form: the zero padding itself)
            if constexpr (U8) {
                padv = b1[W1U_PAD - W1U_BIAS + c];            // 255 mean_c: normalises to zero
                vv = l1_channel(vin.v[r], c);
                if (r == 0 && J.p0 == 0 && keep0 == 0.f) vv = splat4(padv);
            } else {
                vv = (r == 0 && J.p0 == 0) ? vin.v[r][c] * splat4(keep0) : vin.v[r][c];     // kernel row 0 is the zero padding only in the band at the image top
            }
            const float left = lane_left(vv.w);               // a wave starts at a row start: lane 0 has qp = 0
            const float t0[3] = {qp > 0 ? left : padv, vv.x, vv.y}, t1[3] = {vv.y, vv.z, vv.w};
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    a0[j] = fmaf(t0[s], cur[s * 6 + j], a0[j]);
                    a1[j] = fmaf(t1[s], cur[s * 6 + j], a1[j]);
                }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { a0[j] = hardswish(a0[j]); a1[j] = hardswish(a1[j]); }
        f4* dst = ring + lr * J.PITCH;
        dst[qp] = f4{a0[0], a0[1], a0[2], a0[3]};                       // even column 2 qp, channels 0-3
        dst[J.npix1 + qp] = f4{a0[4], a0[5], 0.f, 0.f};                 //                   channels 4-5 (+ padding)
        dst[J.HALF + 1 + qp] = f4{a1[0], a1[1], a1[2], a1[3]};          // odd column 2 qp + 1
        dst[J.npix1 + J.HALF + 1 + qp] = f4{a1[4], a1[5], 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- layer 2: this group's ring -> the frame's layer-2 map ------------------------------------------
    auto layer2 = [&](const auto& J) {
#ifndef VT_F16
        // Layer 2 on v_mfma_f32_16x16x1_4B_f32: four 16x16 blocks per instruction, K = 1.  Its 6 input channels make 54 real
        // k-steps; on the 16x16x4 form (k in quads of 4 channels, chunks of 4 quads) they pad to 80.  Block b = pixel tile 4 gw + b:
        // lane (b, px) SUPPLIES pixel px of that tile as B, every lane supplies W[oc = px][k] as A (the same for the four
        // blocks), and lane (q, px) RECEIVES channels 4q..4q+3 of pixel px of all four tiles (tools/src/probe_mfma4b.hip).
        // Four waves of the group (one per SIMD) cover the band's 16 tiles; the other four go straight to the barrier.
        if (gw < 4) {
            typedef float f16v __attribute__((ext_vector_type(16)));
            const int ln = fresh(lane), q = ln >> 4, px = ln & 15;
            const int op = 16 * (4 * gw + q) + px, yy = op >> J.lgW2, xx = op & ((1 << J.lgW2) - 1);
            const f4* src = ring + 2 * yy * J.PITCH + xx;                     // tap (0,0) of this lane's pixel, channel quad 0
            const f4* wk = cw2 + px;                                          // [tap][channels 0-3 | 4-5][16 output channels] float4: 16 lanes read 16 consecutive entries
            const f4 bv2 = ld4(cb2 + 4 * q);
            f16v acc = {bv2.x, bv2.y, bv2.z, bv2.w, bv2.x, bv2.y, bv2.z, bv2.w, bv2.x, bv2.y, bv2.z, bv2.w, bv2.x, bv2.y, bv2.z, bv2.w};
            auto tapoff = [&](int tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                return dy * J.PITCH + (dx == 1 ? 0 : (dx == 0 ? J.HALF : J.HALF + 1));
            };
            f4 a0 = src[tapoff(0)], a1 = src[J.npix1 + tapoff(0)], w0 = wk[0], w1 = wk[16];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                f4 na0 = a0, na1 = a1, nw0 = w0, nw1 = w1;
                if (tap + 1 < 9) {
                    na0 = src[tapoff(tap + 1)]; na1 = src[J.npix1 + tapoff(tap + 1)];
                    nw0 = wk[32 * (tap + 1)]; nw1 = wk[32 * (tap + 1) + 16];
                    __builtin_amdgcn_sched_barrier(0);        // keep the next tap's reads ahead of this tap's MFMAs
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.x, a0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.y, a0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.z, a0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.w, a0.w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w1.x, a1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w1.y, a1.y, acc, 0, 0, 0);
                a0 = na0; a1 = na1; w0 = nw0; w1 = nw1;
            }
            if (q < 3) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int ob = 16 * (4 * gw + bb) + px, y = ob >> J.lgW2, x = ob & ((1 << J.lgW2) - 1);
                    f4 r = {acc[4 * bb], acc[4 * bb + 1], acc[4 * bb + 2], acc[4 * bb + 3]};
                    r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                    m2[J.m2_off + q * J.npix2 + (J.p0 + y + 1) * J.pitch2 + ((x & 1) ? J.half2 + 1 + (x >> 1) : (x >> 1))] = r;
                }
            }
        }
#else
        f4 w2a[5][1];
#pragma unroll
        for (int c = 0; c < 5; ++c) w2a[c][0] = cw2[c * 64 + lane];
        const f4 bv2 = ld4(cb2 + 4 * q);
        int base[2], yy[2], xx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int op = 16 * (gw + 8 * i) + px;                       // 16 tiles per band: tiles gw and gw + 8
            yy[i] = op >> J.lgW2; xx[i] = op & ((1 << J.lgW2) - 1);
            base[i] = 2 * yy[i] * J.PITCH + xx[i];
        }
        f4 acc[2][1] = {{bv2}, {bv2}};
        auto off2 = [&](int c) { return s2_chunk_off<2>(c, q, J.npix1, J.PITCH, J.HALF); };
        vtc::mma_pass<1, 2, 5, 5, true>(ring, base, w2a, 0, off2, acc);
        if (q < 3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f4 v = acc[i][0];
                v.x = hardswish(v.x); v.y = hardswish(v.y); v.z = hardswish(v.z); v.w = hardswish(v.w);
                const int x = xx[i];
                m2[J.m2_off + q * J.npix2 + (J.p0 + yy[i] + 1) * J.pitch2 + ((x & 1) ? J.half2 + 1 + (x >> 1) : (x >> 1))] = v;
            }
        }
#endif
    };

    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (DIAG) {
            if (stamps != nullptr) {
                unsigned long long tt;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
                if (lane == 0) stamps[((size_t)b * 16 + wave) * 32 + nstamp] = tt;
                ++nstamp;
            }
        }
    };
    stamp();
    // ---- start ---------------------------------------------------------------------------------------------------------
    // Order in the memory queue: (1) this thread's constant (tiny, cache-resident), (2) group A's first band -- the template
    // crop, which the first interval computes on, (3) group B's first band, needed one interval later.  The constant's load is
    // older than the crop fetch, so the LDS write below waits for it alone (counted vmcnt); issued after the fetch it sat behind
    // 26 MB of crop requests from all workgroups (a 5-8 k cycle prologue).
    L1In<U8> v;
    const int t = threadIdx.x;
#ifndef VT_F16
    constexpr int NW2 = 9 * 32;                              // [tap][2][16][4] floats: layer-2 weights for the 4-block MFMA
    const float* const w2src = w2k;
#else
    constexpr int NW2 = 5 * 64;
    const float* const w2src = w2img;
#endif
    const float* csrc = w2src;                               // threads without a constant load a valid address and drop it
    const bool has_c = t < NW2 || (t >= 320 && t < 344);
    if (t < NW2) csrc = w2src + 4 * t;
    else if (t >= 320 && t < 324) csrc = b2 + 4 * (t - 320);
    else if (t >= 324 && t < 332) csrc = b3 + 4 * (t - 324);
    else if (t >= 332 && t < 344) csrc = b4 + 4 * (t - 332);
    const f4 cst = ld4(csrc);
    __builtin_amdgcn_sched_barrier(0);
    // Round 6: wave w of a group issues its first band's loads w x VT_SF_STAGGER x 64 cycles late.  All 256 workgroups start together and ask
    // for 25 MB at once; issued in the same cycle, every wave's nine loads come back with the LAST of them (the ~6.5 k-cycle prologue).
    // Staggered, the memory system serves the waves in order and wave 0 computes while wave 7's rows are still on their way.  Measured
    // (tools/ab_stages.py, two box sessions, us per launch): 0: 21.1-21.2 * 4: 21.1 * 6: 20.45-20.54 * 8: 20.7 * 10: 21.5 * 16: 24.0.
#ifndef VT_SF_STAGGER
#define VT_SF_STAGGER 6
#endif
    auto stagger = [&](int slots) {
        if constexpr (VT_SF_STAGGER > 0)
            for (int i = 0; i < slots * VT_SF_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
    };
    if (grp == 0) {      // every thread that holds a constant is in group A (t < 344): fetch and LDS write in one straight line,
                         // so the wait in front of the write is a counted one (the nine crop loads stay in flight)
        stagger(gw);
        if (do_z) fetch(bz, v);
        if (has_c) cw2[t] = cst;
    }
    {   // zero only what is read without ever being written: the top rows of the rings,
        // row 0 and column -1 of the layer-2 maps (column -1 of the rings is cleared per band in layer1)
        if (t < 192) {                                                  // the layer-3 / layer-4 offset tables, one entry per thread
            const bool l4 = t >= 64;
            const int e = l4 ? t - 64 : t, per = l4 ? 4 * G::OFF4 : 4 * G::OFF3;
            const bool isz = e >= per;
            const int r = isz ? e - per : e, qq = l4 ? r / G::OFF4 : r / G::OFF3, c = l4 ? r - qq * G::OFF4 : r - qq * G::OFF3;
            int v;
            if (!l4) v = isz ? s2_chunk_off<3>(c, qq, G::NPIX2Z, G::TZ / 4 + 1, G::TZ / 8) : s2_chunk_off<3>(c, qq, G::NPIX2X, G::TX / 4 + 1, G::TX / 8);
            else v = isz ? s2_chunk_off<6>(c, qq, G::NPIX3Z, G::TZ / 8 + 1, G::TZ / 16) : s2_chunk_off<6>(c, qq, G::NPIX3X, G::TX / 8 + 1, G::TX / 16);
            otab[t] = v;
        } else if (t < 384) {}
        else if (t >= 384 && t < 384 + 2 * 65) {                        // ring B, row 0 (first used by x0: image top)
            const int e = t - 384, pl = e / 65, col = e - pl * 65;
            ring0[G::RING + pl * G::NPIX1X + col] = splat4(0.f);
        } else if (t >= 520 && t < 520 + 2 * 33) {                      // ring A, row 0 in the z layout (its x bands copy a halo)
            const int e = t - 520, pl = e / 33, col = e - pl * 33;
            ring0[pl * G::NPIX1Z + col] = splat4(0.f);
        } else if (t >= 704 && t < 704 + 3 * 66) {                      // search layer-2 map: row 0 and column -1
            const int e = t - 704, pl = e / 66, k = e - pl * 66;
            m2[pl * G::NPIX2X + (k < 33 ? k : (k - 33) * 33 + 16)] = splat4(0.f);
        } else if (t >= 902 && t < 902 + 3 * 34) {                      // template layer-2 map
            const int e = t - 902, pl = e / 34, k = e - pl * 34;
            m2[M2Z_OFF + pl * G::NPIX2Z + (k < 17 ? k : (k - 17) * 17 + 8)] = splat4(0.f);
        }
    }
    if (grp == 1 && do_x) { stagger(do_z ? 8 + gw : gw); fetch(bx0, v); }     // behind group A's requests (delaying it further changed nothing: measured)
    stamp();
    // No barrier here: nothing written above is read before the first interval's barrier (layer 1 reads no LDS,
    // and its ring writes do not overlap the entries cleared above).
    stamp();

    // ---- the staggered layer-1 / layer-2 pipeline --------------------------------------------------------
    // Written out per group (no loop-carried registers: a prefetched band flows straight into its layer1).
    // Both sequences execute the same six barriers.
    // layer-3 / layer-4 weights of this wave are requested while the pipeline's last interval runs /
    // while layer 3 runs, so their L2 round trips are not exposed
    constexpr int NCH4 = 14;
    constexpr int NCH3 = 7;
    const int ot3 = wave & 1;
    opnd w3a[NCH3][1];
    // L3BF3: layer 3 runs as three-piece bf16 products (vt_bf3.h): every wave needs the WHOLE image, so it goes through LDS.  Group B's
    // eight waves stage it (three LDS-DMA pieces each) in the interval in which they have nothing else to do.
    auto load_w3 = [&]() {
        if constexpr (L3BF3) {
            if (grp == 1) {
                for (int t = gw; t < G::W3L_TILES; t += 8)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w3b + (size_t)t * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(lds + G::W3P_OFF + t * 64), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            vtc::load_weights<1, NCH3, NCH3>(w3img + (size_t)ot3 * NCH3 * 256, 0, NCH3, lane, w3a);
        }
    };
    if (skip & 1) { load_w3(); __syncthreads(); }
    if (!(skip & 1)) {
        const bool l2 = !(skip & 2);
        // zmode: a crop that is not wanted keeps its barriers and drops its work (all conditions are wave-uniform)
        // The next band's input is requested one interval ahead: by the four waves without layer-2 work (gw >= 4) at the start of
        // the interval, by the four that run layer 2 right after it (its accumulators, double-buffered operands and the nine
        // prefetched float4 together do not fit in 128 registers; layer 2 takes a third of the interval, the rest covers the loads).
        // (ONE fetch site: layer 2 is a no-op for the waves with gw >= 4, so "after layer 2" is "at once" for them; two sites made
        // hipcc give the prefetch two register sets and merge them with 18 v_mov_b64 per band)
        auto l2_and_fetch = [&](const auto& J2, bool run2, const auto& Jn, bool run_f) {
            if (run2) layer2(J2);
            if (run_f) fetch(Jn, v);
        };
        if (grp == 0) {
            if (do_z) layer1(bz, v);                    stamp(); __syncthreads(); stamp();   // 0: L1(z)
            l2_and_fetch(bz, l2 && do_z, bx1, do_x);    stamp(); __syncthreads(); stamp();   // 1: L2(z), x1 requested
            if (do_x) layer1(bx1, v);                   stamp(); __syncthreads(); stamp();   // 2: L1(x1)
            l2_and_fetch(bx1, l2 && do_x, bx3, do_x);   stamp(); __syncthreads(); stamp();   // 3: L2(x1), x3 requested
            if (do_x) layer1(bx3, v);                   stamp(); __syncthreads(); stamp();   // 4: L1(x3)
            load_w3();     // the layer-3 weights' L2 round trip runs under this interval (no prefetched band is live any more)
            if (l2 && do_x) layer2(bx3);                stamp(); __syncthreads(); stamp();   // 5: L2(x3)
        } else {
            stamp(); __syncthreads(); stamp();   // 0: (x0 requested at kernel start)
            if (do_x) layer1(bx0, v);                   stamp(); __syncthreads(); stamp();   // 1: L1(x0)
            l2_and_fetch(bx0, l2 && do_x, bx2, do_x);   stamp(); __syncthreads(); stamp();   // 2: L2(x0), x2 requested
            if (do_x) layer1(bx2, v);                   stamp(); __syncthreads(); stamp();   // 3: L1(x2)
            if (l2 && do_x) layer2(bx2);                stamp(); __syncthreads(); stamp();   // 4: L2(x2)
            load_w3();
            __syncthreads();   // 5
        }
    }

    // ---- layer 3 (12 -> 24, Hardswish) on the whole maps, all 16 waves -------------------------------------
    // layer-4 work item of this wave: 15 (pixel tile, output tile) items, one per wave
    const bool z4 = wave >= 12;
    const int item4 = z4 ? wave - 12 : wave;
    const int tile4 = z4 ? 0 : item4 / 3, ot4 = z4 ? item4 : item4 - 3 * tile4;
    opnd w4a[NCH4][1];
    if constexpr (!L3BF3) {
        if (wave < 15) vtc::load_weights<1, NCH4, NCH4>(w4img + (size_t)ot4 * NCH4 * 256, 0, NCH4, lane, w4a);
    }
    {
        // pads of the layer-3 maps (they alias the rings, which hold layer-1 data): row 0 and column -1
        if constexpr (L3BF3) {
            for (int i = threadIdx.x; i < 3 * 6 * (2 * 17 + 2 * 9); i += 1024) {        // every piece plane
                const int pp = i / 52, e = i - pp * 52;                                 // pp = piece * 6 + plane
                const vt3::u32x2 zero = {0u, 0u};
                if (e < 17) m3xp[pp * G::NPIX3X + e] = zero;
                else if (e < 34) m3xp[pp * G::NPIX3X + (e - 17) * 17 + 8] = zero;
                else if (e < 43) m3zp[pp * G::NPIX3Z + (e - 34)] = zero;
                else m3zp[pp * G::NPIX3Z + (e - 43) * 9 + 4] = zero;
            }
        } else {
        for (int i = threadIdx.x; i < 6 * (2 * 17 + 2 * 9); i += 1024) {
            const int plane = i / 52, e = i - plane * 52;
            if (e < 17) m3x[plane * G::NPIX3X + e] = splat4(0.f);                               // row 0
            else if (e < 34) m3x[plane * G::NPIX3X + (e - 17) * 17 + 8] = splat4(0.f);         // column -1
            else if (e < 43) m3z[plane * G::NPIX3Z + (e - 34)] = splat4(0.f);
            else m3z[plane * G::NPIX3Z + (e - 43) * 9 + 4] = splat4(0.f);
        }
        }
        if constexpr (L3BF3) {
        // fp32 build: layer 3 as exact three-piece bf16 products.  The layer-2 maps stay fp32 (pre-split maps do not fit), so a B
        // operand is split where it is read -- which pays only if one split feeds BOTH output tiles: a wave takes one pixel tile
        // (x: map row `wave`; z: waves 0-3, 16 pixels each) and both tiles; 7 splits + 48 bf16 MFMAs per unit instead of 56 fp32
        // MFMAs per (unit, output tile), and the four waves of a SIMD overlap each other's splits and MFMAs.  Chunk 7 does not
        // exist (27 quads): its half of the last pair carries zero weights and re-uses chunk 6's pieces.
        if (!(skip & 4)) {
            using vt3::u32x2;
            using vt3::u32x4;
            const u32x4* const W3L = reinterpret_cast<const u32x4*>(lds + G::W3P_OFF);
            constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
            auto unit3 = [&](const f4* map2, int base, const int* tab, f4 (&acc)[2]) {
                int o3[G::OFF3];
                {
                    const int4* tp = reinterpret_cast<const int4*>(tab);
                    const int4 a = tp[0], bq = tp[1];
                    o3[0] = a.x; o3[1] = a.y; o3[2] = a.z; o3[3] = a.w; o3[4] = bq.x; o3[5] = bq.y; o3[6] = bq.z; o3[7] = bq.w;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    u32x4 A[2][3];
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) A[ot][pc] = W3L[((ot * 4 + p) * 3 + pc) * 64 + lane];
                    u32x2 lo[3], hi[3];
                    vt3::split3(map2[o3[2 * p] + base], lo[0], lo[1], lo[2]);
                    if (p < 3) vt3::split3(map2[o3[2 * p + 1] + base], hi[0], hi[1], hi[2]);
                    u32x4 B[3];
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) B[pc] = p < 3 ? u32x4{lo[pc].x, lo[pc].y, hi[pc].x, hi[pc].y} : u32x4{lo[pc].x, lo[pc].y, lo[pc].x, lo[pc].y};
#pragma unroll
                    for (int e = 0; e < 6; ++e)
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot) acc[ot] = vt3::mma(A[ot][TW[e]], B[TX[e]], acc[ot]);
                }
            };
            const f4 bias0 = ld4(cb3 + 4 * q), bias1 = ld4(cb3 + 16 + 4 * q);
            if (do_x) {   // search: pixel tile = row `wave` of the 16 x 16 map
                constexpr int P2 = G::TX / 4 + 1, P3 = G::TX / 8 + 1, H3 = G::TX / 16;
                f4 acc[2] = {bias0, bias1};
                unit3(m2, 2 * wave * P2 + px, otab + q * G::OFF3, acc);
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    if (16 * ot + 4 * q < 24) {
                        f4 r = acc[ot];
                        r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                        u32x2 pcs[3];
                        vt3::split3(r, pcs[0], pcs[1], pcs[2]);        // split ONCE where it is produced: layer 4 reads pieces
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            m3xp[(pc * 6 + 4 * ot + q) * G::NPIX3X + (wave + 1) * P3 + ((px & 1) ? H3 + 1 + (px >> 1) : (px >> 1))] = pcs[pc];
                    }
            }
            if (wave < 4 && do_z) {   // template: 4 pixel tiles of the 8 x 8 map
                constexpr int P2 = G::TZ / 4 + 1, P3 = G::TZ / 8 + 1, H3 = G::TZ / 16;
                const int op = 16 * wave + px, y = op >> 3, x = op & 7;
                f4 acc[2] = {bias0, bias1};
                unit3(m2 + M2Z_OFF, 2 * y * P2 + x, otab + 4 * G::OFF3 + q * G::OFF3, acc);
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
                    if (16 * ot + 4 * q < 24) {
                        f4 r = acc[ot];
                        r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                        u32x2 pcs[3];
                        vt3::split3(r, pcs[0], pcs[1], pcs[2]);
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            m3zp[(pc * 6 + 4 * ot + q) * G::NPIX3Z + (y + 1) * P3 + ((x & 1) ? H3 + 1 + (x >> 1) : (x >> 1))] = pcs[pc];
                    }
            }
        }
        } else {
        const f4 bv3 = ld4(cb3 + 16 * ot3 + 4 * q);
        if (!(skip & 4)) {
            if (do_x) {   // search: 16 pixel tiles (rows of the 16 x 16 map); this wave: rows wave>>1 and (wave>>1) + 8
                constexpr int P2 = G::TX / 4 + 1, P3 = G::TX / 8 + 1, H3 = G::TX / 16;
                int base[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) base[i] = 2 * ((wave >> 1) + 8 * i) * P2 + px;
                f4 acc[2][1] = {{bv3}, {bv3}};
                int o3[G::OFF3];
                {
                    const int4* tp = reinterpret_cast<const int4*>(otab + q * G::OFF3);
                    const int4 a = tp[0], bq = tp[1];
                    o3[0] = a.x; o3[1] = a.y; o3[2] = a.z; o3[3] = a.w; o3[4] = bq.x; o3[5] = bq.y; o3[6] = bq.z; o3[7] = bq.w;
                }
                auto off3 = [&](int c) { return o3[c]; };
                vtc::mma_pass<1, 2, NCH3, NCH3>(m2, base, w3a, 0, off3, acc);
                if (16 * ot3 + 4 * q < 24) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f4 r = acc[i][0];
                        r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                        const int y = (wave >> 1) + 8 * i;
                        // the store must wait until every wave has finished reading the rings?  No: layer 3 reads
                        // only the layer-2 maps; the rings died at the pipeline's last barrier.
                        m3x[(4 * ot3 + q) * G::NPIX3X + (y + 1) * P3 + ((px & 1) ? H3 + 1 + (px >> 1) : (px >> 1))] = r;
                    }
                }
            }
            if (wave < 8 && do_z) {   // template: 4 pixel tiles of the 8 x 8 map
                constexpr int P2 = G::TZ / 4 + 1, P3 = G::TZ / 8 + 1, H3 = G::TZ / 16;
                const int op = 16 * (wave >> 1) + px, y = op >> 3, x = op & 7;
                int base[1] = {2 * y * P2 + x};
                f4 acc[1][1] = {{bv3}};
                int o3[G::OFF3];
                {
                    const int4* tp = reinterpret_cast<const int4*>(otab + 4 * G::OFF3 + q * G::OFF3);
                    const int4 a = tp[0], bq = tp[1];
                    o3[0] = a.x; o3[1] = a.y; o3[2] = a.z; o3[3] = a.w; o3[4] = bq.x; o3[5] = bq.y; o3[6] = bq.z; o3[7] = bq.w;
                }
                auto off3 = [&](int c) { return o3[c]; };
                vtc::mma_pass<1, 1, NCH3, NCH3>(m2 + M2Z_OFF, base, w3a, 0, off3, acc);
                if (16 * ot3 + 4 * q < 24) {
                    f4 r = acc[0][0];
                    r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                    m3z[(4 * ot3 + q) * G::NPIX3Z + (y + 1) * P3 + ((x & 1) ? H3 + 1 + (x >> 1) : (x >> 1))] = r;
                }
            }
        }
        }
    }
    // L3BF3: this wave's first two chunk pairs of layer-4 weight pieces are requested in FRONT of the barrier that ends layer 3 --
    // their L2 round trip runs under the 2-3 k cycles most waves wait there (the fp32 form holds all of w4a from before layer 3)
    vt3::u32x4 A4[4][3];
    if constexpr (L3BF3) {
        if (wave < 15 && !(skip & 8) && (z4 ? do_z : do_x)) {
            const vt3::u32x4* const wg0 = reinterpret_cast<const vt3::u32x4*>(w4b) + (size_t)ot4 * (NCH4 / 2) * 3 * 64 + lane;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) A4[p][pc] = wg0[(p * 3 + pc) * 64];
        }
    }
    stamp();
    __syncthreads();
    stamp();

    // ---- layer 4 (24 -> 48) + pos-embed -> token rows: 15 (pixel tile, output tile) items, one per wave ------
    if (wave < 15 && !(skip & 8) && (z4 ? do_z : do_x)) {
        const bool is_z = z4;
        const int tile = tile4, ot = ot4;
        const int lgS4 = is_z ? 2 : 3;
        const int P3 = is_z ? G::TZ / 8 + 1 : G::TX / 8 + 1;
        const f4* map3 = is_z ? m3z : m3x;
        const int op = 16 * tile + px, y = op >> lgS4, x = op & ((1 << lgS4) - 1);
        int base[1] = {2 * y * P3 + x};
        f4 acc[1][1] = {{ld4(cb4 + 16 * ot + 4 * q)}};
        const f4 pe = ld4((is_z ? pos_z : pos_x) + (size_t)op * 48 + 16 * ot + 4 * q);   // requested before the MFMAs
        int o4[G::OFF4];
        {
            const int4* tp = reinterpret_cast<const int4*>(otab + 2 * 4 * G::OFF3 + (is_z ? 4 * G::OFF4 : 0) + q * G::OFF4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int4 a = tp[i];
                o4[4 * i] = a.x; o4[4 * i + 1] = a.y; o4[4 * i + 2] = a.z; o4[4 * i + 3] = a.w;
            }
        }
        auto off4 = [&](int c) { return o4[c]; };
        if constexpr (L3BF3) {
            // layer 4 as exact three-piece bf16 products: B operands are the pieces layer 3 wrote (two 8-byte reads per piece and chunk
            // pair), A operands this output tile's weight pieces straight from L2, three pairs ahead; 7 pairs x 6 MFMAs on two
            // accumulators (even / odd pairs) instead of 56 fp32 MFMAs.  Chunk 13 has two real quads and two pad quads (zero weights;
            // their offsets clamp to a valid quad).
            using vt3::u32x2;
            using vt3::u32x4;
            constexpr int NP4 = NCH4 / 2, TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
            const int ps = is_z ? 6 * G::NPIX3Z : 6 * G::NPIX3X;                    // piece stride (entries)
            const u32x2* const mp = is_z ? m3zp : m3xp;
            const u32x4* const wg = reinterpret_cast<const u32x4*>(w4b) + (size_t)ot * NP4 * 3 * 64 + lane;
            auto load_a = [&](int p, u32x4 (&A)[3]) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) A[pc] = wg[(p * 3 + pc) * 64];
            };
            auto read_b = [&](int p, u32x4 (&Bv)[3]) {
                const int o0 = base[0] + o4[2 * p], o1 = base[0] + o4[2 * p + 1];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    const u32x2 lo = mp[pc * ps + o0], hi = mp[pc * ps + o1];
                    Bv[pc] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
            };
            u32x4 (&A)[4][3] = A4;          // three pairs ahead: an L2 round trip is longer than two pairs' MFMAs
            u32x4 Bv[2][3];
            read_b(0, Bv[0]);
            f4 accA = acc[0][0], accB = splat4(0.f);
#pragma unroll
            for (int p = 0; p < NP4; ++p) {
                if (p + 3 < NP4) load_a(p + 3, A[(p + 3) & 3]);
                if (p + 1 < NP4) read_b(p + 1, Bv[(p + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    if (p & 1) accB = vt3::mma(A[p & 3][TW[e]], Bv[p & 1][TX[e]], accB);
                    else accA = vt3::mma(A[p & 3][TW[e]], Bv[p & 1][TX[e]], accA);
                }
            }
            st4(tokens + ((size_t)b * L + (is_z ? 0 : len_z) + op) * 48 + 16 * ot + 4 * q, (accA + accB) + pe);
        } else {
        vtc::mma_pass<1, 1, NCH4, NCH4>(map3, base, w4a, 0, off4, acc);
        st4(tokens + ((size_t)b * L + (is_z ? 0 : len_z) + op) * 48 + 16 * ot + 4 * q, acc[0][0] + pe);
        }
    }
    stamp();
}

// ------------------------------------------------------------------------------------------ stem_pipe
// Layers 1 + 2 of both crops of one frame in one workgroup, for geometries whose layer-2 maps do not fit
// in LDS (G256): the layer-1 / layer-2 half of stem_fused_kernel -- two 8-wave groups half a period apart,
// one on the VALU layer, the other on the MFMA layer of the previous band -- as a loop over the frame's
// bands (template crop first), with layer 2 writing the channel-quad planes that stem_b reads.
// Replaces stem_a, whose four workgroups per CU run the two layers in lock step (their times add up).
template <int TX, int TZ>
struct PipeGeo {
    static constexpr int R2X = 1024 / TX, R2Z = 1024 / TZ;       // 512 pixel pairs per band = one per thread of a group
    static constexpr int NBX = (TX / 4) / R2X, NBZ = (TZ / 4) / R2Z, NB = NBX + NBZ;
    static constexpr int NPIX1X = round16((2 * R2X + 1) * (TX / 2 + 1));
    static constexpr int NPIX1Z = round16((2 * R2Z + 1) * (TZ / 2 + 1));
    static constexpr int RING = 2 * (NPIX1X > NPIX1Z ? NPIX1X : NPIX1Z);
    static constexpr int CONST_F4 = 5 * 64 + 4;                  // layer-2 weight images, b2
    static constexpr int LDS_BYTES = (2 * RING + CONST_F4) * 16;
    static_assert(NB % 2 == 0 && NBZ % 2 == 0, "bands alternate between the two groups, crop by crop");
    static_assert((R2X * (TX / 4)) == 256 && (R2Z * (TZ / 4)) == 256, "16 layer-2 tiles per band");
};

template <int TX, int TZ, int ZMODE, bool DIAG, bool U8 = false>      // U8 (ZMODE 1): xin is the uint8 (B, TX, TX, 3) patch, w1g / b1 the folded image (vt_stem.h: L1In)
__global__ __launch_bounds__(1024) void stem_pipe_kernel(
    const float* __restrict__ zin, const float* __restrict__ xin, const float* __restrict__ w1g, const float* __restrict__ b1,
    const float* __restrict__ w2img, const float* __restrict__ b2, float* __restrict__ act_z, float* __restrict__ act_x, int skip_arg,
    unsigned long long* __restrict__ stamps,     // diagnostic (VT_DBG_STAMPS), null in production: [B][16][32]
    const float* __restrict__ w2k) {             // layer-2 weights as [tap][input channels 0-3 | 4-5 + padding][16 output channels][4] (f32 build)
    // ZMODE 0: both crops; 1: search bands only (template cached downstream); 2: template bands only.  DIAG: see stem_fused_kernel.
    using G = PipeGeo<TX, TZ>;
    static_assert(!U8 || ZMODE == 1, "the uint8 patch form is the search-only (cached template) step");
    const int skip = DIAG ? skip_arg : 0;
    constexpr int s_lo = ZMODE == 1 ? G::NBZ / 2 : 0, s_hi = ZMODE == 2 ? G::NBZ / 2 : G::NB / 2;   // band pairs [s_lo, s_hi)
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    f4* const ring0 = reinterpret_cast<f4*>(lds_f);
    f4* const cw2 = ring0 + 2 * G::RING;
    const float* const cb2 = reinterpret_cast<const float*>(cw2 + 5 * 64);

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 3, gw = wave & 7;
    const int q = lane >> 4, px = lane & 15;
    f4* const ring = ring0 + grp * G::RING;
    const f4* const other_ring = ring0 + (1 - grp) * G::RING;
    const int pair = gw * 64 + lane;

    struct Band {
        const float* in; float* out;
        int lgT, HALF, lgHALF, PITCH, npix1, p0, R2, lgW2;
        bool halo, is_z;
    };
    auto band = [&](int j) {       // j-th band of the frame: template bands first
        Band J;
        const bool is_z = j < G::NBZ;
        const int kb = is_z ? j : j - G::NBZ;
        constexpr int lgTX = TX == 256 ? 8 : 7, lgTZ = TZ == 128 ? 7 : 6;
        static_assert((1 << lgTX) == TX && (1 << lgTZ) == TZ, "crop sides");
        J.in = is_z ? zin + (size_t)b * 3 * TZ * TZ : xin + (size_t)b * 3 * TX * TX;
        J.out = is_z ? act_z + (size_t)b * 3 * (TZ / 4) * (TZ / 4) * 4 : act_x + (size_t)b * 3 * (TX / 4) * (TX / 4) * 4;
        J.lgT = is_z ? lgTZ : lgTX; J.HALF = (is_z ? TZ : TX) >> 2; J.lgHALF = J.lgT - 2; J.PITCH = ((is_z ? TZ : TX) >> 1) + 1;
        J.npix1 = is_z ? G::NPIX1Z : G::NPIX1X; J.R2 = is_z ? G::R2Z : G::R2X; J.p0 = kb * J.R2; J.lgW2 = J.lgT - 2;
        J.halo = kb > 0; J.is_z = is_z;
        return J;
    };
    // buffer loads: scalar descriptor + one 32-bit offset register per kernel row pair (see stem_fused_kernel)
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zin + (size_t)b * 3 * TZ * TZ), 0, 3 * TZ * TZ * 4, 0x00020000);
    const auto rsrc_x = U8 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(xin) + (size_t)b * 3 * TX * TX), 0, 3 * TX * TX, 0x00020000)
                           : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (size_t)b * 3 * TX * TX), 0, 3 * TX * TX * 4, 0x00020000);
    auto fetch = [&](const Band& J, L1In<U8>& vin) {           // raw loads only (see stem_fused_kernel)
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const int lr = 1 + (pair >> J.lgHALF), qp = pair & (J.HALF - 1);
        const int p1 = 2 * J.p0 - 1 + lr;
        if constexpr (U8) {     // row y, pixels 4 qp .. 4 qp + 3 of the uint8 patch = bytes 12 (y T / 4 + qp) .. + 11: one load per kernel row
            const unsigned o1 = 12u * ((((unsigned)(2 * p1)) << (J.lgT - 2)) + (unsigned)qp);
            const unsigned o0 = p1 > 0 ? o1 - (3u << J.lgT) : o1;
            vin.v[0] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o0, 0, 0);
            vin.v[1] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o1, 0, 0);
            vin.v[2] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o1 + (3u << J.lgT), 0, 0);
            return;
        } else {
        auto& v = vin.v;
        const unsigned off1 = ((((unsigned)(2 * p1)) << J.lgT) + 4u * (unsigned)qp) << 2;
        const unsigned off0 = p1 > 0 ? off1 - (4u << J.lgT) : off1;
        const unsigned off2 = off1 + (4u << J.lgT);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned vo = r == 0 ? off0 : (r == 1 ? off1 : off2), so = (unsigned)c << (2 * J.lgT + 2);
                const u4 t = J.is_z ? __builtin_amdgcn_raw_buffer_load_b128(rsrc_z, vo, so, 0) : __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, so, 0);
                v[r][c] = __builtin_bit_cast(f4, t);
            }
        }
    };
    auto layer1 = [&](const Band& J, const L1In<U8>& vin) {
        __builtin_amdgcn_s_setprio(3);
        const int lr = 1 + (pair >> J.lgHALF), qp = pair & (J.HALF - 1);
        const float keep0 = (2 * J.p0 - 1 + lr) > 0 ? 1.f : 0.f;
        const int nrow = 2 * J.R2 + 1;
        if (pair < 2 * nrow) {                                  // column -1 of every ring row
            const int plane = pair >= nrow ? 1 : 0;
            ring[plane * J.npix1 + (pair - plane * nrow) * J.PITCH + J.HALF] = splat4(0.f);
        }
        if (pair >= 128 && pair < 128 + 2 * J.PITCH) {          // row 0: the previous band's last row, or the image top
            const int e = pair - 128, plane = e >= J.PITCH ? 1 : 0, col = e - plane * J.PITCH;
            ring[plane * J.npix1 + col] = J.halo ? other_ring[plane * J.npix1 + 2 * J.R2 * J.PITCH + col] : splat4(0.f);
        }
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
            float padv = 0.f;                                 // what a tap outside the crop reads (fp32 form: the zero padding itself)
            if constexpr (U8) {
                padv = b1[W1U_PAD - W1U_BIAS + c];            // 255 mean_c: normalises to zero
                vv = l1_channel(vin.v[r], c);
                if (r == 0 && J.p0 == 0 && keep0 == 0.f) vv = splat4(padv);
            } else {
                vv = (r == 0 && J.p0 == 0) ? vin.v[r][c] * splat4(keep0) : vin.v[r][c];     // only the band at the image top has a padding row
            }
            const float left = lane_left(vv.w);
            const float t0[3] = {qp > 0 ? left : padv, vv.x, vv.y}, t1[3] = {vv.y, vv.z, vv.w};
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    a0[j] = fmaf(t0[s], cur[s * 6 + j], a0[j]);
                    a1[j] = fmaf(t1[s], cur[s * 6 + j], a1[j]);
                }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { a0[j] = hardswish(a0[j]); a1[j] = hardswish(a1[j]); }
        f4* dst = ring + lr * J.PITCH;
        dst[qp] = f4{a0[0], a0[1], a0[2], a0[3]};
        dst[J.npix1 + qp] = f4{a0[4], a0[5], 0.f, 0.f};
        dst[J.HALF + 1 + qp] = f4{a1[0], a1[1], a1[2], a1[3]};
        dst[J.npix1 + J.HALF + 1 + qp] = f4{a1[4], a1[5], 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
    };
    auto layer2 = [&](const Band& J) {
#ifndef VT_F16
        // Layer 2 on v_mfma_f32_16x16x1_4B_f32: four 16x16 blocks per instruction, K = 1.  Its 6 input channels make 54 real
        // k-steps; on the 16x16x4 form (k in quads of 4 channels, chunks of 4 quads) they pad to 80.  Block b = pixel tile 4 gw + b:
        // lane (b, px) SUPPLIES pixel px of that tile as B, every lane supplies W[oc = px][k] as A (the same for the four
        // blocks), and lane (q, px) RECEIVES channels 4q..4q+3 of pixel px of all four tiles (tools/src/probe_mfma4b.hip).
        // Four waves of the group (one per SIMD) cover the band's 16 tiles; the other four go straight to the barrier.
        if (gw < 4) {
            typedef float f16v __attribute__((ext_vector_type(16)));
            const int op = 16 * (4 * gw + q) + px, yy = op >> J.lgW2, xx = op & ((1 << J.lgW2) - 1);
            const f4* src = ring + 2 * yy * J.PITCH + xx;                     // tap (0,0) of this lane's pixel, channel quad 0
            const f4* wk = cw2 + px;                                          // [tap][channels 0-3 | 4-5][16 output channels] float4: 16 lanes read 16 consecutive entries
            const f4 bv2 = ld4(cb2 + 4 * q);
            f16v acc = {bv2.x, bv2.y, bv2.z, bv2.w, bv2.x, bv2.y, bv2.z, bv2.w, bv2.x, bv2.y, bv2.z, bv2.w, bv2.x, bv2.y, bv2.z, bv2.w};
            auto tapoff = [&](int tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                return dy * J.PITCH + (dx == 1 ? 0 : (dx == 0 ? J.HALF : J.HALF + 1));
            };
            f4 a0 = src[tapoff(0)], a1 = src[J.npix1 + tapoff(0)], w0 = wk[0], w1 = wk[16];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                f4 na0 = a0, na1 = a1, nw0 = w0, nw1 = w1;
                if (tap + 1 < 9) {
                    na0 = src[tapoff(tap + 1)]; na1 = src[J.npix1 + tapoff(tap + 1)];
                    nw0 = wk[32 * (tap + 1)]; nw1 = wk[32 * (tap + 1) + 16];
                    __builtin_amdgcn_sched_barrier(0);        // keep the next tap's reads ahead of this tap's MFMAs
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.x, a0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.y, a0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.z, a0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w0.w, a0.w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w1.x, a1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x1f32(w1.y, a1.y, acc, 0, 0, 0);
                a0 = na0; a1 = na1; w0 = nw0; w1 = nw1;
            }
            if (q < 3) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int ob = 16 * (4 * gw + bb) + px, y = ob >> J.lgW2, x = ob & ((1 << J.lgW2) - 1);
                    f4 r = {acc[4 * bb], acc[4 * bb + 1], acc[4 * bb + 2], acc[4 * bb + 3]};
                    r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                    st4(J.out + ((((size_t)q << (2 * J.lgW2)) + (((size_t)J.p0 + y) << J.lgW2) + x) << 2), r);   // quad plane q
                }
            }
        }
#else
        f4 w2a[5][1];
#pragma unroll
        for (int c = 0; c < 5; ++c) w2a[c][0] = cw2[c * 64 + lane];
        const f4 bv2 = ld4(cb2 + 4 * q);
        int base[2], yy[2], xx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int op = 16 * (gw + 8 * i) + px;
            yy[i] = op >> J.lgW2; xx[i] = op & ((1 << J.lgW2) - 1);
            base[i] = 2 * yy[i] * J.PITCH + xx[i];
        }
        f4 acc[2][1] = {{bv2}, {bv2}};
        auto off2 = [&](int c) { return s2_chunk_off<2>(c, q, J.npix1, J.PITCH, J.HALF); };
        vtc::mma_pass<1, 2, 5, 5, true>(ring, base, w2a, 0, off2, acc);
        if (q < 3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f4 r = acc[i][0];
                r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                st4(J.out + ((((size_t)q << (2 * J.lgW2)) + (((size_t)J.p0 + yy[i]) << J.lgW2) + xx[i]) << 2), r);   // quad plane q
            }
        }
#endif
    };

    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (DIAG) {
            if (stamps != nullptr && nstamp < 32) {
                unsigned long long tt;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
                if (lane == 0) stamps[((size_t)b * 16 + wave) * 32 + nstamp] = tt;
                ++nstamp;
            }
        }
    };
    stamp();
    L1In<U8> v;
    fetch(band(2 * s_lo + grp), v);
#ifndef VT_F16
    if (threadIdx.x < 9 * 32) cw2[threadIdx.x] = ld4(w2k + 4 * threadIdx.x);     // [tap][2][16][4] floats
#else
    if (threadIdx.x < 5 * 64) cw2[threadIdx.x] = ld4(w2img + 4 * threadIdx.x);
#endif
    else if (threadIdx.x < 5 * 64) {}
    else if (threadIdx.x < 5 * 64 + 4) cw2[threadIdx.x] = ld4(b2 + 4 * (threadIdx.x - 320));
    // group B works one interval behind group A; both execute NB + 1 barriers
    if (grp == 1) __syncthreads();
    for (int s = s_lo; s < s_hi; ++s) {
        const int j = 2 * s + grp;
        if (!(skip & 1)) layer1(band(j), v);
        stamp();
        __syncthreads();
        stamp();
        fetch(band(j + 2 < 2 * s_hi ? j + 2 : j), v);          // the next band of this group, a whole interval ahead
        if (!(skip & 2)) layer2(band(j));
        stamp();
        __syncthreads();
        stamp();
    }
    if (grp == 0) __syncthreads();
}

}  // namespace vts
