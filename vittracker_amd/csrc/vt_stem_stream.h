// vt_stem_stream.h -- the whole patch embedding of one frame in one workgroup as a STREAMING pipeline over bands, for crops
// whose layer-2 maps do not fit in LDS (G256: 256 / 128 px) -- and for G128 as well.
//
// Same arithmetic as stem_pipe + stem_b (reference: Conv2d_BN / b16 / LevitPatchEmbedding and the pos-embed add + cat of
// OstrackDist.forward, lib/models/vit_dist/vit_dist.py:10-54,78-84).  A band = 512 layer-1 pixel pairs = R2 = 1024 / T layer-2
// rows of a crop of side T; the dependencies close per band: R2 layer-2 rows -> R2 / 2 layer-3 rows -> R2 / 4 token rows, each
// needing ONE row of the band above (the 3 x 3 stride-2 taps reach one row up).  So nothing but token rows leaves the CU: no
// 62 MB layer-2 intermediate (written by stem_pipe, read back by stem_b: 1.56 x the algorithmic HBM traffic), no second launch,
// and layers 3 / 4 run in the issue slots the layer-1 / layer-2 pipeline leaves free instead of after it.
//
// Roles (16 waves, one barrier per interval; interval j):
//     waves 0-7   (two per SIMD)  layer 1 (3 -> 6, VALU) of band j from registers fetched one interval earlier -> L1 ring j & 1;
//                                 the crop rows of band j + 1 are requested first (a second register set)
//     waves 8-11  (one per SIMD)  layer 2 (6 -> 12, v_mfma_f32_16x16x1_4B_f32) of band j - 1 from ring (j - 1) & 1 -> L2 slot (j - 1) % 3
//     waves 12-15 (one per SIMD)  layer 3 (12 -> 24) of band j - 2: L2 slot (j - 2) % 3 -> L3 slot (j - 2) % 3, then
//                                 layer 4 (24 -> 48) + pos-embed of band j - 3: L3 slot j % 3 -> token rows in HBM
// Bands run template first, then search.  A slot holds a band's rows plus, as row 0, a copy of the last row of the band above (the
// writer of that band stores its last row twice) or zeros at the top of a crop; slots are slot-major ([slot][plane][row][col]), so
// the two crops' different widths never alias inside the rings.  Three slots: the writer of band g + 1, the reader of band g and
// the halo copy for band g + 2 touch three different slots in any interval.
#pragma once
#include "vt_common.h"
#include "vt_conv.h"
#include "vt_stem.h"

#ifndef VT_F16
#ifndef VT_SS_PRIO34
#define VT_SS_PRIO34 2
#endif
#ifndef VT_SS_PRIO2
#define VT_SS_PRIO2 0
#endif
#ifndef VT_SS_SKIP
#define VT_SS_SKIP 0        // timing experiments only (wrong results): 1 = no layer 1, 2 = no layer 2, 4 = no layer 3, 8 = no layer 4, 16 = no fetch
#endif
namespace vts {

template <int TX, int TZ>
struct StreamGeo {
    static constexpr int r2(int T) { return 1024 / T; }
    static constexpr int NBX = (TX / 4) / r2(TX), NBZ = (TZ / 4) / r2(TZ), NB = NBX + NBZ;
    static constexpr int npix1(int T) { return round16((2 * r2(T) + 1) * (T / 2 + 1)); }
    static constexpr int np2(int T) { return round16((r2(T) + 1) * (T / 4 + 1)); }          // one plane of an L2 slot
    static constexpr int np3(int T) { return round16((r2(T) / 2 + 1) * (T / 8 + 1)); }      // one plane of an L3 slot
    static constexpr int imax(int a, int b) { return a > b ? a : b; }
    static constexpr int RING = 2 * imax(npix1(TX), npix1(TZ));                              // f4 per L1 ring (2 planes)
    static constexpr int SLOT2 = 3 * imax(np2(TX), np2(TZ));                                 // f4 per L2 slot (3 planes)
    static constexpr int SLOT3 = 6 * imax(np3(TX), np3(TZ));                                 // f4 per L3 slot (6 planes)
    static constexpr int CONST_F4 = 9 * 32 + 4 + 8 + 12;                                     // layer-2 weights for the 4-block MFMA, b2, b3, b4
    // per-chunk map offsets of layers 3 / 4 as tables [x | z][quad q][chunk] (ints), as in stem_fused: s2_chunk_off is ~15 VALU
    // instructions per chunk and lane, 21 chunks per interval on the waves that carry the interval's longest MFMA chains
    static constexpr int OFF3 = 8, OFF4 = 16;                                                // two 16-bit offsets per int: rows of 4 / 8 ints
    static constexpr int OFFTAB_F4 = (2 * 4 * OFF3 + 2 * 4 * OFF4) / 8;                      // 24
    // the LAST k-chunk of layer 4's weights (3 output tiles x 1 KiB) lives in LDS instead of registers: the layer-3/4 waves hold 84
    // weight registers, and with all 14 chunks of layer 4 in registers the kernel needed 132-140 against the 128 a 1024-thread
    // workgroup gets -- one weight quad was spilled and reloaded from scratch inside the interval loop, behind a vmcnt(0) (round 4)
    static constexpr int W4L_F4 = 3 * 64;
    static constexpr int LDS_F4 = 2 * RING + 3 * SLOT2 + 3 * SLOT3 + CONST_F4 + OFFTAB_F4 + W4L_F4;
    static constexpr int LDS_BYTES = LDS_F4 * 16;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(r2(TX) % 4 == 0 && r2(TZ) % 4 == 0, "a band holds whole token rows");
};

// ZMODE 0: both crops; 1: search bands only (the template's token rows are cached in `tokens`); 2: template bands only.
// U8 (round 6; ZMODE 1 only): `xin` is the uint8 (B, TX, TX, 3) patch of vt_crop_u8, w1g / b1 point into the folded layer-1 image
// w1u (vt_stem.h: L1In) -- the layer-1 waves hold 2 x 9 instead of 2 x 36 prefetch registers and fetch a quarter of the bytes.
template <int TX, int TZ, int ZMODE, bool U8 = false>
__global__ __launch_bounds__(1024) void stem_stream_kernel(
    const float* __restrict__ zin, const float* __restrict__ xin,                       // (B,3,TZ,TZ), (B,3,TX,TX)
    const float* __restrict__ w1g, const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ w3img,
    const float* __restrict__ b3, const float* __restrict__ w4img, const float* __restrict__ b4, const float* __restrict__ pos_z,
    const float* __restrict__ pos_x, float* __restrict__ tokens, int L, int len_z,
    const float* __restrict__ w2k) {               // layer-2 weights as [tap][input channels 0-3 | 4-5 + padding][16 output channels][4]
    using G = StreamGeo<TX, TZ>;
    static_assert(!U8 || ZMODE == 1, "the uint8 patch form is the search-only (cached template) step");
    constexpr int g_lo = ZMODE == 1 ? G::NBZ : 0, g_hi = ZMODE == 2 ? G::NBZ : G::NB;   // bands [g_lo, g_hi)
    constexpr int NIV = g_hi - g_lo + 3, NIV2 = (NIV + 1) / 2;                         // intervals; the loops run two per iteration
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    f4* const lds = reinterpret_cast<f4*>(lds_f);
    f4* const ring0 = lds;
    f4* const l2ring = lds + 2 * G::RING;
    f4* const l3ring = l2ring + 3 * G::SLOT2;
    f4* const cw2 = l3ring + 3 * G::SLOT3;                                            // [9][2][16] f4
    const float* const cb2 = reinterpret_cast<const float*>(cw2 + 9 * 32);            // 16 floats
    const float* const cb3 = cb2 + 16;                                                // 32
    const float* const cb4 = cb3 + 32;                                                // 48
    int* const otab = reinterpret_cast<int*>(cw2 + G::CONST_F4);             // [x | z][4][OFF3], then [x | z][4][OFF4]
    f4* const w4l = cw2 + G::CONST_F4 + G::OFFTAB_F4;                        // [3 output tiles][64 lanes]: layer 4's last k-chunk

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Every phase derives its indices and LDS offsets from a FRESH copy of the lane index: left to itself hipcc hoists those
    // loop-invariant per-lane values (one offset per k-chunk and crop layout) out of the interval loops and spills them.
    auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };

    struct Band {
        bool is_z;
        int kb, nb;                                     // band index inside its crop, bands of the crop
        int lgT, HALF, lgHALF, PITCH, npix1, R2, lgW2;  // layer 1 / 2 (as in stem_pipe_kernel)
        int pitch2, half2, np2;                         // L2 slot layout
        int lgW3, R3, pitch3, half3, np3;               // L3 slot layout
        int lgW4;
    };
    auto band = [&](int g) {       // g-th band of the frame: template bands first
        Band J;
        constexpr int lgTX = TX == 256 ? 8 : 7, lgTZ = TZ == 128 ? 7 : 6;
        static_assert((1 << lgTX) == TX && (1 << lgTZ) == TZ, "crop sides");
        J.is_z = g < G::NBZ;
        J.kb = J.is_z ? g : g - G::NBZ;
        J.nb = J.is_z ? G::NBZ : G::NBX;
        const int T = J.is_z ? TZ : TX;
        J.lgT = J.is_z ? lgTZ : lgTX; J.HALF = T >> 2; J.lgHALF = J.lgT - 2; J.PITCH = (T >> 1) + 1;
        J.npix1 = J.is_z ? G::npix1(TZ) : G::npix1(TX); J.R2 = J.is_z ? G::r2(TZ) : G::r2(TX); J.lgW2 = J.lgT - 2;
        J.pitch2 = (T >> 2) + 1; J.half2 = T >> 3; J.np2 = J.is_z ? G::np2(TZ) : G::np2(TX);
        J.lgW3 = J.lgT - 3; J.R3 = J.R2 >> 1; J.pitch3 = (T >> 3) + 1; J.half3 = T >> 4; J.np3 = J.is_z ? G::np3(TZ) : G::np3(TX);
        J.lgW4 = J.lgT - 4;
        return J;
    };

    // ---- constants -> LDS (before the first barrier; first read in interval g_lo + 1) --------------------------------------
    {
        const int t = threadIdx.x;
        if (t < 9 * 32) cw2[t] = ld4(w2k + 4 * t);
        else if (t < 9 * 32 + 4) cw2[t] = ld4(b2 + 4 * (t - 288));
        else if (t < 9 * 32 + 12) cw2[t] = ld4(b3 + 4 * (t - 292));
        else if (t < 9 * 32 + 24) cw2[t] = ld4(b4 + 4 * (t - 300));
        else if (t >= 320 && t < 320 + 4 * G::OFFTAB_F4) {
            const int e0 = t - 320;                                 // one int = chunks 2 j, 2 j + 1 of a table row
            const bool l4 = e0 >= 4 * G::OFF3;
            const int e = l4 ? e0 - 4 * G::OFF3 : e0, per = l4 ? 2 * G::OFF4 : 2 * G::OFF3;      // ints per crop
            const bool isz = e >= per;
            const int r = isz ? e - per : e, rowi = l4 ? G::OFF4 / 2 : G::OFF3 / 2, qq = r / rowi, c = 2 * (r - qq * rowi);
            const int T = isz ? TZ : TX;
            int v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
                v[h] = l4 ? s2_chunk_off<6>(c + h, qq, isz ? G::np3(TZ) : G::np3(TX), (T >> 3) + 1, T >> 4)
                          : s2_chunk_off<3>(c + h, qq, isz ? G::np2(TZ) : G::np2(TX), (T >> 2) + 1, T >> 3);
            otab[e0] = v[0] | (v[1] << 16);
        }
    }

    if (wave < 8) {
        // =============================================================== layer 1: waves 0-7 =====================================
        const int pair_ = wave * 64 + lane;                      // this thread's pixel pair of a band (0..511)
        const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zin + (size_t)b * 3 * TZ * TZ), 0, 3 * TZ * TZ * 4, 0x00020000);
        const auto rsrc_x = U8 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(xin) + (size_t)b * 3 * TX * TX), 0, 3 * TX * TX, 0x00020000)
                               : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (size_t)b * 3 * TX * TX), 0, 3 * TX * TX * 4, 0x00020000);
        auto fetch = [&](const Band& J, L1In<U8>& vin) {           // raw loads only: nothing here depends on the loaded data
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const int pair = fresh(pair_);
            const int lr = 1 + (pair >> J.lgHALF), qp = pair & (J.HALF - 1);
            const int p1 = 2 * J.kb * J.R2 - 1 + lr;               // layer-1 row (>= 0)
            if constexpr (U8) {     // row y, pixels 4 qp .. 4 qp + 3 of the uint8 patch = bytes 12 (y T / 4 + qp) .. + 11: one load per kernel row
                const unsigned o1 = 12u * ((((unsigned)(2 * p1)) << (J.lgT - 2)) + (unsigned)qp);
                const unsigned o0 = p1 > 0 ? o1 - (3u << J.lgT) : o1;                              // the image top reads row 0 (replaced in layer1)
                vin.v[0] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o0, 0, 0);
                vin.v[1] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o1, 0, 0);
                vin.v[2] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_x, o1 + (3u << J.lgT), 0, 0);
                return;
            } else {
            auto& v = vin.v;
            const unsigned off1 = ((((unsigned)(2 * p1)) << J.lgT) + 4u * (unsigned)qp) << 2;
            const unsigned off0 = p1 > 0 ? off1 - (4u << J.lgT) : off1;   // the image top reads row 0 (zeroed in layer1)
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
        auto layer1 = [&](const Band& J, int g, const L1In<U8>& vin) {
            f4* const ring = ring0 + (g & 1) * G::RING;
            const f4* const other_ring = ring0 + ((g & 1) ^ 1) * G::RING;
            const int pair = fresh(pair_);
            const int lr = 1 + (pair >> J.lgHALF), qp = pair & (J.HALF - 1);
            const float keep0 = (2 * J.kb * J.R2 - 1 + lr) > 0 ? 1.f : 0.f;   // kernel row 0 of layer-1 row 0 is the zero padding
            const int nrow = 2 * J.R2 + 1;
            if (pair < 2 * nrow) {                                  // column -1 of every ring row
                const int plane = pair >= nrow ? 1 : 0;
                ring[plane * J.npix1 + (pair - plane * nrow) * J.PITCH + J.HALF] = splat4(0.f);
            }
            if (pair >= 128 && pair < 128 + 2 * J.PITCH) {          // row 0: the previous band's last row, or the image top
                const int e = pair - 128, plane = e >= J.PITCH ? 1 : 0, col = e - plane * J.PITCH;
                ring[plane * J.npix1 + col] = J.kb > 0 ? other_ring[plane * J.npix1 + 2 * J.R2 * J.PITCH + col] : splat4(0.f);
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
                float padv = 0.f;                                   // what a tap outside the crop reads (fp32 form: the zero padding itself)
                if constexpr (U8) {
                    padv = b1[W1U_PAD - W1U_BIAS + c];              // 255 mean_c: normalises to zero
                    vv = l1_channel(vin.v[r], c);
                    if (r == 0 && J.kb == 0 && keep0 == 0.f) vv = splat4(padv);
                } else {
                    vv = (r == 0 && J.kb == 0) ? vin.v[r][c] * splat4(keep0) : vin.v[r][c];      // only the band at the top of a crop has a padding row
                }
                const float left = lane_left(vv.w);                 // a wave starts at a row start: lane 0 has qp = 0
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
        };
        L1In<U8> va, vb;
        // (stem_fused's wave-staggered first fetch, NOTES R6-4, measured here too: 88.0 against 88.2-88.3 us -- the start-up is 2 % of this kernel)
        fetch(band(g_lo), va);
        __syncthreads();                 // constants in LDS (the other roles' first reads) -- every role executes this barrier
#pragma unroll 1
        for (int it = 0; it < NIV2; ++it) {
            const int j = g_lo + 2 * it;
            if (j + 1 < g_hi && !(VT_SS_SKIP & 16)) fetch(band(j + 1), vb);
            if (j < g_hi && !(VT_SS_SKIP & 1)) layer1(band(j), j, va);
            __syncthreads();
            if (j + 2 < g_hi && !(VT_SS_SKIP & 16)) fetch(band(j + 2), va);
            if (j + 1 < g_hi && !(VT_SS_SKIP & 1)) layer1(band(j + 1), j + 1, vb);
            __syncthreads();
        }
    } else if (wave < 12) {
        // =============================================================== layer 2: waves 8-11 ====================================
        const int gw = wave - 8;
        typedef float f16v __attribute__((ext_vector_type(16)));
        auto layer2 = [&](const Band& J, int g) {
            const int ln = fresh(lane), q = ln >> 4, px = ln & 15, tid2 = gw * 64 + ln;
            const f4* const ring = ring0 + (g & 1) * G::RING;
            f4* const slot = l2ring + (g % 3) * G::SLOT2;
            f4* const nslot = l2ring + ((g + 1) % 3) * G::SLOT2;
            // housekeeping: column -1 of the halo row and of this band's rows; the halo row itself at the top of a crop
            if (tid2 < 3 * (J.R2 + 1)) {
                const int plane = tid2 / (J.R2 + 1), row = tid2 - plane * (J.R2 + 1);
                slot[plane * J.np2 + row * J.pitch2 + J.half2] = splat4(0.f);
            }
            if (J.kb == 0)
                for (int e = tid2; e < 3 * J.pitch2; e += 256) {
                    const int plane = e / J.pitch2, col = e - plane * J.pitch2;
                    slot[plane * J.np2 + col] = splat4(0.f);
                }
            const int op = 16 * (4 * gw + q) + px, yy = op >> J.lgW2, xx = op & ((1 << J.lgW2) - 1);
            const f4* src = ring + 2 * yy * J.PITCH + xx;                     // tap (0,0) of this lane's pixel, channel quad 0
            const f4* wk = cw2 + px;
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
                const bool copy_down = J.kb + 1 < J.nb;       // the band below is of the same crop: it needs this band's last row
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int ob = 16 * (4 * gw + bb) + px, y = ob >> J.lgW2, x = ob & ((1 << J.lgW2) - 1);
                    f4 r = {acc[4 * bb], acc[4 * bb + 1], acc[4 * bb + 2], acc[4 * bb + 3]};
                    r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                    const int col = (x & 1) ? J.half2 + 1 + (x >> 1) : (x >> 1);
                    slot[q * J.np2 + (y + 1) * J.pitch2 + col] = r;
                    if (copy_down && y == J.R2 - 1) nslot[q * J.np2 + col] = r;
                }
            }
        };
        if (VT_SS_PRIO2 > 0) __builtin_amdgcn_s_setprio(VT_SS_PRIO2);
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < NIV2; ++it) {
            const int j = g_lo + 2 * it;
            if (j - 1 >= g_lo && j - 1 < g_hi && !(VT_SS_SKIP & 2)) layer2(band(j - 1), j - 1);
            __syncthreads();
            if (j >= g_lo && j < g_hi && !(VT_SS_SKIP & 2)) layer2(band(j), j);
            __syncthreads();
        }
    } else {
        // ====================================================== layers 3 and 4: waves 12-15 ====================================
        const int w4 = wave - 12;
        constexpr int NCH3 = 7, NCH4 = 14;
        const int ot3 = w4 & 1;
        constexpr int NR4 = NCH4 - 1;                  // layer-4 chunks held in registers; the last one is read from LDS (StreamGeo)
        f4 w3a[NCH3][1], w4a[NR4][1];
        vtc::load_weights<1, NCH3, NCH3>(w3img + (size_t)ot3 * NCH3 * 256, 0, NCH3, lane, w3a);
        if (w4 < 3) {
            vtc::load_weights<1, NR4, NCH4>(w4img + (size_t)w4 * NCH4 * 256, 0, NR4, lane, w4a);
            w4l[w4 * 64 + lane] = ld4(w4img + ((size_t)w4 * NCH4 + NR4) * 256 + lane * 4);      // read back by this wave only
        }
        auto layer3 = [&](const Band& J, int g) {
            const int ln = fresh(lane), q = ln >> 4, px = ln & 15, tid4 = w4 * 64 + ln;
            const f4* const in = l2ring + (g % 3) * G::SLOT2;
            f4* const slot = l3ring + (g % 3) * G::SLOT3;
            f4* const nslot = l3ring + ((g + 1) % 3) * G::SLOT3;
            if (tid4 < 6 * (J.R3 + 1)) {                          // column -1 of the halo row and of this band's rows
                const int plane = tid4 / (J.R3 + 1), row = tid4 - plane * (J.R3 + 1);
                slot[plane * J.np3 + row * J.pitch3 + J.half3] = splat4(0.f);
            }
            if (J.kb == 0)                                         // top of a crop: the halo row is the zero padding
                for (int e = tid4; e < 6 * J.pitch3; e += 256) {
                    const int plane = e / J.pitch3, col = e - plane * J.pitch3;
                    slot[plane * J.np3 + col] = splat4(0.f);
                }
            int base[2], yy[2], xx[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int op = 16 * ((w4 >> 1) + 2 * i) + px;     // the band's 64 layer-3 pixels = 4 tiles; this wave: tiles w4 >> 1 and + 2
                yy[i] = op >> J.lgW3; xx[i] = op & ((1 << J.lgW3) - 1);
                base[i] = 2 * yy[i] * J.pitch2 + xx[i];
            }
            const f4 bv3 = ld4(cb3 + 16 * ot3 + 4 * q);
            f4 acc[2][1] = {{bv3}, {bv3}};
            const int4 o3 = *reinterpret_cast<const int4*>(otab + (J.is_z ? 2 * G::OFF3 : 0) + q * (G::OFF3 / 2));
            auto off3 = [&](int c) { const int w = (c >> 1) == 0 ? o3.x : (c >> 1) == 1 ? o3.y : (c >> 1) == 2 ? o3.z : o3.w; return (c & 1) ? (int)((unsigned)w >> 16) : (w & 0xffff); };
#ifndef VT_SS_PIN3
#define VT_SS_PIN3 false
#endif
            vtc::mma_pass<1, 2, NCH3, NCH3, VT_SS_PIN3>(in, base, w3a, 0, off3, acc);
            if (16 * ot3 + 4 * q < 24) {
                const bool copy_down = J.kb + 1 < J.nb;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f4 r = acc[i][0];
                    r.x = hardswish(r.x); r.y = hardswish(r.y); r.z = hardswish(r.z); r.w = hardswish(r.w);
                    const int col = (xx[i] & 1) ? J.half3 + 1 + (xx[i] >> 1) : (xx[i] >> 1);
                    slot[(4 * ot3 + q) * J.np3 + (yy[i] + 1) * J.pitch3 + col] = r;
                    if (copy_down && yy[i] == J.R3 - 1) nslot[(4 * ot3 + q) * J.np3 + col] = r;
                }
            }
        };
        auto layer4 = [&](const Band& J, int g) {
            if (w4 >= 3) return;
            const int ln = fresh(lane), q = ln >> 4, px = ln & 15;
            const f4* const in = l3ring + (g % 3) * G::SLOT3;
            const int y = px >> J.lgW4, x = px & ((1 << J.lgW4) - 1);          // the band's 16 tokens = one tile
            const int base = 2 * y * J.pitch3 + x;
            const int tk = 16 * J.kb + px;                                      // token index inside this crop
            const int4* const tp4 = reinterpret_cast<const int4*>(otab + 4 * G::OFF3 + (J.is_z ? 2 * G::OFF4 : 0) + q * (G::OFF4 / 2));
            const int4 o4a = tp4[0], o4b = tp4[1];
            auto at = [&](int c) {
                const int4& o = c < 8 ? o4a : o4b;
                const int j = (c & 7) >> 1, w = j == 0 ? o.x : j == 1 ? o.y : j == 2 ? o.z : o.w;
                return base + ((c & 1) ? (int)((unsigned)w >> 16) : (w & 0xffff));
            };
            // One pixel tile x one output tile: a single accumulator would be a chain of 56 dependent MFMAs (40 cycles each instead
            // of 32, and nothing of this wave to fill the gaps).  Even and odd k-chunks go to two accumulators, added at the end;
            // the B operands are read two chunks ahead.
            static_assert(NCH4 % 2 == 0, "chunk pairs");
            f4 acc0 = ld4(cb4 + 16 * w4 + 4 * q), acc1 = splat4(0.f);
            f4 b0 = in[at(0)], b1 = in[at(1)];
            f4 pe = acc1;
#pragma unroll
            for (int k = 0; k < NCH4; k += 2) {
                if (k + 2 < NCH4) {
                    const f4 n0 = in[at(k + 2)], n1 = in[at(k + 3)];
                    // the pos-embed row is requested three pairs (~770 cycles of MFMAs) before it is added, not at the top: with it, the
                    // last chunk's weights and two operand pairs live together this role needed more than its 128 registers
                    if (k + 6 == NCH4) pe = ld4((J.is_z ? pos_z : pos_x) + (size_t)tk * 48 + 16 * w4 + 4 * q);
                    __builtin_amdgcn_sched_barrier(0);        // keep the next pair's reads ahead of this pair's MFMAs
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4a[k][0][r], b0[r], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4a[k + 1 < NR4 ? k + 1 : 0][0][r], b1[r], acc1, 0, 0, 0);
                    }
                    b0 = n0; b1 = n1;
                } else {
                    // last pair: chunk NCH4 - 1's weights come from LDS (StreamGeo), requested here -- no operand prefetch is live any
                    // more -- and covered by the even chunk's four MFMAs
                    const f4 wlast = w4l[w4 * 64 + ln];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4a[k][0][r], b0[r], acc0, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wlast[r], b1[r], acc1, 0, 0, 0);
                }
            }
            st4(tokens + ((size_t)b * L + (J.is_z ? 0 : len_z) + tk) * 48 + 16 * w4 + 4 * q, (acc0 + acc1) + pe);
        };
        // these four waves carry the longest dependent chains of an interval and are the youngest of the workgroup: without
        // priority the issue arbiter serves the layer-1 / layer-2 waves first and the interval ends with this role running alone
        __builtin_amdgcn_s_setprio(VT_SS_PRIO34);
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < NIV2; ++it) {
            const int j = g_lo + 2 * it;
            if (j - 2 >= g_lo && j - 2 < g_hi && !(VT_SS_SKIP & 4)) layer3(band(j - 2), j - 2);
            if (j - 3 >= g_lo && j - 3 < g_hi && !(VT_SS_SKIP & 8)) layer4(band(j - 3), j - 3);
            __syncthreads();
            if (j - 1 >= g_lo && j - 1 < g_hi && !(VT_SS_SKIP & 4)) layer3(band(j - 1), j - 1);
            if (j - 2 >= g_lo && j - 2 < g_hi && !(VT_SS_SKIP & 8)) layer4(band(j - 2), j - 2);
            __syncthreads();
        }
    }
}

}  // namespace vts
#endif  // !VT_F16
