// vt_bf3.h -- fp32 products on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 pieces, x = h + m + l
// (8 mantissa bits each, by truncation: every residual is an exact fp32 subtraction), and a product a b is the six terms
// hh + hm + mh + hl + lh + mm accumulated in fp32 by v_mfma_f32_16x16x32_bf16; what is dropped (ml, lm, ll) is about ONE fp32
// rounding per product -- 2^-24.6 of |a b| on average, below 2^-21 for any operands (tests/test_bf3_arithmetic.py) -- and in a sum it
// disappears in the accumulator's own fp32 rounding (tools/src/probe_bf3.hip, on the hardware: max error / sum |a b| 2.6e-7 against
// 3.0e-7 for v_mfma_f32_16x16x4_f32).  Six 16 x 16 x 32 instructions cover EIGHT times the K of a 16 x 16 x 4 fp32 MFMA in
// 6 x 16 cycles against 8 x 32, and VALU work issues beside them.  Shared by vt_head3.h (towers) and vt_blocks.h (MLP).
//
// Operand convention: a lane's 8 bf16 of a K = 32 instruction are its quad (k = 4 q + 0..3, vt_common.h) of 16-deep chunk 2 p,
// then its quad of chunk 2 p + 1 -- the K order inside an MFMA is free as long as both operands agree -- so the fp32 kernels'
// chunking carries over unchanged; an odd last chunk pairs with zeros.
#pragma once
#include "vt_common.h"

namespace vt3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = h + m + l.  Element r of a result piece = bf16 bits of x[r]'s piece; two elements per dword, low half first.
//
// Round 6, measured and NOT the default (VT_SPLIT_DOT2=1 builds it): 14 instead of 22 vector instructions per four values.  A residual is
// x - trunc_bf16(x): the packed pair [h0 | h1] that the MFMA operand needs anyway is also what v_dot2c_f32_bf16 reads -- D += A.lo B.lo +
// A.hi B.hi with B = [-1 | 0] (or [0 | -1]) and D = x0 (x1) IS the residual, one instruction instead of v_and + v_sub.  It is exact on
// the hardware (tools/src/probe_split.hip: 130,560 values over every exponent, denormals included, 0 pieces differ from the v_and /
// v_sub form) -- but v_dot2c_f32_bf16 issues at about a QUARTER of the plain rate: 111 cycles per split and wave against 89 for the
// 22-instruction form (same probe, four waves per SIMD).  So the split stays 22 plain instructions (NOTES R6-3).
#ifndef VT_SPLIT_DOT2
#define VT_SPLIT_DOT2 0
#endif
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// x0 - lo(pair), x1 - hi(pair): the residuals of the two values whose leading pieces are packed in `pair`
__device__ __forceinline__ void sub_pair(unsigned pair, float x0, float x1, float& r0, float& r1) {
    const bf16x2 a = __builtin_bit_cast(bf16x2, pair);
    // the two selectors live in registers the compiler cannot see through: written as literals, hipcc 7.2 encodes 0x0000bf80 as the
    // INLINE constant -1.0, which the instruction then reads as the fp32 pattern 0xbf800000 = [0 | -1] -- the other element
    unsigned sel_lo = 0x0000bf80u, sel_hi = 0xbf800000u;
    asm volatile("" : "+s"(sel_lo), "+s"(sel_hi));
    r0 = __builtin_amdgcn_fdot2_f32_bf16(a, __builtin_bit_cast(bf16x2, sel_lo), x0, false);      // B = [-1 | 0]
    r1 = __builtin_amdgcn_fdot2_f32_bf16(a, __builtin_bit_cast(bf16x2, sel_hi), x1, false);      // B = [0 | -1]
}
__device__ __forceinline__ void split3(f4 x, u32x2& h, u32x2& m, u32x2& l) {
#ifdef VT_SPLIT_FAKE      // timing builds only (wrong results): the two packs of h, m = l = h -- what the kernels cost WITHOUT the 16 residual instructions
    h = u32x2{__builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u), __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u)};
    m = h; l = h;
    asm volatile("" : "+v"(m), "+v"(l));
    return;
#endif
#if VT_SPLIT_DOT2
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        float r0, r1, s0, s1;
        h[p] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        sub_pair(h[p], x0, x1, r0, r1);
        m[p] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        sub_pair(m[p], r0, r1, s0, s1);
        l[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
#else
    unsigned xb[4], r1b[4], r2b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xb[i] = __float_as_uint(x[i]);
        const float r1 = x[i] - __uint_as_float(xb[i] & 0xffff0000u);
        r1b[i] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(r1b[i] & 0xffff0000u);
        r2b[i] = __float_as_uint(r2);
    }
    h = u32x2{__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
    m = u32x2{__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u)};
    l = u32x2{__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u)};
#endif
}
// the fp32 value back from its pieces (exact: h + m has at most 16 significant bits, + l at most 24)
__device__ __forceinline__ f4 join3(u32x2 h, u32x2 m, u32x2 l) {
    f4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned sh = (i & 1) ? 0u : 16u;
        const float fh = __uint_as_float((h[i >> 1] << sh) & 0xffff0000u), fm = __uint_as_float((m[i >> 1] << sh) & 0xffff0000u),
                    fl = __uint_as_float((l[i >> 1] << sh) & 0xffff0000u);
        v[i] = (fh + fm) + fl;
    }
    return v;
}

__device__ __forceinline__ f4 mma(u32x4 a, u32x4 b, f4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// one 16-deep chunk on its own (an odd last chunk of K): the legacy K = 16 instruction takes the lane's quad as a 64-bit operand and
// costs the matrix pipe what a K = 32 instruction costs (tools/src/probe_bf3.hip), i.e. what the same chunk paired with zeros would --
// without the zero registers: no v_mov to widen an operand, two registers less per operand (round 4).
// RULE: an accumulator chain uses ONE of the two instructions.  A K = 16 MFMA whose SrcC is the result of a K = 32 MFMA issued just
// before it (or the reverse) read a stale accumulator on gfx950 with hipcc 7.2 -- every replay of the G128 block kernel differed
// (tools/race_check.py); with the K = 16 terms on an accumulator of their own, added to the other on the VALU, 40 of 40 replays
// are bit-identical.  The compiler pads same-kind chains and MFMA -> VALU reads correctly; the mixed back-to-back chain it does not.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mma16(u32x2 a, u32x2 b, f4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc, 0, 0, 0);
}
// the three small terms (l h, h l, m m), then the three large ones (m h, h m, h h), of one K = 32 step: pieces [0] = h, [1] = m, [2] = l
__device__ __forceinline__ f4 mma_small(const u32x4 (&a)[3], const u32x4 (&b)[3], f4 acc) {
    acc = mma(a[2], b[0], acc);
    acc = mma(a[0], b[2], acc);
    return mma(a[1], b[1], acc);
}
__device__ __forceinline__ f4 mma_large(const u32x4 (&a)[3], const u32x4 (&b)[3], f4 acc) {
    acc = mma(a[1], b[0], acc);
    acc = mma(a[0], b[1], acc);
    return mma(a[0], b[0], acc);
}

}  // namespace vt3
