// vt_common.h -- shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define VT_WAVE 64

// One 16-deep k-chunk of a 16x16 tile: D += A(16 x 16) B(16 x 16) on operand images.
//
// Operand convention used everywhere ("operand image"): a lane's f4 holds, for its row/column
// index (lane & 15) and quarter q = lane >> 4, the four k-values 4q + r (r = 0..3) of a 16-wide
// k chunk.  Result: D[row = 4q + r][col = lane & 15] in element r.
//
// fp32 build (default): four chained v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains).  MFMA step r
// consumes element r of both operands, so step r's hardware k-slot (lane >> 4) stands for k = 4q + r
// on BOTH sides: the k order is permuted identically for A and B, which leaves the product unchanged.
//
// VT_F16 build (libvittrack_hip_f16.so, BASELINE config 5): the SAME operand image is exactly the
// operand layout of v_mfma_f32_16x16x16_f16 (lane holds A[row = lane & 15][k = 4 (lane >> 4) + j]),
// so each chunk is ONE MFMA on the operands rounded to f16 (RNE), accumulating in f32.  Every
// contraction of every kernel goes through the helpers below, so the whole step switches precision
// with the build flag; LayerNorm, softmax, GELU, Hardswish, sigmoid and the residual stream stay f32.
#ifdef VT_F16
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h4 to_h4(f4 v) { return __builtin_convertvector(v, h4); }
__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(to_h4(a), to_h4(b), acc, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void mfma4_shared_b(const f4 (&a)[N], f4 b, f4 (&acc)[N]) {
    const h4 hb = to_h4(b);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(to_h4(a[n]), hb, acc[n], 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void mfma4_shared_a(f4 a, const f4 (&b)[N], f4 (&acc)[N]) {
    const h4 ha = to_h4(a);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, to_h4(b[n]), acc[n], 0, 0, 0);
}
// Stored operand images (round 4): an operand that is written once and read many times -- weight images, the K / V^T images of the
// block kernel -- is kept in the MFMA's own operand type, so the f16 build converts it ONCE where it is produced (weights: at
// vt_load_weights, on the device, by this same conversion) instead of at every MFMA call; the values are the same (RNE of the same
// fp32 number), the results bit-identical.  The fp32 build stores float4 and `opnd` is f4.
typedef h4 opnd;
__device__ __forceinline__ opnd to_opnd(f4 v) { return to_h4(v); }
__device__ __forceinline__ f4 mfma4(h4 a, f4 b, f4 acc) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, to_h4(b), acc, 0, 0, 0); }
template <int N>
__device__ __forceinline__ void mfma4_shared_b(const h4 (&a)[N], f4 b, f4 (&acc)[N]) {
    const h4 hb = to_h4(b);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[n], hb, acc[n], 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void mfma4_shared_a(f4 a, const h4 (&b)[N], f4 (&acc)[N]) {
    const h4 ha = to_h4(a);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, b[n], acc[n], 0, 0, 0);
}
#define VT_PRECISION_NAME "f16"
constexpr bool VT_IS_F16 = true;
#else
constexpr bool VT_IS_F16 = false;
__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// N independent accumulator chains sharing the B operand (weights / keys as A): the four k-steps
// are issued round-robin over the chains, so consecutive MFMAs never depend on each other
// (v_mfma_f32_16x16x4_f32: 32-cycle issue but 40-cycle dependent-accumulator latency).
template <int N>
__device__ __forceinline__ void mfma4_shared_b(const f4 (&a)[N], f4 b, f4 (&acc)[N]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n][r], b[r], acc[n], 0, 0, 0);
}
// same with a shared A operand (activations as A: the transposed-output form used for V)
template <int N>
__device__ __forceinline__ void mfma4_shared_a(f4 a, const f4 (&b)[N], f4 (&acc)[N]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[n][r], acc[n], 0, 0, 0);
}
typedef f4 opnd;
__device__ __forceinline__ opnd to_opnd(f4 v) { return v; }
#define VT_PRECISION_NAME "f32"
#endif

__device__ __forceinline__ f4 splat4(float v) { return f4{v, v, v, v}; }

// A lane's base address inside an LDS region as ONE opaque VGPR: every access `p[constant index]` through the returned pointer is
// then a ds_read / ds_write with the index in the instruction's 16-bit offset field.  Left to itself hipcc folds the region's own
// offset into the immediate, runs out of the 64 KiB the field covers (the weight staging buffers lie above 30 KiB and are 54 KiB
// long) and builds the addresses one v_add_u32 per access instead (round 4: 177 of the G128 block kernel's ~2000 static VALU
// instructions).  `generic` must point into LDS.
template <typename T>
using lds_cptr = const __attribute__((address_space(3))) T*;
template <typename T>
__device__ __forceinline__ lds_cptr<T> lds_lane_base(const void* generic, unsigned lane_byte_off) {
    unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)generic + lane_byte_off;
    asm volatile("" : "+v"(a));
    return (lds_cptr<T>)(uintptr_t)a;
}

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// streaming store: the line is not kept dirty in this XCD's L2 until the end-of-kernel write-back
__device__ __forceinline__ void st4_nt(float* p, f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); }

// Sum / max over the 4 lanes that share (lane & 15): lanes l, l^16, l^32, l^48.
// gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of one register
// with the even rows (lower half) of another; fed the same value twice they return {v[l & ~16],
// v[l | 16]} (resp. 32): the two xor partners, in VALU latency instead of two ds_bpermute round
// trips through the LDS queue.  vt_selftest_mfma checks this lane map.
__device__ __forceinline__ void xor_pair16(float v, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xor_pair32(float v, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float quad_sum(float v) {
    float a, b;
    xor_pair16(v, a, b);
    v = a + b;
    xor_pair32(v, a, b);
    return a + b;
}
__device__ __forceinline__ float quad_max(float v) {
    float a, b;
    xor_pair16(v, a, b);
    v = fmaxf(a, b);
    xor_pair32(v, a, b);
    return fmaxf(a, b);
}

__device__ __forceinline__ float hsum4(f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float hmax4(f4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }

// nn.GELU() (exact erf form): 0.5 u (1 + erf(u / sqrt 2)) = u Phi(u), written as
//     GELU(u) = max(u, 0) - t * 0.5 erfc(t / sqrt 2),   t = |u|
// (u erf(u / sqrt 2) is even in u), with 0.5 erfc(t / sqrt 2) = 2^(P(t) - 1): P = a degree-6 minimax fit of log2 erfc(t / sqrt 2) on
// [0, 6.5] weighted by the error it causes in GELU (tools/gelu_fit.py; fit error 8e-8).  t is clamped at 6.5, where the subtracted
// term is 2e-10.  Max |error| against the exact function, evaluated in fp32: 5.9e-7 (output rounding of fp32 included; the
// Abramowitz-Stegun 7.1.26 form it replaces: 5.5e-7), below the net's own fp32 noise floor of ~2e-6.
// Cost per TWO elements: 2 v_max (-|u| as source modifiers) + 6 v_pk_fma + 2 v_exp + 2 v_max + 1 v_pk_fma = 13 instructions, one
// transcendental per element -- against 23 with two transcendentals (rcp + exp2) per element before: GELU was ~45 % of the block
// kernel's VALU issue.
__device__ __forceinline__ f2 gelu_pair(f2 u) {
    // nt = -t = max(-|u|, -6.5): negation and absolute value are source modifiers of the v_max; the polynomial is written in nt (odd
    // coefficients change sign) and the last step is fma(nt, e, max(u, 0)) -- no separate negation anywhere
    const f2 nt = {__builtin_fmaxf(-__builtin_fabsf(u.x), -6.5f), __builtin_fmaxf(-__builtin_fabsf(u.y), -6.5f)};
    f2 p = __builtin_elementwise_fma(nt, f2{2.992413958e-05f, 2.992413958e-05f}, f2{7.398738213e-04f, 7.398738213e-04f});
    p = __builtin_elementwise_fma(p, nt, f2{7.977461502e-03f, 7.977461502e-03f});
    p = __builtin_elementwise_fma(p, nt, f2{5.323818492e-02f, 5.323818492e-02f});
    p = __builtin_elementwise_fma(p, nt, f2{-4.589156874e-01f, -4.589156874e-01f});
    p = __builtin_elementwise_fma(p, nt, f2{1.151147082e+00f, 1.151147082e+00f});
    p = __builtin_elementwise_fma(p, nt, f2{-1.0f, -1.0f});
    const f2 e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    const f2 m = {fmaxf(u.x, 0.0f), fmaxf(u.y, 0.0f)};
    return __builtin_elementwise_fma(nt, e, m);
}
__device__ __forceinline__ f4 gelu4(f4 u) {
    const f2 a = gelu_pair(f2{u.x, u.y}), b = gelu_pair(f2{u.z, u.w});
    return f4{a.x, a.y, b.x, b.y};
}
// Hardswish: y * clamp(y + 3, 0, 6) / 6  =  y * clamp(y / 6 + 0.5, 0, 1): the second form is ONE fma with the [0, 1] clamp as
// its output modifier (v_fma_f32 ... clamp, v_pk_fma_f32 ... clamp) and one multiply -- 2 instead of 4-6 instructions (round 3).
__device__ __forceinline__ float hardswish(float y) {
    return y * __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(y, 1.0f / 6.0f, 0.5f), 0.0f), 1.0f);
}
__device__ __forceinline__ float sigmoid_clamped(float v) {
    // torch.clamp(x.sigmoid_(), 1e-4, 1 - 1e-4)   (lib/models/layers/head.py:177-179)
    float y = 1.0f / (1.0f + expf(-v));
    return fminf(fmaxf(y, 1e-4f), 1.0f - 1e-4f);
}

// ---- the tail of Vit_dist.track() for one sequence (lib/test/tracker/vit_dist.py:107-111,150-156; clip_box,
// lib/utils/box_ops.py:97-106).  Shared by update_state_kernel (vt_track.h) and the decode kernel's fused form (vt_head.h).
struct TrackTail {
    const double* resize_factor;   // (B) from crop_kernel
    double* states;                // (B,4) [x,y,w,h], updated in place
    double* record;                // optional (B,5) [x,y,w,h,confidence]: device or device-mapped pinned host memory
    int search_size, H, W, margin;
    int keep;                      // open loop (vt_set_open_loop): the new box goes to `record` only, `states` stay as they are
};

__device__ __forceinline__ void update_state_one(int b, const float (&hann_box)[4], float conf, const TrackTail& t) {
    const double rf = t.resize_factor[b];
    // (pred_boxes.mean(0) * search_size / resize_factor).tolist(): float32 arithmetic, then Python floats
    double p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = (double)((hann_box[k] * (float)t.search_size) / (float)rf);
    // map_box_back (lib/test/tracker/vit_dist.py:150-156)
    const double sx = t.states[4 * b + 0], sy = t.states[4 * b + 1], sw = t.states[4 * b + 2], sh = t.states[4 * b + 3];
    const double cx_prev = sx + 0.5 * sw, cy_prev = sy + 0.5 * sh;
    const double half_side = 0.5 * t.search_size / rf;
    const double cx_real = p[0] + (cx_prev - half_side), cy_real = p[1] + (cy_prev - half_side);
    double bx1 = cx_real - 0.5 * p[2], by1 = cy_real - 0.5 * p[3];
    const double w = p[2], h = p[3];
    // clip_box(box, H, W, margin) (lib/utils/box_ops.py:97-106)
    double bx2 = bx1 + w, by2 = by1 + h;
    bx1 = fmin(fmax(0.0, bx1), (double)(t.W - t.margin));
    bx2 = fmin(fmax((double)t.margin, bx2), (double)t.W);
    by1 = fmin(fmax(0.0, by1), (double)(t.H - t.margin));
    by2 = fmin(fmax((double)t.margin, by2), (double)t.H);
    const double nw = fmax((double)t.margin, bx2 - bx1), nh = fmax((double)t.margin, by2 - by1);
    if (!t.keep) {
        t.states[4 * b + 0] = bx1;
        t.states[4 * b + 1] = by1;
        t.states[4 * b + 2] = nw;
        t.states[4 * b + 3] = nh;
    }
    if (t.record != nullptr) {
        t.record[5 * b + 0] = bx1;
        t.record[5 * b + 1] = by1;
        t.record[5 * b + 2] = nw;
        t.record[5 * b + 3] = nh;
        t.record[5 * b + 4] = (double)conf;
    }
}
