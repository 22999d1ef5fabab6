// vt_track.h -- the steps either side of the network in Vit_dist.track(), on the device.
//
// crop_kernel replaces, per sequence, sample_target (lib/train/data/processing_utils.py:12-79:
// square crop of side ceil(sqrt(w*h)*factor) around the previous box, zero padding, cv.resize to
// T x T) followed by Preprocessor.process (lib/test/tracker/data_utils.py:11-17: /255, -mean, /std,
// HWC -> NCHW).  update_state_kernel replaces the tail of track() (lib/test/tracker/vit_dist.py:
// 107-111,150-156 and clip_box, lib/utils/box_ops.py:97-106).
//
// Numerics follow the reference's host code: box / crop geometry in double (Python floats),
// round-half-even for the crop origin (Python round()), OpenCV's INTER_LINEAR uint8 path in 11-bit
// fixed point (weights = round(w * 2048), horizontal pass in int, vertical pass
// (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), float32 normalisation in the reference's
// operation order (as torch executes it on a GPU).  Integer stages are bit-exact against the host port in vittracker_amd/host_ops.py
// (which is itself unpinned against cv2: SURVEY.md 8(f) rank 1).
#pragma once
#include "vt_common.h"

namespace vtt {

struct CropGeom {     // per sequence, written by crop_kernel's first lane for update_state_kernel
    double resize_factor;   // T / crop_sz
};

// Source index and 11-bit weights of output coordinate d (OpenCV resize, linear, pixel centres).
__device__ __forceinline__ void lin_coeff(int d, int src, double scale, int& s0, int& s1, int& a0, int& a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
    a1 = (int)rintf(f * 2048.f);
    a0 = (int)rintf((1.f - f) * 2048.f);
    s0 = s;
    s1 = s + 1 < src ? s + 1 : src - 1;
}

// grid (ceil(T * ceil(T/4) / 256), B); frames (B,H,W,3) uint8; states (B,4) double [x,y,w,h]; out (B,3,T,T) float.
// One thread = four consecutive output pixels of a row (all three channels): the vertical coefficients are computed once,
// and a channel's four values leave as ONE 16-byte store when T is a multiple of 4 (the crop sizes the tracker uses are:
// 64 / 128 / 256), i.e. whole 256-byte row segments per quarter-wave instead of 4-byte stores.
// BYTES = true: the same kernel with every 8-byte window assembled from eight single-byte loads -- the form that needs nothing of the
// device's unaligned-access mode; vt_create's self test (vittrack.hip: crop_selftest) selects it when the fast form's result differs.
// U8OUT (round 6): `out` is the uint8 (B, T, T, 3) patch itself -- sample_target's return value, before Preprocessor.process -- which
// the stems' uint8 forms consume (vt_stem.h: L1In); no normalisation table, a thread's 12 values leave as one 12-byte store.
template <bool BYTES = false, bool U8OUT = false>
__global__ __launch_bounds__(256) void crop_kernel(const unsigned char* __restrict__ frames, int H, int W,
                                                   const double* __restrict__ states, double factor, int T,
                                                   float m0, float m1, float m2, float s0, float s1, float s2,
                                                   float* __restrict__ out, double* __restrict__ resize_factor) {
    const int b = blockIdx.y;
    // Preprocessor.process maps a uint8 value to (v / 255 - mean) / std: 256 x 3 possible results.  They are computed ONCE per
    // workgroup with the reference's arithmetic (three separately rounded fp32 ops, below) into an LDS table -- per output value
    // one LDS read instead of a convert, a multiply, a subtract and an IEEE division sequence (~14 VALU instructions of the ~74 a
    // value cost).
    __shared__ float norm_lut[U8OUT ? 1 : 3 * 256];
    if constexpr (!U8OUT) {
        const float meanv[3] = {m0, m1, m2}, stdq[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // torch's CUDA `tensor / 255.0` multiplies by the float reciprocal (div_true with a CPU scalar);
            // Preprocessor.process runs on the GPU, so that is the reference arithmetic
            // Three separately rounded ops, as three torch kernels: the empty asm keeps hipcc from
            // contracting the multiply and the subtraction into one fma (the _rn intrinsics do not).
            float scaled = (float)(int)threadIdx.x * (1.0f / 255.0f);
            asm volatile("" : "+v"(scaled));
            float centred = scaled - meanv[c];
            asm volatile("" : "+v"(centred));
            norm_lut[c * 256 + threadIdx.x] = centred / stdq[c];
        }
        __syncthreads();
    }
    unsigned char* const out8 = reinterpret_cast<unsigned char*>(out) + (size_t)b * T * T * 3;      // U8OUT: this frame's patch
    const double bx = states[4 * b + 0], by = states[4 * b + 1], bw = states[4 * b + 2], bh = states[4 * b + 3];
    const int crop_sz = (int)ceil(sqrt(bw * bh) * factor);
    const int T4 = (T + 3) >> 2;                      // pixel groups per row
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (!(crop_sz >= 1)) {
        // The reference raises 'Too small bounding box.' here (processing_utils.py:33-34).  A kernel cannot
        // raise: the crop and its resize factor are poisoned with NaN, so every box derived from them is NaN and
        // the caller sees it (BatchedVitTracker checks user-supplied boxes on the host before they get here; boxes
        // produced by vt_update_state are at least `margin` wide and never take this branch).
        if (idx == 0) resize_factor[b] = __builtin_nan("");
        if (idx < T * T4) {
            const int oy = idx / T4, ox0 = (idx - oy * T4) * 4;
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 4 && ox0 + k < T; ++k) {
                    if constexpr (U8OUT) out8[((size_t)oy * T + ox0 + k) * 3 + c] = 0;      // bytes cannot carry the poison: the NaN resize factor does
                    else out[(((size_t)b * 3 + c) * T + oy) * T + ox0 + k] = __builtin_nanf("");
                }
        }
        return;
    }
    const int x1 = (int)rint(bx + 0.5 * bw - crop_sz * 0.5);     // Python round(): half to even
    const int y1 = (int)rint(by + 0.5 * bh - crop_sz * 0.5);
    const int x2 = x1 + crop_sz, y2 = y1 + crop_sz;
    // valid source range of the padded crop (the reference's pad formula keeps max(x2 - W + 1, 0)
    // columns on the right, i.e. drops the last image column when the crop reaches the border)
    const int vx0 = x1 < 0 ? 0 : x1, vx1 = x2 - (x2 - W + 1 > 0 ? x2 - W + 1 : 0);
    const int vy0 = y1 < 0 ? 0 : y1, vy1 = y2 - (y2 - H + 1 > 0 ? y2 - H + 1 : 0);
    if (idx == 0) resize_factor[b] = (double)T / (double)crop_sz;
    if (idx >= T * T4) return;
    const int oy = idx / T4, ox0 = (idx - oy * T4) * 4;
    const double scale = (double)crop_sz / (double)T;
    int sy0, sy1, by0, by1;
    lin_coeff(oy, crop_sz, scale, sy0, sy1, by0, by1);
    // Source pixels: an RGB pixel is 3 consecutive bytes, and the two columns a bilinear sample reads are neighbours (or the same
    // pixel at the crop's edge), so ONE 8-byte load at byte offset 3 x covers both -- 8 loads per thread instead of 48 single-byte
    // loads, which were the kernel's cost (3072 vector-memory instructions per 128 x 128 crop: 33 us at batch 256, a quarter of the
    // tracker step).  Buffer loads at byte-unaligned offsets (tools/src/probe_unaligned.hip: the hardware returns the right bytes;
    // a load that crosses the end of the buffer returns zeros, so the frame's last pixels are read 8 bytes back and shifted).
    const size_t frame_bytes = (size_t)H * W * 3, rest = (size_t)(gridDim.y - b) * frame_bytes;      // bytes from this frame to the end of the batch
    const unsigned nrec = rest > 0xfffffff0ull ? 0xfffffff0u : (unsigned)rest;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(frames + (size_t)b * frame_bytes), 0, (int)nrec, 0x00020000);
    const int yy0 = y1 + sy0, yy1 = y1 + sy1;
    const bool vr0 = yy0 >= vy0 && yy0 < vy1, vr1 = yy1 >= vy0 && yy1 < vy1;
    const unsigned rowo0 = (unsigned)(vr0 ? yy0 : 0) * (unsigned)(W * 3), rowo1 = (unsigned)(vr1 ? yy1 : 0) * (unsigned)(W * 3);
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    auto load8 = [&](unsigned off) -> unsigned long long {      // bytes off .. off + 7 of the frame (the last bytes of the batch: shifted in)
        if constexpr (BYTES) {
            unsigned long long r = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)       // out-of-range bytes read as zero (buffer bounds), as in the fast form after its shift
                r |= (unsigned long long)(__builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)(off + j), 0, 0) & 0xffu) << (8 * j);
            return r;
        }
        const unsigned over = off + 8u > nrec ? off + 8u - nrec : 0u;
        const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(off - over), 0, 0);
        return (((unsigned long long)v.y << 32) | v.x) >> (8u * over);
    };
    float res[3][4];
    unsigned pk[3] = {0u, 0u, 0u};      // U8OUT: the 12 bytes of this thread's four pixels, HWC
    unsigned long long q0[4], q1[4];
    int ax0a[4], ax1a[4], sh1[4];
    bool vc0a[4], vc1a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {       // all eight loads first
        const int ox = ox0 + k < T ? ox0 + k : T - 1;
        int sx0, sx1;
        lin_coeff(ox, crop_sz, scale, sx0, sx1, ax0a[k], ax1a[k]);
        const int xx0 = x1 + sx0, xx1 = x1 + sx1;
        vc0a[k] = xx0 >= vx0 && xx0 < vx1; vc1a[k] = xx1 >= vx0 && xx1 < vx1;
        // base pixel of the 8-byte window: the left column when it is inside the frame, else the right one (then the left is padding)
        const int xb = vc0a[k] ? xx0 : (vc1a[k] ? xx1 : 0);
        sh1[k] = vc1a[k] ? 24 * (xx1 - xb) : 0;                     // bit offset of the right column's pixel inside the window: 0 or 24
        q0[k] = load8(rowo0 + 3u * (unsigned)xb);
        q1[k] = load8(rowo1 + 3u * (unsigned)xb);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ax0 = ax0a[k], ax1 = ax1a[k];
        const bool vc0 = vc0a[k], vc1 = vc1a[k];
        // the left column is at bit 0 of the window when it is valid (it is the base); the right one at sh1 (0 when it is the base itself)
        // pixel (cy, cx) of the zero-padded crop: the frame inside the valid range, 0 outside (masked once per pixel, all channels)
        const unsigned l0 = vr0 && vc0 ? (unsigned)q0[k] : 0u, l1 = vr1 && vc0 ? (unsigned)q1[k] : 0u;
        const unsigned r0w = vr0 && vc1 ? (unsigned)(q0[k] >> sh1[k]) : 0u, r1w = vr1 && vc1 ? (unsigned)(q1[k] >> sh1[k]) : 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p00 = (int)((l0 >> (8 * c)) & 0xffu), p01 = (int)((r0w >> (8 * c)) & 0xffu);
            const int p10 = (int)((l1 >> (8 * c)) & 0xffu), p11 = (int)((r1w >> (8 * c)) & 0xffu);
            // every factor is below 2^24 (8-bit pixels, 12-bit weights, 15-bit row sums): the 24-bit multiplier gives the same integers
            const int r0 = __mul24(p00, ax0) + __mul24(p01, ax1);
            const int r1 = __mul24(p10, ax0) + __mul24(p11, ax1);
            int v = ((__mul24(by0, r0 >> 4) >> 16) + (__mul24(by1, r1 >> 4) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            if constexpr (U8OUT) pk[(3 * k + c) >> 2] |= (unsigned)v << (8 * ((3 * k + c) & 3));
            else res[c][k] = norm_lut[c * 256 + v];
        }
    }
    if constexpr (U8OUT) {
        unsigned char* o = out8 + ((size_t)oy * T + ox0) * 3;
        if ((T & 3) == 0) {
            typedef unsigned u3a __attribute__((ext_vector_type(3), aligned(4)));
            *reinterpret_cast<u3a*>(o) = u3a{pk[0], pk[1], pk[2]};
        } else {
            for (int i = 0; i < 12 && ox0 + i / 3 < T; ++i) o[i] = (unsigned char)(pk[i >> 2] >> (8 * (i & 3)));
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float* o = out + (((size_t)b * 3 + c) * T + oy) * T + ox0;
        if ((T & 3) == 0) {
            st4(o, f4{res[c][0], res[c][1], res[c][2], res[c][3]});
        } else {
            for (int k = 0; k < 4 && ox0 + k < T; ++k) o[k] = res[c][k];
        }
    }
}

// The crop as the tracker's step runs it (T a multiple of 4, T <= CROP_FAST_MAX_T): same arithmetic, same results bit for bit as
// crop_kernel (which stays: any T, the byte-load form, the reference of the device self test).  What differs:
//   - a workgroup computes the T column entries of its frame ONCE into LDS (byte offset of the 8-byte window, the two 11-bit weights
//     packed for v_dot2, the right column's bit offset) and walks G groups of 256 items (an item = four consecutive pixels of a row):
//     the column table, the normalisation table and the fp64 box geometry are paid once per G x 12 output values of a thread
//   - the walk is software-pipelined (group g + 1's eight window loads are issued before group g's arithmetic).  MEASURED (256 sequences,
//     rocprofv3 of tracking/track_batch_demo.py, us per launch at G128 / G256; crop_kernel: 25.1 / 111.3): G = 1: 22.3 / 106.3,
//     G = 2: 22.6 / 129.0, G = 4: 26.1 / 126.5 -- one item per thread stays the best shape, so G = 1 is what launch_crop uses;
//     windows fetched as three ALIGNED dwords + funnel shift (half the address-path cycles by tools/src/probe_gather.hip): 26.2 / 111.5
//     at G = 1, i.e. slower, and removed again.  The kernel is bound by neither its instruction count (a third of crop_kernel's in
//     the arithmetic) nor the gather's address path alone; NOTES R5-8.
//   - zero padding lives in the WEIGHTS: a column / row of the padded crop outside the frame gets weight 0, so the arithmetic carries
//     no validity masks
//   - the horizontal pass of a (row, channel) is v_perm_b32 (the channel's byte of the left and the right pixel into the two halves
//     of a dword) + v_dot2_u32_u16 with the packed weights; the vertical pass's (b (r >> 4)) >> 16 is one v_mul_hi_u32_u24
//   - a load that could cross the end of the buffer (the last rows of the last frame) shifts its window as crop_kernel does, on a
//     slow path a whole wave takes or skips
constexpr int CROP_FAST_MAX_T = 512;
#ifndef VT_CROPF_DBG
#define VT_CROPF_DBG 0      // timing builds only (wrong results): 1 = no frame loads, 2 = no stores, 4 = every workgroup reads frame 0, 8 = no arithmetic
#endif
__device__ __forceinline__ unsigned mulhi24(unsigned a, unsigned b) {      // (a b) >> 32 for a, b < 2^24
    unsigned r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// dst's byte N = (v >> 2) & 0xff, its other bytes kept: the vertical pass's final shift written straight into the packed patch bytes
// (SDWA destination select; `two` = a register holding 2)
__device__ __forceinline__ void put_byte_shr2(unsigned& dst, unsigned v, unsigned two, int n) {      // n: a constant once the callers' loops are unrolled
    switch (n) {
        case 0: asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(two), "v"(v)); break;
        case 1: asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(two), "v"(v)); break;
        case 2: asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(two), "v"(v)); break;
        default: asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(two), "v"(v)); break;
    }
}
template <int G, bool U8OUT = false>
__global__ __launch_bounds__(256) void crop_fast_kernel(const unsigned char* __restrict__ frames, int H, int W,
                                                        const double* __restrict__ states, double factor, int T,
                                                        float m0, float m1, float m2, float s0, float s1, float s2,
                                                        float* __restrict__ out, double* __restrict__ resize_factor) {
    const int b = blockIdx.y, tid = threadIdx.x;
    __shared__ float norm_lut[U8OUT ? 1 : 3 * 256];
    __shared__ __attribute__((aligned(16))) unsigned xtab[CROP_FAST_MAX_T * 4];      // per output column: window byte offset, weights, right column's shift, -
    unsigned char* const out8 = reinterpret_cast<unsigned char*>(out) + (size_t)b * T * T * 3;      // U8OUT: this frame's (T, T, 3) patch
    typedef unsigned u3a __attribute__((ext_vector_type(3), aligned(4)));
    if constexpr (!U8OUT) {
        const float meanv[3] = {m0, m1, m2}, stdq[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {       // Preprocessor.process on the 256 possible values: see crop_kernel
            float scaled = (float)tid * (1.0f / 255.0f);
            asm volatile("" : "+v"(scaled));
            float centred = scaled - meanv[c];
            asm volatile("" : "+v"(centred));
            norm_lut[c * 256 + tid] = centred / stdq[c];
        }
    }
    const double bx = states[4 * b + 0], by = states[4 * b + 1], bw = states[4 * b + 2], bh = states[4 * b + 3];
    const int crop_sz = (int)ceil(sqrt(bw * bh) * factor);
    const int T4 = T >> 2, nitems = T * T4;
    const int item0 = blockIdx.x * G * 256;
    if (!(crop_sz >= 1)) {                  // 'Too small bounding box.': NaN poison, as crop_kernel
        if (blockIdx.x == 0 && tid == 0) resize_factor[b] = __builtin_nan("");
        for (int g = 0; g < G; ++g) {
            const int idx = item0 + g * 256 + tid;
            if (idx < nitems) {
                const int oy = idx / T4, ox0 = (idx - oy * T4) * 4;
                if constexpr (U8OUT) *reinterpret_cast<u3a*>(out8 + ((size_t)oy * T + ox0) * 3) = u3a{0u, 0u, 0u};      // the NaN resize factor carries the poison
                else
                    for (int c = 0; c < 3; ++c) st4(out + (((size_t)b * 3 + c) * T + oy) * T + ox0, splat4(__builtin_nanf("")));
            }
        }
        return;
    }
    const int x1 = (int)rint(bx + 0.5 * bw - crop_sz * 0.5);
    const int y1 = (int)rint(by + 0.5 * bh - crop_sz * 0.5);
    const int x2 = x1 + crop_sz, y2 = y1 + crop_sz;
    const int vx0 = x1 < 0 ? 0 : x1, vx1 = x2 - (x2 - W + 1 > 0 ? x2 - W + 1 : 0);
    const int vy0 = y1 < 0 ? 0 : y1, vy1 = y2 - (y2 - H + 1 > 0 ? y2 - H + 1 : 0);
    if (blockIdx.x == 0 && tid == 0) resize_factor[b] = (double)T / (double)crop_sz;
    const double scale = (double)crop_sz / (double)T;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    for (int ox = tid; ox < T; ox += 256) {
        int sx0, sx1, ax0, ax1;
        lin_coeff(ox, crop_sz, scale, sx0, sx1, ax0, ax1);
        const int xx0 = x1 + sx0, xx1 = x1 + sx1;
        const bool vc0 = xx0 >= vx0 && xx0 < vx1, vc1 = xx1 >= vx0 && xx1 < vx1;
        const int xb = vc0 ? xx0 : (vc1 ? xx1 : 0);         // base pixel of the window: the left column when it is inside the frame
        *reinterpret_cast<u4v*>(xtab + 4 * ox) = u4v{3u * (unsigned)xb, (unsigned)(vc0 ? ax0 : 0) | ((unsigned)(vc1 ? ax1 : 0) << 16),
                                                      (unsigned)(vc1 ? 24 * (xx1 - xb) : 0), 0u};
    }
    __syncthreads();
    const size_t frame_bytes = (size_t)H * W * 3, rest = (size_t)(gridDim.y - b) * frame_bytes;
    const unsigned nrec = rest > 0xfffffff0ull ? 0xfffffff0u : (unsigned)rest;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(frames + ((VT_CROPF_DBG & 4) ? 0 : (size_t)b * frame_bytes)), 0, (int)nrec, 0x00020000);
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    struct Item {
        unsigned long long q0[4], q1[4];
        unsigned wp[4], sh[4], byw0, byw1;
        int oy, ox0;
        bool live;
    };
    auto fetch = [&](int g, Item& it) {
        const int idx = item0 + g * 256 + tid;
        it.live = idx < nitems;
        const int idc = it.live ? idx : nitems - 1;
        it.oy = idc / T4;
        it.ox0 = (idc - it.oy * T4) * 4;
        int sy0, sy1, by0, by1;
        lin_coeff(it.oy, crop_sz, scale, sy0, sy1, by0, by1);
        const int yy0 = y1 + sy0, yy1 = y1 + sy1;
        const bool vr0 = yy0 >= vy0 && yy0 < vy1, vr1 = yy1 >= vy0 && yy1 < vy1;
        const unsigned rowo0 = (unsigned)(vr0 ? yy0 : 0) * (unsigned)(W * 3), rowo1 = (unsigned)(vr1 ? yy1 : 0) * (unsigned)(W * 3);
        it.byw0 = vr0 ? (unsigned)by0 << 12 : 0u;      // a padded row weighs nothing
        it.byw1 = vr1 ? (unsigned)by1 << 12 : 0u;
        // can a window of this item cross the end of the buffer?  (3 (W - 1) is the largest column offset)
        const unsigned far = (rowo0 > rowo1 ? rowo0 : rowo1) + 3u * (unsigned)(W - 1) + 8u;
        const bool slow = __builtin_amdgcn_ballot_w64(far > nrec) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u4v e = *reinterpret_cast<const u4v*>(xtab + 4 * (it.ox0 + k));
            it.wp[k] = e.y; it.sh[k] = e.z;
            const unsigned o0 = rowo0 + e.x, o1 = rowo1 + e.x;
            if (VT_CROPF_DBG & 1) {
                it.q0[k] = ((unsigned long long)o0 << 32) | o1;
                it.q1[k] = ((unsigned long long)o1 << 32) | o0;
            } else
            if (slow) {
                const unsigned ov0 = o0 + 8u > nrec ? o0 + 8u - nrec : 0u, ov1 = o1 + 8u > nrec ? o1 + 8u - nrec : 0u;
                const u2v a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(o0 - ov0), 0, 0);
                const u2v c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(o1 - ov1), 0, 0);
                it.q0[k] = (((unsigned long long)a.y << 32) | a.x) >> (8u * ov0);
                it.q1[k] = (((unsigned long long)c.y << 32) | c.x) >> (8u * ov1);
            } else {
                const u2v a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)o0, 0, 0);
                const u2v c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)o1, 0, 0);
                it.q0[k] = ((unsigned long long)a.y << 32) | a.x;
                it.q1[k] = ((unsigned long long)c.y << 32) | c.x;
            }
        }
    };
    auto finish = [&](const Item& it) {
        float res[3][4];
        unsigned pk[3] = {0u, 0u, 0u};      // U8OUT: the 12 bytes of the item's four pixels, HWC
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned l0 = (unsigned)it.q0[k], l1 = (unsigned)it.q1[k];
            const unsigned r0w = __builtin_amdgcn_alignbit((unsigned)(it.q0[k] >> 32), l0, it.sh[k]);       // the window >> 0 or 24 bits
            const unsigned r1w = __builtin_amdgcn_alignbit((unsigned)(it.q1[k] >> 32), l1, it.sh[k]);
            const us2 wv = __builtin_bit_cast(us2, it.wp[k]);
            if ((VT_CROPF_DBG & 8) && U8OUT) { pk[k % 3] ^= l0 ^ l1 ^ r0w ^ r1w ^ it.wp[k]; continue; }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // bytes of the result: [left pixel's channel c, 0, right pixel's channel c, 0]
                const unsigned sel = 0x0c040c00u + 0x00010001u * (unsigned)c;
                const unsigned r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(r0w, l0, sel)), wv, 0u, false);
                const unsigned r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(r1w, l1, sel)), wv, 0u, false);
                // (b (r >> 4)) >> 16 = ((b << 12) (r with its low 4 bits cleared)) >> 32: one 24-bit high multiply (both factors < 2^24)
                const unsigned t0 = mulhi24(it.byw0, r0 & ~15u), t1 = mulhi24(it.byw1, r1 & ~15u);
                // v = (t0 + t1 + 2) >> 2, clamped at 255; its table entry is at byte 4 v
                if constexpr (U8OUT) {
                    unsigned v = (t0 + t1 + 2u) >> 2;
                    v = v > 255u ? 255u : v;
                    pk[(3 * k + c) >> 2] |= v << (8 * ((3 * k + c) & 3));
                } else {
                unsigned v4 = (t0 + t1 + 2u) & ~3u;
                v4 = v4 > 1020u ? 1020u : v4;
                res[c][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(norm_lut) + c * 1024 + v4);
                }
            }
        }
        if constexpr (U8OUT) {
            if ((VT_CROPF_DBG & 2) ? (pk[0] == 0x12345678u && pk[1] == 0x9abcdef0u) : it.live) *reinterpret_cast<u3a*>(out8 + ((size_t)it.oy * T + it.ox0) * 3) = u3a{pk[0], pk[1], pk[2]};
        } else
        if (it.live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) st4(out + (((size_t)b * 3 + c) * T + it.oy) * T + it.ox0, f4{res[c][0], res[c][1], res[c][2], res[c][3]});
        }
    };
    Item buf[2];
    fetch(0, buf[0]);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) fetch(g + 1, buf[(g + 1) & 1]);
        finish(buf[g & 1]);
    }
}

// crop_band_kernel (round 6) -- the crop at the tracker's sizes (T = 64 / 128 / 256: T / 4 = 2^LGT4 column groups), same arithmetic and
// results bit for bit as crop_fast_kernel / crop_kernel.  Measured on timing builds of crop_fast_kernel (tools/crop_ab.py, T = 128, 256
// frames, us per launch alone): everything 13.8 * no frame loads 11.6 * no loads and no stores 9.3 * no arithmetic 12.9; T = 256: 39.2 for
// 4 x the items -- the kernel is bound by the vector instructions it issues (~500 per item of 12 values, of which the fp64 box geometry,
// the column entry, the row coefficients and the index arithmetic every thread repeats are ~210), not by its traffic.  Here a workgroup
// owns a BAND of IPT x 256 items and a thread IPT items of ONE column group:
//   - geometry, the column table and the band's row table (row byte offsets + vertical weights, one entry per output row) are computed
//     once per IPT x 256 items; a thread reads its four column entries into registers once and one row entry per item
//   - the IPT x 8 window loads of a thread are all in flight before the first item's arithmetic (what a thread per item had in flight
//     as IPT threads), instead of crop_fast_kernel<G>'s two-deep software pipeline, which halved the loads in flight and lost (NOTES R5-8)
//   - only the LAST frame of the batch can read past the buffer's end: the shifted-window slow path is a workgroup-uniform branch
//   - ALIGNED: stamps (tools/crop_stamps.py, -DVT_CROPF_DBG=16) put 12.3 k of a band's 16.6 k cycles into ISSUING its 32 window loads per
//     thread: a byte-aligned 8-byte gather costs the CU's address path ~34 cycles per wave-load against 18 for any dword-aligned load of
//     up to 16 bytes (tools/src/probe_gather.hip), and 512 of them per CU and round is what the kernel waits for.  So a window is fetched
//     as the 12 ALIGNED bytes that contain it (the two pixels' 6 bytes start at byte 0..3 of them) and shifted into place with two
//     v_alignbit; the batch's last frame keeps the byte-aligned form (its shifted-back windows).
template <bool U8OUT, int LGT4, int IPT, bool ALIGNED>
__global__ __launch_bounds__(256) void crop_band_kernel(const unsigned char* __restrict__ frames, int H, int W,
                                                        const double* __restrict__ states, double factor,
                                                        float m0, float m1, float m2, float s0, float s1, float s2,
                                                        float* __restrict__ out, double* __restrict__ resize_factor) {
    constexpr int T4 = 1 << LGT4, T = 4 * T4, RPG = 256 >> LGT4, NROWS = IPT * RPG;      // rows per group of 256 items, rows per band
    static_assert(T <= 256 && (T * T4) % (IPT * 256) == 0, "a band is whole rows and the frame whole bands");
    const int b = blockIdx.y, tid = threadIdx.x;
    // VT_CROPF_DBG & 16 (timing build, uint8 form): s_memtime at the phase boundaries of every workgroup, written over the first 48 bytes of its band
    unsigned long long stamp_[7] = {0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) {
        if constexpr ((VT_CROPF_DBG & 16) != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[i])::"memory");
    };
    stamp(0);
    __shared__ float norm_lut[U8OUT ? 1 : 3 * 256];
    __shared__ __attribute__((aligned(16))) unsigned xtab[T * 4];          // per output column: window byte offset, packed weights, right column's shift, -
    __shared__ __attribute__((aligned(16))) unsigned ytab[NROWS * 4];      // per output row of the band: byte offsets of its two source rows, their weights << 12
    unsigned char* const out8 = reinterpret_cast<unsigned char*>(out) + (size_t)b * T * T * 3;
    typedef unsigned u3a __attribute__((ext_vector_type(3), aligned(4)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    if constexpr (!U8OUT) {
        const float meanv[3] = {m0, m1, m2}, stdq[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {       // Preprocessor.process on the 256 possible values: see crop_kernel
            float scaled = (float)tid * (1.0f / 255.0f);
            asm volatile("" : "+v"(scaled));
            float centred = scaled - meanv[c];
            asm volatile("" : "+v"(centred));
            norm_lut[c * 256 + tid] = centred / stdq[c];
        }
    }
    const double bx = states[4 * b + 0], by = states[4 * b + 1], bw = states[4 * b + 2], bh = states[4 * b + 3];
    const int crop_sz = (int)ceil(sqrt(bw * bh) * factor);
    const int row0 = blockIdx.x * NROWS;                 // first output row of this band
    const int cg = tid & (T4 - 1), rl = tid >> LGT4;     // this thread's column group and its row inside a group of 256 items
    if (!(crop_sz >= 1)) {                  // 'Too small bounding box.': NaN poison, as crop_kernel
        if (blockIdx.x == 0 && tid == 0) resize_factor[b] = __builtin_nan("");
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int oy = row0 + j * RPG + rl;
            if constexpr (U8OUT) *reinterpret_cast<u3a*>(out8 + ((size_t)oy * T + 4 * cg) * 3) = u3a{0u, 0u, 0u};
            else
                for (int c = 0; c < 3; ++c) st4(out + (((size_t)b * 3 + c) * T + oy) * T + 4 * cg, splat4(__builtin_nanf("")));
        }
        return;
    }
    const int x1 = (int)rint(bx + 0.5 * bw - crop_sz * 0.5);
    const int y1 = (int)rint(by + 0.5 * bh - crop_sz * 0.5);
    const int x2 = x1 + crop_sz, y2 = y1 + crop_sz;
    const int vx0 = x1 < 0 ? 0 : x1, vx1 = x2 - (x2 - W + 1 > 0 ? x2 - W + 1 : 0);
    const int vy0 = y1 < 0 ? 0 : y1, vy1 = y2 - (y2 - H + 1 > 0 ? y2 - H + 1 : 0);
    if (blockIdx.x == 0 && tid == 0) resize_factor[b] = (double)T / (double)crop_sz;
    const double scale = (double)crop_sz / (double)T;
    if ((VT_CROPF_DBG & 16) != 0) { asm volatile("" ::"v"(x1), "v"(y1), "v"(scale)); stamp(1); }
    if (tid < T) {                          // column entries (as crop_fast_kernel)
        int sx0, sx1, ax0, ax1;
        lin_coeff(tid, crop_sz, scale, sx0, sx1, ax0, ax1);
        const int xx0 = x1 + sx0, xx1 = x1 + sx1;
        const bool vc0 = xx0 >= vx0 && xx0 < vx1, vc1 = xx1 >= vx0 && xx1 < vx1;
        const int xb = vc0 ? xx0 : (vc1 ? xx1 : 0);
        *reinterpret_cast<u4v*>(xtab + 4 * tid) = u4v{3u * (unsigned)xb, (unsigned)(vc0 ? ax0 : 0) | ((unsigned)(vc1 ? ax1 : 0) << 16),
                                                       (unsigned)(vc1 ? 24 * (xx1 - xb) : 0), 0u};
    }
    if (tid >= 256 - NROWS) {               // row entries of the band, by the LAST threads (the first T are busy with the columns)
        const int r = tid - (256 - NROWS);
        int sy0, sy1, by0, by1;
        lin_coeff(row0 + r, crop_sz, scale, sy0, sy1, by0, by1);
        const int yy0 = y1 + sy0, yy1 = y1 + sy1;
        const bool vr0 = yy0 >= vy0 && yy0 < vy1, vr1 = yy1 >= vy0 && yy1 < vy1;
        *reinterpret_cast<u4v*>(ytab + 4 * r) = u4v{(unsigned)(vr0 ? yy0 : 0) * (unsigned)(W * 3), (unsigned)(vr1 ? yy1 : 0) * (unsigned)(W * 3),
                                                     vr0 ? (unsigned)by0 << 12 : 0u, vr1 ? (unsigned)by1 << 12 : 0u};      // a padded row weighs nothing
    }
    __syncthreads();
    stamp(2);
    const size_t frame_bytes = (size_t)H * W * 3, rest = (size_t)(gridDim.y - b) * frame_bytes;
    const unsigned nrec = rest > 0xfffffff0ull ? 0xfffffff0u : (unsigned)rest;
    const unsigned char* const fb = frames + (size_t)b * frame_bytes;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(fb), 0, (int)nrec, 0x00020000);
    // ALIGNED: the same bytes through a descriptor whose base is the frame's address rounded DOWN to a dword; offsets carry the remainder
    const unsigned mis = (unsigned)(reinterpret_cast<unsigned long long>(fb) & 3ull);
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(fb - mis), 0, (int)(nrec > 0xfffffff0u - 4u ? nrec : nrec + mis), 0x00020000);
    unsigned xo[4], wp[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u4v e = *reinterpret_cast<const u4v*>(xtab + 4 * (4 * cg + k));
        xo[k] = e.x + (ALIGNED ? mis : 0u); wp[k] = e.y; sh[k] = e.z;
    }
    typedef unsigned u3v __attribute__((ext_vector_type(3)));
    // a window can cross the end of the buffer only in the batch's last frame (its last rows): there every load takes the byte-aligned,
    // shifted-back form.  The two forms are two instantiations of one body (AL: aligned 12-byte loads), so neither holds the other's registers.
    // ... and there only in a band that reads the frame's last source row (its windows end at most 11 + 3 bytes past their first byte: inside
    // the next row): every other band of the last frame takes the aligned form too -- the shifted form waits for each pair of windows before it
    // requests the next (the shift is part of the request), a chain of 4 IPT memory round trips that the whole launch waited for
    bool last = b == (int)gridDim.y - 1;
    if (ALIGNED && last) {
        int sl0, sl1, al0, al1;
        lin_coeff(row0 + NROWS - 1, crop_sz, scale, sl0, sl1, al0, al1);
        const int ymax = y1 + sl1 < vy1 - 1 ? y1 + sl1 : vy1 - 1;       // rows beyond the valid range read row 0
        if (ymax <= H - 2 && W * 3 >= 16) last = false;
    }
    unsigned two = 2u;
    asm volatile("" : "+v"(two));      // put_byte_shr2's shift operand has to live in a vector register
    const bool col_live = (wp[0] | wp[1] | wp[2] | wp[3]) != 0u;
    auto body = [&](auto al_c) {
        constexpr bool AL = decltype(al_c)::value;
        u2v q0[AL ? 1 : IPT][4], q1[AL ? 1 : IPT][4];
        u3v ra0[AL ? IPT : 1][4], ra1[AL ? IPT : 1][4];      // AL: the 12 aligned bytes around each window ...
        unsigned ro0[IPT], ro1[IPT];                          // ... and the item's row offsets: a window's shift is 8 x its offset's low two bits
        unsigned byw0[IPT], byw1[IPT];
        auto issue = [&](int j) {
            const u4v e = *reinterpret_cast<const u4v*>(ytab + 4 * (j * RPG + rl));
            byw0[j] = e.z; byw1[j] = e.w; ro0[j] = e.x; ro1[j] = e.y;
            // an item whose two rows or whose four columns all lie in the crop's zero padding (a window reaching over the frame's
            // border) weighs nothing: its loads are skipped, the products below are 0 x 0
            const bool live = ((e.z | e.w) != 0u) && col_live;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned o0 = e.x + xo[k], o1 = e.y + xo[k];
                if (!live) {
                    if constexpr (AL) { ra0[j][k] = u3v{0u, 0u, 0u}; ra1[j][k] = u3v{0u, 0u, 0u}; }
                    else { q0[j][k] = u2v{0u, 0u}; q1[j][k] = u2v{0u, 0u}; }
                } else
                if constexpr (AL) {
                    // raw 12 aligned bytes now; the funnel shift where they are used
                    ra0[j][k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_a, (int)(o0 & ~3u), 0, 0);
                    ra1[j][k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc_a, (int)(o1 & ~3u), 0, 0);
                } else {
                    if (ALIGNED) { o0 -= mis; o1 -= mis; }
                    if (last) {
                        const unsigned ov0 = o0 + 8u > nrec ? o0 + 8u - nrec : 0u, ov1 = o1 + 8u > nrec ? o1 + 8u - nrec : 0u;
                        const u2v a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(o0 - ov0), 0, 0);
                        const u2v c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(o1 - ov1), 0, 0);
                        const unsigned long long wa = (((unsigned long long)a.y << 32) | a.x) >> (8u * ov0), wc = (((unsigned long long)c.y << 32) | c.x) >> (8u * ov1);
                        q0[j][k] = u2v{(unsigned)wa, (unsigned)(wa >> 32)};
                        q1[j][k] = u2v{(unsigned)wc, (unsigned)(wc >> 32)};
                    } else {
                        q0[j][k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)o0, 0, 0);
                        q1[j][k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)o1, 0, 0);
                    }
                }
            }
        };
        auto math = [&](int j) {
            const int oy = row0 + j * RPG + rl;
            float res[3][4];
            unsigned pk[3] = {0u, 0u, 0u};      // U8OUT: the 12 bytes of the item's four pixels, HWC
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u2v w0, w1;
                if constexpr (AL) {      // v_alignbit reads bits [4:0] of its shift: 8 x (offset & 3)
                    const unsigned b0 = (ro0[j] + xo[k]) << 3, b1 = (ro1[j] + xo[k]) << 3;
                    w0 = u2v{__builtin_amdgcn_alignbit(ra0[j][k].y, ra0[j][k].x, b0), __builtin_amdgcn_alignbit(ra0[j][k].z, ra0[j][k].y, b0)};
                    w1 = u2v{__builtin_amdgcn_alignbit(ra1[j][k].y, ra1[j][k].x, b1), __builtin_amdgcn_alignbit(ra1[j][k].z, ra1[j][k].y, b1)};
                } else {
                    w0 = q0[j][k]; w1 = q1[j][k];
                }
                const unsigned l0 = w0.x, l1 = w1.x;
                const unsigned r0w = __builtin_amdgcn_alignbit(w0.y, l0, sh[k]);       // the window >> 0 or 24 bits
                const unsigned r1w = __builtin_amdgcn_alignbit(w1.y, l1, sh[k]);
                const us2 wv = __builtin_bit_cast(us2, wp[k]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned sel = 0x0c040c00u + 0x00010001u * (unsigned)c;       // [left pixel's channel c, 0, right pixel's channel c, 0]
                    const unsigned r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(r0w, l0, sel)), wv, 0u, false);
                    const unsigned r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(r1w, l1, sel)), wv, 0u, false);
                    const unsigned t0 = mulhi24(byw0[j], r0 & ~15u), t1 = mulhi24(byw1[j], r1 & ~15u);      // (b (r >> 4)) >> 16, see crop_fast_kernel
                    // t0 + t1 <= (by0 + by1) * 32640 >> 16 = 1020 (the weights of a pair sum to 2048, r >> 4 <= 255 * 128): the reference's
                    // saturation can never act, so no clamp here (crop_kernel / crop_fast_kernel keep theirs: same results)
                    if constexpr (U8OUT) {
                        put_byte_shr2(pk[(3 * k + c) >> 2], t0 + t1 + 2u, two, (3 * k + c) & 3);
                    } else {
                        const unsigned v4 = (t0 + t1 + 2u) & ~3u;
                        res[c][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(norm_lut) + c * 1024 + v4);
                    }
                }
            }
            if constexpr (U8OUT) {
                *reinterpret_cast<u3a*>(out8 + ((size_t)oy * T + 4 * cg) * 3) = u3a{pk[0], pk[1], pk[2]};
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) st4(out + (((size_t)b * 3 + c) * T + oy) * T + 4 * cg, f4{res[c][0], res[c][1], res[c][2], res[c][3]});
            }
        };
        // two items at a time (the conditional loads are waited for as a whole anyway): half the window registers, so that the kernel
        // stays within 128 registers = four workgroups per CU = ONE round of the 1024 bands of 256 G128 frames
        constexpr int HS = IPT < 2 ? IPT : 2;
#pragma unroll
        for (int h = 0; h < IPT; h += HS) {
#pragma unroll
            for (int j = 0; j < HS; ++j) issue(h + j);
            if ((VT_CROPF_DBG & 16) != 0 && h == 0) {
                stamp(3);                                   // the first half's loads issued
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp(4);                                   // ... and here
            }
#pragma unroll
            for (int j = 0; j < HS; ++j) math(h + j);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (ALIGNED && !last) body(std::integral_constant<bool, ALIGNED>{});
    else body(std::false_type{});
    if constexpr ((VT_CROPF_DBG & 16) != 0 && U8OUT) {
        stamp(5);
        {   // where this workgroup ran: HW_ID (wave / SIMD / CU / SH / SE ids) and XCC_ID
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
            stamp_[6] = ((unsigned long long)xcc << 32) | hw;
        }
        __syncthreads();
        if (tid == 0)
            for (int i = 0; i < 7; ++i) reinterpret_cast<unsigned long long*>(out8 + (size_t)row0 * T * 3)[i] = stamp_[i];
    }
}

// One thread per sequence.  hann_boxes (B,4) float [cx,cy,w,h] in [0,1]; states (B,4) double in/out.
// `record` (optional, (B,5) double, device memory or device-mapped pinned host memory): [x, y, w, h, confidence] of the new state --
// what track() returns; written here, the step needs no copy kernel and no device -> host copy after it.
__global__ void update_state_kernel(const float* __restrict__ hann_boxes, const double* __restrict__ resize_factor,
                                    int search_size, int H, int W, int margin, int B, double* __restrict__ states,
                                    const float* __restrict__ conf, double* __restrict__ record) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float hb[4] = {hann_boxes[4 * b + 0], hann_boxes[4 * b + 1], hann_boxes[4 * b + 2], hann_boxes[4 * b + 3]};
    const TrackTail t{resize_factor, states, record, search_size, H, W, margin, 0};
    update_state_one(b, hb, conf != nullptr ? conf[b] : 0.f, t);
}

}  // namespace vtt
