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
template <bool BYTES = false>
__global__ __launch_bounds__(256) void crop_kernel(const unsigned char* __restrict__ frames, int H, int W,
                                                   const double* __restrict__ states, double factor, int T,
                                                   float m0, float m1, float m2, float s0, float s1, float s2,
                                                   float* __restrict__ out, double* __restrict__ resize_factor) {
    const int b = blockIdx.y;
    // Preprocessor.process maps a uint8 value to (v / 255 - mean) / std: 256 x 3 possible results.  They are computed ONCE per
    // workgroup with the reference's arithmetic (three separately rounded fp32 ops, below) into an LDS table -- per output value
    // one LDS read instead of a convert, a multiply, a subtract and an IEEE division sequence (~14 VALU instructions of the ~74 a
    // value cost).
    __shared__ float norm_lut[3 * 256];
    {
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
    }
    __syncthreads();
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
                for (int k = 0; k < 4 && ox0 + k < T; ++k) out[(((size_t)b * 3 + c) * T + oy) * T + ox0 + k] = __builtin_nanf("");
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
            res[c][k] = norm_lut[c * 256 + v];
        }
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

// One thread per sequence.  hann_boxes (B,4) float [cx,cy,w,h] in [0,1]; states (B,4) double in/out.
// `record` (optional, (B,5) double, device memory or device-mapped pinned host memory): [x, y, w, h, confidence] of the new state --
// what track() returns; written here, the step needs no copy kernel and no device -> host copy after it.
__global__ void update_state_kernel(const float* __restrict__ hann_boxes, const double* __restrict__ resize_factor,
                                    int search_size, int H, int W, int margin, int B, double* __restrict__ states,
                                    const float* __restrict__ conf, double* __restrict__ record) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float hb[4] = {hann_boxes[4 * b + 0], hann_boxes[4 * b + 1], hann_boxes[4 * b + 2], hann_boxes[4 * b + 3]};
    const TrackTail t{resize_factor, states, record, search_size, H, W, margin};
    update_state_one(b, hb, conf != nullptr ? conf[b] : 0.f, t);
}

}  // namespace vtt
