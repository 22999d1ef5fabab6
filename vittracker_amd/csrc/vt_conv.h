// vt_conv.h -- shared implicit-GEMM core for the 3x3 convolutions of the stem and the head.
//
// A conv layer is D[oc][pixel] = sum_k W[oc][k] * im2col[k][pixel] on v_mfma_f32_16x16x4_f32:
//   A operand = folded weights, pre-packed on the host as images [oc_tile][chunk][64 lanes][4]
//               (chunk = 16 k-values = 4 "quads" of 4 input channels at one tap),
//   B operand = activations read from a quad-planar LDS map: lane (px = lane & 15, q = lane >> 4)
//               fetches quad Q = 4*chunk + q with ONE ds_read_b128; the layer supplies the map
//               offset of (tap, channel-quad) through `off_of_chunk`.
//
// Scheduling (what keeps the MFMA pipe fed): the weights of a pass of up to MAXC chunks are
// requested from L2 in one burst BEFORE the pass (one exposed latency per pass instead of one per
// chunk -- at 8-32 MFMAs per chunk a single chunk of cover is shorter than an L2 round trip),
// the B operands are read one chunk ahead, and inside a chunk the MFMAs go round-robin over the
// NPT x NOT independent accumulators so no MFMA waits on its predecessor's result.
#pragma once
#include "vt_common.h"

namespace vtc {

// Decode quad Q of a layer with NQ channel-quads per tap: (tap, icq); pad quads clamp to the last
// real one (their weights are zero in the packed image).
template <int NQ>
__device__ __forceinline__ void decode_quad(int Q, int& tap, int& icq) {
    constexpr int NQT = 9 * NQ;
    Q = Q < NQT ? Q : NQT - 1;
    tap = Q / NQ;
    icq = Q - tap * NQ;
}

// Request the weights of chunks [c0, c0 + n), n <= MAXC, of NOT consecutive output tiles.
// NCH = chunks per output tile in the weight image (image stride).  Issue this as early as the
// data flow allows: it does not depend on activations.
// The image holds STORED operands (vt_common.h `opnd`): float4 in the fp32 build; in the f16 build vt_load_weights converts the
// head's and the stem's layer-3 / layer-4 images in place -- a lane's four values as h4 in the first 8 bytes of its 16-byte slot,
// so every float offset into an image stays valid -- and a load fetches 8 bytes and needs no conversion (round 4).
template <int NOT, int MAXC, int NCH>
__device__ __forceinline__ void load_weights(const float* __restrict__ wimg, int c0, int n, int lane,
                                             opnd (&a)[MAXC][NOT]) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
        if (k < n)
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot) a[k][ot] = *reinterpret_cast<const opnd*>(wimg + ((size_t)(ot * NCH + c0 + k) * 64 + lane) * 4);
    __builtin_amdgcn_sched_barrier(0);      // keep the burst where it was written (the scheduler would sink it)
}
// the same from an image that stays float4 in both builds (the stem's layer 2, which other kernels also copy to LDS as it is)
template <int NOT, int MAXC, int NCH>
__device__ __forceinline__ void load_weights_f4(const float* __restrict__ wimg, int c0, int n, int lane,
                                                f4 (&a)[MAXC][NOT]) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
        if (k < n)
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot) a[k][ot] = ld4(wimg + ((size_t)(ot * NCH + c0 + k) * 64 + lane) * 4);
    __builtin_amdgcn_sched_barrier(0);
}

// Accumulate chunks [c0, c0 + N) with preloaded weights for NPT pixel tiles x NOT output tiles.
// N and NPT are compile-time: no branch may sit between MFMAs (a wave-uniform `if` around each
// MFMA costs a basic block + register shuffling per instruction and halves the pipe's duty).
//   base[i]  per-lane map offset of tap (0,0) for pixel tile i
//   off(c)   per-lane map offset of this lane's quad of chunk c (tap + channel-quad part)
// PIN: keep the next chunk's B reads ahead of this chunk's MFMAs (see below); costs registers, so opt-in.
// General form: at(c, i) = per-lane map offset of this lane's quad of chunk c for pixel tile i.
template <int NOT, int NPT, int MAXC, int N, bool PIN = false, typename WT, typename AtFn>
__device__ __forceinline__ void mma_pass_at(const f4* in_map, const WT (&a)[MAXC][NOT], int c0, AtFn at, f4 (&acc)[NPT][NOT]) {
    static_assert(N <= MAXC, "chunk count");
    f4 b[2][NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) b[0][i] = in_map[at(c0, i)];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k + 1 < N) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) b[(k + 1) & 1][i] = in_map[at(c0 + k + 1, i)];
            // keep these reads ahead of chunk k's MFMAs: left alone, the scheduler sinks them to just before
            // their first use and every chunk then starts with a full LDS round trip
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        }
#ifdef VT_F16
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int i = 0; i < NPT; ++i) acc[i][ot] = mfma4(a[k][ot], b[k & 1][i], acc[i][ot]);
#else
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
                for (int i = 0; i < NPT; ++i)
                    acc[i][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][ot][r], b[k & 1][i][r], acc[i][ot], 0, 0, 0);
#endif
    }
}

// The common case: offset = base[i] (tap (0,0) of pixel tile i) + off(c) (tap + channel-quad part of chunk c).
template <int NOT, int NPT, int MAXC, int N, bool PIN = false, typename WT, typename OffFn>
__device__ __forceinline__ void mma_pass(const f4* in_map, const int (&base)[NPT], const WT (&a)[MAXC][NOT], int c0,
                                         OffFn off, f4 (&acc)[NPT][NOT]) {
    int oc = 0, ov = 0;      // off() of the last chunk asked for (at() is called for i = 0 .. NPT-1 of the same chunk in a row)
    bool have = false;
    auto at = [&](int c, int i) {
        if (!have || c != oc) { ov = off(c); oc = c; have = true; }
        return ov + base[i];
    };
    mma_pass_at<NOT, NPT, MAXC, N, PIN>(in_map, a, c0, at, acc);
}

// All NCH chunks of a layer in passes of MAXC (weights fetched per pass); NCH compile-time.
template <int NOT, int NPT, int MAXC, int NCH, typename OffFn>
__device__ __forceinline__ void conv_all_chunks(const f4* in_map, const int (&base)[NPT],
                                                const float* __restrict__ wimg, int lane, OffFn off,
                                                f4 (&acc)[NPT][NOT]) {
    constexpr int FULL = NCH / MAXC, REM = NCH % MAXC;
#pragma unroll
    for (int p = 0; p < FULL; ++p) {
        opnd a[MAXC][NOT];
        load_weights<NOT, MAXC, NCH>(wimg, p * MAXC, MAXC, lane, a);
        mma_pass<NOT, NPT, MAXC, MAXC>(in_map, base, a, p * MAXC, off, acc);
    }
    if constexpr (REM > 0) {
        opnd a[MAXC][NOT];
        load_weights<NOT, MAXC, NCH>(wimg, FULL * MAXC, REM, lane, a);
        mma_pass<NOT, NPT, MAXC, REM>(in_map, base, a, FULL * MAXC, off, acc);
    }
}

}  // namespace vtc
