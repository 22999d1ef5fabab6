// vt_blocks.h -- the joint template+search transformer stack, one workgroup per frame.
//
// Replaces, for C = 48 / 1 head / mlp_ratio 4:
//   timm Block.forward x depth + final LayerNorm   (lib/models/vit_dist/vit_dist.py:88-94;
//   block arithmetic restated in-tree at lib/models/layers/attn_blocks.py:130-133 and
//   lib/models/layers/attn.py:33-59).
//
// Design (MI355X): every contraction runs on v_mfma_f32_16x16x4_f32 (exact f32).  Each wave owns
// 16-token tiles and keeps their residual stream in registers as three "operand images"
// (vt_common.h): lane (tok = lane & 15, q = lane >> 4) holds features 16c + 4q + {0..3} of its
// token for chunk c.  With the weights as the A operand (rows = output features) and the
// activations as the B operand (columns = tokens), an MFMA result is again an operand image of
// the next layer, so LN -> QKV -> softmax -> PV -> proj -> LN -> fc1 -> GELU -> fc2 chains
// entirely in registers.  Only K and V cross waves, through LDS, as lane-linear 1 KiB tiles
// (conflict-free ds_write_b128 / ds_read_b128):
//   Kimg[J][c]   = k features of key tile J          (from W_k as A, h as B)
//   Vimg[t][J]   = V^T: feature tile t x key tile J  (from h as A, W_v as B: swapped operands,
//                  which lands V already transposed for the P.V product)
// S^T = K q^T is computed with keys as rows, so a softmax row lives in one lane's registers plus
// the 3 partner lanes (two xor-shuffles), never in LDS.
#pragma once
#include <type_traits>
#include "vt_common.h"
#include "vt_bf3.h"

namespace vtb {

constexpr int C = 48;          // embed dim
constexpr int NC = C / 16;     // 3 feature chunks
constexpr int HID = 4 * C;     // 192
constexpr int NH = HID / 16;   // 12 hidden chunks
constexpr float LN_EPS = 1e-5f;

// per-block packed parameter offsets (floats); images are [out_tile][k_chunk][lane][4]
constexpr int O_LN1G = 0;
constexpr int O_LN1B = O_LN1G + C;
constexpr int O_WQKV = O_LN1B + C;                 // 9 x 3 tiles
constexpr int O_BQKV = O_WQKV + 9 * NC * 256;
constexpr int O_WPROJ = O_BQKV + 3 * C;            // 3 x 3 tiles
constexpr int O_BPROJ = O_WPROJ + NC * NC * 256;
constexpr int O_LN2G = O_BPROJ + C;
constexpr int O_LN2B = O_LN2G + C;
constexpr int O_W1 = O_LN2B + C;                   // 12 x 3 tiles
constexpr int O_B1 = O_W1 + NH * NC * 256;
constexpr int O_W2 = O_B1 + HID;                   // 3 x 12 tiles
constexpr int O_B2 = O_W2 + NC * NH * 256;
constexpr int BLOCK_STRIDE = O_B2 + C;             // 28272 floats
static_assert(BLOCK_STRIDE % 4 == 0 && O_WQKV % 4 == 0 && O_W1 % 4 == 0 && O_W2 % 4 == 0, "16B alignment");

// BF3 (G128 frame form): the MLP's two GEMMs -- 52 % of a block's MACs -- run on the bf16 matrix pipe as exact three-piece
// products (vt_bf3.h).  Their weights come from a second parameter buffer of pre-split images, per block:
//   fc1 (K = 48 = a chunk pair + one odd chunk): [out tile 12][ pair 0: piece 3 x 64 lanes x 16 B | chunk 2: piece 3 x 64 lanes x 8 B ]  = 54 KiB
//   fc2 (K = 192 = 6 chunk pairs):               [out tile 3][pair 6][piece 3][64 lanes x 16 B]                                     = 54 KiB
//   qkv (K = 48; LN1 folded), fc1's layout:      [out tile 9: q 0-2, k 3-5, v 6-8][ pair 0 | chunk 2 ]                              = 40.5 KiB
// (v: the operands swap roles -- tokens as rows, so V lands transposed -- which the 16 x 16 x 32 instruction's identical A / B
// register layouts make free: the same image serves.)  BF3 alone leaves proj (9 KiB) and the attention products on fp32 MFMAs.
// A3 (round 5, G128 frame form): those too -- K, V^T and the guests' q are PUBLISHED as pieces (split once by the wave that computes
// them), S^T = K q^T, O^T = V^T P^T and proj run as six-term products; nothing in the kernel issues an fp32 MFMA any more.  LDS:
//   K pieces   [key tile NT][ pair 0: piece 3 x 64 x 16 B | chunk 2: piece 3 x 64 x 8 B ]     (a key tile = one fc1-layout "output tile")
//   V^T pieces [feature tile NC][ key-chunk pair NT / 2: piece 3 x 64 x 16 B | odd last chunk: piece 3 x 64 x 8 B ]
//   proj       fc1's layout, 3 output tiles (13.5 KiB, in buffer B)
// = 45 instead of 30 KiB for K / V^T; the room comes from the guests' exchange areas, which move into space that is idle in
// their phase: q pieces / attention partials / softmax statistics behind proj's image in buffer B (free until fc2 is staged
// after the attention barrier), the fc2 partial sums behind the next block's qkv image in buffer A (free until fc1 is staged).
constexpr int W3_FC1_TILES = NH * 3 + NH * 3 / 2;     // KiB tiles (the LDS-DMA unit)
constexpr int W3_FC1_OT16 = 3 * 64 + 3 * 32;          // 16-byte units per output tile (288)
constexpr int W3_FC2_TILES = NC * (NH / 2) * 3;
constexpr int W3_QKV_TILES = (9 * W3_FC1_OT16 + 63) / 64;                // qkv (K = 48, 9 output tiles) in fc1's layout: 40.5 KiB, staged as 41
constexpr int W3_PROJ_TILES = (NC * W3_FC1_OT16 + 63) / 64;               // proj (K = 48, 3 output tiles) in fc1's layout: 13.5 KiB, staged as 14 (A3)
constexpr int BLOCK3_STRIDE = (W3_FC1_TILES + W3_FC2_TILES + W3_QKV_TILES + W3_PROJ_TILES) * 256;      // floats: [fc1 | fc2 | qkv | proj]
static_assert(W3_FC1_TILES == 54 && W3_FC2_TILES == 54 && NH * W3_FC1_OT16 == W3_FC1_TILES * 64, "three-piece MLP images");

// The small parameters of a block (LayerNorm gamma / beta and the four bias vectors, 624 floats) are
// copied to LDS once when the kernel starts.  Read from global memory where they are used -- as the
// accumulator initialisers in front of a GEMM -- each one exposes an L2 round trip of several hundred
// cycles with nothing to hide it behind; from LDS they cost what an operand image costs.
constexpr int S_LN1G = 0;
constexpr int S_LN1B = S_LN1G + C;
constexpr int S_BQKV = S_LN1B + C;
constexpr int S_BPROJ = S_BQKV + 3 * C;
constexpr int S_LN2G = S_BPROJ + C;
constexpr int S_LN2B = S_LN2G + C;
constexpr int S_B1 = S_LN2B + C;
constexpr int S_B2 = S_B1 + HID;
constexpr int SMALL_STRIDE = S_B2 + C;             // 624 floats per block; the final norm (2 C) follows the last block
static_assert(SMALL_STRIDE % 4 == 0 && S_BQKV % 4 == 0 && S_BPROJ % 4 == 0 && S_B1 % 4 == 0 && S_B2 % 4 == 0, "16B alignment");
static_assert(O_LN1B == O_LN1G + C && O_LN2G == O_BPROJ + C && O_LN2B == O_LN2G + C, "contiguous sections");
__host__ __device__ constexpr int small_floats(int depth) { return depth * SMALL_STRIDE + 2 * C; }

// global offset (floats, relative to params) of small-parameter float `f` of the LDS copy
__device__ __forceinline__ size_t small_src(int f, int depth) {
    const int blk = f / SMALL_STRIDE, r = f - blk * SMALL_STRIDE;
    if (blk >= depth) return (size_t)depth * BLOCK_STRIDE + r;    // norm.weight, norm.bias
    const int o = r < S_BQKV ? O_LN1G + r : r < S_BPROJ ? O_BQKV + (r - S_BQKV) : r < S_B1 ? O_BPROJ + (r - S_BPROJ)
                : r < S_B2 ? O_B1 + (r - S_B1) : O_B2 + (r - S_B2);
    return (size_t)blk * BLOCK_STRIDE + o;
}

__device__ __forceinline__ f4 wimg(const float* __restrict__ base, int tile, int lane) {
    return ld4(base + (size_t)tile * 256 + lane * 4);
}

// the same from an image of stored operands (vt_common.h `opnd`: f4, or the f16 build's pre-converted h4 images)
__device__ __forceinline__ opnd wimg_o(const opnd* __restrict__ base, int tile, int lane) { return base[(size_t)tile * 64 + lane]; }

// Burst-load N weight operand images (tile indices first + j * stride) and pin the burst where it
// is written, so it is in flight while the code that follows (LayerNorm, the previous GEMM's MFMAs)
// executes instead of stalling the GEMM that consumes it.
template <int N>
__device__ __forceinline__ void wburst(f4 (&a)[N], const float* __restrict__ base, int first, int stride, int lane) {
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = wimg(base, first + j * stride, lane);
    __builtin_amdgcn_sched_barrier(0);
}

// Tell the scheduler to issue `n` rounds of {1 MFMA, `valu` VALU/transcendental ops} (the fp32-MFMA forms' MLP: GELU of one hidden
// group written between the MFMAs of the next).  What that buys is instruction ORDER, not overlap: nothing issues on a SIMD while
// one of its waves executes v_mfma_f32_16x16x4_f32 (tools/src/probe_coexec.hip, NOTES R4-1); beside a bf16 MFMA about two plain
// vector instructions per MFMA do.
template <int N, int VALU_PER_MFMA>
__device__ __forceinline__ void interleave_mfma_valu() {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
    }
}

// LayerNorm of one token held as NC operand-image chunks.  AFFINE = true: g / b point at gamma / beta (C floats) -- the final
// norm, whose output is the head's input.  AFFINE = false: the normalised token only -- norm1 and norm2 feed a linear layer, and
// vt_load_weights folds their affine part into it in double precision (W' = W diag(gamma), b' = b + W beta: qkv and fc1), which
// takes 6 packed multiplies / fmas and 6 LDS reads out of every LayerNorm.
template <bool AFFINE = true>
__device__ __forceinline__ void layer_norm_img(const f4 (&x)[NC], f4 (&h)[NC], const float* __restrict__ g,
                                               const float* __restrict__ b, int q) {
    // sums over a token's 48 features: packed adds / fmas over the three chunks first, then one horizontal sum and the two
    // cross-lane steps (the four lanes sharing lane & 15 hold the token)
    static_assert(NC == 3, "three feature chunks");
    const float mean = quad_sum(hsum4((x[0] + x[1]) + x[2])) * (1.0f / C);
    f4 d[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) d[c] = x[c] - splat4(mean);
    const f4 v4 = __builtin_elementwise_fma(d[2], d[2], __builtin_elementwise_fma(d[1], d[1], d[0] * d[0]));
    const float inv = __builtin_amdgcn_rsqf(quad_sum(hsum4(v4)) * (1.0f / C) + LN_EPS);   // v_rsq_f32: 1 ulp
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if constexpr (AFFINE) {
            const f4 gg = ld4(g + 16 * c + 4 * q), bb = ld4(b + 16 * c + 4 * q);
            h[c] = d[c] * splat4(inv) * gg + bb;
        } else {
            h[c] = d[c] * splat4(inv);
        }
    }
}
__device__ __forceinline__ void layer_norm_plain(const f4 (&x)[NC], f4 (&h)[NC]) { layer_norm_img<false>(x, h, nullptr, nullptr, 0); }

// Weight staging through LDS (WLDS).  Every wave of a workgroup needs every weight tile of a GEMM.
// Fetched per wave from L2 that is NW x 110 KB per block and workgroup -- with 256 workgroups reading
// the same lines at the same time the L2 channels, not the MFMA pipe, set the pace (measured: 48
// MFMAs took ~3000 cycles instead of 1536).  With WLDS each 1 KiB operand image is copied ONCE per
// workgroup by LDS-DMA (global_load_lds_dwordx4: lane-linear, exactly the image layout) into one of
// two staging buffers, one GEMM ahead of its use, and the A operands are ds_read_b128 from there:
//     buffer A: qkv (27 tiles) -> fc1 (36) -> next block's qkv ...
//     buffer B: proj (9 tiles) -> fc2 (36) -> next block's proj ...
// A buffer is refilled only after the barrier that ends its last reader; __syncthreads() waits
// for the DMA (vmcnt) before it releases the readers.
constexpr int WBUF_TILES = NH * NC;   // 36: fc1 / fc2
// A barrier that publishes LDS-DMA staged weights to the other waves.  global_load_lds is tracked by vmcnt,
// while at workgroup scope the memory model only promises lgkmcnt(0) in front of a barrier: hipcc (ROCm 7.2)
// happens to emit vmcnt(0) there as well, but that is compiler behaviour, so the wait is written out.
template <bool STAGED>
__device__ __forceinline__ void barrier_publish() {
    if constexpr (STAGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// timing experiments only (wrong results): -DVT_BLK_GUESTS=0 compiles the guest waves' work out of the balanced frame form (they keep
// the barriers), -DVT_BLK_OWNERS=0 the owners' -- what each role costs alone, and what the pairing costs (tools/block_stamps.py)
#ifndef VT_BLK_GUESTS
#define VT_BLK_GUESTS 1
#endif
#ifndef VT_BLK_OWNERS
#define VT_BLK_OWNERS 1
#endif
#ifndef VT_BLK_NOSTAGE
#define VT_BLK_NOSTAGE 0       // timing experiments only (wrong results): 1 = no weight staging at all -- the MFMAs then run on whatever the LDS
#endif                         // held, mostly zeros, cooler and at higher clocks: NOT a measure of what staging costs; 2 (BF3L form) = staging in
                               // block 0 only, its valid weights re-used by the later blocks: that is (38.0 against 38.9 us: ~1.3 us for three blocks)
__device__ __forceinline__ void stage_tiles(f4* dst, const float* __restrict__ src, int ntiles, int w, int nw, int lane, bool first_block = true) {
    if (VT_BLK_NOSTAGE == 1 || (VT_BLK_NOSTAGE == 2 && !first_block)) return;
    for (int t = w; t < ntiles; t += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)t * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dst + t * 64), 16, 0, 0);
}

// One GEMM stage: NCHUNK k-chunks, N independent accumulator chains.  `opa(c, a)` fills the N
// per-chain operands of chunk c, `opb(c)` returns the operand all chains share.  The operands of
// chunk c+1 are requested BEFORE the MFMAs of chunk c are issued (explicit double buffer): left to
// itself the scheduler puts each ds_read / global_load right in front of its first use and the wave
// pays the full read latency once per chunk (measured: ~47 instead of 32 cycles per MFMA).
// AT: type of the per-chain operands: `opnd` for stored images (weights, K, V^T), f4 for values computed in this wave.
template <int NCHUNK, int N, bool SHARED_IS_B, bool PRELOAD_ALL = true, typename AT = opnd, typename OpA, typename OpS>
__device__ __forceinline__ void gemm_stage(OpA opa, OpS ops, f4 (&acc)[N]) {
    if constexpr (PRELOAD_ALL && NCHUNK * N <= 24) {
        // small stage: request every operand first and pin the requests ahead of the MFMAs -- one
        // read round trip per stage instead of one per chunk
        AT a[NCHUNK][N];
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) opa(c, a[c]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            if constexpr (SHARED_IS_B) mfma4_shared_b(a[c], ops(c), acc);
            else mfma4_shared_a(ops(c), a[c], acc);
        }
    } else {
        AT a[2][N];
        opa(0, a[0]);
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            if (c + 1 < NCHUNK) opa(c + 1, a[(c + 1) & 1]);
            if constexpr (SHARED_IS_B) mfma4_shared_b(a[c & 1], ops(c), acc);
            else mfma4_shared_a(ops(c), a[c & 1], acc);
        }
    }
}

// NT = token tiles per frame (L / 16), NW = waves per workgroup, TPW = tiles per wave.
// (min waves per SIMD: the lean 5-wave variant is capped at 168 VGPRs so two or three workgroups share a CU)
//
// BAL (G128: 5 tiles on a CU's 4 SIMDs).  One wave per tile puts two tile-waves on SIMD0, which then
// carries 2/5 of the MFMAs and sets the kernel time.  The balanced variant runs 8 waves: waves 0-3
// ("owners", one per SIMD) own tiles 0-3 exactly as before; waves 4-7 ("guests", again one per SIMD)
// share tile 4 four ways and keep its residual stream redundantly in registers:
//     LN1 + QKV   guest 0: q -> Qg (LDS), guest 1: k -> Kimg, guest 2: v -> Vimg          (36 MFMAs each)
//     attention   guest g: keys of tile g (guest 3: tiles 3 and 4) -> partial (max, sum, P.V) in LDS,
//                 guests-only rendezvous on an LDS counter, flash-style merge               (24 / 48)
//     proj        guest g < 3: output tile g -> Dg (LDS); every guest adds all three        (12)
//     LN2 + fc1   guest g: hidden tiles 3g..3g+2, GELU                                      (36)
//     fc2         K-split: guest g contracts its own hidden tiles -> partial in LDS; summed (36)
// Each SIMD then issues ~708 instead of 1104 MFMAs per block, and has a second instruction stream that
// fills the owner's waits (f32 MFMA and VALU issue add up even across waves: tools/src/probe_overlap.hip).
template <int NT, int NW, int TPW, bool WLDS, bool BAL = false, bool ZC = false, bool BF3 = false, bool A3 = false>   // ZC: template-cache variant (config 5); BF3: qkv + MLP on the bf16 pipe; A3: attention + proj too
__global__ __launch_bounds__(NW * 64, (NW == 5 && !WLDS) ? 3 : 1) void blocks_kernel(const float* __restrict__ tokens,   // (B, L, C)
                                                         const float* __restrict__ params,   // packed, see O_*
                                                         float* __restrict__ feat,           // (B, Lx, C)
                                                         float* __restrict__ resid,          // (B, L, C) or null
                                                         int len_z, int depth_total, int nblocks,
                                                         int dbg_skip_tile,                  // timing experiments only (-1)
                                                         unsigned long long* __restrict__ stamps,   // diagnostic, null in production
                                                         // exact template cache (BASELINE config 5): block 0's LN1 + qkv of the template rows is
                                                         // frame-invariant (the template tokens are; LN and the projections act per token).
                                                         // [B][len_z/16][9][64] f4: q, k, v^T images of each template tile.
                                                         float* __restrict__ zcache,
                                                         int zcache_mode,     // 0: off, 1: compute and store, 2: load instead of computing
                                                         const float* __restrict__ params3,   // BF3: the MLP's three-piece images, BLOCK3_STRIDE per block
                                                         unsigned* __restrict__ vlscr) {      // VP2L (the G256 A3 form): the low pieces of V^T, [B][depth][NC][NT / 2][64] x 16 B
    static_assert(NW * TPW >= NT, "tiles must be covered");
    // BF3L: the LDS-staged form (G128 balanced frame form: qkv + MLP); BF3G: weights from L2 as in every !WLDS form (G256: MLP only)
    constexpr bool BF3L = BF3 && WLDS, BF3G = BF3 && !WLDS;
    static_assert(!BF3L || (BAL && 2 * NT * NC >= W3_FC2_TILES - WBUF_TILES), "BF3 with staging: written for the balanced frame form; fc2's third output tile is staged in the K / V area");
    static_assert(!BAL || (WLDS && TPW == 1 && NW == 2 * (NT - 1)), "balanced variant: NT-1 owners + NT-1 guests");
    static_assert(!A3 || BF3, "A3: written for the three-piece forms");
    // A3: 16-byte units per key tile of the K pieces / per feature tile of the V^T pieces.  VP3 = V^T as pieces too (the staged G128 form);
    // the G256 form (BF3G) has LDS for the K pieces only (90 KiB) beside 60 KiB for V^T: q k^T and proj run on the bf16 pipe, and so does
    // P.V since VP2L (below), with V^T's low pieces outside LDS
    constexpr bool VP3 = A3 && BF3L;
    // VP2L (round 5): the G256 form's V^T as pieces after all -- the high and middle pieces take exactly the fp32 image's 60 KiB of LDS, the
    // LOW pieces (used by one of a product's six terms) go through a per-frame, per-block scratch in global memory: written by the wave
    // that computes them, read back (L2) as that term's A operand.  P.V then runs on the bf16 pipe: 180 instead of 240 matrix instructions
    // per query tile, each half as long.
    constexpr bool VP2L = A3 && BF3G;
    constexpr int KP_T16 = W3_FC1_OT16, VP_PAIRS = NT / 2, VP_T16 = VP_PAIRS * 3 * 64 + (NT & 1) * 3 * 32;
    constexpr int KV_UNITS = A3 ? NT * KP_T16 + (VP3 ? NC * VP_T16 : NC * NT * 64) : 2 * NT * NC * 64;
    constexpr int L = NT * 16;
    constexpr int NOWN = BAL ? NT - 1 : NT;                // tiles handled by owner waves
    constexpr int GT = NT - 1;                             // BAL: the guests' tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* Kimg = reinterpret_cast<f4*>(lds);                 // [NT][NC][64]   (BF3L: also fc2's third output tile, as pieces)
    f4* Vimg = Kimg + (A3 ? NT * KP_T16 : NT * NC * 64);   // [NC][NT][64]   (A3: behind the K pieces; VP3: pieces, addressed from Kimg)
    // the K and V^T images as STORED operands (f16 build: h4, half the bytes of the same area; written once per block, read by every wave)
    opnd* const Ko = reinterpret_cast<opnd*>(Kimg);
    opnd* const Vo = reinterpret_cast<opnd*>(Vimg);
    f4* Wa = Kimg + KV_UNITS;                              // WLDS: [36][64] staging buffer A
    constexpr int WA_TILES = BF3L ? W3_FC1_TILES : WBUF_TILES;      // BF3: fc1's three-piece image is 54 KiB
    f4* Wb = Wa + WA_TILES * 64;                           // WLDS: [36][64] staging buffer B (BF3: fc2's output tiles 0 and 1; tile 2 goes to the K / V area, free during the MLP)
    float* Sp = reinterpret_cast<float*>(WLDS ? Wb + WBUF_TILES * 64 : Wa);   // small parameters, small_floats(depth)
    // BAL: guest exchange areas.  A3: only Dg and the counter keep room of their own (header comment)
    f4* const Gx = reinterpret_cast<f4*>(Sp + small_floats(depth_total));
    f4* Qg = A3 ? Wb + W3_PROJ_TILES * 64 : Gx;                       // [NC][64]      q of the guest tile (A3: as pieces, KP_T16 units)
    f4* Pg = Qg + (A3 ? KP_T16 : NC * 64);                            // [4][NC][64]   attention / fc2 partials
    f4* Dg = A3 ? Gx : Pg + 4 * NC * 64;                              // [NC][64]      proj output tiles
    float* Mg = reinterpret_cast<float*>(A3 ? Pg + 4 * NC * 64 : Dg + NC * 64);              // [4][2][64]    partial softmax max / sum
    int* gflag = A3 ? reinterpret_cast<int*>(Dg + NC * 64) : reinterpret_cast<int*>(Mg + 4 * 2 * 64);   // guests' rendezvous counter
    f4* Pg2 = A3 ? Wa + W3_QKV_TILES * 64 : Pg;                       // [4][NC][64]   fc2 partial sums
    static_assert(!A3 || (W3_PROJ_TILES * 64 + KP_T16 + 4 * NC * 64 + 4 * 2 * 64 / 4 <= WBUF_TILES * 64 && W3_QKV_TILES * 64 + 4 * NC * 64 <= W3_FC1_TILES * 64),
                  "A3: the exchange areas fit behind proj's / qkv's image");

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok = lane & 15, q = lane >> 4;
    const int g = w - (NT - 1);                            // BAL: guest index of waves >= NT-1
    const float scale = 0.14433756729740643f;  // 48^-0.5  (head_dim ** -0.5, attn.py:15)
    constexpr float SCALE_LOG2E = 0.14433756729740643f * 1.4426950408889634f;
    const int guest_prio = dbg_skip_tile == -3 ? 0 : 1;   // -3: experiment, guests at default priority
    const bool fine = dbg_skip_tile == -2 || dbg_skip_tile <= -100;          // -2: per-stage stamps; -(100 + t): per-stage stamps and tile t skipped
    if (dbg_skip_tile <= -100) dbg_skip_tile = -100 - dbg_skip_tile;
    // diagnostic phase stamps: shader-clock reads by lane 0 of every wave, [b][w][64]
    int nstamp = 0;
    auto stamp = [&]() {
        if (stamps != nullptr) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (lane == 0) stamps[((size_t)b * NW + w) * 64 + nstamp] = t;
            ++nstamp;
        }
    };
    auto fstamp = [&]() { if (fine) stamp(); };     // per-stage stamps (VT_DBG_STAMPS=2)
    stamp();
    // staging of an image of `ntiles` stored-operand tiles of block `blk_of` into an LDS buffer, as 1 KiB DMA pieces (fp32 build:
    // one per float4 tile of `params`; f16 build: one per TWO h4 tiles of the pre-converted images in `params3`, BLOCK_STRIDE halves
    // per block at the float layout's offsets -- an odd tile count reads 512 bytes past the image, inside the buffer)
    auto stage_img = [&](f4* dst, int off, int ntiles, int blk_of) {
        if constexpr (VT_IS_F16)
            stage_tiles(dst, reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(params3) + (size_t)blk_of * BLOCK_STRIDE + off), (ntiles + 1) / 2, w, NW, lane);
        else
            stage_tiles(dst, params + (size_t)blk_of * BLOCK_STRIDE + off, ntiles, w, NW, lane);
    };
    if constexpr (BF3L) stage_tiles(Wa, params3 + (W3_FC1_TILES + W3_FC2_TILES) * 256, W3_QKV_TILES, w, NW, lane);
    else if constexpr (WLDS) stage_img(Wa, O_WQKV, 9 * NC, 0);   // block 0's qkv weights
    for (int i = threadIdx.x; 4 * i < small_floats(depth_total); i += NW * 64)
        st4(Sp + 4 * i, ld4(params + small_src(4 * i, depth_total)));

    f4 x[TPW][NC];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int T = w + NW * i;
        if (T < NOWN && T != dbg_skip_tile) {
            const float* src = tokens + ((size_t)b * L + 16 * T + tok) * C + 4 * q;
#pragma unroll
            for (int c = 0; c < NC; ++c) x[i][c] = ld4(src + 16 * c);
        }
    }
    f4 x4[NC];                                   // BAL: every guest's copy of the guest tile's residual stream
    if constexpr (BAL) {
        if (w >= NOWN) {
            const float* src = tokens + ((size_t)b * L + 16 * GT + tok) * C + 4 * q;
#pragma unroll
            for (int c = 0; c < NC; ++c) x4[c] = ld4(src + 16 * c);
        }
        if (threadIdx.x == 0) *gflag = 0;
        // the guests' short chains sit on every phase's critical path (the owners wait for them at the
        // barriers), but the issue arbiter favours the older owner waves: raise the guests' priority
        if (w >= NOWN && guest_prio > 0) __builtin_amdgcn_s_setprio(2);
    }
    barrier_publish<WLDS>();

    const int lane_k = lane;
    for (int blk = 0; blk < nblocks; ++blk) {
        // BF3: every per-lane address inside a block is derived from a fresh (opaque) copy of the lane index -- the kernel sits at the
        // 256-register cap, and as invariants of the block loop hipcc computed ~40 LDS addresses up front and spilled them; their
        // reloads wait on vmcnt, i.e. on the weight staging in flight.  Recomputing them costs a few dozen VALU instructions per block.
        int lane_s = lane_k;
        if constexpr (BF3) asm volatile("" : "+v"(lane_s));
        const int lane = lane_s, tok = lane & 15, q = lane >> 4;
        const float* __restrict__ P = params + (size_t)blk * BLOCK_STRIDE;
        const float* S = Sp + blk * SMALL_STRIDE;
        // weight operand image `t` of each GEMM: from the staging buffers (WLDS) or straight from L2
        // (stored operands, vt_common.h `opnd`: the fp32 build reads the float4 images of `params`; the f16 build the same images
        // converted once at vt_load_weights -- `params3` then holds them, BLOCK_STRIDE halves per block at the float layout's offsets)
        const opnd* const Po = VT_IS_F16 ? reinterpret_cast<const opnd*>(reinterpret_cast<const _Float16*>(params3) + (size_t)blk * BLOCK_STRIDE)
                                         : reinterpret_cast<const opnd*>(P);
        const opnd* const Wa_o = reinterpret_cast<const opnd*>(Wa);
        const opnd* const Wb_o = reinterpret_cast<const opnd*>(Wb);
        auto w_qkv = [&](int t) { return WLDS ? Wa_o[t * 64 + lane] : wimg_o(Po + O_WQKV / 4, t, lane); };
        auto w_proj = [&](int t) { return WLDS ? Wb_o[t * 64 + lane] : wimg_o(Po + O_WPROJ / 4, t, lane); };
        auto w_fc1 = [&](int t) { return WLDS ? Wa_o[t * 64 + lane] : wimg_o(Po + O_W1 / 4, t, lane); };
        auto w_fc2 = [&](int t) { return WLDS ? Wb_o[t * 64 + lane] : wimg_o(Po + O_W2 / 4, t, lane); };
        // BF3 staging: the same four bursts per block as the fp32 form, issued by all eight waves at the start of the phase before the
        // one that reads them (fc1 54 KiB -> Wa during attention; fc2 36 KiB -> Wb + 18 KiB -> the K / V area during the fc1 phase).
        // The bursts cost ~1.3 us per launch (VT_BLK_NOSTAGE = 2); every other placement measured slower or equal.
        const float* __restrict__ P3 = BF3 ? params3 + (size_t)blk * BLOCK3_STRIDE : nullptr;
        constexpr int PROJ_TILES = NC * NC;
        using vt3::u32x2;
        using vt3::u32x4;
        // the images' per-lane addresses are built from a fresh copy of the lane index in every block: as invariants of the block
        // loop hipcc computed ~38 of them up front and spilled them (reloads wait on vmcnt, i.e. on the weight staging in flight)
        int lane3 = lane;
        if constexpr (BF3) asm volatile("" : "+v"(lane3));
        // one opaque base register per (region, lane stride): the images' tile / piece offsets then sit in the instructions' offset fields
        const auto Wa3 = BF3L ? lds_lane_base<u32x4>(Wa, 16u * lane3) : nullptr;                       // pair-0 pieces, 16 B per lane
        const auto Wa3h = BF3L ? lds_lane_base<u32x2>(Wa, 192u * 16u + 8u * lane3) : nullptr;          // chunk-2 pieces, 8 B per lane
        const auto Wb3 = BF3L ? lds_lane_base<u32x4>(Wb, 16u * lane3) : nullptr;
        const auto Wk3 = BF3L ? lds_lane_base<u32x4>(Kimg, 16u * lane3) : nullptr;
        auto w1_load3 = [&](int t, u32x4 (&a0)[3], u32x2 (&a2)[3]) {       // fc1 pieces of output tile t: chunk pair 0, chunk 2
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                a0[pc] = Wa3[t * W3_FC1_OT16 + pc * 64];
                a2[pc] = Wa3h[t * W3_FC1_OT16 * 2 + pc * 64];
            }
        };
        auto w2_load3 = [&](int p, u32x4 (&a)[NC][3]) {                     // fc2 pieces of chunk pair p, all three output tiles
#pragma unroll
            for (int ot = 0; ot < NC; ++ot)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    a[ot][pc] = ot < 2 ? Wb3[((ot * (NH / 2) + p) * 3 + pc) * 64] : Wk3[(p * 3 + pc) * 64];
        };
        auto split_h3 = [&](const f4 (&h)[NC], u32x4 (&hb)[3], u32x2 (&hc)[3]) {       // LN2's output as fc1's B operands
            u32x2 a[3], b2[3];
            vt3::split3(h[0], a[0], a[1], a[2]);
            vt3::split3(h[1], b2[0], b2[1], b2[2]);
            vt3::split3(h[2], hc[0], hc[1], hc[2]);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) hb[pc] = u32x4{a[pc].x, a[pc].y, b2[pc].x, b2[pc].y};
        };
        // one 16 x 16 output tile of a K = 48 layer: 12 MFMAs as two chains (chunk pair 0 onto `init`; chunk 2, on the K = 16
        // instruction, onto zero), added.  SWAP: the activation pieces are the A operand (rows = tokens): v, so that V^T comes out.
        auto tile48 = [&](auto swap, const u32x4 (&a0)[3], const u32x2 (&a2)[3], const u32x4 (&hb)[3], const u32x2 (&hc)[3], f4 init) {
            constexpr bool SWAP = decltype(swap)::value;
            constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
            f4 accA = init, accB = splat4(0.f);
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                accB = SWAP ? vt3::mma16(hc[TX[e]], a2[TW[e]], accB) : vt3::mma16(a2[TW[e]], hc[TX[e]], accB);
                accA = SWAP ? vt3::mma(hb[TX[e]], a0[TW[e]], accA) : vt3::mma(a0[TW[e]], hb[TX[e]], accA);
            }
            return accA + accB;
        };
        // ---- A3: K / V^T / q / proj as three-piece operands (layouts: header comment)
        const auto Kp3 = A3 ? lds_lane_base<u32x4>(Kimg, 16u * lane3) : nullptr;                                   // key tile J, pair-0 piece pc: [J * KP_T16 + pc * 64]
        const auto Kp3h = A3 ? lds_lane_base<u32x2>(Kimg, 192u * 16u + 8u * lane3) : nullptr;                      // chunk-2 piece: [J * KP_T16 * 2 + pc * 64]
        const auto Vp3 = A3 ? lds_lane_base<u32x4>(Kimg + NT * KP_T16, 16u * lane3) : nullptr;                     // feature tile t, pair p: [t * VP_T16 + (p * 3 + pc) * 64]
        const auto Vp3h = A3 ? lds_lane_base<u32x2>(Kimg + NT * KP_T16, VP_PAIRS * 3 * 64 * 16u + 8u * lane3) : nullptr;   // odd last chunk: [t * VP_T16 * 2 + pc * 64]
        auto k_load3 = [&](int J, u32x4 (&a0)[3], u32x2 (&a2)[3]) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                a0[pc] = Kp3[J * KP_T16 + pc * 64];
                a2[pc] = Kp3h[J * KP_T16 * 2 + pc * 64];
            }
        };
        auto wp_load3 = [&](int t, u32x4 (&a0)[3], u32x2 (&a2)[3]) {       // proj pieces of output tile t (fc1's layout): buffer B, or straight from L2 (BF3G)
            if constexpr (BF3L) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    a0[pc] = Wb3[t * W3_FC1_OT16 + pc * 64];
                    a2[pc] = reinterpret_cast<const u32x2*>(Wb)[(t * W3_FC1_OT16 + 192) * 2 + pc * 64 + lane3];
                }
            } else {
                const u32x4* const Gp = reinterpret_cast<const u32x4*>(P3) + (W3_FC1_TILES + W3_FC2_TILES + W3_QKV_TILES) * 64;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    a0[pc] = Gp[t * W3_FC1_OT16 + pc * 64 + lane3];
                    a2[pc] = reinterpret_cast<const u32x2*>(Gp)[(t * W3_FC1_OT16 + 192) * 2 + pc * 64 + lane3];
                }
            }
        };
        auto store_k3 = [&](f4* dst_tile, const f4 (&kr)[NC]) {            // a token tile's k (or the guests' q) as pieces: one "output tile" of fc1's layout
            u32x4 kb[3];
            u32x2 kc[3];
            split_h3(kr, kb, kc);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                reinterpret_cast<u32x4*>(dst_tile)[pc * 64 + lane3] = kb[pc];
                reinterpret_cast<u32x2*>(dst_tile + 192)[pc * 64 + lane3] = kc[pc];
            }
        };
        auto store_k3c = [&](f4* dst_tile, int ot, f4 r) {                  // the same, one feature chunk at a time (no three-chunk collection)
            u32x2 pk[3];
            vt3::split3(r, pk[0], pk[1], pk[2]);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                if (ot < 2) reinterpret_cast<u32x2*>(dst_tile)[(pc * 64 + lane3) * 2 + ot] = pk[pc];
                else reinterpret_cast<u32x2*>(dst_tile + 192)[pc * 64 + lane3] = pk[pc];
            }
        };
        // unit offset (u32x2) of key chunk J's half inside feature tile t's V^T pieces, piece pc at + pc * 128 (pair) / + pc * 64 (odd chunk)
        auto store_v3 = [&](int T, int ot, f4 r) {                          // V^T chunk T (this wave's token tile) of feature tile ot
            u32x2 pv[3];
            vt3::split3(r, pv[0], pv[1], pv[2]);
            u32x2* base = reinterpret_cast<u32x2*>(Kimg + NT * KP_T16 + ot * VP_T16);
            if ((NT & 1) && T == NT - 1) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) base[VP_PAIRS * 3 * 64 * 2 + pc * 64 + lane3] = pv[pc];
            } else {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) base[(((T >> 1) * 3 + pc) * 64 + lane3) * 2 + (T & 1)] = pv[pc];
            }
        };
        // VP2L: V^T chunk T of feature tile ot -- h and m to LDS as [feature tile][chunk pair][piece 2][64 lanes] x 16 B (a pair's two chunks are
        // the two halves of a lane's 16 bytes), l to this block's scratch plane [feature tile][chunk pair][64 lanes] x 16 B
        constexpr int VL_BLK_U4 = NC * VP_PAIRS * 64;
        u32x4* const vl_blk = VP2L ? reinterpret_cast<u32x4*>(vlscr) + ((size_t)b * depth_total + blk) * VL_BLK_U4 : nullptr;
        auto store_v2l = [&](int T, int ot, f4 r) {
            u32x2 pv[3];
            vt3::split3(r, pv[0], pv[1], pv[2]);
            const int pp = T >> 1, hf = T & 1;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) reinterpret_cast<u32x2*>(Vimg)[((((ot * VP_PAIRS + pp) * 2 + pc) * 64) + lane3) * 2 + hf] = pv[pc];
            reinterpret_cast<u32x2*>(vl_blk)[(((ot * VP_PAIRS + pp) * 64) + lane3) * 2 + hf] = pv[2];
        };
        if constexpr (A3 && BF3L) {
            stage_tiles(Wb, P3 + (W3_FC1_TILES + W3_FC2_TILES + W3_QKV_TILES) * 256, W3_PROJ_TILES, w, NW, lane, blk == 0);
        } else if constexpr (BF3L) {
            stage_tiles(Wb, P + O_WPROJ, PROJ_TILES, w, NW, lane, blk == 0);
        } else if constexpr (WLDS) stage_img(Wb, O_WPROJ, NC * NC, blk);   // proj: free since the last barrier

        f4 qr[TPW][NC];
        // ---- LN1 + QKV; publish K / V^T images ------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int T = w + NW * i;
            const bool z_tile = ZC && blk == 0 && 16 * T < len_z;       // wave-uniform; compiled out of the default kernel
            // the cache pointer is rebuilt from a fresh copy of the lane index where it is used (block 0 only): as a loop invariant
            // of the block loop it stayed in registers for the whole kernel (the G256 variant spilled 68 B / lane)
            int lane_z = lane;
            if constexpr (ZC) asm volatile("" : "+v"(lane_z));
            f4* const zc = reinterpret_cast<f4*>(zcache) + (((size_t)b * (len_z >> 4) + T) * 3 * NC) * 64 + lane_z;
            if (VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && z_tile && zcache_mode == 2) {
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) {
                    qr[i][ot] = zc[ot * 64];
                    if constexpr (VP3) store_v3(T, ot, zc[(2 * NC + ot) * 64]);
                    else if constexpr (VP2L) store_v2l(T, ot, zc[(2 * NC + ot) * 64]);
                    else Vo[(ot * NT + T) * 64 + lane] = to_opnd(zc[(2 * NC + ot) * 64]);
                    if constexpr (!A3) Ko[(T * NC + ot) * 64 + lane] = to_opnd(zc[(NC + ot) * 64]);
                }
                if constexpr (A3) {
                    f4 kr[NC];
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) kr[ot] = zc[(NC + ot) * 64];
                    store_k3(Kimg + T * KP_T16, kr);
                }
            } else if (VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile) {
                f4 h[NC];
                layer_norm_plain(x[i], h);
                fstamp();
                if constexpr (BF3) {
                    u32x4 hb[3];
                    u32x2 hc[3];
                    split_h3(h, hb, hc);
                    // qkv pieces of output tile t: from the staging buffer (BF3L) or straight from L2 (BF3G), the next tile's in flight
                    const u32x4* const Gq = reinterpret_cast<const u32x4*>(P3) + (W3_FC1_TILES + W3_FC2_TILES) * 64;
                    const u32x2* const Gqh = reinterpret_cast<const u32x2*>(Gq);
                    auto wq_load3 = [&](int t, u32x4 (&a0)[3], u32x2 (&a2)[3]) {
                        if constexpr (BF3L) w1_load3(t, a0, a2);
                        else {
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) {
                                a0[pc] = Gq[t * W3_FC1_OT16 + pc * 64 + lane3];
                                a2[pc] = Gqh[(t * W3_FC1_OT16 + 192) * 2 + pc * 64 + lane3];
                            }
                        }
                    };
                    u32x4 a0[2][3];
                    u32x2 a2[2][3];
                    // a tile's accumulator initialiser (its bias) is requested WITH its weight pieces, one tile ahead: requested beside
                    // the next tile's pieces it was the youngest LDS read in flight, so the wait in front of the tile's first MFMA was
                    // lgkmcnt(0) -- for the prefetch just issued: one exposed LDS round trip per tile (round 4)
                    auto bias_q = [&](int t) { return t < 2 * NC ? ld4(S + S_BQKV + 16 * t + 4 * q) : splat4(S[S_BQKV + 2 * C + 16 * (t % NC) + tok]); };
                    f4 bias[2];
                    f4 kr[NC];           // A3: k of this token tile, published as pieces once all three feature chunks are there
                    wq_load3(0, a0[0], a2[0]);
                    bias[0] = bias_q(0);
#pragma unroll
                    for (int t = 0; t < 3 * NC; ++t) {
                        if (t + 1 < 3 * NC) {
                            wq_load3(t + 1, a0[(t + 1) & 1], a2[(t + 1) & 1]);
                            bias[(t + 1) & 1] = bias_q(t + 1);
                        }
                        const int ot = t % NC;
                        f4 r;
                        __builtin_amdgcn_sched_barrier(0);
                        if (t < 2 * NC) r = tile48(std::false_type{}, a0[t & 1], a2[t & 1], hb, hc, bias[t & 1]);
                        else r = tile48(std::true_type{}, a0[t & 1], a2[t & 1], hb, hc, bias[t & 1]);
                        if (t < NC) qr[i][ot] = r;
                        else if constexpr (A3) {
                            if (t < 2 * NC) {
                                if constexpr (VP3) {
                                    kr[ot] = r;
                                    if (ot == NC - 1) store_k3(Kimg + T * KP_T16, kr);
                                } else store_k3c(Kimg + T * KP_T16, ot, r);      // the 20-tile form has no registers for kr
                            } else if constexpr (VP3) store_v3(T, ot, r);
                            else if constexpr (VP2L) store_v2l(T, ot, r);
                            else Vo[(ot * NT + T) * 64 + lane] = to_opnd(r);
                        }
                        else if (t < 2 * NC) Ko[(T * NC + ot) * 64 + lane] = to_opnd(r);
                        else Vo[(ot * NT + T) * 64 + lane] = to_opnd(r);
                        if (z_tile && zcache_mode == 1) zc[t * 64] = r;
                        if (t == 2 * NC - 1) fstamp();
                    }
                } else {
                {   // q and k: 6 independent chains, rows = features, cols = tokens (B = h shared)
                    f4 acc[2 * NC];
#pragma unroll
                    for (int ot = 0; ot < 2 * NC; ++ot) acc[ot] = ld4(S + S_BQKV + 16 * ot + 4 * q);
                    gemm_stage<NC, 2 * NC, true, WLDS>(
                        [&](int c, opnd (&a)[2 * NC]) {
#pragma unroll
                            for (int ot = 0; ot < 2 * NC; ++ot) a[ot] = w_qkv(ot * NC + c);
                        },
                        [&](int c) { return h[c]; }, acc);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) {
                        qr[i][ot] = acc[ot];
                        Ko[(T * NC + ot) * 64 + lane] = to_opnd(acc[NC + ot]);
                        if (z_tile && zcache_mode == 1) { zc[ot * 64] = acc[ot]; zc[(NC + ot) * 64] = acc[NC + ot]; }
                    }
                }
                fstamp();
                {   // v, operands swapped: rows = tokens, cols = v features (A = h shared) -> V^T image
                    f4 acc[NC];
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) acc[ot] = splat4(S[S_BQKV + 2 * C + 16 * ot + tok]);
                    gemm_stage<NC, NC, false, WLDS>(
                        [&](int c, opnd (&bw)[NC]) {
#pragma unroll
                            for (int ot = 0; ot < NC; ++ot) bw[ot] = w_qkv((2 * NC + ot) * NC + c);
                        },
                        [&](int c) { return h[c]; }, acc);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) {
                        Vo[(ot * NT + T) * 64 + lane] = to_opnd(acc[ot]);
                        if (z_tile && zcache_mode == 1) zc[(2 * NC + ot) * 64] = acc[ot];
                    }
                }
                }
            }
        }
        if constexpr (BAL) {
            if (BF3L && VT_BLK_GUESTS && w >= NOWN && g < 3) {
                f4 h[NC];
                layer_norm_plain(x4, h);
                u32x4 hb[3];
                u32x2 hc[3];
                split_h3(h, hb, hc);
                u32x4 a0[NC][3];
                u32x2 a2[NC][3];
#pragma unroll
                for (int j = 0; j < NC; ++j) w1_load3(NC * g + j, a0[j], a2[j]);
                f4 r[NC];
                if (g < 2) {
                    f4 bias[NC];
#pragma unroll
                    for (int j = 0; j < NC; ++j) bias[j] = ld4(S + S_BQKV + 16 * (g * NC + j) + 4 * q);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NC; ++j) r[j] = tile48(std::false_type{}, a0[j], a2[j], hb, hc, bias[j]);
                    if constexpr (A3) {
                        store_k3(g == 0 ? Qg : Kimg + GT * KP_T16, r);
                    } else {
#pragma unroll
                        for (int j = 0; j < NC; ++j) {
                            if (g == 0) Qg[j * 64 + lane] = r[j];
                            else Ko[(GT * NC + j) * 64 + lane] = to_opnd(r[j]);
                        }
                    }
                } else {
                    f4 bias[NC];
#pragma unroll
                    for (int j = 0; j < NC; ++j) bias[j] = splat4(S[S_BQKV + 2 * C + 16 * j + tok]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NC; ++j) r[j] = tile48(std::true_type{}, a0[j], a2[j], hb, hc, bias[j]);
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        if constexpr (A3) store_v3(GT, j, r[j]);
                        else Vo[(j * NT + GT) * 64 + lane] = to_opnd(r[j]);
                    }
                }
            } else if (VT_BLK_GUESTS && w >= NOWN && g < 3) {      // guest 0: q, guest 1: k, guest 2: v of the guest tile
                f4 h[NC];
                layer_norm_plain(x4, h);
                f4 acc[NC];
                if (g < 2) {
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) acc[ot] = ld4(S + S_BQKV + 16 * (g * NC + ot) + 4 * q);
                    gemm_stage<NC, NC, true, true>(
                        [&](int c, opnd (&a)[NC]) {
#pragma unroll
                            for (int ot = 0; ot < NC; ++ot) a[ot] = w_qkv((g * NC + ot) * NC + c);
                        },
                        [&](int c) { return h[c]; }, acc);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) {
                        if (g == 0) Qg[ot * 64 + lane] = acc[ot];
                        else Ko[(GT * NC + ot) * 64 + lane] = to_opnd(acc[ot]);
                    }
                } else {
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) acc[ot] = splat4(S[S_BQKV + 2 * C + 16 * ot + tok]);
                    gemm_stage<NC, NC, false, true>(
                        [&](int c, opnd (&bw)[NC]) {
#pragma unroll
                            for (int ot = 0; ot < NC; ++ot) bw[ot] = w_qkv((2 * NC + ot) * NC + c);
                        },
                        [&](int c) { return h[c]; }, acc);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) Vo[(ot * NT + GT) * 64 + lane] = to_opnd(acc[ot]);
                }
            }
        }
        stamp();            // QKV done
        if constexpr (VP2L) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the low V^T pieces have reached L2 (measured: free -- the phase's stamps do not move)
        barrier_publish<WLDS>();    // K/V published; proj weights landed; buffer A free
        stamp();
        if constexpr (BF3L) {
            stage_tiles(Wa, P3, W3_FC1_TILES, w, NW, lane, blk == 0);
        } else if constexpr (WLDS) stage_img(Wa, O_W1, NH * NC, blk);      // fc1 weights
        // In the last block the template rows only matter as keys / values: their attention
        // output, proj and MLP never reach the head (vit_dist.py:126 keeps the search rows only),
        // so those tiles stop after publishing K / V -- unless the caller asked for the residual.
        const bool last_skip_z = (blk == depth_total - 1) && (resid == nullptr);
        // ---- attention + proj (residual add) --------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int T = w + NW * i;
            if (A3 && VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) {
                // A3: the same attention + proj as six-term bf16 products.  q is split here (it stayed fp32 across the barrier: 12 instead
                // of 18 registers), the scores' exponentials are split as they are produced, o after its normalisation.
                constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
                u32x4 qb[3];
                u32x2 qc[3];
                split_h3(qr[i], qb, qc);
                f4 s[NT];
                float m0 = -3.0e38f, m1 = -3.0e38f;
                {
                    // the next key tile's pieces in flight (two register sets) where the registers are there; the 20-tile form, which holds
                    // three tiles' residual streams and q, reads a tile's pieces where it uses them (its SIMD partner covers the wait)
                    constexpr int NB = VP3 ? 2 : 1;
                    u32x4 a0[NB][3];
                    u32x2 a2[NB][3];
                    if (NB == 2) k_load3(0, a0[0], a2[0]);
#pragma unroll
                    for (int J = 0; J < NT; ++J) {       // S^T tiles: rows = keys (A = K pieces), cols = queries (B = q pieces)
                        if (NB == 2 && J + 1 < NT) k_load3(J + 1, a0[(J + 1) % NB], a2[(J + 1) % NB]);
                        if (NB == 1) k_load3(J, a0[0], a2[0]);
                        __builtin_amdgcn_sched_barrier(0);
                        s[J] = tile48(std::false_type{}, a0[J % NB], a2[J % NB], qb, qc, splat4(0.f));
                        m0 = fmaxf(fmaxf(m0, s[J].x), s[J].y);
                        m1 = fmaxf(fmaxf(m1, s[J].z), s[J].w);
                    }
                }
                fstamp();
                const float m = quad_max(fmaxf(m0, m1));
                const f2 k2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-m * SCALE_LOG2E, -m * SCALE_LOG2E};
                f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
                f4 o[NC];
                // proj's pieces of the first output tile are requested under the last MFMAs of P.V (VP3; the 20-tile form reads them where
                // it uses them, one register set)
                constexpr int NWB = VP3 ? 2 : 1;
                u32x4 wa0[NWB][3];
                u32x2 wa2[NWB][3];
                if constexpr (VP3) {
                    u32x4 pb[VP_PAIRS > 0 ? VP_PAIRS : 1][3];      // P^T as pieces: key-chunk pairs ...
                    u32x2 po[3], plo[3];                            // ... the odd last chunk; a pair's first chunk
                    // one key-chunk pair's V^T pieces (three feature tiles: 36 registers) at a time; the first pair's are requested in front
                    // of the exponentials
                    u32x4 v[NC][3];
                    auto v_load3 = [&](int pp) {
#pragma unroll
                        for (int t = 0; t < NC; ++t)
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) v[t][pc] = Vp3[t * VP_T16 + (pp * 3 + pc) * 64];
                    };
                    if (VP_PAIRS > 0) v_load3(0);
#pragma unroll
                    for (int J = 0; J < NT; ++J) {
                        const f2 a = __builtin_elementwise_fma(f2{s[J].x, s[J].y}, k2, nm2);
                        const f2 c = __builtin_elementwise_fma(f2{s[J].z, s[J].w}, k2, nm2);
                        const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                        const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
                        d0 += ea;
                        d1 += ec;
                        u32x2 pj[3];
                        vt3::split3(f4{ea.x, ea.y, ec.x, ec.y}, pj[0], pj[1], pj[2]);
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            if ((NT & 1) && J == NT - 1) po[pc] = pj[pc];
                            else if (J & 1) pb[J >> 1][pc] = u32x4{plo[pc].x, plo[pc].y, pj[pc].x, pj[pc].y};
                            else plo[pc] = pj[pc];
                        }
                    }
                    const f2 dd = d0 + d1;
                    const float rden = __builtin_amdgcn_rcpf(quad_sum(dd.x + dd.y));     // v_rcp_f32: 1 ulp
                    fstamp();
                    f4 oA[NC], oB[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) oA[t] = oB[t] = splat4(0.f);
                    u32x2 vh[NC][3];
                    if (NT & 1) {
#pragma unroll
                        for (int t = 0; t < NC; ++t)
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) vh[t][pc] = Vp3h[t * VP_T16 * 2 + pc * 64];
                    }
#pragma unroll
                    for (int pp = 0; pp < VP_PAIRS; ++pp) {        // O^T = V^T P^T: 3 feature-tile chains share B = a pair's P pieces
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < 6; ++e)
#pragma unroll
                            for (int t = 0; t < NC; ++t) oA[t] = vt3::mma(v[t][TW[e]], pb[pp][TX[e]], oA[t]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (pp + 1 < VP_PAIRS) v_load3(pp + 1);
                    }
                    wp_load3(0, wa0[0], wa2[0]);
                    if (NT & 1) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < 6; ++e)
#pragma unroll
                            for (int t = 0; t < NC; ++t) oB[t] = vt3::mma16(vh[t][TW[e]], po[TX[e]], oB[t]);
                    }
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = (oA[t] + oB[t]) * splat4(rden);
                } else if constexpr (VP2L) {
                    // O^T = V^T P^T pair by pair: a key-chunk pair's exponentials are split where they are produced (12 registers of pieces
                    // alive at a time, not the 120 of all ten pairs) and go straight into the pair's 18 MFMAs; V^T's h / m pieces from LDS, its l
                    // pieces from the block's scratch plane, requested a pair ahead
                    const auto Vp2 = lds_lane_base<u32x4>(Vimg, 16u * lane3);      // [((t * VP_PAIRS + pp) * 2 + pc) * 64]
                    const u32x4* const vlg = vl_blk + lane3;
                    u32x4 vl[2][NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) vl[0][t] = vlg[(t * VP_PAIRS) * 64];
                    f4 oA[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) oA[t] = splat4(0.f);
#pragma unroll
                    for (int pp = 0; pp < VP_PAIRS; ++pp) {
                        if (pp + 1 < VP_PAIRS) {
#pragma unroll
                            for (int t = 0; t < NC; ++t) vl[(pp + 1) & 1][t] = vlg[(t * VP_PAIRS + pp + 1) * 64];
                        }
                        u32x4 v[NC][2];
#pragma unroll
                        for (int t = 0; t < NC; ++t)
#pragma unroll
                            for (int pc = 0; pc < 2; ++pc) v[t][pc] = Vp2[((t * VP_PAIRS + pp) * 2 + pc) * 64];
                        u32x2 pj[2][3];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int J = 2 * pp + jj;
                            const f2 a = __builtin_elementwise_fma(f2{s[J].x, s[J].y}, k2, nm2);
                            const f2 c = __builtin_elementwise_fma(f2{s[J].z, s[J].w}, k2, nm2);
                            const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                            const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
                            d0 += ea;
                            d1 += ec;
                            vt3::split3(f4{ea.x, ea.y, ec.x, ec.y}, pj[jj][0], pj[jj][1], pj[jj][2]);
                        }
                        u32x4 pb[3];
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) pb[pc] = u32x4{pj[0][pc].x, pj[0][pc].y, pj[1][pc].x, pj[1][pc].y};
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < 6; ++e)
#pragma unroll
                            for (int t = 0; t < NC; ++t) oA[t] = vt3::mma(TW[e] == 2 ? vl[pp & 1][t] : v[t][TW[e] & 1], pb[TX[e]], oA[t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f2 dd = d0 + d1;
                    const float rden = __builtin_amdgcn_rcpf(quad_sum(dd.x + dd.y));
                    fstamp();
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = oA[t] * splat4(rden);
                } else {
                    // V^T stays an fp32 image: exponentials in place, O^T = V^T P^T on fp32 MFMAs
#pragma unroll
                    for (int J = 0; J < NT; ++J) {
                        const f2 a = __builtin_elementwise_fma(f2{s[J].x, s[J].y}, k2, nm2);
                        const f2 c = __builtin_elementwise_fma(f2{s[J].z, s[J].w}, k2, nm2);
                        const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                        const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
                        s[J] = f4{ea.x, ea.y, ec.x, ec.y};
                        d0 += ea;
                        d1 += ec;
                    }
                    const f2 dd = d0 + d1;
                    const float rden = __builtin_amdgcn_rcpf(quad_sum(dd.x + dd.y));
                    fstamp();
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
                    gemm_stage<NT, NC, true, WLDS>(
                        [&](int J, opnd (&a)[NC]) {
#pragma unroll
                            for (int t = 0; t < NC; ++t) a[t] = Vo[(t * NT + J) * 64 + lane];
                        },
                        [&](int J) { return s[J]; }, o);
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = o[t] * splat4(rden);
                }
                fstamp();
                u32x4 ob[3];
                u32x2 oc[3];
                split_h3(o, ob, oc);
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) {
                    if (NWB == 2 && ot + 1 < NC) wp_load3(ot + 1, wa0[(ot + 1) % NWB], wa2[(ot + 1) % NWB]);
                    if (NWB == 1) wp_load3(ot, wa0[0], wa2[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    x[i][ot] = tile48(std::false_type{}, wa0[ot % NWB], wa2[ot % NWB], ob, oc, x[i][ot] + ld4(S + S_BPROJ + 16 * ot + 4 * q));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (!A3 && VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) {
                // softmax((q k^T) * scale) (attn.py:40-41) as exp2(raw * (scale log2 e) - max_raw * (scale log2 e)): the scale, the
                // subtraction and exp's own log2 e factor become ONE packed fma per two scores, the row maximum runs on v_max3 and the
                // row sum on packed adds -- 10 instead of 22 VALU instructions per score tile (scale > 0: the raw maximum is the maximum)
                f4 s[NT];
                float m0 = -3.0e38f, m1 = -3.0e38f;
                constexpr int JG = 5;                    // key tiles per group = independent chains
                static_assert(NT % JG == 0, "NT must be a multiple of 5");
#pragma unroll
                for (int j0 = 0; j0 < NT; j0 += JG) {    // S^T tiles: rows = keys, cols = queries (B = q shared)
                    f4 acc[JG];
#pragma unroll
                    for (int j = 0; j < JG; ++j) acc[j] = splat4(0.f);
                    gemm_stage<NC, JG, true, WLDS>(
                        [&](int c, opnd (&a)[JG]) {
#pragma unroll
                            for (int j = 0; j < JG; ++j) a[j] = Ko[((j0 + j) * NC + c) * 64 + lane];
                        },
                        [&](int c) { return qr[i][c]; }, acc);
#pragma unroll
                    for (int j = 0; j < JG; ++j) {
                        s[j0 + j] = acc[j];
                        m0 = fmaxf(fmaxf(m0, acc[j].x), acc[j].y);
                        m1 = fmaxf(fmaxf(m1, acc[j].z), acc[j].w);
                    }
                }
                fstamp();
                const float m = quad_max(fmaxf(m0, m1));
                const f2 k2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-m * SCALE_LOG2E, -m * SCALE_LOG2E};
                f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    const f2 a = __builtin_elementwise_fma(f2{s[J].x, s[J].y}, k2, nm2);
                    const f2 c = __builtin_elementwise_fma(f2{s[J].z, s[J].w}, k2, nm2);
                    const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                    const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
                    s[J] = f4{ea.x, ea.y, ec.x, ec.y};
                    d0 += ea;
                    d1 += ec;
                }
                const f2 dd = d0 + d1;
                const float den = dd.x + dd.y;
                const float rden = __builtin_amdgcn_rcpf(quad_sum(den));     // v_rcp_f32: 1 ulp
                fstamp();
                f4 o[NC];
#pragma unroll
                for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
                gemm_stage<NT, NC, true, WLDS>(                // O^T = V^T P^T: 3 feature-tile chains share B = P_J
                    [&](int J, opnd (&a)[NC]) {
#pragma unroll
                        for (int t = 0; t < NC; ++t) a[t] = Vo[(t * NT + J) * 64 + lane];
                    },
                    [&](int J) { return s[J]; }, o);
#pragma unroll
                for (int t = 0; t < NC; ++t) o[t] = o[t] * splat4(rden);
                fstamp();
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(S + S_BPROJ + 16 * ot + 4 * q);
                gemm_stage<NC, NC, true, WLDS>(
                    [&](int c, opnd (&a)[NC]) {
#pragma unroll
                        for (int ot = 0; ot < NC; ++ot) a[ot] = w_proj(ot * NC + c);
                    },
                    [&](int c) { return o[c]; }, x[i]);
            }
        }
        if constexpr (BAL) {
            if (VT_BLK_GUESTS && w >= NOWN) {
                // partial attention of the guest queries over this guest's key tiles
                auto attn_part = [&](auto njc, int J0) {
                    constexpr int NJ = decltype(njc)::value;
                    f4 qv[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) qv[c] = Qg[c * 64 + lane];
                    f4 sc[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) sc[j] = splat4(0.f);
                    gemm_stage<NC, NJ, true, true>(
                        [&](int c, opnd (&a)[NJ]) {
#pragma unroll
                            for (int j = 0; j < NJ; ++j) a[j] = Ko[((J0 + j) * NC + c) * 64 + lane];
                        },
                        [&](int c) { return qv[c]; }, sc);
                    float m0 = -3.0e38f, m1 = -3.0e38f;          // raw scores, as in the owners' path
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        m0 = fmaxf(fmaxf(m0, sc[j].x), sc[j].y);
                        m1 = fmaxf(fmaxf(m1, sc[j].z), sc[j].w);
                    }
                    const float mraw = quad_max(fmaxf(m0, m1));
                    const float m = mraw * scale;                 // published in natural-log units: the merge uses exp(m_k - M)
                    const f2 k2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-mraw * SCALE_LOG2E, -mraw * SCALE_LOG2E};
                    f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f2 a = __builtin_elementwise_fma(f2{sc[j].x, sc[j].y}, k2, nm2);
                        const f2 c = __builtin_elementwise_fma(f2{sc[j].z, sc[j].w}, k2, nm2);
                        const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                        const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
                        sc[j] = f4{ea.x, ea.y, ec.x, ec.y};
                        d0 += ea;
                        d1 += ec;
                    }
                    const f2 dd = d0 + d1;
                    float den = dd.x + dd.y;
                    den = quad_sum(den);
                    f4 o[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
                    gemm_stage<NJ, NC, true, true>(
                        [&](int J, opnd (&a)[NC]) {
#pragma unroll
                            for (int t = 0; t < NC; ++t) a[t] = Vo[(t * NT + J0 + J) * 64 + lane];
                        },
                        [&](int J) { return sc[J]; }, o);
#pragma unroll
                    for (int t = 0; t < NC; ++t) Pg[(g * NC + t) * 64 + lane] = o[t];
                    Mg[(g * 2 + 0) * 64 + lane] = m;
                    Mg[(g * 2 + 1) * 64 + lane] = den;
                };
                // A3: the same partial attention on pieces; every key chunk of a guest is a K = 16 product (one chain per feature tile)
                auto attn_part3 = [&](auto njc, int J0) {
                    constexpr int NJ = decltype(njc)::value;
                    constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
                    u32x4 qb[3];
                    u32x2 qc[3];
                    u32x4 a0[NJ][3];
                    u32x2 a2[NJ][3];
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        qb[pc] = reinterpret_cast<const u32x4*>(Qg)[pc * 64 + lane3];
                        qc[pc] = reinterpret_cast<const u32x2*>(Qg + 192)[pc * 64 + lane3];
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) k_load3(J0 + j, a0[j], a2[j]);
                    // V^T pieces of key chunk J0 + j: a half of its pair's 16 bytes, or the odd last chunk
                    u32x2 vh[NJ][NC][3];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int J = J0 + j;
                        const bool oddc = (NT & 1) && J == NT - 1;
#pragma unroll
                        for (int t = 0; t < NC; ++t) {
                            const u32x2* base = reinterpret_cast<const u32x2*>(Kimg + NT * KP_T16 + t * VP_T16);
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc)
                                vh[j][t][pc] = oddc ? base[VP_PAIRS * 3 * 64 * 2 + pc * 64 + lane3] : base[(((J >> 1) * 3 + pc) * 64 + lane3) * 2 + (J & 1)];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f4 sc[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) sc[j] = tile48(std::false_type{}, a0[j], a2[j], qb, qc, splat4(0.f));
                    float m0 = -3.0e38f, m1 = -3.0e38f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        m0 = fmaxf(fmaxf(m0, sc[j].x), sc[j].y);
                        m1 = fmaxf(fmaxf(m1, sc[j].z), sc[j].w);
                    }
                    const float mraw = quad_max(fmaxf(m0, m1));
                    const float m = mraw * scale;
                    const f2 k2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-mraw * SCALE_LOG2E, -mraw * SCALE_LOG2E};
                    f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
                    u32x2 pp[NJ][3];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f2 a = __builtin_elementwise_fma(f2{sc[j].x, sc[j].y}, k2, nm2);
                        const f2 c = __builtin_elementwise_fma(f2{sc[j].z, sc[j].w}, k2, nm2);
                        const f2 ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                        const f2 ec = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
                        d0 += ea;
                        d1 += ec;
                        vt3::split3(f4{ea.x, ea.y, ec.x, ec.y}, pp[j][0], pp[j][1], pp[j][2]);
                    }
                    const f2 dd = d0 + d1;
                    const float den = quad_sum(dd.x + dd.y);
                    f4 o[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int e = 0; e < 6; ++e)
#pragma unroll
                            for (int t = 0; t < NC; ++t) o[t] = vt3::mma16(vh[j][t][TW[e]], pp[j][TX[e]], o[t]);
#pragma unroll
                    for (int t = 0; t < NC; ++t) Pg[(g * NC + t) * 64 + lane] = o[t];
                    Mg[(g * 2 + 0) * 64 + lane] = m;
                    Mg[(g * 2 + 1) * 64 + lane] = den;
                };
                if constexpr (A3) {
                    if (g == 3) attn_part3(std::integral_constant<int, NT - 3>{}, 3);
                    else attn_part3(std::integral_constant<int, 1>{}, g);
                } else {
                    if (g == 3) attn_part(std::integral_constant<int, NT - 3>{}, 3);
                    else attn_part(std::integral_constant<int, 1>{}, g);
                }
                // guests-only rendezvous: every guest has published its partial
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(gflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (g < 3) {
                    while (__hip_atomic_load(gflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (blk + 1))
                        __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    // merge the four partial softmaxes; output tile g of proj
                    float mk[4], lk[4], M = -3.0e38f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        mk[k] = Mg[(k * 2 + 0) * 64 + lane];
                        lk[k] = Mg[(k * 2 + 1) * 64 + lane];
                        M = fmaxf(M, mk[k]);
                    }
                    float Lsum = 0.f;
                    f4 o[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = splat4(0.f);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float f = __expf(mk[k] - M);
                        Lsum = fmaf(f, lk[k], Lsum);
#pragma unroll
                        for (int t = 0; t < NC; ++t) o[t] = o[t] + splat4(f) * Pg[(k * NC + t) * 64 + lane];
                    }
                    const float rl = __builtin_amdgcn_rcpf(Lsum);
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[t] = o[t] * splat4(rl);
                    if constexpr (A3) {
                        u32x4 ob[3], wa0[3];
                        u32x2 oc[3], wa2[3];
                        wp_load3(g, wa0, wa2);
                        split_h3(o, ob, oc);
                        Dg[g * 64 + lane] = tile48(std::false_type{}, wa0, wa2, ob, oc, splat4(0.f));
                    } else {
                        f4 acc[1] = {splat4(0.f)};
                        gemm_stage<NC, 1, true, true>([&](int c, opnd (&a)[1]) { a[0] = w_proj(g * NC + c); },
                                                      [&](int c) { return o[c]; }, acc);
                        Dg[g * 64 + lane] = acc[0];
                    }
                }
            }
        }
        stamp();            // attention + proj done
        barrier_publish<WLDS>();    // K/V and buffer B free; fc1 weights landed
        stamp();
        if constexpr (BF3L) {
            stage_tiles(Wb, P3 + W3_FC1_TILES * 256, WBUF_TILES, w, NW, lane, blk == 0);                                          // fc2, output tiles 0 and 1
            stage_tiles(Kimg, P3 + (W3_FC1_TILES + WBUF_TILES) * 256, W3_FC2_TILES - WBUF_TILES, w, NW, lane, blk == 0);           // output tile 2
        } else if constexpr (WLDS) stage_img(Wb, O_W2, NC * NH, blk);      // fc2 weights
        // ---- LN2 + MLP (residual add) ---------------------------------------------------------
        // fc1 -> GELU -> fc2 in three groups of HG = 4 hidden tiles, software-pipelined so GELU (VALU)
        // of one group issues in the shadow of the next group's MFMAs:
        //   fc1(g0) | fc1(g1) || GELU(g0) | fc1(g2) || GELU(g1) | fc2(k in g0) || GELU(g2) | fc2(g1) | fc2(g2)
        constexpr int HG = 4;
        static_assert(NH / HG == 3, "pipeline written for 12 hidden tiles");
        constexpr int NHID = WLDS ? TPW : 1;       // without the mid-MLP barrier each tile finishes before the next starts
        f4 hid[NHID][NH];
        f4 acc2[NHID][HG];
        auto fc1 = [&](const f4 (&h)[NC], int g, f4 (&acc)[HG]) {
#pragma unroll
            for (int j = 0; j < HG; ++j) acc[j] = ld4(S + S_B1 + 16 * (HG * g + j) + 4 * q);
            gemm_stage<NC, HG, true, WLDS>(
                [&](int c, opnd (&a)[HG]) {
#pragma unroll
                    for (int j = 0; j < HG; ++j) a[j] = w_fc1((HG * g + j) * NC + c);
                },
                [&](int c) { return h[c]; }, acc);
        };
        auto gelu_group = [&](const f4 (&acc)[HG], f4 (&hd)[NH], int g) {
#pragma unroll
            for (int j = 0; j < HG; ++j)
                hd[HG * g + j] = gelu4(acc[j]);
        };
        auto fc2 = [&](const f4 (&hd)[NH], int g, f4 (&xo)[NC]) {
            gemm_stage<HG, NC, true, WLDS>(
                [&](int cc, opnd (&a)[NC]) {
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) a[ot] = w_fc2(ot * NH + HG * g + cc);
                },
                [&](int cc) { return hd[HG * g + cc]; }, xo);
        };
        auto mlp_first = [&](int i, int hi) {          // LN2, fc1 (all groups), GELU of groups 0 and 1
            f4 h[NC];
            layer_norm_plain(x[i], h);
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(S + S_B2 + 16 * ot + 4 * q);
            f4 acc0[HG], acc1[HG];
            fstamp();
            fc1(h, 0, acc0);
            __builtin_amdgcn_sched_barrier(0);
            fstamp();
            fc1(h, 1, acc1);
            gelu_group(acc0, hid[hi], 0);
            interleave_mfma_valu<12 * HG, 3>();
            __builtin_amdgcn_sched_barrier(0);
            fstamp();
            fc1(h, 2, acc2[hi]);
            gelu_group(acc1, hid[hi], 1);
            interleave_mfma_valu<12 * HG, 3>();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mlp_second = [&](int i, int hi) {         // fc2 (+ GELU of group 2 behind its first third)
            fc2(hid[hi], 0, x[i]);
            gelu_group(acc2[hi], hid[hi], 2);
            interleave_mfma_valu<12 * HG, 3>();
            __builtin_amdgcn_sched_barrier(0);
            fstamp();
            fc2(hid[hi], 1, x[i]);
            fstamp();
            fc2(hid[hi], 2, x[i]);
        };
        // ---- BF3: the same MLP as exact three-piece bf16 products (vt_bf3.h).  fc1 walks the 12 hidden tiles in units of two
        // (two independent accumulator chains; 12 MFMAs per tile: six terms on chunk pair (0, 1), six on chunk 2 with a zero
        // upper half), the next unit's weight pieces requested ahead, GELU + the split of the previous unit under the current
        // unit's MFMAs.  Unit u's two hidden tiles ARE fc2's chunk pair u, so the split results are fc2's B operands as they stand.
        // N chains x 12 MFMAs: the small terms of both K steps first, then the large ones
        auto fc1_terms3 = [&](auto nc, const u32x4 (*a0)[3], const u32x2 (*a2)[3], const u32x4 (&hb)[3], const u32x2 (&hc)[3], f4* acc) {
            constexpr int N = decltype(nc)::value;
            constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
            f4 accB[N];
#pragma unroll
            for (int j = 0; j < N; ++j) accB[j] = splat4(0.f);
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            const int e = 3 * half + t3;
                            if (k == 0) accB[j] = vt3::mma16(a2[j][TW[e]], hc[TX[e]], accB[j]);
                            else acc[j] = vt3::mma(a0[j][TW[e]], hb[TX[e]], acc[j]);
                        }
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] = acc[j] + accB[j];
        };
        u32x4 hq[NHID][NH / 2][3];       // GELU(fc1) as pieces: [chunk pair][piece] = {quad of chunk 2 p | quad of chunk 2 p + 1}
        auto mlp_first3 = [&](int i, int hi) {
            f4 h[NC];
            layer_norm_plain(x[i], h);
#pragma unroll
            for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(S + S_B2 + 16 * ot + 4 * q);
            u32x4 hb[3];
            u32x2 hc[3];
            split_h3(h, hb, hc);
            fstamp();
            // one hidden tile per step, its 12 MFMAs as two independent chains (chunk pair 0 onto the bias, chunk 2 onto zero) that
            // are added afterwards: the next tile's 18 weight registers are requested a step ahead (36 in flight, not 72).  GELU + the
            // split of the PREVIOUS tile (48 VALU instructions) are written out as six stages of ~8, one behind each pair of MFMAs,
            // with a scheduling barrier per stage: left to itself hipcc issues the 12 MFMAs as one block (the wave then sits on the
            // accumulator chains for ~190 cycles with nothing else to issue) and the VALU work behind it; sched_group_barrier
            // requests did not change that here, and without the asm pin LLVM sinks the whole chain below the following tiles.
            constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
            u32x4 a0[2][3];
            u32x2 a2[2][3];
            w1_load3(0, a0[0], a2[0]);
            f4 bias1[2];
            bias1[0] = ld4(S + S_B1 + 4 * q);
            f4 prev = splat4(0.f);
            u32x2 pl[3];            // pieces of the pair's first tile
            f2 ua, ub, na, nb, pa, pb, ea, eb;      // GELU state between stages (pairs (x, y) and (z, w) of `prev`)
            unsigned xb[4], r1b[4], r2b[4];
            [[maybe_unused]] u32x2 hpk, mpk;      // VT_SPLIT_DOT2: the packed h / m pieces, which the residuals are computed from (vt_bf3.h)
            auto gstage = [&](int e) {
                const auto fma2 = [](f2 p, f2 n, float c) { return __builtin_elementwise_fma(p, n, f2{c, c}); };
                if (e == 0) {
                    ua = f2{prev.x, prev.y}; ub = f2{prev.z, prev.w};
                    na = f2{__builtin_fmaxf(-__builtin_fabsf(ua.x), -6.5f), __builtin_fmaxf(-__builtin_fabsf(ua.y), -6.5f)};
                    nb = f2{__builtin_fmaxf(-__builtin_fabsf(ub.x), -6.5f), __builtin_fmaxf(-__builtin_fabsf(ub.y), -6.5f)};
                    pa = __builtin_elementwise_fma(na, f2{2.992413958e-05f, 2.992413958e-05f}, f2{7.398738213e-04f, 7.398738213e-04f});
                    pb = __builtin_elementwise_fma(nb, f2{2.992413958e-05f, 2.992413958e-05f}, f2{7.398738213e-04f, 7.398738213e-04f});
                    pa = fma2(pa, na, 7.977461502e-03f); pb = fma2(pb, nb, 7.977461502e-03f);
                } else if (e == 1) {
                    pa = fma2(pa, na, 5.323818492e-02f); pb = fma2(pb, nb, 5.323818492e-02f);
                    pa = fma2(pa, na, -4.589156874e-01f); pb = fma2(pb, nb, -4.589156874e-01f);
                    pa = fma2(pa, na, 1.151147082e+00f); pb = fma2(pb, nb, 1.151147082e+00f);
                    pa = fma2(pa, na, -1.0f); pb = fma2(pb, nb, -1.0f);
                } else if (e == 2) {
                    ea = f2{__builtin_amdgcn_exp2f(pa.x), __builtin_amdgcn_exp2f(pa.y)};
                    eb = f2{__builtin_amdgcn_exp2f(pb.x), __builtin_amdgcn_exp2f(pb.y)};
                    ua = f2{fmaxf(ua.x, 0.0f), fmaxf(ua.y, 0.0f)};
                    ub = f2{fmaxf(ub.x, 0.0f), fmaxf(ub.y, 0.0f)};
                } else if (e == 3) {
                    const f2 ga = __builtin_elementwise_fma(na, ea, ua), gb = __builtin_elementwise_fma(nb, eb, ub);
                    const float gv[4] = {ga.x, ga.y, gb.x, gb.y};
#if VT_SPLIT_DOT2
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        float r0, r1;
                        hpk[p2] = __builtin_amdgcn_perm(__float_as_uint(gv[2 * p2 + 1]), __float_as_uint(gv[2 * p2]), 0x07060302u);
                        vt3::sub_pair(hpk[p2], gv[2 * p2], gv[2 * p2 + 1], r0, r1);
                        r1b[2 * p2] = __float_as_uint(r0); r1b[2 * p2 + 1] = __float_as_uint(r1);
                    }
#else
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xb[k] = __float_as_uint(gv[k]);
                        r1b[k] = __float_as_uint(gv[k] - __uint_as_float(xb[k] & 0xffff0000u));
                    }
#endif
                } else if (e == 4) {
#if VT_SPLIT_DOT2
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        float r0, r1;
                        mpk[p2] = __builtin_amdgcn_perm(r1b[2 * p2 + 1], r1b[2 * p2], 0x07060302u);
                        vt3::sub_pair(mpk[p2], __uint_as_float(r1b[2 * p2]), __uint_as_float(r1b[2 * p2 + 1]), r0, r1);
                        r2b[2 * p2] = __float_as_uint(r0); r2b[2 * p2 + 1] = __float_as_uint(r1);
                    }
#else
#pragma unroll
                    for (int k = 0; k < 4; ++k) r2b[k] = __float_as_uint(__uint_as_float(r1b[k]) - __uint_as_float(r1b[k] & 0xffff0000u));
#endif
                }
            };
            auto gfinish = [&](int t) {      // the six packs; t = the tile the pieces belong to
                u32x2 pc3[3];
#if VT_SPLIT_DOT2
                pc3[0] = hpk;
                pc3[1] = mpk;
#else
                pc3[0] = u32x2{__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
                pc3[1] = u32x2{__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u)};
#endif
                pc3[2] = u32x2{__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u)};
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) asm volatile("" : "+v"(pc3[pc]));
                if (t & 1) {
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) hq[hi][t >> 1][pc] = u32x4{pl[pc].x, pl[pc].y, pc3[pc].x, pc3[pc].y};
                } else {
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) pl[pc] = pc3[pc];
                }
            };
#pragma unroll
            for (int t = 0; t <= NH; ++t) {
                if (t == 4 || t == 8) fstamp();
                f4 accA = splat4(0.f), accB = splat4(0.f);
                if (t < NH) {
                    accA = bias1[t & 1];
                    if (t + 1 < NH) {       // the next tile's pieces AND its bias, a tile ahead (see the qkv loop)
                        w1_load3(t + 1, a0[(t + 1) & 1], a2[(t + 1) & 1]);
                        bias1[(t + 1) & 1] = ld4(S + S_B1 + 16 * (t + 1) + 4 * q);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    if (t < NH) {
                        accB = vt3::mma16(a2[t & 1][TW[e]], hc[TX[e]], accB);
                        accA = vt3::mma(a0[t & 1][TW[e]], hb[TX[e]], accA);
                    }
                    if (t > 0) {
                        if (e < 5) gstage(e);
                        else gfinish(t - 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                prev = accA + accB;
            }
        };
        auto mlp_second3 = [&](int i, int hi) {
            constexpr int NP = NH / 2;
            constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
            u32x4 a[2][NC][3];
            w2_load3(0, a[0]);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (p + 1 < NP) w2_load3(p + 1, a[(p + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 6; ++e)
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) x[i][ot] = vt3::mma(a[p & 1][ot][TW[e]], hq[hi][p][TX[e]], x[i][ot]);
                if (p == 1 || p == 3) fstamp();
            }
        };
        if constexpr (BF3L) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int T = w + NW * i;
                if (VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) mlp_first3(i, i);
            }
            u32x2 gq[NC][3];     // the guest's GELU(fc1) tiles 3 g .. 3 g + 2 as pieces
            if (VT_BLK_GUESTS && w >= NOWN) {
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) x4[ot] = x4[ot] + ld4(S + S_BPROJ + 16 * ot + 4 * q) + Dg[ot * 64 + lane];
                f4 h[NC];
                layer_norm_plain(x4, h);
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) x4[ot] = x4[ot] + ld4(S + S_B2 + 16 * ot + 4 * q);
                u32x4 hb[3];
                u32x2 hc[3];
                split_h3(h, hb, hc);
                u32x4 a0[NC][3];
                u32x2 a2[NC][3];
#pragma unroll
                for (int j = 0; j < NC; ++j) w1_load3(NC * g + j, a0[j], a2[j]);
                f4 acc[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[j] = ld4(S + S_B1 + 16 * (NC * g + j) + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
                fc1_terms3(std::integral_constant<int, NC>{}, a0, a2, hb, hc, acc);
#pragma unroll
                for (int j = 0; j < NC; ++j) vt3::split3(gelu4(acc[j]), gq[j][0], gq[j][1], gq[j][2]);
            }
                stamp();            // fc1 done
            barrier_publish<true>();    // fc2 weights landed; buffer A free
            if (blk + 1 < nblocks) stage_tiles(Wa, P3 + BLOCK3_STRIDE + (W3_FC1_TILES + W3_FC2_TILES) * 256, W3_QKV_TILES, w, NW, lane, false);   // next block's qkv
            stamp();
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int T = w + NW * i;
                if (VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) mlp_second3(i, i);
            }
            if (VT_BLK_GUESTS && w >= NOWN) {
                // fc2 restricted to this guest's hidden tiles 3 g .. 3 g + 2 = one whole chunk pair and half of another (the other
                // half of that pair belongs to the neighbouring guest: zeros in this guest's B operand)
                const bool odd = g & 1;
                const int p_full = (3 * g + (odd ? 1 : 0)) >> 1, p_half = (3 * g + (odd ? 0 : 2)) >> 1;
                u32x4 bf[3];
                u32x2 bh[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    bf[pc] = odd ? u32x4{gq[1][pc].x, gq[1][pc].y, gq[2][pc].x, gq[2][pc].y} : u32x4{gq[0][pc].x, gq[0][pc].y, gq[1][pc].x, gq[1][pc].y};
                    bh[pc] = odd ? gq[0][pc] : gq[2][pc];
                }
                u32x4 af[NC][3];
                u32x2 ah[NC][3];       // the half pair: the weights' quad of this guest's chunk only (K = 16 instruction)
                w2_load3(p_full, af);
                {
                    // chunk 2 p + 1 = the upper quad of the lane's 16 bytes, chunk 2 p the lower; p_half is wave-uniform, not constant
                    const unsigned hoff = 16u * lane3 + (odd ? 8u : 0u) + (unsigned)p_half * (3u * 64u * 16u);
                    const auto Wb3h = lds_lane_base<u32x2>(Wb, hoff);
                    const auto Wk3h = lds_lane_base<u32x2>(Kimg, hoff);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            ah[ot][pc] = ot < 2 ? Wb3h[((ot * (NH / 2)) * 3 + pc) * 64 * 2] : Wk3h[pc * 64 * 2];
                }
                f4 part[NC], partB[NC];
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) part[ot] = partB[ot] = splat4(0.f);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                            for (int ot = 0; ot < NC; ++ot) {
                                const int e = 3 * half + t3;
                                if (k == 0) partB[ot] = vt3::mma16(ah[ot][TW[e]], bh[TX[e]], partB[ot]);
                                else part[ot] = vt3::mma(af[ot][TW[e]], bf[TX[e]], part[ot]);
                            }
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) part[ot] = part[ot] + partB[ot];
#pragma unroll
                for (int ot = 0; ot < NC; ++ot) Pg2[(g * NC + ot) * 64 + lane] = part[ot];
            }
        } else if constexpr (WLDS) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int T = w + NW * i;
                if (VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) mlp_first(i, i);
            }
            f4 ghid[NC];        // BAL: GELU(fc1) of this guest's three hidden tiles
            if constexpr (BAL) {
                if (VT_BLK_GUESTS && w >= NOWN) {
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) x4[ot] = x4[ot] + ld4(S + S_BPROJ + 16 * ot + 4 * q) + Dg[ot * 64 + lane];
                    f4 h[NC];
                    layer_norm_plain(x4, h);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) x4[ot] = x4[ot] + ld4(S + S_B2 + 16 * ot + 4 * q);
#pragma unroll
                    for (int j = 0; j < NC; ++j) ghid[j] = ld4(S + S_B1 + 16 * (NC * g + j) + 4 * q);
                    gemm_stage<NC, NC, true, true>(
                        [&](int c, opnd (&a)[NC]) {
#pragma unroll
                            for (int j = 0; j < NC; ++j) a[j] = w_fc1((NC * g + j) * NC + c);
                        },
                        [&](int c) { return h[c]; }, ghid);
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        ghid[j] = gelu4(ghid[j]);
                }
            }
            stamp();            // fc1 done
            barrier_publish<WLDS>();    // fc2 weights landed; buffer A free
            if (blk + 1 < nblocks) stage_img(Wa, O_WQKV, 9 * NC, blk + 1);   // next block's qkv
            stamp();
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int T = w + NW * i;
                if (VT_BLK_OWNERS && T < NOWN && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) mlp_second(i, i);
            }
            if constexpr (BAL) {
                if (VT_BLK_GUESTS && w >= NOWN) {   // fc2 restricted to this guest's hidden tiles: a partial sum of the update
                    f4 part[NC];
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) part[ot] = splat4(0.f);
                    gemm_stage<NC, NC, true, true>(
                        [&](int cc, opnd (&a)[NC]) {
#pragma unroll
                            for (int ot = 0; ot < NC; ++ot) a[ot] = w_fc2(ot * NH + NC * g + cc);
                        },
                        [&](int cc) { return ghid[cc]; }, part);
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) Pg[(g * NC + ot) * 64 + lane] = part[ot];
                }
            }
        } else {
            // many tiles per wave (G256): plain per-tile MLP in two groups of 6 hidden tiles -- the
            // pipelined form above costs registers this variant does not have
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int T = w + NW * i;
                if (BF3G && T < NT && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) {
                    // BF3G: the same per-tile MLP as exact three-piece bf16 products, weight pieces straight from L2 (vt_bf3.h; images
                    // as in the staged form): fc1 one hidden tile per step, the pieces of the tile after next in flight; GELU + split
                    // once per hidden tile; fc2 per chunk pair with the next pair's pieces in flight
                    const u32x4* const G1 = reinterpret_cast<const u32x4*>(P3);
                    const u32x2* const G1h = reinterpret_cast<const u32x2*>(P3);
                    const u32x4* const G2 = G1 + W3_FC1_TILES * 64;
                    auto g1_load3 = [&](int t, u32x4 (&a0)[3], u32x2 (&a2)[3]) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            a0[pc] = G1[t * W3_FC1_OT16 + pc * 64 + lane3];
                            a2[pc] = G1h[(t * W3_FC1_OT16 + 192) * 2 + pc * 64 + lane3];
                        }
                    };
                    f4 h[NC];
                    layer_norm_plain(x[i], h);
                    u32x4 hb[3];
                    u32x2 hc[3];
                    split_h3(h, hb, hc);
                    u32x4 hq3[NH / 2][3];
                    {
                        u32x4 a0[2][3];
                        u32x2 a2[2][3];
                        g1_load3(0, a0[0], a2[0]);
                        u32x2 pl[3];
#pragma unroll
                        for (int t = 0; t < NH; ++t) {
                            if (t + 1 < NH) g1_load3(t + 1, a0[(t + 1) & 1], a2[(t + 1) & 1]);
                            const f4 bias = ld4(S + S_B1 + 16 * t + 4 * q);
                            __builtin_amdgcn_sched_barrier(0);
                            const f4 r = tile48(std::false_type{}, a0[t & 1], a2[t & 1], hb, hc, bias);
                            u32x2 pc3[3];
                            vt3::split3(gelu4(r), pc3[0], pc3[1], pc3[2]);
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) {
                                if (t & 1) hq3[t >> 1][pc] = u32x4{pl[pc].x, pl[pc].y, pc3[pc].x, pc3[pc].y};
                                else pl[pc] = pc3[pc];
                            }
                        }
                    }
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(S + S_B2 + 16 * ot + 4 * q);
                    {
                        constexpr int NP = NH / 2;
                        constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
                        // 18 steps of (chunk pair p, output tile ot): 6 MFMAs on x[ot] each, the pieces of the step three ahead in flight
                        auto g2_load3 = [&](int st, u32x4 (&a)[3]) {
                            const int p = st / NC, ot = st - NC * p;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) a[pc] = G2[((ot * NP + p) * 3 + pc) * 64 + lane3];
                        };
                        constexpr int NS = NP * NC, AHEAD = 3;
                        u32x4 a[AHEAD + 1][3];
#pragma unroll
                        for (int st = 0; st < AHEAD; ++st) g2_load3(st, a[st]);
#pragma unroll
                        for (int st = 0; st < NS; ++st) {
                            if (st + AHEAD < NS) g2_load3(st + AHEAD, a[(st + AHEAD) % (AHEAD + 1)]);
                            __builtin_amdgcn_sched_barrier(0);
                            const int p = st / NC, ot = st - NC * p;
#pragma unroll
                            for (int e = 0; e < 6; ++e) x[i][ot] = vt3::mma(a[st % (AHEAD + 1)][TW[e]], hq3[p][TX[e]], x[i][ot]);
                        }
                    }
                } else if (T < NT && T != dbg_skip_tile && !(last_skip_z && 16 * T < len_z)) {
                    f4 h[NC];
                    layer_norm_plain(x[i], h);
                    f4 hd[NH];
                    constexpr int G6 = 6;
#pragma unroll
                    for (int g = 0; g < NH; g += G6) {
                        f4 acc[G6];
#pragma unroll
                        for (int j = 0; j < G6; ++j) acc[j] = ld4(S + S_B1 + 16 * (g + j) + 4 * q);
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            opnd a[G6];
#pragma unroll
                            for (int j = 0; j < G6; ++j) a[j] = w_fc1((g + j) * NC + c);
                            mfma4_shared_b(a, h[c], acc);
                        }
#pragma unroll
                        for (int j = 0; j < G6; ++j)
                            hd[g + j] = gelu4(acc[j]);
                    }
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) x[i][ot] = x[i][ot] + ld4(S + S_B2 + 16 * ot + 4 * q);
#pragma unroll
                    for (int c = 0; c < NH; ++c) {
                        opnd a[NC];
#pragma unroll
                        for (int ot = 0; ot < NC; ++ot) a[ot] = w_fc2(ot * NH + c);
                        mfma4_shared_b(a, hd[c], x[i]);
                    }
                }
            }
        }
        stamp();            // MLP done
        if constexpr (WLDS) barrier_publish<true>();   // buffer B free; next qkv weights landed
        if constexpr (BAL) {
            if (w >= NOWN) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int ot = 0; ot < NC; ++ot) x4[ot] = x4[ot] + Pg2[(k * NC + ot) * 64 + lane];
            }
        }
    }

    stamp();
    // ---- epilogue: optional residual dump; final LayerNorm on the search tokens ----------------
    const float* PF = Sp + depth_total * SMALL_STRIDE;   // norm.weight, norm.bias
    // BF3 (at the register cap): the output addresses are built here from a fresh copy of the lane index, not kept across the block loop
    int lane_e = lane_k;
    if constexpr (BF3) asm volatile("" : "+v"(lane_e));
    const int tok_e = lane_e & 15, q_e = lane_e >> 4;
    const int Lx = L - len_z;
    if constexpr (BAL) {
        if (g == 0) {
            if (resid != nullptr) {
                float* dst = resid + ((size_t)b * L + 16 * GT + tok_e) * C + 4 * q_e;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(dst + 16 * c, x4[c]);
            }
            f4 h[NC];
            layer_norm_img(x4, h, PF, PF + C, q_e);
            float* dst = feat + ((size_t)b * Lx + (16 * GT - len_z) + tok_e) * C + 4 * q_e;
#pragma unroll
            for (int c = 0; c < NC; ++c) st4(dst + 16 * c, h[c]);
        }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int T = w + NW * i;
        if (T < NOWN) {
            if (resid != nullptr) {
                float* dst = resid + ((size_t)b * L + 16 * T + tok_e) * C + 4 * q_e;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(dst + 16 * c, x[i][c]);
            }
            if (16 * T >= len_z) {   // len_z is a multiple of 16 for both supported geometries
                f4 h[NC];
                layer_norm_img(x[i], h, PF, PF + C, q_e);
                float* dst = feat + ((size_t)b * Lx + (16 * T - len_z) + tok_e) * C + 4 * q_e;
#pragma unroll
                for (int c = 0; c < NC; ++c) st4(dst + 16 * c, h[c]);
            }
        }
    }
}

}  // namespace vtb
