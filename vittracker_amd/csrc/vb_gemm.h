// vb_gemm.h -- bf16 MFMA GEMM for the ViT-Base path (BASELINE config 4), gfx950.
//
//   out[m][n] (+)= sum_k X[m][k] * W[n][k] + bias[n]        X: activations (K contiguous), W: nn.Linear / folded conv weights
//
// One workgroup of 8 waves computes a BM x BN tile with v_mfma_f32_16x16x32_bf16, K in steps of 64.  Weights are the
// MFMA A operand and activations the B operand, so a result tile is D[n = 4q + r][m = lane & 15]: every lane holds FOUR
// CONSECUTIVE output features of one token -- 8-byte bf16 / 16-byte f32 stores into row-major [M][N] buffers with no
// transpose.  (For the V third of the qkv projection the operands swap roles, which hands every lane four consecutive
// TOKENS of one feature: V is stored transposed, [frame][head][d][token], exactly what the attention kernel's P.V
// contraction wants -- vb_attn.h.)
//
// Operand panels go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into 16-row x 32-k
// sub-tiles of 1 KiB whose two 8-row halves are XOR-ed by 32 bytes ("st_16x32": ds_read_b128 of a fragment is then
// conflict-free); the swizzle is applied to the per-lane SOURCE address and to the fragment read, the LDS destination
// of a DMA stays lane-linear.
//
// Pipeline (256 x 256 tile): the software-pipelined k-loop below (two LDS stages of one 64-deep k-tile, one DMA piece per
// group of 8 MFMAs, two barriers per k-tile, counted vmcnt).  The narrow tile (256 x 64, head convs 2-4) keeps a simple
// two-buffer 64-deep loop.  (Rounds 2-3 also carried a 4-stage, a whole-line two-stage and a phase-interleaved form of the wide
// loop; all measured slower -- NOTES_r1_r3.md -- and were removed in round 5.)
//
// XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so id -> (id % 8) * per_xcd +
// id / 8 gives every XCD a contiguous run of tiles, n fastest: the X panel of a tile row is fetched into that XCD's L2
// once and re-used by all its column tiles.
//
// Per-feature bias and per-token LayerNorm factor (round 5).  Neither lives in registers across the k-loop: behind a tile's
// epilogue every wave requests the NEXT tile's 64 bias values and (LayerNorm-folded GEMMs) 128 per-token rstd values by
// three 4-byte LDS-DMA instructions into the head of its own 4 KiB epilogue staging area, which nothing else touches until
// that tile's epilogue; the wait in front of the tile's first k-tile covers them.  The accumulators start at zero.
//
// LayerNorm folded into the GEMM that consumes it (qkv, fc1; lib/models/ostrack/vit.py:88-90):
//     LN(x) W^T + b = rstd_m * (x (W diag(g) (I - 11^T / K))^T) + (b + W beta)
// i.e. the k-centred, gamma-scaled weight rows W' (folded in fp64 at vt_load_weights) absorb the mean subtraction, the GEMM
// reads the RAW residual row rounded to bf16, and the epilogue multiplies by the row's rstd.  The GEMM that PRODUCES a residual
// row (patch embedding, proj, fc2: EPI_PATCH / EPI_RESID) writes that bf16 copy next to its f32 read-modify-write and, per
// (row, 64-column wave slice), the slice's (sum, centred sum of squares); vbm::ln_finalize_kernel merges the 12 slices of a
// row (Chan's update) into rstd.  What this removes: the LayerNorm kernel's pass over the residual stream (252 MB read + 126 MB
// written per LayerNorm at B = 256; 24 of 25 launches).  What it costs in accuracy: the common mode of a row is rounded with
// the row instead of being subtracted first -- the error of the product grows by sqrt(1 + mean^2 / var) of the row
// (tools/vitb_lnfold_emul.py: identical to the unfolded form at the fixtures' mean^2 / var << 1; 2.3x at mean = 2 std).
// VB_LN_FOLD=0 (read at vt_create) keeps the separate LayerNorm kernel and unfolded weights.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vt_common.h"

#ifndef VB_SWP_NO_MFMA
#define VB_SWP_NO_MFMA 0     // timing experiments only: the software-pipelined k-loop without its MFMAs (wrong results)
#endif
#ifndef VB_SWP_NO_DMA
#define VB_SWP_NO_DMA 0      // timing experiments only: no operand staging inside the k-loop (wrong results)
#endif
#ifndef VB_SWP_NO_BARRIER
#define VB_SWP_NO_BARRIER 0  // timing experiments only: no rendezvous per k-tile (races by design)
#endif
#ifndef VB_RESID_AHEAD
#define VB_RESID_AHEAD 2      // residual-row groups (16 rows x 256 B per wave) requested ahead of their use in the RESID / PATCH epilogue
#endif
#ifndef VB_EPI_NT
#define VB_EPI_NT 0          // 1: every epilogue's stores (and the residual's loads) carry the non-temporal hint (measured: +-0 except
#endif                       //    the transposed V stores, which always carry it: v projection 100 -> 92 us)
#ifndef VB_SWP_HALFISSUE
#define VB_SWP_HALFISSUE 0   // 1: only the wm = 0 waves issue the DMA instructions (two per slot)
#endif

namespace vbg {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int NWAVES = 8;

enum AMode { A_PLAIN = 0, A_CONV = 1 };
enum Epi {
    EPI_PATCH = 0,   // resid[m][n] = acc + bias[n] + pos[m % L][n]                     (f32 out)
    EPI_BF16 = 1,    // out[m][n] = bf16(acc + bias[n])                                  (q (pre-scaled) and k projections)
    EPI_RESID = 2,   // resid[m][n] += acc + bias[n]                                     (f32 read-modify-write)
    EPI_GELU = 3,    // out[m][n] = bf16(gelu(acc + bias[n]))
    EPI_CONV = 4,    // out[map row of m][n] = bf16(relu(acc + bias[n]))                 (head towers, BN folded)
    EPI_VT = 5,      // vt[frame][n][token] = bf16(acc + bias[n]): the v projection, stored transposed (operands swapped)
};

struct Args {
    const bf16* X;        // activations: [M][K] (A_PLAIN) or zero-bordered NHWC map [B][F+2][F+2][C] (A_CONV)
    const bf16* W;        // [N (padded to BN)][K]
    const float* bias;    // [N]
    void* out;            // bf16 output (QKV: qk buffer; GELU / CONV) -- unused by PATCH / RESID
    float* resid;         // f32 residual stream [M][N] (PATCH, RESID)
    const float* pos;     // PATCH: [L][N] position embeddings (template rows first)
    bf16* vt;             // EPI_VT: transposed V [B][N][L]
    int M, N, K;
    int ldo;              // row stride of `out` in elements
    int L;                // tokens per frame (PATCH, VT)
    int C, F;             // A_CONV: channels per tap, map side
    int out_padded;       // EPI_CONV: 1 = write into a zero-bordered (F+2)^2 map, 0 = plain [M][ldo]
    // grouped launch (blockIdx.y = group): element strides between groups
    long long gX, gW, gOut;
    int gBias;
    int n_split;          // EPI_CONV with towers concatenated along N: columns per tower (0 = none); tower t writes out + t * gOut
    int rb;               // tile order: > 1 = tiles are walked in blocks of `rb` tile rows, column by column inside a block (so the
                          // 32 concurrent tiles of an XCD are ~rb rows x 32 / rb columns); 0 / 1 = row-major
    int desync_ticks;     // > 0: workgroup phase groups -- group g = (blockIdx.x / 8) % desync_groups starts g * desync_ticks / desync_groups
    int desync_groups;    // ticks of the 100 MHz clock late, so the CUs' epilogues (HBM) and k-loops (MFMA) do not all coincide
    // LayerNorm folded into the consuming GEMM (header comment)
    const float* rstd;    // consumer (BF16 / VT / GELU): [M] per-row 1 / sqrt(var + eps); nullptr = plain bias epilogue
    bf16* xb;             // producer (PATCH / RESID): [M][N] bf16 copy of the rows just written to `resid`; nullptr = none
    f2* stats;            // producer: [tiles_n * WN][ldstats] per-row (sum, centred sum of squares) of each 64-column wave slice
    int ldstats;
    // producer, round 6: the bf16 copy is written as x - c_row, c_row = cm[cm_mod ? m % cm_mod : m] -- an estimate of the row's mean
    // (RESID: the mean ln_finalize found for the row BEFORE this update; PATCH: mean of the token's pos-embed row + bias).  The folded
    // weights' rows sum to zero, so subtracting ANY per-row constant leaves x W'^T unchanged in exact arithmetic; in bf16 it takes the
    // rows' common mode out of what is rounded: the fold's error stops growing with mean^2 / var (tests/test_gpu_vitb.py, cm fixtures).
    const float* cm;
    int cm_mod;
    int dbg;              // timing experiments only (VB_DBG, wrong results by design; 0 in production):
                          // 1 = every tile loads the X panel of tile row 0, 2 = ... the W panel of tile column 0,
                          // 4 = no MFMAs, 8 = no epilogue, 16 = no W staging, 32 = no X staging (wide tile)
};

// The epilogue staging area is written and read back through differently typed pointers by the same wave: the accesses
// are declared may_alias and fenced, so neither type-based alias analysis nor the scheduler can move a read above the
// write that feeds it (DS operations of one wave execute in order; the waits are the compiler's).
typedef bf16x4 __attribute__((may_alias)) bf16x4a;
typedef uint4 __attribute__((may_alias)) uint4a;
typedef f4 __attribute__((may_alias)) f4a;
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); asm volatile("" ::: "memory"); }

__device__ __forceinline__ int swz_byte(int p) { return p ^ (((p >> 9) & 1) << 5); }   // st_16x32, involution on [0, 1024)

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 to_bf16x4(f4 v) {
    const bf16x2 lo = __builtin_convertvector(f2{v.x, v.y}, bf16x2), hi = __builtin_convertvector(f2{v.z, v.w}, bf16x2);
    return bf16x4{lo.x, lo.y, hi.x, hi.y};
}

__device__ __forceinline__ f4 fma4(f4 a, f4 b, f4 c) { return __builtin_elementwise_fma(a, b, c); }

// sum over the 16 lanes of a DPP row (lane & 15 varies), result in every lane: two quad permutes, row_half_mirror, row_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1, 0, 3, 2]
    v += dpp_f<0x4E>(v);      // quad_perm [2, 3, 0, 1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    return v;
}

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(NWAVES * 64) void gemm_kernel(const Args a) {
    static_assert(WM * WN == NWAVES, "8 waves");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;    // 16x16 tiles per wave
    constexpr int SX = BM / 16 * 2, SW = BN / 16 * 2;      // 1 KiB sub-tiles per panel and k-tile
    constexpr int NS = (SX + SW) / NWAVES;                 // DMA instructions per wave and k-tile
    static_assert((SX + SW) % NWAVES == 0 && SX % NWAVES == 0, "panel split");
    constexpr int NSX = SX / NWAVES;
    constexpr int BUF_BYTES = (SX + SW) * 1024;
    constexpr bool WIDE = (BM / 16) % NWAVES == 0 && (BN / 16) % NWAVES == 0;      // the 256 x 256 tile
    constexpr bool SWP = WIDE;
    constexpr int EP_OFF = 2 * BUF_BYTES;                                           // epilogue staging: 8 waves x 4 KiB behind the stages
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_outer = lane;
    const int wm = w / WN, wn = w % WN;
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM, nwg = tiles_m * tiles_n;
    const int grp = blockIdx.y;
    const bf16* __restrict__ X = a.X + grp * a.gX;
    const bf16* __restrict__ W = a.W + grp * a.gW;
    const float* __restrict__ bias = a.bias + grp * a.gBias;

    // ---- XCD-aware tile order: virtual block id -> tile
    auto tile_of = [&](int vb, int& m0, int& n0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        int tm, tn;
        if (a.rb > 1) {
            const int per = a.rb * tiles_n, blk = t / per, rem = t - blk * per;
            const int left = tiles_m - blk * a.rb, rows = left < a.rb ? left : a.rb;
            tn = rem / rows;
            tm = blk * a.rb + rem - tn * rows;
        } else {
            tm = t / tiles_n;
            tn = t - tm * tiles_n;
        }
        m0 = tm * BM;
        n0 = tn * BN;
    };
    // ---- per-lane DMA sources of this wave's sub-tiles (element offsets at k-tile 0)
    const int pl = swz_byte(lane * 16), prow = pl >> 6, pk = (pl & 63) >> 1;     // row in sub-tile, k element in sub-tile
    unsigned src_off[NS];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = w + NWAVES * i;
            if (i < NSX) {
                int m = m0 + (s >> 1) * 16 + prow;
                m = m < a.M ? m : a.M - 1;                                          // rows past M: a valid row, result unused
                unsigned base;
                if constexpr (AMODE == A_CONV) {
                    const int FF = a.F * a.F, b = m / FF, yx = m - b * FF, y = yx / a.F, x = yx - y * a.F, P = a.F + 2;
                    base = (unsigned)(((b * P + y) * P + x) * a.C);
                } else {
                    base = (unsigned)m * (unsigned)a.K;
                }
                src_off[i] = base + (s & 1) * 32 + pk;
            } else {
                const int sw = s - SX;
                src_off[i] = (unsigned)(n0 + (sw >> 1) * 16 + prow) * (unsigned)a.K + (sw & 1) * 32 + pk;   // W rows are padded to BN
            }
        }
    };
    auto stage = [&](int kt, char* buf) {
        unsigned kx;     // element offset of k-tile kt along an X row
        if constexpr (AMODE == A_CONV) {
            const int per_tap = a.C / BK, tap = kt / per_tap, c0 = (kt - tap * per_tap) * BK, r = tap / 3, s = tap - 3 * r;
            kx = (unsigned)((r * (a.F + 2) + s) * a.C + c0);
        } else {
            kx = (unsigned)kt * BK;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = w + NWAVES * i;
            const bf16* g = i < NSX ? X + src_off[i] + kx : W + src_off[i] + (unsigned)kt * BK;
            glds16(g, buf + s * 1024 + lane * 16);
        }
    };

    f4 acc[TN][TM];
    // Normal tiles: lane holds n = nb + 4q + {0..3} of token m = mb + (lane & 15); EPI_VT: tokens mb + 4q + {0..3} of feature
    // n = nb + (lane & 15).
    // Per-feature bias / per-token rstd of the NEXT tile -> the head of this wave's epilogue staging area (header comment):
    // ep[0, 256) = bias of the wave's 64 features, ep[256, 256 + 64 TM) = rstd of its 16 TM tokens.  Rows / columns beyond M / N
    // read a valid element (result unused).
    constexpr bool LNC = EPI == EPI_BF16 || EPI == EPI_VT || EPI == EPI_GELU;      // may consume a folded LayerNorm
    char* const ep = smem + EP_OFF + w * 4096;
    auto stage_vec = [&](int m0, int n0) {
        int n = n0 + wn * 64 + lane;
        n = n < a.N ? n : a.N - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bias + n),
                                         (__attribute__((address_space(3))) void*)ep, 4, 0, 0);
        if constexpr (LNC) {
            if (a.rstd) {
#pragma unroll
                for (int h = 0; h < TM * 16 / 64; ++h) {
                    int m = m0 + wm * TM * 16 + h * 64 + lane;
                    m = m < a.M ? m : a.M - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.rstd + m),
                                                     (__attribute__((address_space(3))) void*)(ep + 256 + h * 256), 4, 0, 0);
                }
            }
        }
    };
    // Epilogue through LDS: every wave owns a 4 KiB staging area (behind the pipeline stages) and moves its (TM x 16) x 64
    // result sub-tile through it in 4 KiB chunks, so that global memory sees WHOLE rows -- 16 B per lane, 8 lanes per 128-byte
    // bf16 row / 16 lanes per 256-byte f32 row -- instead of the accumulator layout's 8-byte pieces at a row stride (measured:
    // those cost 12 us per 256 x 256 tile, a third of the K = 768 GEMMs).  Chunks are XOR-swizzled by row so the column-wise
    // writes spread over the banks.  CHECK = false on tiles wholly inside M (all but the last tile row): no per-store branch.
    static_assert(TN == 4, "a wave's sub-tile is 64 output features wide");
    auto epilogue_impl = [&](int m0, int n0, auto check) {
        constexpr bool CHECK = decltype(check)::value;
        // per-lane epilogue addresses come from a FRESH copy of the lane index: computed from the kernel's own `lane` they are
        // invariants of the persistent tile loop, hipcc keeps them all in registers across the k-loop and spills fragments instead
        int lane_e = lane_outer;
        asm volatile("" : "+v"(lane_e));
        const int lane = lane_e, l15 = lane & 15, q4 = (lane >> 4) * 4;
        const int mw = m0 + wm * TM * 16, nw = n0 + wn * 64;           // this wave's sub-tile origin
        // this tile's bias (and rstd) out of the staging area's head, BEFORE the first staging write overwrites it (DS
        // operations of one wave execute in order)
        if constexpr (EPI == EPI_VT) {
            // swapped tile: features on lanes.  Chunk = 16 features x 128 tokens (256-byte rows)
            static_assert(EPI != EPI_VT || TM == 8, "128 tokens per row");
            float bs[TN];
            f4 rs[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) bs[i] = *reinterpret_cast<const volatile float*>(ep + (i * 16 + l15) * 4);
#pragma unroll
            for (int j = 0; j < TM; ++j) rs[j] = a.rstd ? *reinterpret_cast<const f4a*>(ep + 256 + (j * 16 + q4) * 4) : splat4(1.f);
            lds_fence();
#pragma unroll
            for (int i = 0; i < TN; ++i) {
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    *reinterpret_cast<bf16x4a*>(ep + l15 * 256 + (((j * 2 + (q4 >> 3)) ^ l15) << 4) + (q4 & 4) * 2) =
                        to_bf16x4(fma4(acc[i][j], rs[j], splat4(bs[i])));
                lds_fence();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 4 * t + (lane >> 4), ch = lane & 15;
                    const uint4 v = *reinterpret_cast<const uint4a*>(ep + row * 256 + ((ch ^ row) << 4));
                    const int m = mw + ch * 8, n = nw + i * 16 + row;
                    if (!CHECK || m < a.M) {      // M is a multiple of 8 (L is): 8 tokens never straddle a frame or the end
                        const int f = m / a.L, tk = m - f * a.L;
                        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&v), reinterpret_cast<u32x4*>(a.vt + ((size_t)f * a.N + n) * a.L + tk));
                    }
                }
                lds_fence();
            }
            return;
        }
        if (nw >= a.N) return;                                           // W rows beyond N are zero padding (BN = 64, N = 32)
        f4 bs[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) bs[i] = *reinterpret_cast<const f4a*>(ep + (i * 16 + q4) * 4);
        if constexpr (EPI == EPI_RESID || EPI == EPI_PATCH) {
            lds_fence();
            // f32 residual stream: chunk = 16 rows x 256 B; read-modify-write in whole rows, loads before stores.  Next to it
            // (LayerNorm folded into the next GEMM): the rows' bf16 copy and, per row, this wave's 64-column (sum, centred M2)
            const int tn = n0 / BN;
            // the rows' old values (the position table for PATCH) are requested VB_RESID_AHEAD row groups ahead of their use: asked for
            // where they are added, every one of a tile's TM row groups waits a whole memory latency per wave
            constexpr int AH = VB_RESID_AHEAD;
            f4 oldq[AH + 1][4];
            float oc[4];      // the rows' centring constants (Args::cm): ONE set, requested at the top of a row group's iteration and used at its end
                              // (the same few KB for all twelve column tiles of a row: L1 / L2 hits; a ring of AH + 1 sets spilled 40 B)
            const int ch = lane & 15, r0 = lane >> 4;
            auto load_c = [&](int j, float (&oc)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = mw + j * 16 + 4 * t + r0;
                    const int mc = CHECK ? (m < a.M ? m : a.M - 1) : m;
                    oc[t] = (a.xb && a.cm) ? a.cm[a.cm_mod ? mc % a.cm_mod : mc] : 0.f;
                }
            };
            auto load_old = [&](int j, f4 (&o)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = mw + j * 16 + 4 * t + r0;
                    const int mc = CHECK ? (m < a.M ? m : a.M - 1) : m;
                    if constexpr (EPI == EPI_RESID)
                        o[t] = VB_EPI_NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.resid + (size_t)mc * a.N + nw + ch * 4))
                                         : ld4(a.resid + (size_t)mc * a.N + nw + ch * 4);
                    else o[t] = ld4(a.pos + (size_t)(mc % a.L) * a.N + nw + ch * 4);
                }
            };
#pragma unroll
            for (int j = 0; j < AH && j < TM; ++j) load_old(j, oldq[j]);

#pragma unroll
            for (int j = 0; j < TM; ++j) {
#pragma unroll
                for (int i = 0; i < TN; ++i) *reinterpret_cast<f4a*>(ep + l15 * 256 + (((i * 4 + (q4 >> 2)) ^ l15) << 4)) = acc[i][j] + bs[i];
                if (j + AH < TM) load_old(j + AH, oldq[(j + AH) % (AH + 1)]);
                load_c(j, oc);
                f4 (&old)[4] = oldq[j % (AH + 1)];
                lds_fence();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 4 * t + r0, m = mw + j * 16 + row;
                    const f4 v = *reinterpret_cast<const f4a*>(ep + row * 256 + ((ch ^ row) << 4));
                    const f4 nv = old[t] + v;
                    const bool ok = !CHECK || m < a.M;
                    if (ok) {
                        if (VB_EPI_NT) st4_nt(a.resid + (size_t)m * a.N + nw + ch * 4, nv);
                        else st4(a.resid + (size_t)m * a.N + nw + ch * 4, nv);
                    }
                    if (a.xb) {
                        if (ok) *reinterpret_cast<bf16x4*>(a.xb + (size_t)m * a.N + nw + ch * 4) = to_bf16x4(nv - splat4(oc[t]));
                        const float s = row16_sum(hsum4(nv));
                        const f4 d = nv - splat4(s * (1.0f / 64.0f));
                        const float m2 = row16_sum(hsum4(d * d));
                        if (ok && ch == 0) a.stats[(size_t)(tn * WN + wn) * a.ldstats + m] = f2{s, m2};
                    }
                }
                lds_fence();
            }
            return;
        }
        // bf16 outputs: chunk = 32 rows x 128 B
        float rs[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) rs[j] = 1.f;
        if constexpr (LNC) {
            if (a.rstd) {
#pragma unroll
                for (int j = 0; j < TM; ++j) rs[j] = *reinterpret_cast<const volatile float*>(ep + 256 + (j * 16 + l15) * 4);
            }
        }
        lds_fence();
        bf16* obase = static_cast<bf16*>(a.out);
        int ncol = nw;
        if constexpr (EPI == EPI_CONV) {
            obase += grp * a.gOut;
            if (a.n_split) { const int t = nw / a.n_split; obase += (size_t)t * a.gOut; ncol = nw - t * a.n_split; }
        }
#pragma unroll
        for (int c = 0; c < TM / 2; ++c) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int row = jj * 16 + l15;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    f4 v = LNC ? fma4(acc[i][2 * c + jj], splat4(rs[2 * c + jj]), bs[i]) : acc[i][2 * c + jj] + bs[i];
                    if constexpr (EPI == EPI_GELU) v = gelu4(v);
                    if constexpr (EPI == EPI_CONV) v = f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                    *reinterpret_cast<bf16x4a*>(ep + row * 128 + (((i * 2 + (q4 >> 3)) ^ (row & 7)) << 4) + (q4 & 4) * 2) = to_bf16x4(v);
                }
            }
            lds_fence();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 8 * t + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4a*>(ep + row * 128 + ((ch ^ (row & 7)) << 4));
                const int m = mw + c * 32 + row;
                if (CHECK && m >= a.M) continue;
                if (nw + ch * 8 >= a.N) continue;              // zero-padded weight rows (N = 32 under a 64-wide tile)
                size_t orow = (size_t)m;
                if constexpr (EPI == EPI_CONV) {
                    if (a.out_padded) {
                        const int FF = a.F * a.F, bb = m / FF, yx = m - bb * FF, y = yx / a.F, x = yx - y * a.F, P = a.F + 2;
                        orow = (size_t)((bb * P + y + 1) * P + x + 1);
                    }
                }
                if (VB_EPI_NT) __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&v), reinterpret_cast<u32x4*>(obase + orow * a.ldo + ncol + ch * 8));
                else *reinterpret_cast<uint4*>(obase + orow * a.ldo + ncol + ch * 8) = v;
            }
            lds_fence();
        }
    };
    auto epilogue = [&](int m0, int n0) {
        if (m0 + BM <= a.M) epilogue_impl(m0, n0, std::false_type{});
        else epilogue_impl(m0, n0, std::true_type{});
    };

    const int fr = swz_byte((lane & 15) * 64 + (lane >> 4) * 16);   // fragment byte inside a sub-tile
    auto mfma_step = [&](const char* xp, const char* wp, int stride) {     // one 32-deep k-step from LDS
        bf16x8 fw[TN], fx[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(wp + i * stride);
#pragma unroll
        for (int j = 0; j < TM; ++j) fx[j] = *reinterpret_cast<const bf16x8*>(xp + j * stride);
        if constexpr (EPI == EPI_VT) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[j], fw[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = splat4(0.f);
    };
    int vb = blockIdx.x, m0, n0;
    if (vb >= nwg) return;
    tile_of(vb, m0, n0);
    stage_vec(m0, n0);

    if constexpr (SWP) {
        // ---- Software-pipelined k-loop, TWO barriers per k-tile, one DMA instruction per stage.  LDS as in the phase-interleaved
        // form: two stages of one 64-deep k-tile (64 KiB each: X region 32 KiB | W region 32 KiB), pieces of 8 rows x 128 B (whole
        // cache lines), chunk ^ row swizzle.  A wave's 64 MFMAs of a k-tile run as EIGHT stages of 8 (one accumulator quadrant x
        // one 32-deep k-step) in snake order over the two k-steps
        //     k0: Q00 Q01 Q11 Q10   k1: Q10 Q11 Q01 Q00        (Qxw = token half x, feature half w of the wave's 128 x 64)
        // so consecutive stages share an operand, and each stage's MFMAs cover the fragment reads of LATER stages (X: two buffers
        // XA / XB of 4 token tiles; W: all four (half, k-step) sets, so the W region is done with after S1) and ONE DMA instruction:
        //     stage  MFMAs        fragment reads                         DMA (k-tile t is the current one)
        //     S1  Q00 XA  W00     W10, W01, W11                          X(t+1) piece 2
        //     S2  Q01 XA  W10     XB <- X1[k0]                           X(t+1) piece 3      B2: lgkmcnt(0), barrier -> W region of stage t free
        //     S3  Q11 XB  W10     XA <- X1[k1]                           W(t+2) piece 0
        //     S4  Q10 XB  W00                                            W(t+2) piece 1
        //     S5  Q10 XA  W01     XB <- X0[k1]                           W(t+2) piece 2
        //     S6  Q11 XA  W11                                            W(t+2) piece 3      B1: vmcnt(4) lgkmcnt(0), barrier -> stage t + 1 readable,
        //     S7  Q01 XB  W11     XA <- X0[k0] of t + 1                  X(t+2) piece 0          X region of stage t free
        //     S8  Q00 XB  W01     W00 <- W0[k0] of t + 1                 X(t+2) piece 1
        // Why this shape (measured, tools/gpu_vbvar.sh): a DMA instruction is accepted only while the CU's vector-memory queue has
        // room (the fill path moves 1 KiB per ~31-45 clk), so a wave that issues its 8 pieces back to back sits in front of its own
        // MFMAs for most of the fill time -- with all 8 behind one barrier the k-loop took the SUM of its MFMA-only (0.99 us) and
        // DMA-only (1.13 us) forms; paced at one piece per stage (8 waves x 1 per ~256 clk = the queue's rate) nothing queues.  The
        // lead comes from releasing the W region early (B2) and the X region at B1: W runs a k-tile and a half ahead, X half a
        // k-tile.  vmcnt(4) at B1 leaves W(t+2)'s four pieces in flight across the barrier.  The two waves of a SIMD drift
        // apart between barriers, so one's LDS waits and DMA issue sit under the other's MFMAs.  Across tiles: the last k-tile's
        // S7 / S8 stage the next tile's X(0), W(0), W(1) and X(1) pieces 0, 1 (both stages are free behind its B1), in front of the epilogue.
        constexpr int PX = BM / 8, PW = BN / 8, STAGE_BYTES = (PX + PW) * 1024;
        constexpr int NQ = (PX + PW) / NWAVES;
        static_assert(2 * STAGE_BYTES == EP_OFF && NQ == 8 && TM == 8 && TN == 4, "256 x 256 tile, 2 x 4 waves");
        const int drow = lane >> 3, dk = ((lane & 7) ^ drow) * 8;
        // VB_SWP_HALFISSUE: only the wm = 0 waves (one per SIMD) issue DMA instructions, two per slot (16 per k-tile); their SIMD
        // partners (wm = 1) issue none and keep the matrix pipe busy meanwhile.
        constexpr bool HALF = VB_SWP_HALFISSUE != 0;
        constexpr int NSRC = HALF ? 2 * NQ : NQ;
        auto piece_of = [&](int j) { return HALF ? wn + (NWAVES / 2) * j : w + NWAVES * j; };     // j-th piece of this wave: 0 .. PX-1 X, then W
        unsigned soff[NSRC];                        // per-lane BYTE offsets of this wave's DMA sources at k-tile 0
        auto set_sources_s = [&](int m0, int n0) {
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                const int p = piece_of(i);
                if (i < NSRC / 2) {
                    int m = m0 + p * 8 + drow;
                    m = m < a.M ? m : a.M - 1;
                    unsigned base;
                    if constexpr (AMODE == A_CONV) {
                        const int FF = a.F * a.F, b = m / FF, yx = m - b * FF, y = yx / a.F, x = yx - y * a.F, P = a.F + 2;
                        base = (unsigned)(((b * P + y) * P + x) * a.C);
                    } else {
                        base = (unsigned)m * (unsigned)a.K;
                    }
                    soff[i] = (base + dk) * 2u;
                } else {
                    soff[i] = ((unsigned)(n0 + (p - PX) * 8 + drow) * (unsigned)a.K + dk) * 2u;
                }
            }
        };
        auto kx_of = [&](int kt) -> unsigned {      // element offset of k-tile kt along an X row
            if constexpr (AMODE == A_CONV) {
                const int per_tap = a.C / BK, tap = kt / per_tap, c0 = (kt - tap * per_tap) * BK, r = tap / 3, sx = tap - 3 * r;
                return (unsigned)((r * (a.F + 2) + sx) * a.C + c0);
            } else {
                return (unsigned)kt * BK;
            }
        };
        auto issue1 = [&](int stg, int kt, unsigned kx, int i) {      // DMA slot i (0-3: X pieces, 4-7: W pieces) of k-tile kt -> stage stg
#if VB_SWP_NO_DMA
            return;
#endif
            if (HALF && wm != 0) return;
            // wave-uniform 64-bit base (pinned into an SGPR pair) + 32-bit per-lane byte offset: the saddr form of the instruction,
            // one VGPR per source and no address arithmetic on the VALU; the LDS destination is wave-uniform too (M0; the hardware
            // adds lane x 16)
            const unsigned long long b64 = i < NQ / 2 ? reinterpret_cast<unsigned long long>(X) + (unsigned long long)kx * 2
                                                      : reinterpret_cast<unsigned long long>(W) + (unsigned long long)((unsigned)kt * BK) * 2;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b64), hi = __builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
            const char* base = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
#pragma unroll
            for (int jj = 0; jj < (HALF ? 2 : 1); ++jj) {
                const int j = HALF ? 2 * i + jj : i;
                unsigned off = soff[j];
                asm volatile("" : "+v"(off));    // opaque: otherwise LICM hoists base + offset as 64-bit per-lane pointers (16 registers) out of the k-loop
                glds16(base + (size_t)off, smem + stg * STAGE_BYTES + piece_of(j) * 1024);
            }
        };
        const int r7 = lane & 7, fbase = ((lane & 15) >> 3) * 1024 + r7 * 128;
        const int fk0 = fbase + ((((lane >> 4)) ^ r7) << 4), fk1 = fbase + (((4 + (lane >> 4)) ^ r7) << 4);
        const int nk = a.K / BK;                    // even (checked by the host)
        bf16x8 XA[4], XB[4], W00[2], W10[2], W01[2], W11[2];     // W<half><k-step>
        // fragment addresses: one opaque base register per (operand, k-step, stage) + an immediate below 16 KiB.  Opaque, because
        // stage 1 lies beyond the 16-bit offset field and hipcc otherwise materialises one address register PER READ (~30)
        int fxa[2][2], fwa[2][2];                    // [stage][k-step]
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                fxa[st][kk] = (wm * TM * 2) * 1024 + (kk ? fk1 : fk0) + st * STAGE_BYTES;
                fwa[st][kk] = (PX + wn * TN * 2) * 1024 + (kk ? fk1 : fk0) + st * STAGE_BYTES;
                asm volatile("" : "+v"(fxa[st][kk]), "+v"(fwa[st][kk]));
            }
        auto ldx = [&](bf16x8 (&fx)[4], auto stg, int half, int kk) {
            const char* xp = smem + fxa[decltype(stg)::value][kk] + half * 8 * 1024;
#pragma unroll
            for (int j = 0; j < 4; ++j) fx[j] = *reinterpret_cast<const bf16x8*>(xp + j * 2048);
        };
        auto ldw = [&](bf16x8 (&fw)[2], auto stg, int half, int kk) {
            const char* wp = smem + fwa[decltype(stg)::value][kk] + half * 4 * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(wp + i * 2048);
        };
        auto mma8 = [&](int xh, int wh, const bf16x8 (&fx)[4], const bf16x8 (&fw)[2]) {
#if !VB_SWP_NO_MFMA
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (EPI == EPI_VT)
                        acc[2 * wh + i][4 * xh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[j], fw[i], acc[2 * wh + i][4 * xh + j], 0, 0, 0);
                    else
                        acc[2 * wh + i][4 * xh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[2 * wh + i][4 * xh + j], 0, 0, 0);
                }
#endif
        };
        auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
        auto bar = [&]() {
#if !VB_SWP_NO_BARRIER
            __builtin_amdgcn_s_barrier();
#endif
        };
        bool more = true;
        int cm0 = m0, cn0 = n0;
        const std::integral_constant<int, 0> S0{};
        const std::integral_constant<int, 1> S1{};
        // One k-tile in stage STG -- straight-line code, no branch inside (a branch would put the MFMAs and the loads they are
        // meant to cover into different scheduling regions).  K is a multiple of 128, so a tile starts in stage 0 and ends in
        // stage 1; the last two k-tiles of a tile are their own instantiations: FILL1 = k-tile kt + 1 exists (X pieces 2, 3 in
        // S1, S2), FILL2 = k-tile kt + 2 exists (W pieces in S3-S6, X pieces 0, 1 in S7, S8), LAST = the tile's last k-tile.
        auto ktile = [&](int kt, auto stg, auto fill1, auto fill2, auto last_) {
            constexpr int STG = decltype(stg)::value;
            constexpr bool FILL1 = decltype(fill1)::value, FILL2 = decltype(fill2)::value, LAST = decltype(last_)::value;
            const std::integral_constant<int, STG ^ 1> nstg{};
            const unsigned kx1 = FILL1 ? kx_of(kt + 1) : 0u, kx2 = FILL2 ? kx_of(kt + 2) : 0u;
            // S1
            mma8(0, 0, XA, W00);
            ldw(W10, stg, 1, 0);
            ldw(W01, stg, 0, 1);
            ldw(W11, stg, 1, 1);
            if constexpr (FILL1) issue1(STG ^ 1, kt + 1, kx1, 2);
            SB();
            // S2
            mma8(0, 1, XA, W10);
            ldx(XB, stg, 1, 0);
            if constexpr (FILL1) issue1(STG ^ 1, kt + 1, kx1, 3);
            SB();
            // B2: every W fragment of this k-tile is in registers (read in S1) -> the W region of this stage may be re-staged
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            SB();
            // S3
            mma8(1, 1, XB, W10);
            ldx(XA, stg, 1, 1);
            if constexpr (FILL2) issue1(STG, kt + 2, kx2, 4);
            SB();
            // S4
            mma8(1, 0, XB, W00);
            if constexpr (FILL2) issue1(STG, kt + 2, kx2, 5);
            SB();
            // S5
            mma8(1, 0, XA, W01);
            ldx(XB, stg, 0, 1);
            if constexpr (FILL2) issue1(STG, kt + 2, kx2, 6);
            SB();
            // S6
            mma8(1, 1, XA, W11);
            if constexpr (FILL2) issue1(STG, kt + 2, kx2, 7);
            SB();
            // B1: k-tile kt + 1 has landed (all but the four W pieces just issued), this stage's X fragments are all in registers
            if constexpr (FILL2 && HALF) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");      // (wm = 1 waves have nothing outstanding)
            else if constexpr (FILL2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            bar();
            SB();
            if constexpr (!LAST) {
                // S7
                mma8(0, 1, XB, W11);
                ldx(XA, nstg, 0, 0);
                if constexpr (FILL2) issue1(STG, kt + 2, kx2, 0);
                SB();
                // S8
                mma8(0, 0, XB, W01);
                ldw(W00, nstg, 0, 0);
                if constexpr (FILL2) issue1(STG, kt + 2, kx2, 1);
                SB();
            } else {
                // both stages are free: the next tile's X(0), W(0), W(1) and X(1) pieces 0, 1 go out under the last 16 MFMAs
                vb += gridDim.x;
                more = vb < nwg;
                mma8(0, 1, XB, W11);
                if (more) {
                    tile_of(vb, m0, n0);
                    set_sources_s((a.dbg & 1) ? 0 : m0, (a.dbg & 2) ? 0 : n0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) issue1(0, 0, kx_of(0), i);
                }
                SB();
                mma8(0, 0, XB, W01);
                if (more) {
#pragma unroll
                    for (int i = 4; i < 8; ++i) issue1(1, 1, kx_of(1), i);
                    issue1(1, 1, kx_of(1), 0);
                    issue1(1, 1, kx_of(1), 1);
                }
                SB();
            }
        };
        if (a.desync_ticks > 0) {       // phase offset of this workgroup (all its tiles take the same time, so the offset persists)
            const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) % a.desync_groups) * a.desync_ticks / a.desync_groups;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
        set_sources_s((a.dbg & 1) ? 0 : m0, (a.dbg & 2) ? 0 : n0);
#pragma unroll
        for (int i = 0; i < 8; ++i) issue1(0, 0, kx_of(0), i);
#pragma unroll
        for (int i = 4; i < 8; ++i) issue1(1, 1, kx_of(1), i);
        issue1(1, 1, kx_of(1), 0);
        issue1(1, 1, kx_of(1), 1);
        const std::true_type T{};
        const std::false_type Fa{};
        for (;;) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ldx(XA, S0, 0, 0);
            ldw(W00, S0, 0, 0);
            zero_acc();
            cm0 = m0; cn0 = n0;
            for (int kt = 0; kt + 2 < nk; kt += 2) {
                ktile(kt, S0, T, T, Fa);
                ktile(kt + 1, S1, T, T, Fa);
            }
            ktile(nk - 2, S0, T, Fa, Fa);
            ktile(nk - 1, S1, Fa, Fa, T);
            if (!(a.dbg & 8)) epilogue(cm0, cn0);
            if (!more) break;
            stage_vec(m0, n0);           // the next tile's bias / rstd; the wait at the top of the loop covers them
        }
    } else {
        const int nk = a.K / BK;
        // ---- two buffers of one 64-deep k-tile; the DMA of the next tile's first k-tile is issued ahead of the epilogue
        set_sources(m0, n0);
        stage(0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (;;) {
            zero_acc();
            for (int kt = 0; kt < nk; ++kt) {
                char* cur = smem + (kt & 1) * BUF_BYTES;
                if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * BUF_BYTES);
                const char* xp = cur + (wm * TM * 2) * 1024 + fr;
                const char* wp = cur + (SX + wn * TN * 2) * 1024 + fr;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) mfma_step(xp + kk * 1024, wp + kk * 1024, 2048);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // every wave has passed the barrier that ends the last k-tile: both LDS buffers are free
            const int cm0 = m0, cn0 = n0;
            vb += gridDim.x;
            const bool more = vb < nwg;
            if (more) {
                tile_of(vb, m0, n0);
                set_sources(m0, n0);
                stage(0, smem);
            }
            epilogue(cm0, cn0);
            if (!more) break;
            stage_vec(m0, n0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

template <int BM, int BN>
constexpr int lds_bytes() { return 2 * (BM / 16 * 2 + BN / 16 * 2) * 1024 + NWAVES * 4096; }   // pipeline stages + epilogue staging

}  // namespace vbg
