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
// of a DMA stays lane-linear.  Two LDS buffers: the DMA of k-tile t+1 is issued before the MFMAs of tile t, one
// vmcnt(0) + barrier per k-tile.
//
// XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so id -> (id % 8) * per_xcd +
// id / 8 gives every XCD a contiguous run of tiles, n fastest: the X panel of a tile row is fetched into that XCD's L2
// once and re-used by all its column tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vt_common.h"

namespace vbg {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int BK = 64;
constexpr int NWAVES = 8;

enum AMode { A_PLAIN = 0, A_CONV = 1 };
enum Epi {
    EPI_PATCH = 0,   // resid[m][n] = acc + bias[n] + pos[m % L][n]                     (f32 out)
    EPI_BF16 = 1,    // out[m][n] = bf16(acc + bias[n])                                  (q (pre-scaled) and k projections)
    EPI_RESID = 2,   // resid[m][n] += acc + bias[n]                                     (f32 read-modify-write)
    EPI_GELU = 3,    // out[m][n] = bf16(gelu_erf(acc + bias[n]))
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
};

__device__ __forceinline__ int swz_byte(int p) { return p ^ (((p >> 9) & 1) << 5); }   // st_16x32, involution on [0, 1024)

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 to_bf16x4(f4 v) {
    const bf16x2 lo = __builtin_convertvector(f2{v.x, v.y}, bf16x2), hi = __builtin_convertvector(f2{v.z, v.w}, bf16x2);
    return bf16x4{lo.x, lo.y, hi.x, hi.y};
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
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
        const int tm = t / tiles_n;
        m0 = tm * BM;
        n0 = (t - tm * tiles_n) * BN;
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
    const int q4 = (lane >> 4) * 4, l15 = lane & 15;
    // Normal tiles: lane holds n = nb + 4q + {0..3} of token m = mb + (lane & 15).
    auto epilogue = [&](int m0, int n0) {
        if constexpr (EPI == EPI_VT) {
            // swapped tile: lane holds tokens m = mb + 4q + {0..3} of feature n = nb + (lane & 15)
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int n = n0 + wn * TN * 16 + i * 16 + l15;
                const float b = bias[n];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int m = m0 + (wm * TM + j) * 16 + q4;
                    if (m < a.M) {     // M is a multiple of 4 (L is)
                        const int f = m / a.L, t = m - f * a.L;
                        *reinterpret_cast<bf16x4*>(a.vt + ((size_t)f * a.N + n) * a.L + t) = to_bf16x4(acc[i][j] + splat4(b));
                    }
                }
            }
            return;
        }
        f4 bv[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + (wn * TN + i) * 16 + q4;
            bv[i] = n < a.N ? ld4(bias + n) : splat4(0.f);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + (wn * TN + i) * 16 + q4;
            if (n >= a.N) continue;                           // N is a multiple of 4; W rows beyond N are zero padding
            if constexpr (EPI == EPI_RESID || EPI == EPI_PATCH) {
                // read-modify-write of the f32 residual stream: all TM loads of a column tile are issued before the first
                // store (a store to `resid` may alias the next load as far as the compiler knows, which would otherwise
                // serialise TM x TN full memory round trips)
                f4 old[TM];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    int m = m0 + (wm * TM + j) * 16 + l15;
                    m = m < a.M ? m : a.M - 1;
                    if constexpr (EPI == EPI_RESID) old[j] = ld4(a.resid + (size_t)m * a.N + n);
                    else old[j] = ld4(a.pos + (size_t)(m % a.L) * a.N + n);
                }
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int m = m0 + (wm * TM + j) * 16 + l15;
                    if (m < a.M) st4(a.resid + (size_t)m * a.N + n, old[j] + acc[i][j] + bv[i]);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + (wm * TM + j) * 16 + l15;
                if (m >= a.M) continue;
                const f4 v = acc[i][j] + bv[i];
                if constexpr (EPI == EPI_BF16) {
                    *reinterpret_cast<bf16x4*>(static_cast<bf16*>(a.out) + (size_t)m * a.ldo + n) = to_bf16x4(v);
                } else if constexpr (EPI == EPI_GELU) {
                    const f4 g = {gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
                    *reinterpret_cast<bf16x4*>(static_cast<bf16*>(a.out) + (size_t)m * a.ldo + n) = to_bf16x4(g);
                } else if constexpr (EPI == EPI_CONV) {
                    const f4 r = {fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                    size_t row = (size_t)m;
                    if (a.out_padded) {
                        const int FF = a.F * a.F, bb = m / FF, yx = m - bb * FF, y = yx / a.F, x = yx - y * a.F, P = a.F + 2;
                        row = (size_t)((bb * P + y + 1) * P + x + 1);
                    }
                    bf16* o = static_cast<bf16*>(a.out) + grp * a.gOut;
                    int nn = n;
                    if (a.n_split) { const int t = n / a.n_split; o += (size_t)t * a.gOut; nn = n - t * a.n_split; }
                    *reinterpret_cast<bf16x4*>(o + row * a.ldo + nn) = to_bf16x4(r);
                }
            }
        }
    };

    const int fr = swz_byte((lane & 15) * 64 + (lane >> 4) * 16);   // fragment byte inside a sub-tile
    const int nk = a.K / BK;
    // ---- persistent loop over this workgroup's tiles: the DMA of the next tile's first k-tile is issued before the
    // epilogue of the current one, so its latency hides behind the epilogue's stores
    int vb = blockIdx.x, m0, n0;
    if (vb >= nwg) return;
    tile_of(vb, m0, n0);
    set_sources(m0, n0);
    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (;;) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = splat4(0.f);
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + (kt & 1) * BUF_BYTES;
            if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * BUF_BYTES);
            const char* xp = cur + (wm * TM * 2) * 1024 + fr;
            const char* wp = cur + (SX + wn * TN * 2) * 1024 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fw[TN], fx[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(wp + (i * 2 + kk) * 1024);
#pragma unroll
                for (int j = 0; j < TM; ++j) fx[j] = *reinterpret_cast<const bf16x8*>(xp + (j * 2 + kk) * 1024);
                if constexpr (EPI == EPI_VT) {
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[j], fw[i], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
                }
            }
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <int BM, int BN>
constexpr int lds_bytes() { return 2 * (BM / 16 * 2 + BN / 16 * 2) * 1024; }

}  // namespace vbg
