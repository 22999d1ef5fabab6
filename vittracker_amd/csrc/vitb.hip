// vitb.hip -- host side of the ViT-Base OSTrack path (BASELINE config 4): weight packing (bf16, BatchNorm folded, conv
// weights as [cout][tap][cin], attention scale folded into W_q), workspace, launch sequence.  Kernels: vb_gemm.h,
// vb_attn.h, vb_misc.h.  Reached through the same C ABI as the vit_48 path (vt_create with channels = 768).
#include "vb_api.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vb_attn.h"
#include "vb_gemm.h"
#include "vb_misc.h"
#include "vb_qkvattn.h"

using vbg::bf16;

namespace {

constexpr int C = 768, HEADS = 12, HD = 64, HID = 3072, L = 320, LZ = 64, LX = 256, F = 16, PATCH_K = 768, HW = 256 /* head width */;
constexpr float LN_EPS = 1e-6f;
constexpr int HEAD_CH[5] = {768, 256, 128, 64, 32};

struct Err {
    std::string* e;
    int fail(int code, const std::string& msg) const { if (e) *e = msg; return code; }
};

#define VB_HIP(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return E.fail(VT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

uint16_t f2bf(float f) {   // round to nearest even, NaN kept quiet
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T>
struct Buf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) { n = count; return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)); }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
};

struct BlockW {
    Buf<float> ln1g, ln1b, ln2g, ln2b, bqkv, bproj, b1, b2;
    Buf<bf16> wqkv, wproj, w1, w2;
};

}  // namespace

struct VbModel {
    vt_config cfg{};
    int depth = 12, maxB = 0;
    bool loaded = false;
    // parameters
    Buf<bf16> wpatch; Buf<float> bpatch, pos, ng, nb;
    std::vector<BlockW> blk;
    Buf<bf16> wc[4]; Buf<float> bc[4], w5, b5;     // head: conv1 (towers along N), conv2..4 ([3][cout_padded][9 cin])
    // workspace
    Buf<bf16> xn, qk, vt, ao, hid, map0, map1, map2, map3, t4;
    Buf<float> resid;
    // LayerNorm folded into qkv / fc1 (vb_gemm.h): xn then holds the RAW residual rows in bf16, written by the GEMM epilogues that
    // produce them, rstd their 1 / sqrt(var + eps), stats the epilogues' per-slice (sum, M2) pairs [C / 64][max rows]
    bool fold = true;
    bool fused_qkv = true;          // VB_FUSED_QKV: the qkv projection inside the attention kernel (vb_qkvattn.h) instead of qk GEMM + v GEMM + attention
    Buf<float> rstd;
    Buf<float> rmean;        // per-row mean of the residual stream as of the last finalize: the next producer centres its bf16 copy on it
    Buf<float> cpos;         // [L]: mean over channels of (pos-embed row + patch bias): the patch GEMM's centring constant per token
    bool center = true;      // VB_LN_CENTER
    Buf<vbg::f2> stats;
};

namespace {

int need(const vb::TensorMap& tm, const std::string& name, int64_t numel, const float** out, const Err& E) {
    auto it = tm.find(name);
    if (it == tm.end()) return E.fail(VT_ERR_MISSING_KEY, "missing key in state dict: " + name);
    if (it->second.second != numel)
        return E.fail(VT_ERR_MISSING_KEY, "shape mismatch for " + name + ": got " + std::to_string(it->second.second) +
                                              " elements, want " + std::to_string(numel));
    *out = it->second.first;
    return VT_OK;
}

template <typename T>
int upload(Buf<T>& d, const void* h, size_t count, const Err& E) {
    if (!d.p || d.n != count) {
        d.release();
        VB_HIP(d.alloc(count));
    }
    VB_HIP(hipMemcpy(d.p, h, count * sizeof(T), hipMemcpyHostToDevice));
    return VT_OK;
}

int upload_bf16(Buf<bf16>& d, const std::vector<float>& h, const Err& E) {
    std::vector<uint16_t> t(h.size());
    for (size_t i = 0; i < h.size(); ++i) t[i] = f2bf(h[i]);
    return upload(d, t.data(), t.size(), E);
}
int upload_f32(Buf<float>& d, const float* h, size_t n, const Err& E) { return upload(d, h, n, E); }

int num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t prop;
        n = hipGetDeviceProperties(&prop, 0) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return n;
}

inline int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
int launch_gemm(const vbg::Args& a, int groups, hipStream_t st, const Err& E, int cus = 0) {
    if (a.M < 1 || a.K % vbg::BK != 0 || a.N % 8 != 0)
        return E.fail(VT_ERR_ARG, "gemm shape: K must be a multiple of 64 and N of 8 (got M=" + std::to_string(a.M) + " N=" +
                                      std::to_string(a.N) + " K=" + std::to_string(a.K) + ")");
    if (BM == 256 && BN == 256 && a.K % (2 * vbg::BK) != 0) return E.fail(VT_ERR_ARG, "gemm shape: the 256 x 256 tile needs K to be a multiple of 128");
    static const int dbg = [] { const char* v = std::getenv("VB_DBG"); return v ? std::atoi(v) : 0; }();
    vbg::Args ad = a;
    ad.dbg = dbg;
    {   // experiment hook: VB_RB_<epilogue id>=rows overrides the tile-row block of that GEMM kind
        static const int rbs[6] = {env_int("VB_RB_0", -1), env_int("VB_RB_1", -1), env_int("VB_RB_2", -1), env_int("VB_RB_3", -1),
                                   env_int("VB_RB_4", -1), env_int("VB_RB_5", -1)};
        if (rbs[EPI] >= 0) ad.rb = rbs[EPI];
    }
    {   // experiment hook: VB_DESYNC_<epilogue id>=<us>[:groups] -- phase groups of workgroups (Args::desync_ticks)
        static const struct D { int us[6], g[6]; D() {
            for (int e = 0; e < 6; ++e) {
                const std::string n = "VB_DESYNC_" + std::to_string(e);
                const char* v = std::getenv(n.c_str());
                us[e] = v ? std::atoi(v) : 0;
                const char* c = v ? std::strchr(v, ':') : nullptr;
                g[e] = c ? std::max(2, std::atoi(c + 1)) : 2;
            } } } ds;
        if (ds.us[EPI] > 0) { ad.desync_ticks = ds.us[EPI] * 100; ad.desync_groups = ds.g[EPI]; }
    }
    // persistent workgroups: one per CU, each walks tiles blockIdx.x, blockIdx.x + grid, ...
    const int ntiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static const int max_cus = env_int("VB_MAX_CUS", 0);      // experiment hook: persistent grid of at most this many workgroups
    if (max_cus > 0) cus = cus > 0 ? std::min(cus, max_cus) : max_cus;
    const int tiles = std::min(ntiles, std::max(8, (cus > 0 ? std::min(cus, num_cus()) : num_cus()) / groups / 8 * 8));
    constexpr int lds = vbg::lds_bytes<BM, BN>();
    hipLaunchKernelGGL((vbg::gemm_kernel<BM, BN, WM, WN, AMODE, EPI>), dim3(tiles, groups), dim3(512), lds, st, ad);
    VB_HIP(hipGetLastError());
    return VT_OK;
}

template <typename K>
hipError_t allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int run_layernorm(const float* resid, const float* g, const float* b, int B, hipStream_t st, bf16* xn, bf16* map, float* feat, const Err& E,
                  bf16* xb = nullptr, float* rstd = nullptr, float* mean = nullptr) {
    const int M = B * L;
    hipLaunchKernelGGL((vbm::layernorm_kernel<C>), dim3((M + 3) / 4), dim3(256), 0, st, resid, g, b, LN_EPS, M, L, LZ, F, xn, map, feat, xb,
                       rstd, mean);
    VB_HIP(hipGetLastError());
    return VT_OK;
}

constexpr int STAT_P = C / 64;       // (sum, M2) pairs per residual row: one per 64-column wave slice of the 256-wide GEMM tiles

int run_finalize(const VbModel* m, size_t r0, int M, hipStream_t st, const Err& E) {
    hipLaunchKernelGGL(vbm::ln_finalize_kernel<STAT_P>, dim3((M + 255) / 256), dim3(256), 0, st, m->stats.p + r0, (int)(m->stats.n / STAT_P), M, LN_EPS,
                       m->rstd.p + r0, m->rmean.p + r0);
    VB_HIP(hipGetLastError());
    return VT_OK;
}

// LayerNorm(gamma, beta) folded into the Linear(W [N][K], b) that consumes it, in double (vb_gemm.h header):
//     W'[n][k] = W[n][k] gamma[k] - mean_k(W[n][.] gamma[.]),   b'[n] = b[n] + sum_k W[n][k] beta[k]
void fold_layernorm(std::vector<float>& w, std::vector<float>& b, const float* gamma, const float* beta, int N, int K) {
    for (int n = 0; n < N; ++n) {
        float* row = w.data() + (size_t)n * K;
        double sb = 0.0, sg = 0.0;
        for (int k = 0; k < K; ++k) { sb += (double)row[k] * beta[k]; sg += (double)row[k] * gamma[k]; }
        const double mu = sg / K;
        for (int k = 0; k < K; ++k) row[k] = (float)((double)row[k] * gamma[k] - mu);
        b[n] = (float)((double)b[n] + sb);
    }
}

}  // namespace

namespace vb {

int create(const vt_config* cfg, VbModel** out, std::string* err) {
    const Err E{err};
    if (cfg->channels != C || cfg->heads != HEADS || cfg->head_channels != HW || cfg->stride != 16 || cfg->template_size != 128 ||
        cfg->search_size != 256)
        return E.fail(VT_ERR_ARG, "unsupported ViT-Base configuration: this build implements CHANNELS=768, HEADS=12, "
                                  "HEAD.NUM_CHANNELS=256, STRIDE=16, template 128 / search 256 (OSTrack-256)");
    if (cfg->depth < 1 || cfg->depth > 24 || cfg->max_batch < 1) return E.fail(VT_ERR_ARG, "bad depth / max_batch");
    VbModel* m = new VbModel();
    m->cfg = *cfg;
    m->depth = cfg->depth;
    m->maxB = cfg->max_batch;
    m->blk.resize(m->depth);
    m->fold = env_int("VB_LN_FOLD", 1) != 0;
    m->center = env_int("VB_LN_CENTER", 1) != 0;
    m->fused_qkv = env_int("VB_FUSED_QKV", 1) != 0;
    const size_t B = (size_t)cfg->max_batch, M = B * L, P2 = (size_t)(F + 2) * (F + 2);
    hipError_t e = hipSuccess;
    auto A = [&](auto& buf, size_t n) { if (e == hipSuccess) e = buf.alloc(n); };
    A(m->xn, M * C); A(m->resid, M * C); A(m->qk, M * 2 * C); A(m->vt, M * C); A(m->ao, M * C); A(m->hid, M * HID);
    A(m->map0, B * P2 * C); A(m->map1, 3 * B * P2 * HEAD_CH[1]); A(m->map2, 3 * B * P2 * HEAD_CH[2]);
    A(m->map3, 3 * B * P2 * HEAD_CH[3]); A(m->t4, 3 * B * LX * HEAD_CH[4]);
    A(m->rstd, M); A(m->stats, (size_t)STAT_P * M); A(m->rmean, M);
    // zero borders of the padded maps (kernels only ever write interiors)
    if (e == hipSuccess) e = hipMemset(m->map0.p, 0, m->map0.n * 2);
    if (e == hipSuccess) e = hipMemset(m->map1.p, 0, m->map1.n * 2);
    if (e == hipSuccess) e = hipMemset(m->map2.p, 0, m->map2.n * 2);
    if (e == hipSuccess) e = hipMemset(m->map3.p, 0, m->map3.n * 2);
    using namespace vbg;
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 256, 2, 4, A_PLAIN, EPI_PATCH>, lds_bytes<256, 256>());
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 256, 2, 4, A_PLAIN, EPI_BF16>, lds_bytes<256, 256>());
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 256, 2, 4, A_PLAIN, EPI_VT>, lds_bytes<256, 256>());
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 256, 2, 4, A_PLAIN, EPI_RESID>, lds_bytes<256, 256>());
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 256, 2, 4, A_PLAIN, EPI_GELU>, lds_bytes<256, 256>());
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 256, 2, 4, A_CONV, EPI_CONV>, lds_bytes<256, 256>());
    if (e == hipSuccess) e = allow_lds(gemm_kernel<256, 64, 8, 1, A_CONV, EPI_CONV>, lds_bytes<256, 64>());
    if (e == hipSuccess) e = allow_lds(vba::attn_kernel<L, HD>, vba::Geo<L, HD>::LDS_BYTES);
    if (e == hipSuccess) e = allow_lds(vbq::qkv_attn_kernel, vbq::LDS_BYTES);
    if (e != hipSuccess) {
        destroy(m);
        return E.fail(VT_ERR_HIP, std::string("ViT-Base workspace: ") + hipGetErrorString(e));
    }
    *out = m;
    return VT_OK;
}

void destroy(VbModel* m) {
    if (!m) return;
    m->wpatch.release(); m->bpatch.release(); m->pos.release(); m->ng.release(); m->nb.release();
    for (BlockW& b : m->blk) {
        b.ln1g.release(); b.ln1b.release(); b.ln2g.release(); b.ln2b.release(); b.bqkv.release(); b.bproj.release();
        b.b1.release(); b.b2.release(); b.wqkv.release(); b.wproj.release(); b.w1.release(); b.w2.release();
    }
    for (int i = 0; i < 4; ++i) { m->wc[i].release(); m->bc[i].release(); }
    m->w5.release(); m->b5.release();
    m->xn.release(); m->qk.release(); m->vt.release(); m->ao.release(); m->hid.release(); m->map0.release();
    m->map1.release(); m->map2.release(); m->map3.release(); m->t4.release(); m->resid.release();
    m->rstd.release(); m->stats.release(); m->rmean.release(); m->cpos.release();
    delete m;
}

// Key layout: the reference's OSTrack ckpt['net'] -- backbone.{patch_embed.proj, pos_embed_z, pos_embed_x, blocks.N.*, norm},
// box_head.* (lib/models/ostrack/vit.py:94-139, base_backbone.py:83-84, lib/models/layers/head.py:98-128).
int load_weights(VbModel* m, const TensorMap& tm, std::string* err) {
    const Err E{err};
    int rc;
    const float* p;
    const std::string bb = "backbone.";
    if ((rc = need(tm, bb + "patch_embed.proj.weight", (int64_t)C * PATCH_K, &p, E))) return rc;
    if ((rc = upload_bf16(m->wpatch, std::vector<float>(p, p + (size_t)C * PATCH_K), E))) return rc;
    if ((rc = need(tm, bb + "patch_embed.proj.bias", C, &p, E))) return rc;
    if ((rc = upload_f32(m->bpatch, p, C, E))) return rc;
    {
        std::vector<float> pos((size_t)L * C);
        if ((rc = need(tm, bb + "pos_embed_z", (int64_t)LZ * C, &p, E))) return rc;
        std::memcpy(pos.data(), p, (size_t)LZ * C * 4);
        if ((rc = need(tm, bb + "pos_embed_x", (int64_t)LX * C, &p, E))) return rc;
        std::memcpy(pos.data() + (size_t)LZ * C, p, (size_t)LX * C * 4);
        if ((rc = upload_f32(m->pos, pos.data(), pos.size(), E))) return rc;
        // the patch GEMM's centring constants: a token row is patches W^T + bias + pos row; what is known of its mean before the GEMM runs
        // is mean(bias) + mean(pos row) (the projection of a normalised patch is zero-mean over channels to first order)
        const float* bp = nullptr;
        if ((rc = need(tm, bb + "patch_embed.proj.bias", C, &bp, E))) return rc;
        double bmean = 0;
        for (int k = 0; k < C; ++k) bmean += bp[k];
        std::vector<float> cpos(L);
        for (int t = 0; t < L; ++t) {
            double s = 0;
            for (int k = 0; k < C; ++k) s += pos[(size_t)t * C + k];
            cpos[t] = (float)((s + bmean) / C);
        }
        if ((rc = upload_f32(m->cpos, cpos.data(), cpos.size(), E))) return rc;
    }
    const float scale = 1.0f / std::sqrt((float)HD);     // 0.125: a power of two, folding it into W_q / b_q is exact
    for (int i = 0; i < m->depth; ++i) {
        BlockW& b = m->blk[i];
        const std::string pre = bb + "blocks." + std::to_string(i) + ".";
        struct V { const char* name; Buf<float>* dst; int n; };
        const V vecs[] = {{"norm1.weight", &b.ln1g, C}, {"norm1.bias", &b.ln1b, C}, {"norm2.weight", &b.ln2g, C}, {"norm2.bias", &b.ln2b, C},
                          {"attn.proj.bias", &b.bproj, C}, {"mlp.fc2.bias", &b.b2, C}};
        for (const V& v : vecs) {
            if ((rc = need(tm, pre + v.name, v.n, &p, E))) return rc;
            if ((rc = upload_f32(*v.dst, p, v.n, E))) return rc;
        }
        const float *g1, *be1, *g2, *be2;
        if ((rc = need(tm, pre + "norm1.weight", C, &g1, E)) || (rc = need(tm, pre + "norm1.bias", C, &be1, E))) return rc;
        if ((rc = need(tm, pre + "norm2.weight", C, &g2, E)) || (rc = need(tm, pre + "norm2.bias", C, &be2, E))) return rc;
        if ((rc = need(tm, pre + "attn.qkv.weight", (int64_t)3 * C * C, &p, E))) return rc;
        std::vector<float> w(p, p + (size_t)3 * C * C);
        if ((rc = need(tm, pre + "attn.qkv.bias", 3 * C, &p, E))) return rc;
        std::vector<float> bq(p, p + 3 * C);
        if (m->fold) fold_layernorm(w, bq, g1, be1, 3 * C, C);
        for (size_t k = 0; k < (size_t)C * C; ++k) w[k] *= scale;
        for (int k = 0; k < C; ++k) bq[k] *= scale;
        if ((rc = upload_bf16(b.wqkv, w, E))) return rc;
        if ((rc = upload_f32(b.bqkv, bq.data(), bq.size(), E))) return rc;
        if ((rc = need(tm, pre + "attn.proj.weight", (int64_t)C * C, &p, E))) return rc;
        if ((rc = upload_bf16(b.wproj, std::vector<float>(p, p + (size_t)C * C), E))) return rc;
        if ((rc = need(tm, pre + "mlp.fc1.weight", (int64_t)HID * C, &p, E))) return rc;
        std::vector<float> w1(p, p + (size_t)HID * C);
        if ((rc = need(tm, pre + "mlp.fc1.bias", HID, &p, E))) return rc;
        std::vector<float> b1(p, p + HID);
        if (m->fold) fold_layernorm(w1, b1, g2, be2, HID, C);
        if ((rc = upload_bf16(b.w1, w1, E))) return rc;
        if ((rc = upload_f32(b.b1, b1.data(), b1.size(), E))) return rc;
        if ((rc = need(tm, pre + "mlp.fc2.weight", (int64_t)HID * C, &p, E))) return rc;
        if ((rc = upload_bf16(b.w2, std::vector<float>(p, p + (size_t)HID * C), E))) return rc;
    }
    if ((rc = need(tm, bb + "norm.weight", C, &p, E))) return rc;
    if ((rc = upload_f32(m->ng, p, C, E))) return rc;
    if ((rc = need(tm, bb + "norm.bias", C, &p, E))) return rc;
    if ((rc = upload_f32(m->nb, p, C, E))) return rc;
    // ---- head: Conv3x3(+bias) + BatchNorm(eval, eps 1e-5) folded in double (head.py:8-21), weights as [cout][tap][cin]
    const char* towers[3] = {"ctr", "offset", "size"};
    for (int li = 0; li < 4; ++li) {
        const int cin = HEAD_CH[li], cout = HEAD_CH[li + 1], K = 9 * cin;
        // conv1: towers along N.  conv2 (N = 128 per tower, K = 2304) runs on the 256 x 256 software-pipelined tile, one group per tower, with
        // its weight rows zero-padded to 256 (half of every MFMA column block is padding, and it still beats the two-buffer 256 x 64 loop:
        // 268 -> ~180 us, round 5); conv3 / conv4 keep the narrow tile (BN = 64: rows padded to 64)
        const int rows = li == 0 ? cout : (li == 1 ? 256 : (cout < 64 ? 64 : cout));
        std::vector<float> w((size_t)3 * rows * K, 0.f), bias((size_t)3 * cout);
        for (int t = 0; t < 3; ++t) {
            const std::string cn = std::string("box_head.conv") + std::to_string(li + 1) + "_" + towers[t];
            const float *pw, *pb, *g, *beta, *mu, *var;
            if ((rc = need(tm, cn + ".0.weight", (int64_t)cout * cin * 9, &pw, E))) return rc;
            if ((rc = need(tm, cn + ".0.bias", cout, &pb, E))) return rc;
            if ((rc = need(tm, cn + ".1.weight", cout, &g, E))) return rc;
            if ((rc = need(tm, cn + ".1.bias", cout, &beta, E))) return rc;
            if ((rc = need(tm, cn + ".1.running_mean", cout, &mu, E))) return rc;
            if ((rc = need(tm, cn + ".1.running_var", cout, &var, E))) return rc;
            for (int o = 0; o < cout; ++o) {
                const double k = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
                float* dst = w.data() + ((size_t)t * rows + o) * K;
                for (int c = 0; c < cin; ++c)
                    for (int tap = 0; tap < 9; ++tap) dst[(size_t)tap * cin + c] = (float)((double)pw[((size_t)o * cin + c) * 9 + tap] * k);
                bias[(size_t)t * cout + o] = (float)(((double)pb[o] - (double)mu[o]) * k + (double)beta[o]);
            }
        }
        if ((rc = upload_bf16(m->wc[li], w, E))) return rc;
        if ((rc = upload_f32(m->bc[li], bias.data(), bias.size(), E))) return rc;
    }
    {
        const int CW = HEAD_CH[4];
        std::vector<float> w5((size_t)5 * CW), b5(5);
        int row = 0;
        for (int t = 0; t < 3; ++t) {
            const int nout = t == 0 ? 1 : 2;
            const std::string c5 = std::string("box_head.conv5_") + towers[t];
            if ((rc = need(tm, c5 + ".weight", (int64_t)nout * CW, &p, E))) return rc;
            std::memcpy(w5.data() + (size_t)row * CW, p, (size_t)nout * CW * 4);
            if ((rc = need(tm, c5 + ".bias", nout, &p, E))) return rc;
            std::memcpy(b5.data() + row, p, nout * 4);
            row += nout;
        }
        if ((rc = upload_f32(m->w5, w5.data(), w5.size(), E))) return rc;
        if ((rc = upload_f32(m->b5, b5.data(), 5, E))) return rc;
    }
    m->loaded = true;
    return VT_OK;
}

static int check(VbModel* m, int B, const Err& E) {
    if (!m->loaded) return E.fail(VT_ERR_STATE, "vt_load_weights has not been called");
    if (B < 1 || B > m->maxB) return E.fail(VT_ERR_STATE, "batch " + std::to_string(B) + " outside [1, max_batch=" + std::to_string(m->maxB) + "]");
    return VT_OK;
}

static int check_slice(VbModel* m, int B, const Slice* sl, const Err& E) {
    if (sl && (sl->Btot < 1 || sl->Btot > m->maxB || sl->f0 + (size_t)B > (size_t)sl->Btot))
        return E.fail(VT_ERR_STATE, "frame slice outside the batch");
    return VT_OK;
}

int stem(VbModel* m, const float* z, const float* x, int B, hipStream_t st, float* tokens_out, std::string* err, const Slice* sl) {
    const Err E{err};
    int rc = check(m, B, E);
    if (rc || (rc = check_slice(m, B, sl, E))) return rc;
    const int M = B * L;
    const size_t r0 = (sl ? sl->f0 : 0) * L;          // first token row of the slice
    const int cus = sl ? sl->cus : 0;
    bf16* const xn = m->xn.p + r0 * C;
    bf16* const patches = (m->fold ? m->ao.p : m->xn.p) + r0 * C;     // folded: xn receives the tokens' bf16 copy from the GEMM epilogue
    float* const resid = m->resid.p + r0 * C;
    const size_t items = (size_t)M * 96;
    hipLaunchKernelGGL(vbm::patchify_kernel, dim3((unsigned)std::min<size_t>((items + 255) / 256, 16384)), dim3(256), 0, st, z, x, patches,
                       B, 128, 256);
    VB_HIP(hipGetLastError());
    vbg::Args a{};
    a.X = patches; a.W = m->wpatch.p; a.bias = m->bpatch.p; a.resid = resid; a.pos = m->pos.p;
    a.M = M; a.N = C; a.K = PATCH_K; a.L = L;
    if (m->fold) { a.xb = xn; a.stats = m->stats.p + r0; a.ldstats = (int)(m->stats.n / STAT_P); }
    if (m->fold && m->center) { a.cm = m->cpos.p; a.cm_mod = L; }
    if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_PLAIN, vbg::EPI_PATCH>(a, 1, st, E, cus))) return rc;
    if (m->fold && (rc = run_finalize(m, r0, M, st, E))) return rc;
    if (tokens_out) VB_HIP(hipMemcpyAsync(tokens_out, resid, (size_t)M * C * 4, hipMemcpyDeviceToDevice, st));
    return VT_OK;
}

int blocks(VbModel* m, const float* tokens_in, int B, int nblocks, hipStream_t st, float* feat_out, float* resid_out, std::string* err,
           const Slice* sl) {
    const Err E{err};
    int rc = check(m, B, E);
    if (rc || (rc = check_slice(m, B, sl, E))) return rc;
    const int M = B * L;
    const size_t f0 = sl ? sl->f0 : 0, r0 = f0 * L;
    const int cus = sl ? sl->cus : 0;
    bf16* const xn = m->xn.p + r0 * C;
    float* const resid = m->resid.p + r0 * C;
    bf16* const qk = m->qk.p + r0 * 2 * C;
    bf16* const vt = m->vt.p + r0 * C;                 // [frame][C][L]
    bf16* const ao = m->ao.p + r0 * C;
    bf16* const hid = m->hid.p + r0 * HID;
    bf16* const map0 = m->map0.p + f0 * (size_t)(F + 2) * (F + 2) * C;
    if (nblocks < 0 || nblocks > m->depth) nblocks = m->depth;
    if (tokens_in && tokens_in != resid)
        VB_HIP(hipMemcpyAsync(resid, tokens_in, (size_t)M * C * 4, hipMemcpyDeviceToDevice, st));
    const bool fold = m->fold;
    float* const rstd = fold ? m->rstd.p + r0 : nullptr;
    vbg::f2* const stats = m->stats.p + r0;
    const int ldstats = (int)(m->stats.n / STAT_P);
    // folded LayerNorms: a residual stream from outside has no bf16 copy / rstd yet (vb::stem leaves both behind its GEMM)
    const float* const cmean = (fold && m->center) ? m->rmean.p + r0 : nullptr;
    if (fold && tokens_in && nblocks > 0 && (rc = run_layernorm(resid, nullptr, nullptr, B, st, nullptr, nullptr, nullptr, E, xn, rstd, m->rmean.p + r0))) return rc;
    for (int i = 0; i < nblocks; ++i) {
        const BlockW& b = m->blk[i];
        if (!fold && (rc = run_layernorm(resid, b.ln1g.p, b.ln1b.p, B, st, xn, nullptr, nullptr, E))) return rc;
        if (m->fused_qkv) {          // projection + attention of a (frame, head) in one workgroup: q / k / v^T never leave the CU
            vbq::Args qa{};
            qa.X = xn; qa.W = b.wqkv.p; qa.bias = b.bqkv.p; qa.rstd = rstd; qa.out = ao; qa.B = B; qa.heads = HEADS;
            { static const int hgv = env_int("VB_QA_HGROUP", 6); qa.hgroup = (hgv > 0 && HEADS % hgv == 0) ? hgv : HEADS; }
            const int grid = std::max(8, (cus > 0 ? std::min(cus, num_cus()) : num_cus()) / 8 * 8);
            hipLaunchKernelGGL(vbq::qkv_attn_kernel, dim3(grid), dim3(512), vbq::LDS_BYTES, st, qa);
            VB_HIP(hipGetLastError());
        } else {
        vbg::Args a{};
        a.rstd = rstd;
        a.X = xn; a.W = b.wqkv.p; a.bias = b.bqkv.p; a.out = qk;          // q | k: rows 0 .. 2C of W_qkv
        a.M = M; a.N = 2 * C; a.K = C; a.ldo = 2 * C; a.rb = 4;    // 4 tile rows x 8 columns per XCD: measured 4 % faster than row-major
        if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_PLAIN, vbg::EPI_BF16>(a, 1, st, E, cus))) return rc;
        vbg::Args v{};
        v.X = xn; v.W = b.wqkv.p + (size_t)2 * C * C; v.bias = b.bqkv.p + 2 * C; v.vt = vt;   // v: rows 2C .. 3C, stored transposed
        v.M = M; v.N = C; v.K = C; v.L = L; v.rstd = rstd;
        if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_PLAIN, vbg::EPI_VT>(v, 1, st, E, cus))) return rc;
        constexpr int attn_lds = vba::Geo<L, HD>::LDS_BYTES;
        hipLaunchKernelGGL((vba::attn_kernel<L, HD>), dim3(B * HEADS), dim3(256), attn_lds, st, qk, vt, ao, HEADS);
        VB_HIP(hipGetLastError());
        }
        vbg::Args p{};
        p.X = ao; p.W = b.wproj.p; p.bias = b.bproj.p; p.resid = resid; p.M = M; p.N = C; p.K = C;
        if (fold) { p.xb = xn; p.stats = stats; p.ldstats = ldstats; p.cm = cmean; }
        if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_PLAIN, vbg::EPI_RESID>(p, 1, st, E, cus))) return rc;
        if (fold ? (rc = run_finalize(m, r0, M, st, E)) : (rc = run_layernorm(resid, b.ln2g.p, b.ln2b.p, B, st, xn, nullptr, nullptr, E))) return rc;
        vbg::Args f1{};
        f1.rstd = rstd;
        f1.X = xn; f1.W = b.w1.p; f1.bias = b.b1.p; f1.out = hid; f1.M = M; f1.N = HID; f1.K = C; f1.ldo = HID; f1.rb = 4;
        if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_PLAIN, vbg::EPI_GELU>(f1, 1, st, E, cus))) return rc;
        vbg::Args f2{};
        f2.X = hid; f2.W = b.w2.p; f2.bias = b.b2.p; f2.resid = resid; f2.M = M; f2.N = C; f2.K = HID;
        const bool feeds_ln = fold && i + 1 < nblocks;            // the final norm reads the f32 stream itself
        if (feeds_ln) { f2.xb = xn; f2.stats = stats; f2.ldstats = ldstats; f2.cm = cmean; }
        if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_PLAIN, vbg::EPI_RESID>(f2, 1, st, E, cus))) return rc;
        if (feeds_ln && (rc = run_finalize(m, r0, M, st, E))) return rc;
    }
    if (resid_out) VB_HIP(hipMemcpyAsync(resid_out, resid, (size_t)M * C * 4, hipMemcpyDeviceToDevice, st));
    return run_layernorm(resid, m->ng.p, m->nb.p, B, st, nullptr, map0, feat_out, E);
}

int head(VbModel* m, const float* feat_in, int B, hipStream_t st, float* score, float* size, float* offset, std::string* err,
         const Slice* sl) {
    const Err E{err};
    int rc = check(m, B, E);
    if (rc || (rc = check_slice(m, B, sl, E))) return rc;
    const size_t f0 = sl ? sl->f0 : 0;
    const long long Bt = sl ? sl->Btot : B;              // tower-major buffers: [tower][Bt frames]...
    const int cus = sl ? sl->cus : 0;
    bf16* const map0 = m->map0.p + f0 * (size_t)(F + 2) * (F + 2) * C;
    if (feat_in) {
        const size_t items = (size_t)B * LX * (C / 4);
        hipLaunchKernelGGL(vbm::feat_to_map_kernel, dim3((unsigned)std::min<size_t>((items + 255) / 256, 16384)), dim3(256), 0, st, feat_in,
                           map0, B, F, C);
        VB_HIP(hipGetLastError());
    }
    const int M = B * LX;
    const long long P2 = (long long)(F + 2) * (F + 2);
    {   // conv1 of the three towers as one GEMM: N = 3 x 256, K = 9 x 768
        vbg::Args a{};
        a.X = map0; a.W = m->wc[0].p; a.bias = m->bc[0].p; a.out = m->map1.p + f0 * (size_t)P2 * HEAD_CH[1];
        a.M = M; a.N = 3 * HEAD_CH[1]; a.K = 9 * HEAD_CH[0]; a.ldo = HEAD_CH[1]; a.C = HEAD_CH[0]; a.F = F; a.out_padded = 1;
        a.n_split = HEAD_CH[1]; a.gOut = Bt * P2 * HEAD_CH[1];
        if ((rc = launch_gemm<256, 256, 2, 4, vbg::A_CONV, vbg::EPI_CONV>(a, 1, st, E, cus))) return rc;
    }
    bf16* maps[4] = {m->map1.p + f0 * (size_t)P2 * HEAD_CH[1], m->map2.p + f0 * (size_t)P2 * HEAD_CH[2],
                     m->map3.p + f0 * (size_t)P2 * HEAD_CH[3], m->t4.p + f0 * (size_t)LX * HEAD_CH[4]};
    for (int li = 1; li < 4; ++li) {   // conv2..4: one launch per layer, blockIdx.y = tower
        const int cin = HEAD_CH[li], cout = HEAD_CH[li + 1], rows = li == 1 ? 256 : (cout < 64 ? 64 : cout);
        vbg::Args a{};
        a.X = maps[li - 1]; a.W = m->wc[li].p; a.bias = m->bc[li].p; a.out = maps[li];
        a.M = M; a.N = cout; a.K = 9 * cin; a.ldo = cout; a.C = cin; a.F = F; a.out_padded = li < 3;
        a.gX = Bt * P2 * cin; a.gW = (long long)rows * 9 * cin; a.gBias = cout;
        a.gOut = li < 3 ? Bt * P2 * cout : Bt * LX * cout;
        if (li == 1 ? (rc = launch_gemm<256, 256, 2, 4, vbg::A_CONV, vbg::EPI_CONV>(a, 3, st, E, cus))
                    : (rc = launch_gemm<256, 64, 8, 1, vbg::A_CONV, vbg::EPI_CONV>(a, 3, st, E, cus))) return rc;
    }
    hipLaunchKernelGGL((vbm::conv5_kernel<32>), dim3((M + 255) / 256), dim3(256), 0, st, maps[3], m->w5.p, m->b5.p, M, LX,
                       (size_t)(Bt * LX * HEAD_CH[4]), score, size, offset);
    VB_HIP(hipGetLastError());
    return VT_OK;
}

}  // namespace vb
