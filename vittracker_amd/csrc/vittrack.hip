// vittrack.hip -- C-ABI runtime of libvittrack_hip.so (see include/vittrack.h).
// Host side: config checks, BatchNorm folding, MFMA operand-image packing, workspace, launches,
// hipGraph capture.  Device side: vt_stem.h / vt_blocks.h / vt_head.h.
#include "../../include/vittrack.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "vb_api.h"
#include "vt_blocks.h"
#include "vt_blocks_tile.h"
#include "vt_head.h"
#ifndef VT_SEQ3_MAXP
#define VT_SEQ3_MAXP 2      // head_seq3: conv1 weight pairs per register pass
#endif
#include "vt_head3.h"
#include "vt_generic.h"
#include "vt_stem.h"
#include "vt_stem_fused.h"
#include "vt_stem_stream.h"
#include "vt_track.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(VT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    int alloc(size_t floats) {
        n = floats;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), floats * sizeof(float));
        if (e != hipSuccess) return fail(VT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        return VT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
};

constexpr int STEM_CH[5] = {3, 6, 12, 24, 48};

}  // namespace

struct vt_model {
    VbModel* vb = nullptr;           // ViT-Base path (channels = 768): backbone + head towers live in vitb.hip
    // any other stride-16 geometry of the vit_48_h32 surface: the shape-generic kernels of vt_generic.h
    bool generic = false;
    vtg::Dims gd{48, 1, 32};         // widths of the shape-generic path (round 6: any CHANNELS / HEADS / HEAD.NUM_CHANNELS)
    DevBuf g_stem_w[4], g_stem_b[4]; // folded conv weights [cout][cin][9] / bias
    DevBuf g_blocks;                 // depth * gd.block_stride() + 2 C (final norm)
    DevBuf g_head;                   // 3 * gd.tower_stride()
    DevBuf g_a, g_b;                 // stem ping-pong maps; then the head's
    DevBuf g_qkv, g_ao, g_hid, g_x;  // (B L, 3 C), (B L, C), (B L, 4 C); the residual stream being updated
    vt_config cfg{};
    int len_z = 0, len_x = 0, L = 0, F = 0, Fz = 0;
    bool weights_loaded = false;
    // parameters on the device
    DevBuf stem_w[4], stem_b[4];     // folded, [group][tap][cin][OCG] / [cout]
    DevBuf stem_w3b;                 // layer 3 as three-piece bf16 images (stem_fused, fp32 build)
    DevBuf stem_w4b;                 // layer 4, the same way: [out tile 3][chunk pair 7][piece 3][64 lanes][8 bf16]
    DevBuf stem_w2k;                 // layer 2 again as [tap][input-channel quad][16 output channels][4] for the 4-block f32 MFMA
    // layer 1 with Preprocessor.process folded in, for uint8 patches (vt_stem.h: L1In): [162 weights][6 biases][3 pad values]
    DevBuf stem_w1u;
    std::vector<double> stem_w1_f64, stem_b1_f64;   // layer 1 with BN folded, kept for vt_set_normalization
    float norm_mean[3] = {0.485f, 0.456f, 0.406f}, norm_std[3] = {0.229f, 0.224f, 0.225f};   // lib/test/tracker/data_utils.py:8-9
    int track_u8 = 1;                // VT_TRACK_U8: vt_track_step hands the crop to the stem as a uint8 patch (0: the fp32 crop of vt_crop)
    DevBuf pos_z, pos_x;             // (len, C)
    DevBuf blocks;                   // depth * BLOCK_STRIDE + 2C (final norm)
    DevBuf blocks3;                  // depth * BLOCK3_STRIDE: the MLP's three-piece bf16 images (vt_blocks.h, BF3)
    DevBuf head;                     // 3 * TOWER_STRIDE
    DevBuf head3;                    // F = 8, fp32 build: the towers' weights as three-piece bf16 images (vt_head3.h)
    int head_bf3 = 1;                // VT_HEAD_BF3: the towers on the bf16 matrix pipe at fp32 accuracy (F = 8); 0 = fp32 MFMA towers
    DevBuf window;                   // F*F
    // workspace sized for max_batch
    DevBuf act_x, act_z;             // layer-2 activations, NHWC(12)
    DevBuf tokens, feat;
    DevBuf tokens_c;                 // token matrix of the cached-template step: its template rows are written by vt_set_template only
    // small batches (vt_blocks_tile.h): q / K image / V^T image of every tile and the residual stream between the per-block launches
    DevBuf tile_q, tile_k, tile_v, tile_x;
    DevBuf head_m1;                  // F = 16, small batches: conv1 output of the three towers, [frames][3][8][NPIX] float4, zero borders
    int head_m1_frames = 0;
    int head_split = -1;             // F = 16: conv1 as a launch of its own over row strips (1 / 0 force, -1: by batch size)
    int tile_frames = 0;             // frames those workspaces are sized for
    int blocks_tile = -1;            // 1 / 0 force the tile-parallel form of the blocks / forbid it, -1 (default): by batch size
    int open_loop = 0;               // vt_set_open_loop: vt_track_step leaves states_dev untouched (the step's box is in `record`)
    DevBuf zcache;                   // block-0 q / k / v^T images of the template tiles (vt_set_template)
    DevBuf vlscr;                    // G256 frame-form block kernel (A3): the low pieces of V^T, [B][depth][3][L / 32][64] x 16 B (vt_blocks.h VP2L)
    int tmpl_frames = 0;             // frames whose template rows (tokens + zcache) are cached
    int tmpl_form_batch = 0;         // the form batch vt_set_template ran under (the cache holds THAT form's operands)
    int graphs_captured = 0;         // live graphs captured from this model (vt_graph_destroy takes them off again): they hold the forms of their capture
    std::vector<vt_graph*> graphs;   // those graphs; vt_destroy orphans them, so a graph destroyed after its model touches nothing of it
    DevBuf score, size, offset, pred, hann, conf;
    hipStream_t cap_stream = nullptr;
    hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};   // extra capture streams for graph chains
    hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
    unsigned long long* dbg_stamps = nullptr;   // VT_DBG_STAMPS=1: per-wave phase stamps of the block kernel
    // diagnostic switches, read from the environment ONCE at vt_create (all 0 / -1 in production)
    int skip_stem_a = 0, skip_stem_b = 0, skip_head = 0, dbg_skip_tile = -1, graph_chains = 1, chain_cus = 0, chain_delay_us = 0;
    // Kernel form per stage: 1 / 0 force it, -1 (default) = by batch size.  The one-workgroup-per-frame forms win once the batch
    // fills the chip; below that the multi-workgroup forms spread a frame over several CUs (measured, us per step, tools/
    // small_batch_sweep.py: G128 B=1 97.6 -> 81.1, B=64 100.4 -> 86.1; G256 B=1 337 -> 287; crossovers at the thresholds below).
    int head_fused = -1;   // F = 8: head_fused_kernel (towers + decode in one workgroup per frame); auto: B > 176
                           // F = 16: head_seq_kernel (one workgroup per frame runs the three towers in turn, then decodes); auto: B > 176
    int stem_pipe = -1;    // G256: stem_pipe_kernel (layers 1 + 2 per frame) instead of stem_a; auto: B > 176
    int stem_fused = -1;   // G128: stem_fused_kernel (one workgroup per frame) instead of stem_a + stem_b; auto: B > 80
    int stem_stream = -1;  // stem_stream_kernel (all four layers of a frame streamed band by band through one workgroup) instead of
                           // stem_pipe + stem_b (G256) / stem_fused (G128); auto: G256 B > 176 (fp32 build)
    int stem_fuse = 1;     // stem_a: one workgroup = band k of both crops (G128: 4 instead of 5 workgroups per frame)
    int stem_bf3 = 1;      // VT_STEM_BF3: layer 3 of stem_fused as exact three-piece bf16 products (fp32 build); 0 = fp32 MFMAs
    int blocks_bf3_g256 = 1;   // the same switch at G256 (MLP only, weights from L2)
    int blocks_bf3 = 2;    // VT_BLOCKS_BF3: the G128 frame form's contractions as exact three-piece bf16 products: 2 = all of them
                           // (A3: attention + proj too), 1 = qkv + MLP, 0 = fp32 MFMAs
    int blocks_bal = 1;    // G128 block kernel: balanced 4 owner + 4 guest waves (1) or one wave per tile (0)
    int blocks_wlds = 1;   // G128 block kernel: weights staged through LDS (1) or read from L2 per wave (0)
    int form_batch = 0;    // vt_set_form_batch: kernel forms are chosen as for a batch of this size (0: by the batch of each call)
    int plan_r2[2] = {0, 0}, plan_r4[2] = {0, 0};   // band plan for (search, template) crops
    bool r4_128_forced = false;   // VT_STEM_R4_128 was set: keep that band height at every batch size
};

struct vt_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    vt_model* owner = nullptr;      // its capture counts in owner->graphs_captured until vt_graph_destroy (the model must outlive its graphs)
};

namespace {

// ------------------------------------------------------------------------------- weight packing
using TensorMap = std::map<std::string, std::pair<const float*, int64_t>>;

int need(const TensorMap& tm, const std::string& name, int64_t numel, const float** out) {
    auto it = tm.find(name);
    if (it == tm.end()) return fail(VT_ERR_MISSING_KEY, "missing key in state dict: " + name);
    if (it->second.second != numel)
        return fail(VT_ERR_MISSING_KEY, "shape mismatch for " + name + ": got " + std::to_string(it->second.second) +
                                            " elements, want " + std::to_string(numel));
    *out = it->second.first;
    return VT_OK;
}

// BN(eval) folded into the preceding conv, in double:  w' = w * g / sqrt(var + eps),
// b' = (b_conv - mean) * g / sqrt(var + eps) + beta      (Conv2d_BN.fuse, vit_dist.py:22-33)
int fold_conv_bn(const TensorMap& tm, const std::string& conv, const std::string& bn, bool conv_bias, int cout,
                 int cin, std::vector<double>& w, std::vector<double>& b) {
    const float *pw, *pb = nullptr, *g, *beta, *mu, *var;
    int rc;
    if ((rc = need(tm, conv + ".weight", (int64_t)cout * cin * 9, &pw))) return rc;
    if (conv_bias && (rc = need(tm, conv + ".bias", cout, &pb))) return rc;
    if ((rc = need(tm, bn + ".weight", cout, &g))) return rc;
    if ((rc = need(tm, bn + ".bias", cout, &beta))) return rc;
    if ((rc = need(tm, bn + ".running_mean", cout, &mu))) return rc;
    if ((rc = need(tm, bn + ".running_var", cout, &var))) return rc;
    w.resize((size_t)cout * cin * 9);
    b.resize(cout);
    for (int o = 0; o < cout; ++o) {
        const double k = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
        for (int i = 0; i < cin * 9; ++i) w[(size_t)o * cin * 9 + i] = (double)pw[(size_t)o * cin * 9 + i] * k;
        b[o] = ((conv_bias ? (double)pb[o] : 0.0) - (double)mu[o]) * k + (double)beta[o];
    }
    return VT_OK;
}

// [cout][cin][3][3] -> [r][cin][s][cout]: the scalar-weight sections of vt_stem.h (stem_a)
std::vector<float> pack_conv_sections(const std::vector<double>& w, int cout, int cin) {
    std::vector<float> out((size_t)cout * cin * 9);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < cin; ++c)
            for (int s = 0; s < 3; ++s)
                for (int j = 0; j < cout; ++j)
                    out[(((size_t)r * cin + c) * 3 + s) * cout + j] = (float)w[((size_t)j * cin + c) * 9 + r * 3 + s];
    return out;
}

// [cout][cin][3][3] -> [group][tap][cin][ocg]
std::vector<float> pack_conv_groups(const std::vector<double>& w, int cout, int cin, int ocg) {
    std::vector<float> out((size_t)cout * cin * 9);
    const int ng = cout / ocg;
    for (int g = 0; g < ng; ++g)
        for (int tap = 0; tap < 9; ++tap)
            for (int c = 0; c < cin; ++c)
                for (int j = 0; j < ocg; ++j)
                    out[(((size_t)g * 9 + tap) * cin + c) * ocg + j] = (float)w[((size_t)(g * ocg + j) * cin + c) * 9 + tap];
    return out;
}

// folded conv [cout][cin][3][3] -> [oc group of 4][k][oc 4], k = tap * cin + channel, zero-padded to a multiple of 16: the A operands
// of the 4 x 4-block MFMA form (vt_head3.h SeqConvQ) -- lane l of register kg holds output channel l & 3 at k = 16 kg + (l >> 2)
void pack_conv_quads(const std::vector<double>& w, int cout, int cin, float* dst) {
    const int kp = vth::kpad16(cin);
    for (int g = 0; g < cout / 4; ++g)
        for (int k = 0; k < kp; ++k)
            for (int oc = 0; oc < 4; ++oc)
                dst[((size_t)g * kp + k) * 4 + oc] = k < 9 * cin ? (float)w[((size_t)(4 * g + oc) * cin + k % cin) * 9 + k / cin] : 0.f;
}

// folded conv [cout][cin][3][3] -> MFMA A-operand images [oc_tile][chunk][64 lanes][4] for the
// implicit GEMM of vt_head.h: element r of lane l of (ot, c) = w[oc = 16 ot + (l & 15)][ic = 4 icq + r][tap]
// with quad Q = 4 c + (l >> 4), (tap, icq) = divmod(Q, cin / 4); zero beyond cout or 9 * cin / 4 quads.
void pack_conv_image(const std::vector<double>& w, int cout, int cin, float* dst) {
    const int nq = (cin + 3) / 4, nqt = 9 * nq, nch = (nqt + 3) / 4, not_ = (cout + 15) / 16;   // cin padded to quads
    for (int ot = 0; ot < not_; ++ot)
        for (int c = 0; c < nch; ++c)
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int oc = 16 * ot + (l & 15), Q = 4 * c + (l >> 4);
                    float v = 0.f;
                    if (oc < cout && Q < nqt) {
                        const int tap = Q / nq, ic = 4 * (Q % nq) + r;
                        if (ic < cin) v = (float)w[((size_t)oc * cin + ic) * 9 + tap];
                    }
                    dst[(((size_t)ot * nch + c) * 64 + l) * 4 + r] = v;
                }
}

// The same weights as three-piece bf16 images for vt_head3.h: [oc_tile][chunk pair][piece][64 lanes][8 bf16]; a lane's 8 values
// are its quad of chunk 2 p, then its quad of chunk 2 p + 1 (zero beyond the last chunk).  w = h + m + l exactly, by truncation
// (the split the kernels apply to activations: vth3::split3).
// x = h + m + l by truncation (vt3::split3): the bf16 bit patterns of the three pieces
void split3_host(float v, uint16_t (&pieces)[3]) {
    uint32_t xb, r1b, r2b;
    std::memcpy(&xb, &v, 4);
    float hf; const uint32_t hb = xb & 0xffff0000u; std::memcpy(&hf, &hb, 4);
    const float r1 = v - hf; std::memcpy(&r1b, &r1, 4);
    float mf; const uint32_t mb = r1b & 0xffff0000u; std::memcpy(&mf, &mb, 4);
    const float r2 = r1 - mf; std::memcpy(&r2b, &r2, 4);
    pieces[0] = (uint16_t)(xb >> 16); pieces[1] = (uint16_t)(r1b >> 16); pieces[2] = (uint16_t)(r2b >> 16);
}

// The MLP's weights as three-piece images for the block kernel's BF3 form (layout: vt_blocks.h), from the fp32 operand images
// [out tile][chunk][64 lanes][4] of the same (folded) weights.
// a K = 48 layer (fc1, qkv) from its fp32 operand image [ot][chunk 3][64 lanes][4]: [ot][ pair 0: piece x lane x 8 | chunk 2: piece x lane x 4 ]
void pack_k48_image3(const float* img1, int ntiles, uint16_t* dst) {
    constexpr int NC = vtb::NC;
    uint16_t pcs[3];
    for (int ot = 0; ot < ntiles; ++ot) {
        uint16_t* o = dst + (size_t)ot * vtb::W3_FC1_OT16 * 8;
        for (int l = 0; l < 64; ++l) {
            for (int e = 0; e < 8; ++e) {
                split3_host(img1[(((size_t)ot * NC + (e >> 2)) * 64 + l) * 4 + (e & 3)], pcs);
                for (int pc = 0; pc < 3; ++pc) o[((size_t)pc * 64 + l) * 8 + e] = pcs[pc];
            }
            for (int e = 0; e < 4; ++e) {
                split3_host(img1[(((size_t)ot * NC + 2) * 64 + l) * 4 + e], pcs);
                for (int pc = 0; pc < 3; ++pc) o[(size_t)192 * 8 + ((size_t)pc * 64 + l) * 4 + e] = pcs[pc];
            }
        }
    }
}

void pack_mlp_images3(const float* img1, const float* img2, const float* imgqkv, const float* imgproj, uint16_t* dst) {
    constexpr int NC = vtb::NC, NH = vtb::NH;
    uint16_t pcs[3];
    pack_k48_image3(imgqkv, 9, dst + (size_t)(vtb::W3_FC1_TILES + vtb::W3_FC2_TILES) * 512);
    pack_k48_image3(imgproj, NC, dst + (size_t)(vtb::W3_FC1_TILES + vtb::W3_FC2_TILES + vtb::W3_QKV_TILES) * 512);      // A3
    for (int ot = 0; ot < NH; ++ot) {           // fc1: [ot][ pair 0: piece x lane x 8 | chunk 2: piece x lane x 4 ]
        uint16_t* o = dst + (size_t)ot * vtb::W3_FC1_OT16 * 8;
        for (int l = 0; l < 64; ++l) {
            for (int e = 0; e < 8; ++e) {
                split3_host(img1[(((size_t)ot * NC + (e >> 2)) * 64 + l) * 4 + (e & 3)], pcs);
                for (int pc = 0; pc < 3; ++pc) o[((size_t)pc * 64 + l) * 8 + e] = pcs[pc];
            }
            for (int e = 0; e < 4; ++e) {
                split3_host(img1[(((size_t)ot * NC + 2) * 64 + l) * 4 + e], pcs);
                for (int pc = 0; pc < 3; ++pc) o[(size_t)192 * 8 + ((size_t)pc * 64 + l) * 4 + e] = pcs[pc];
            }
        }
    }
    uint16_t* d2 = dst + (size_t)vtb::W3_FC1_TILES * 512;
    for (int ot = 0; ot < NC; ++ot)             // fc2: [ot][pair][piece][lane][8]
        for (int p = 0; p < NH / 2; ++p)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    split3_host(img2[(((size_t)ot * NH + 2 * p + (e >> 2)) * 64 + l) * 4 + (e & 3)], pcs);
                    for (int pc = 0; pc < 3; ++pc) d2[((((size_t)ot * (NH / 2) + p) * 3 + pc) * 64 + l) * 8 + e] = pcs[pc];
                }
}

void pack_conv_image3(const std::vector<double>& w, int cout, int cin, uint16_t* dst) {
    const int nq = (cin + 3) / 4, nqt = 9 * nq, nch = (nqt + 3) / 4, ncp = (nch + 1) / 2, not_ = (cout + 15) / 16;
    for (int ot = 0; ot < not_; ++ot)
        for (int cp = 0; cp < ncp; ++cp)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int c = 2 * cp + (e >> 2), r = e & 3;
                    const int oc = 16 * ot + (l & 15), Q = 4 * c + (l >> 4);
                    float v = 0.f;
                    if (oc < cout && c < nch && Q < nqt) {
                        const int tap = Q / nq, ic = 4 * (Q % nq) + r;
                        if (ic < cin) v = (float)w[((size_t)oc * cin + ic) * 9 + tap];
                    }
                    uint16_t pieces[3];
                    split3_host(v, pieces);
                    for (int pc = 0; pc < 3; ++pc) dst[((((size_t)ot * ncp + cp) * 3 + pc) * 64 + l) * 8 + e] = pieces[pc];
                }
}

// nn.Linear weight (OUT, IN) -> MFMA operand images [OUT/16][IN/16][64 lanes][4]:
// element r of lane l of tile (ot, c) = W[16 ot + (l & 15)][16 c + 4 (l >> 4) + r]   (vt_common.h)
void pack_linear_image(const float* W, int OUT, int IN, float* dst) {
    const int nc = IN / 16;
    for (int ot = 0; ot < OUT / 16; ++ot)
        for (int c = 0; c < nc; ++c)
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r)
                    dst[(((size_t)ot * nc + c) * 64 + l) * 4 + r] = W[(size_t)(16 * ot + (l & 15)) * IN + 16 * c + 4 * (l >> 4) + r];
}

int upload(DevBuf& d, const std::vector<float>& h) {
    if (!d.p || d.n != h.size()) {
        d.release();
        int rc = d.alloc(h.size());
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpy(d.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return VT_OK;
}

// lib/test/utils/hann.py:6-16, float32 like torch
std::vector<float> hann2d(int F) {
    std::vector<float> w1(F), w((size_t)F * F);
    const float k = (float)(2.0 * M_PI / (F + 1));
    for (int i = 0; i < F; ++i) w1[i] = 0.5f * (1.0f - cosf(k * (float)(i + 1)));
    for (int y = 0; y < F; ++y)
        for (int x = 0; x < F; ++x) w[(size_t)y * F + x] = w1[y] * w1[x];
    return w;
}

// Preprocessor.process folded into layer 1 (vt_stem.h: L1In): x = u / (255 std_c) - mean_c / std_c per channel, so
// w' = w / (255 std_c), b' = b + sum_{c,tap} w (-mean_c / std_c), in fp64 from the BN-folded weights; pad value 255 mean_c.
int fold_w1u(vt_model* m, const float* mean3, const float* std3) {
    const std::vector<double>& w = m->stem_w1_f64;
    const std::vector<double>& b = m->stem_b1_f64;
    if (w.size() != 6 * 3 * 9 || b.size() != 6) return fail(VT_ERR_STATE, "layer-1 weights not loaded");
    std::vector<double> wf(w.size());
    std::vector<float> img(vts::W1U_FLOATS, 0.f);
    for (int j = 0; j < 6; ++j) {
        double bias = b[j];
        for (int c = 0; c < 3; ++c)
            for (int t = 0; t < 9; ++t) {
                const double wv = w[((size_t)j * 3 + c) * 9 + t];
                wf[((size_t)j * 3 + c) * 9 + t] = wv / (255.0 * (double)std3[c]);
                bias += wv * (-(double)mean3[c] / (double)std3[c]);
            }
        img[vts::W1U_BIAS + j] = (float)bias;
    }
    const std::vector<float> sec = pack_conv_sections(wf, 6, 3);
    std::copy(sec.begin(), sec.end(), img.begin());
    for (int c = 0; c < 3; ++c) {
        img[vts::W1U_PAD + c] = (float)(255.0 * (double)mean3[c]);
        m->norm_mean[c] = mean3[c];
        m->norm_std[c] = std3[c];
    }
    return upload(m->stem_w1u, img);
}

// ------------------------------------------------------------------------------------- launches
int check_ready(vt_model* m, int B) {
    if (!m) return fail(VT_ERR_ARG, "null model");
    if (!m->weights_loaded) return fail(VT_ERR_STATE, "vt_load_weights has not been called");
    if (B < 1 || B > m->cfg.max_batch)
        return fail(VT_ERR_STATE, "batch " + std::to_string(B) + " outside [1, max_batch=" + std::to_string(m->cfg.max_batch) + "]");
    return VT_OK;
}

// The batch size the kernel FORMS of a call are chosen by: the call's own, or the model's form batch (vt_set_form_batch) -- a shard
// of a larger group of sequences then runs the forms the whole group would, so its results do not depend on how the group is sharded.
int form_b(const vt_model* m, int B) { return std::max(B, m->form_batch); }

// Band sizes per crop side.  stem_a: r2 layer-2 rows per workgroup (256 output pixels);
// stem_b: r4 token rows per workgroup (LDS <= ~50 KB so three workgroups share a CU).
struct StemPlan { int r2, r4; };
int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}
StemPlan stem_plan_default(int T);
StemPlan stem_plan(int T) {   // VT_STEM_R2_<T> / VT_STEM_R4_<T> override the defaults (tuning aid)
    StemPlan p = stem_plan_default(T);
    const std::string t = std::to_string(T);
    p.r2 = env_int(("VT_STEM_R2_" + t).c_str(), p.r2);
    p.r4 = env_int(("VT_STEM_R4_" + t).c_str(), p.r4);
    return p;
}
StemPlan stem_plan_default(int T) {
    switch (T) {
        case 64: return {16, 4};    // layer-2 map 16x16, tokens 4x4: one band each
        case 128: return {8, 4};    // 32x32 -> 4 bands; tokens 8x8 -> 2 bands
        case 256: return {4, 2};    // 64x64 -> 16 bands; tokens 16x16 -> 8 bands
        default: return {0, 0};
    }
}


// ---------------------------------------------------------------------------------------- shape-generic path (vt_generic.h)
inline unsigned gen_grid(size_t n) { return (unsigned)((n + 255) / 256); }

int gen_stem(vt_model* m, const float* z, const float* x, int B, hipStream_t st, float* tokens, int zmode, bool xu8 = false) {
    for (int side = 0; side < 2; ++side) {       // 0: template rows, 1: search rows
        const float* img = side == 0 ? z : x;
        if ((side == 0 && zmode == 1) || (side == 1 && zmode == 2) || !img) continue;
        const int T = side == 0 ? m->cfg.template_size : m->cfg.search_size;
        const float* in = img;
        int S = T;
        const int C = m->gd.C, chans[5] = {3, C / 8, C / 4, C / 2, C};      // b16 (vit_dist.py:36-54)
        for (int i = 0; i < 4; ++i) {
            const int cin = chans[i], cout = chans[i + 1], So = S / 2;
            float* out = (i & 1) ? m->g_b.p : m->g_a.p;
            const size_t total = (size_t)B * cout * So * So;
            if (i == 0 && side == 1 && xu8) {      // the search crop arrives as sample_target's uint8 patch: Preprocessor.process per tap, same weights
                hipLaunchKernelGGL(vtg::stem_conv_u8_kernel, dim3(gen_grid(total)), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(img), m->g_stem_w[0].p,
                                   m->g_stem_b[0].p, B, cout, S, m->norm_mean[0], m->norm_mean[1], m->norm_mean[2], m->norm_std[0], m->norm_std[1],
                                   m->norm_std[2], out);
                in = out;
                S = So;
                continue;
            }
            hipLaunchKernelGGL(vtg::stem_conv_kernel, dim3(gen_grid(total)), dim3(256), 0, st, in, m->g_stem_w[i].p, m->g_stem_b[i].p, B, cin, cout, S,
                               i < 3 ? out : nullptr, i < 3 ? nullptr : tokens, side == 0 ? m->pos_z.p : m->pos_x.p, m->L, side == 0 ? 0 : m->len_z);
            in = out;
            S = So;
        }
    }
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int gen_blocks(vt_model* m, const float* tokens, int B, int nblocks, hipStream_t st, float* feat, float* resid) {
    if (nblocks < 0 || nblocks > m->cfg.depth) nblocks = m->cfg.depth;
    const size_t rows = (size_t)B * m->L;
    // the residual stream is updated in place in a buffer of its own: the caller's tokens -- and the cached template rows of
    // vt_set_template's token matrix -- stay untouched
    float* x = m->g_x.p;
    const vtg::Dims d = m->gd;
    const int C = d.C, HID = d.hid();
    HIP_TRY(hipMemcpyAsync(x, tokens, rows * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < nblocks; ++i) {
        const float* P = m->g_blocks.p + (size_t)i * d.block_stride();
        hipLaunchKernelGGL(vtg::ln_linear_kernel<0>, dim3(gen_grid(rows * 3 * C)), dim3(256), 0, st, x, P + d.o_wqkv(), P + d.o_bqkv(), rows, C, 3 * C, m->g_qkv.p);
        const unsigned ag = gen_grid(rows * d.heads);
        switch (d.hd()) {       // head_dim as a template value where it is a common one (registers), else the run-time form
            case 16: hipLaunchKernelGGL(vtg::attn_kernel<16>, dim3(ag), dim3(256), 0, st, m->g_qkv.p, B, m->L, C, d.heads, m->g_ao.p); break;
            case 32: hipLaunchKernelGGL(vtg::attn_kernel<32>, dim3(ag), dim3(256), 0, st, m->g_qkv.p, B, m->L, C, d.heads, m->g_ao.p); break;
            case 48: hipLaunchKernelGGL(vtg::attn_kernel<48>, dim3(ag), dim3(256), 0, st, m->g_qkv.p, B, m->L, C, d.heads, m->g_ao.p); break;
            case 64: hipLaunchKernelGGL(vtg::attn_kernel<64>, dim3(ag), dim3(256), 0, st, m->g_qkv.p, B, m->L, C, d.heads, m->g_ao.p); break;
            default: hipLaunchKernelGGL(vtg::attn_kernel<0>, dim3(ag), dim3(256), 0, st, m->g_qkv.p, B, m->L, C, d.heads, m->g_ao.p); break;
        }
        hipLaunchKernelGGL(vtg::linear_resid_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, m->g_ao.p, P + d.o_wproj(), P + d.o_bproj(), rows, C, C, x);
        hipLaunchKernelGGL(vtg::ln_linear_kernel<1>, dim3(gen_grid(rows * HID)), dim3(256), 0, st, x, P + d.o_w1(), P + d.o_b1(), rows, C, HID, m->g_hid.p);
        hipLaunchKernelGGL(vtg::linear_resid_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, m->g_hid.p, P + d.o_w2(), P + d.o_b2(), rows, C, HID, x);
    }
    if (resid) HIP_TRY(hipMemcpyAsync(resid, x, rows * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    const float* N = m->g_blocks.p + (size_t)m->cfg.depth * d.block_stride();
    hipLaunchKernelGGL(vtg::final_norm_kernel, dim3(gen_grid((size_t)B * m->len_x * C)), dim3(256), 0, st, x, N, N + C, B, m->L, m->len_z, C, feat);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int gen_head(vt_model* m, const float* feat, int B, hipStream_t st, float* score, float* size, float* offset) {
    const int F = m->F;
    const vtg::Dims d = m->gd;
    const size_t npx = (size_t)B * F * F;
    const float* in = feat;
    size_t in_stride = 0;
    for (int i = 0; i < 4; ++i) {
        float* out = (i & 1) ? m->g_b.p : m->g_a.p;
        hipLaunchKernelGGL(vtg::head_conv_kernel, dim3(gen_grid(3 * npx * d.hch(i + 1))), dim3(256), 0, st, in, in_stride, m->g_head.p, d.tower_stride(),
                           d.ho_w(i), d.ho_b(i), B, F, d.hch(i), d.hch(i + 1), out);
        in = out;
        in_stride = npx * d.hch(i + 1);
    }
    hipLaunchKernelGGL(vtg::head_out_kernel, dim3(gen_grid(npx)), dim3(256), 0, st, in, m->g_head.p, d.tower_stride(), d.ho_w5(), d.ho_b5(), d.hch(4), B, F,
                       score, size, offset);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

// Does the stem form a batch of B (under the model's form batch) selects read uint8 patches?  Every form of the tuned geometries does
// (stem_fused, stem_stream, stem_pipe, stem_a); the diagnostic builds and the shape-generic kernels do not.
bool stem_takes_u8(const vt_model* m, int B) {
    (void)B;
    if (m->vb) return false;
    if (m->generic) return true;       // vt_generic.h: stem_conv_u8_kernel (the reference's own normalisation per tap)
    return m->stem_w1u.p != nullptr && m->skip_stem_a == 0 && m->skip_stem_b == 0 && m->dbg_stamps == nullptr;
}

int run_stem(vt_model* m, const float* z, const float* x, int B, hipStream_t st, float* tokens, size_t f0 = 0, int zmode = 0, bool xu8 = false) {
    // f0: first frame of this slice in the model workspace (z, x, tokens already point at the slice)
    // zmode 0: both crops; 1: search crop only (template token rows already in `tokens`); 2: template crop only
    // xu8: x is a uint8 (B, Tx, Tx, 3) patch (vt_crop_u8) and layer 1 runs on the folded weights w1u; zmode 1 only
    if (xu8 && (zmode != 1 || !stem_takes_u8(m, B))) return fail(VT_ERR_STATE, "this stem form has no uint8-patch variant");
    if (m->generic) return gen_stem(m, z, x, B, st, tokens, zmode, xu8);
    const float* const w1 = xu8 ? m->stem_w1u.p : m->stem_w[0].p;
    const float* const b1 = xu8 ? m->stem_w1u.p + vts::W1U_BIAS : m->stem_b[0].p;
    const int Tx = m->cfg.search_size, Tz = m->cfg.template_size;
    float* const act_x = m->act_x.p + f0 * (size_t)(Tx / 4) * (Tx / 4) * 12;
    float* const act_z = m->act_z.p + f0 * (size_t)(Tz / 4) * (Tz / 4) * 12;
    StemPlan px{m->plan_r2[0], m->plan_r4[0]};
    const StemPlan pz{m->plan_r2[1], m->plan_r4[1]};
    // small batches of the 128-px search crop: stem_b in bands of 2 token rows (4 workgroups per crop instead of 2) shortens the
    // latency chain of a band (B=1 step 63.5 -> 59.9 us); at large batches the halo rows it recomputes cost more than that
    const int Bf = form_b(m, B);
    if (Tx == 128 && Bf <= 80 && px.r4 == 4 && !m->r4_128_forced) px.r4 = 2;
    for (const auto& pr : {std::make_pair(Tx, px), std::make_pair(Tz, pz)}) {
        const int T = pr.first, r2 = pr.second.r2, r4 = pr.second.r4;
        const int nt4 = r4 > 0 ? (r4 * (T / 16) + 15) / 16 : 0;
        if (r2 < 1 || r4 < 1 || (T / 4) % r2 || (T / 16) % r4 || (r2 * (T / 4)) % 256 || !(nt4 == 1 || nt4 == 2 || nt4 == 4) ||
            (r4 * (T / 16)) % 16 || (((2 * r4 + 1) * (T / 8)) % 16 && ((2 * r4) * (T / 8)) % 16) ||
            3 * vts::stem_b_npix2(T / 4, r4) < 4 * nt4 * 3 * 64 ||
            (T / 4) > 256 || (4 * r4 + 3) > 5 * (256 / (T / 4)))     // stem_b stages <= 5 layer-2 rows per thread and plane
            return fail(VT_ERR_ARG, "unsupported stem band plan for crop side " + std::to_string(T));
    }
#ifndef VT_F16
    {   // the streaming form: one workgroup per frame, nothing but token rows leaves the CU
        const bool g256 = Tx == 256 && Tz == 128, g128 = Tx == 128 && Tz == 64;
        const bool want_stream = m->stem_stream < 0 ? (g256 && Bf > 176) : m->stem_stream != 0;
        const bool diag = m->skip_stem_a != 0 || m->skip_stem_b != 0 || m->dbg_stamps != nullptr;
        if (want_stream && !diag && (g256 || g128)) {
            auto go = [&](auto kernel, size_t lds) {
                hipLaunchKernelGGL(kernel, dim3(B), dim3(1024), lds, st, z, x, w1, b1, m->stem_b[1].p,
                                   m->stem_w[2].p, m->stem_b[2].p, m->stem_w[3].p, m->stem_b[3].p, m->pos_z.p, m->pos_x.p, tokens, m->L,
                                   m->len_z, m->stem_w2k.p);
            };
            if (g256) {
                constexpr size_t lds = vts::StreamGeo<256, 128>::LDS_BYTES;
                if (xu8) go(&vts::stem_stream_kernel<256, 128, 1, true>, lds);
                else if (zmode == 0) go(&vts::stem_stream_kernel<256, 128, 0>, lds);
                else if (zmode == 1) go(&vts::stem_stream_kernel<256, 128, 1>, lds);
                else go(&vts::stem_stream_kernel<256, 128, 2>, lds);
            } else {
                constexpr size_t lds = vts::StreamGeo<128, 64>::LDS_BYTES;
                if (xu8) go(&vts::stem_stream_kernel<128, 64, 1, true>, lds);
                else if (zmode == 0) go(&vts::stem_stream_kernel<128, 64, 0>, lds);
                else if (zmode == 1) go(&vts::stem_stream_kernel<128, 64, 1>, lds);
                else go(&vts::stem_stream_kernel<128, 64, 2>, lds);
            }
            HIP_TRY(hipGetLastError());
            return VT_OK;
        }
    }
#endif
    const bool want_fused = m->stem_fused < 0 ? Bf > 80 : m->stem_fused != 0;
    const bool want_pipe = m->stem_pipe < 0 ? Bf > 176 : m->stem_pipe != 0;
    if (want_fused && Tx == vts::FusedGeo::TX && Tz == vts::FusedGeo::TZ) {
        // whole patch embedding of a frame in one workgroup; only token rows leave the CU
        const bool diag = m->skip_stem_a != 0 || m->dbg_stamps != nullptr;
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(B), dim3(1024), vts::FusedGeo::LDS_BYTES_P, st, z, x, w1, b1,
                               m->stem_w[1].p, m->stem_b[1].p, m->stem_w[2].p, m->stem_b[2].p, m->stem_w[3].p, m->stem_b[3].p,
                               m->pos_z.p, m->pos_x.p, tokens, m->L, m->len_z, m->skip_stem_a, m->dbg_stamps, m->stem_w2k.p, m->stem_w3b.p, m->stem_w4b.p);
        };
        if (diag && zmode != 0) return fail(VT_ERR_STATE, "the diagnostic stem build has no template-cache form");
        if (diag) go(&vts::stem_fused_kernel<0, true>);
        else if (xu8) {
            if (m->stem_bf3) go(&vts::stem_fused_kernel<1, false, true, true>);
            else go(&vts::stem_fused_kernel<1, false, false, true>);
        } else if (!m->stem_bf3) {      // VT_STEM_BF3=0: layer 3 on fp32 MFMAs (the all-fp32-MFMA step bench.py reports beside the default)
            if (zmode == 0) go(&vts::stem_fused_kernel<0, false, false>);
            else if (zmode == 1) go(&vts::stem_fused_kernel<1, false, false>);
            else go(&vts::stem_fused_kernel<2, false, false>);
        } else if (zmode == 0) go(&vts::stem_fused_kernel<0, false>);
        else if (zmode == 1) go(&vts::stem_fused_kernel<1, false>);
        else go(&vts::stem_fused_kernel<2, false>);
        HIP_TRY(hipGetLastError());
        return VT_OK;
    }
    const bool pipe = want_pipe && Tx == 256 && Tz == 128;
    if (pipe) {   // layers 1 + 2 of a frame in one workgroup (two wave groups half a period apart); stem_b follows
        constexpr size_t lds_p = vts::PipeGeo<256, 128>::LDS_BYTES;
        const bool diag = m->skip_stem_a != 0 || m->dbg_stamps != nullptr;
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(B), dim3(1024), lds_p, st, z, x, w1, b1, m->stem_w[1].p,
                               m->stem_b[1].p, act_z, act_x, m->skip_stem_a, m->dbg_stamps, m->stem_w2k.p);
        };
        if (diag && zmode != 0) return fail(VT_ERR_STATE, "the diagnostic stem build has no template-cache form");
        if (diag) go(&vts::stem_pipe_kernel<256, 128, 0, true>);
        else if (xu8) go(&vts::stem_pipe_kernel<256, 128, 1, false, true>);
        else if (zmode == 0) go(&vts::stem_pipe_kernel<256, 128, 0, false>);
        else if (zmode == 1) go(&vts::stem_pipe_kernel<256, 128, 1, false>);
        else go(&vts::stem_pipe_kernel<256, 128, 2, false>);
        HIP_TRY(hipGetLastError());
    }
    vts::CropA ax{x, act_x, Tx, px.r2, (Tx / 4) / px.r2}, az{z, act_z, Tz, pz.r2, (Tz / 4) / pz.r2};
    // zmode (template cache): a crop that is not wanted gets zero bands -- stem_a / stem_b index their workgroups by
    // (frame, band of x | band of z), so its workgroups simply do not exist
    // fused form: band k of both crops in one workgroup, when the template band then has exactly one
    // layer-2 tile per wave and the workgroup count fills whole rounds of 4 per CU better than the split form
    const int r2z_f = (Tz / 4) / ax.bands;
    const bool fuse = zmode == 0 && m->stem_fuse && r2z_f >= 1 && r2z_f * ax.bands == Tz / 4 && r2z_f * (Tz / 4) == 64;
    if (zmode == 1) az.bands = 0;
    if (zmode == 2) ax.bands = 0;
    if (pipe) {
    } else if (fuse) {
        az.r2 = r2z_f; az.bands = ax.bands;
        const size_t lds_a = sizeof(float) * (vts::stem_a_lds_floats(Tx, ax.r2) + vts::stem_a_lds_floats(Tz, az.r2));
        hipLaunchKernelGGL(vts::stem_a2_kernel, dim3(B * ax.bands), dim3(256), lds_a, st, ax, az, m->stem_w[0].p,
                           m->stem_b[0].p, m->stem_w[1].p, m->stem_b[1].p, m->skip_stem_a);
    } else {
        const size_t lds_a = sizeof(float) * std::max(vts::stem_a_lds_floats(Tx, px.r2), vts::stem_a_lds_floats(Tz, pz.r2));
        if (xu8)
            hipLaunchKernelGGL(vts::stem_a_kernel<true>, dim3(B * (ax.bands + az.bands)), dim3(256), lds_a, st, ax, az, w1, b1, m->stem_w[1].p,
                               m->stem_b[1].p, m->skip_stem_a);
        else
            hipLaunchKernelGGL(vts::stem_a_kernel<false>, dim3(B * (ax.bands + az.bands)), dim3(256), lds_a, st, ax, az, m->stem_w[0].p,
                               m->stem_b[0].p, m->stem_w[1].p, m->stem_b[1].p, m->skip_stem_a);
    }
    HIP_TRY(hipGetLastError());
    vts::CropB bx{act_x, m->pos_x.p, Tx / 4, px.r4, zmode == 2 ? 0 : (Tx / 16) / px.r4, m->len_z};
    vts::CropB bz{act_z, m->pos_z.p, Tz / 4, pz.r4, zmode == 1 ? 0 : (Tz / 16) / pz.r4, 0};
    const size_t lds_b = std::max(vts::stem_b_lds_bytes(Tx / 4, px.r4), vts::stem_b_lds_bytes(Tz / 4, pz.r4));
    if (m->skip_stem_b)
        hipLaunchKernelGGL(vts::stem_b_kernel<true>, dim3(B * (bx.bands + bz.bands)), dim3(256), lds_b, st, bx, bz, m->stem_w[2].p,
                           m->stem_b[2].p, m->stem_w[3].p, m->stem_b[3].p, tokens, m->L, m->skip_stem_b);
    else
        hipLaunchKernelGGL(vts::stem_b_kernel<false>, dim3(B * (bx.bands + bz.bands)), dim3(256), lds_b, st, bx, bz, m->stem_w[2].p,
                           m->stem_b[2].p, m->stem_w[3].p, m->stem_b[3].p, tokens, m->L, m->skip_stem_b);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

// dynamic LDS of blocks_kernel<NT, ., ., WLDS, BAL> at a given depth: K/V images, weight staging buffers,
// the small parameters (LayerNorm vectors + biases of every block) and the guests' exchange area
size_t blocks_lds_bytes(int NT, bool WLDS, bool BAL, int depth, bool BF3 = false, bool A3 = false) {
    if (A3 && !WLDS)    // the G256 form: K as pieces, V^T an fp32 image, no staging buffers, no guests
        return ((size_t)NT * vtb::W3_FC1_OT16 + (size_t)vtb::NC * NT * 64) * sizeof(f4) + (size_t)vtb::small_floats(depth) * sizeof(float);
    if (A3)     // K / V^T as pieces (vt_blocks.h: KV_UNITS); of the guests' areas only Dg and the counter keep room of their own
        return ((size_t)NT * vtb::W3_FC1_OT16 + (size_t)vtb::NC * ((NT / 2) * 3 * 64 + (NT & 1) * 3 * 32) +
                (size_t)(vtb::W3_FC1_TILES + vtb::WBUF_TILES) * 64) * sizeof(f4) +
               (size_t)vtb::small_floats(depth) * sizeof(float) + (size_t)vtb::NC * 64 * sizeof(f4) + 64;
    return ((size_t)2 * NT * vtb::NC + (WLDS ? (BF3 ? vtb::W3_FC1_TILES : vtb::WBUF_TILES) + vtb::WBUF_TILES : 0)) * 64 * sizeof(f4) +
           (size_t)vtb::small_floats(depth) * sizeof(float) +
           (BAL ? (size_t)(vtb::NC + 4 * vtb::NC + vtb::NC) * 64 * sizeof(f4) + 4 * 2 * 64 * sizeof(float) + 64 : 0);
}
constexpr size_t LDS_PER_CU = 160 * 1024;

// floats of V^T low-piece scratch per frame (vt_blocks.h VP2L): depth x NC feature tiles x L / 32 chunk pairs x 64 lanes x 16 B
size_t vlscr_floats_per_frame(const vt_model* m) { return (size_t)m->cfg.depth * vtb::NC * (m->L / 32) * 64 * 4; }

template <int NT, int NW, int TPW, bool WLDS, bool BAL = false, bool ZC = false, bool BF3 = false, bool A3 = false>
int launch_blocks(vt_model* m, hipStream_t st, const float* tokens, int B, int nblocks, float* feat, float* resid, int zcache_mode, size_t f0 = 0) {
    if (zcache_mode != 0 && !ZC)
        return fail(VT_ERR_STATE, "the template cache needs the default block kernel (VT_BLOCKS_BAL = 1)");
    const size_t lds = blocks_lds_bytes(NT, WLDS, BAL, m->cfg.depth, BF3, A3);
    hipLaunchKernelGGL((vtb::blocks_kernel<NT, NW, TPW, WLDS, BAL, ZC, BF3, A3>), dim3(B), dim3(NW * 64), lds, st, tokens, m->blocks.p, feat,
                       resid, m->len_z, m->cfg.depth, nblocks, m->dbg_skip_tile, m->dbg_stamps, m->zcache.p, zcache_mode, m->blocks3.p,
                       m->vlscr.p ? reinterpret_cast<unsigned*>(m->vlscr.p) + f0 * vlscr_floats_per_frame(m) : nullptr);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

// Small batches: one wave per (tile, frame), two launches per block (vt_blocks_tile.h).
template <int NT>
int launch_blocks_tile(vt_model* m, hipStream_t st, const float* tokens, int B, int nblocks, float* feat, float* resid, int zc, size_t f0) {
    // two workspace sets: a block reads q / K / V^T from one while its workgroups write the next block's into the other.
    // f0 = first frame of this slice in the model workspace: the chains of a multi-chain graph (vt_graph_capture_steps) run
    // concurrently on different slices, so each works in its own part of every workspace.
    const size_t set = (size_t)m->tile_frames * m->L * 48 / 4;                  // float4 per set
    const size_t sl = f0 * m->L * 48 / 4;                                       // float4 offset of the slice inside a set
    f4* const qb = reinterpret_cast<f4*>(m->tile_q.p) + sl;
    f4* const kb = reinterpret_cast<f4*>(m->tile_k.p) + sl;
    f4* const vb = reinterpret_cast<f4*>(m->tile_v.p) + sl;
    float* const tile_x = m->tile_x.p + f0 * m->L * 48;
    const float* const normP = m->blocks.p + (size_t)m->cfg.depth * vtb::BLOCK_STRIDE;      // norm.weight, norm.bias
    hipLaunchKernelGGL((vtb::tile_qkv_kernel<NT>), dim3(NT, B), dim3(64), 0, st, tokens, m->blocks.p, qb, kb, vb, m->zcache.p, zc, m->len_z);
    for (int blk = 0; blk < nblocks; ++blk) {
        const float* P = m->blocks.p + (size_t)blk * vtb::BLOCK_STRIDE;
        const float* xin = blk == 0 ? tokens : tile_x;
        const bool last = blk == nblocks - 1;
        const int skip_z = (blk == m->cfg.depth - 1 && resid == nullptr) ? 1 : 0;
        const size_t cur = (size_t)(blk & 1) * set, nxt = (size_t)((blk & 1) ^ 1) * set;
        hipLaunchKernelGGL((vtb::tile_attn_mlp_kernel<NT>), dim3(NT, B), dim3(256), 0, st, xin, tile_x, P, qb + cur, kb + cur, vb + cur,
                           last ? normP : nullptr, feat, last ? resid : nullptr, m->len_z, skip_z,
                           last ? nullptr : P + vtb::BLOCK_STRIDE, qb + nxt, kb + nxt, vb + nxt);
    }
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int run_blocks(vt_model* m, const float* tokens, int B, int nblocks, hipStream_t st, float* feat, float* resid, int zc = 0, size_t f0 = 0) {
    // zc: template cache mode of block 0 (0 off, 1 store, 2 load); f0: first frame of this slice in the model workspace
    if (zc != 0 && f0 != 0) return fail(VT_ERR_STATE, "the template cache is not sliced");
    if (m->generic) return zc == 1 ? VT_OK      // vt_set_template: the template's token rows are the cache; block 0 is recomputed every frame
                                   : gen_blocks(m, tokens, B, nblocks, st, feat, resid);
    if (nblocks < 0 || nblocks > m->cfg.depth) nblocks = m->cfg.depth;
    // Kernel form by batch size: with few frames a workgroup per frame leaves most of the chip idle (a frame's latency is one
    // CU's worth of MFMA issue); one wave per tile spreads frames x tiles over the SIMDs instead.
    const int NTr = m->L / 16;
    const bool diag = m->dbg_skip_tile != -1 || m->dbg_stamps != nullptr;
    // measured (tools/small_batch_sweep.py, SWEEP_TILE=1; us per step, frame form -> tile form): G256 B=1 281 -> 86, B=32 300 -> 136,
    // B=64 314 -> 183, B=128 371 -> 315; G128 B=1 78 -> 59, B=16 79 -> 62, B=64 84 -> 83, B=80 88 -> 86, B=96 95 -> 95
    const int Bf = form_b(m, B);
    const bool want_tile = m->blocks_tile < 0 ? (NTr == 20 ? Bf <= 128 : Bf <= 80) : m->blocks_tile != 0;
    if (want_tile && !diag && nblocks >= 1 && f0 + (size_t)B <= (size_t)m->tile_frames && (NTr == 5 || NTr == 20))
        return NTr == 5 ? launch_blocks_tile<5>(m, st, tokens, B, nblocks, feat, resid, zc, f0)
                        : launch_blocks_tile<20>(m, st, tokens, B, nblocks, feat, resid, zc, f0);
    switch (m->L / 16) {
        case 5:
#ifndef VT_F16
            if (m->blocks_bal && m->blocks_bf3 >= 2)
                return zc ? launch_blocks<5, 8, 1, true, true, true, true, true>(m, st, tokens, B, nblocks, feat, resid, zc)
                          : launch_blocks<5, 8, 1, true, true, false, true, true>(m, st, tokens, B, nblocks, feat, resid, zc);
            if (m->blocks_bal && m->blocks_bf3)
                return zc ? launch_blocks<5, 8, 1, true, true, true, true>(m, st, tokens, B, nblocks, feat, resid, zc)
                          : launch_blocks<5, 8, 1, true, true, false, true>(m, st, tokens, B, nblocks, feat, resid, zc);
#endif
            if (m->blocks_bal) return zc ? launch_blocks<5, 8, 1, true, true, true>(m, st, tokens, B, nblocks, feat, resid, zc)
                                         : launch_blocks<5, 8, 1, true, true>(m, st, tokens, B, nblocks, feat, resid, zc);
            return m->blocks_wlds ? launch_blocks<5, 5, 1, true>(m, st, tokens, B, nblocks, feat, resid, zc)
                                  : launch_blocks<5, 5, 1, false>(m, st, tokens, B, nblocks, feat, resid, zc);
        case 20:   // 8 waves: waves s and s+4 share SIMD s with 3 + 2 tiles, so each SIMD has two instruction streams
#ifndef VT_F16
            if (m->blocks_bal && m->blocks_bf3_g256 >= 2)
                return zc ? launch_blocks<20, 8, 3, false, false, true, true, true>(m, st, tokens, B, nblocks, feat, resid, zc, f0)
                          : launch_blocks<20, 8, 3, false, false, false, true, true>(m, st, tokens, B, nblocks, feat, resid, zc, f0);
            if (m->blocks_bal && m->blocks_bf3_g256)
                return zc ? launch_blocks<20, 8, 3, false, false, true, true>(m, st, tokens, B, nblocks, feat, resid, zc)
                          : launch_blocks<20, 8, 3, false, false, false, true>(m, st, tokens, B, nblocks, feat, resid, zc);
#endif
            if (m->blocks_bal) return zc ? launch_blocks<20, 8, 3, false, false, true>(m, st, tokens, B, nblocks, feat, resid, zc)
                                         : launch_blocks<20, 8, 3, false>(m, st, tokens, B, nblocks, feat, resid, zc);
            return launch_blocks<20, 4, 5, false>(m, st, tokens, B, nblocks, feat, resid, zc);
        default: return fail(VT_ERR_ARG, "unsupported token count " + std::to_string(m->L));
    }
}

// tail (optional): the tracker's map back / clip / state update / record of vt_track_step, run by the decode kernel itself
int run_decode(vt_model* m, hipStream_t st, const float* score, const float* size, const float* offset,
               const float* window, int B, float* pred, float* hann, float* conf, const TrackTail* tail = nullptr) {
    hipLaunchKernelGGL(vth::decode_kernel, dim3(B), dim3(64), 0, st, score, size, offset, window, m->F, pred, hann, conf,
                       tail ? *tail : TrackTail{}, tail ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

// F = 16: batches up to this size run conv1 as its own launch.  SWEEP_HEAD=1 tools/small_batch_sweep.py, us per step, per-tower
// form -> split form: B=1 86.0 -> 78.6, B=8 96.6 -> 91.5, B=16 110.3 -> 103.5, B=32 135.2 -> 132.9, B=64 181.7 -> 186.2
constexpr int HEAD_SPLIT_MAX_B = 32;

int run_head(vt_model* m, const float* feat, int B, hipStream_t st, const vt_outputs* o, size_t f0 = 0, const TrackTail* tail = nullptr) {
    // outputs of the slice starting at frame f0 (feat already points at the slice)
    const size_t n = (size_t)m->len_x;
    float* score = ((o && o->score_map) ? o->score_map : m->score.p) + f0 * n;
    float* size = ((o && o->size_map) ? o->size_map : m->size.p) + f0 * 2 * n;
    float* offset = ((o && o->offset_map) ? o->offset_map : m->offset.p) + f0 * 2 * n;
    float* pred = ((o && o->pred_boxes) ? o->pred_boxes : m->pred.p) + f0 * 4;
    float* hann = ((o && o->hann_boxes) ? o->hann_boxes : m->hann.p) + f0 * 4;
    float* conf = ((o && o->conf) ? o->conf : m->conf.p) + f0;
    if (m->generic) {
        if (int rcg = gen_head(m, feat, B, st, score, size, offset)) return rcg;
        return run_decode(m, st, score, size, offset, m->window.p, B, pred, hann, conf, tail);
    }
    const int Bf = form_b(m, B);
#ifndef VT_F16
    if (m->F == 8 && m->head_bf3 && !m->skip_head) {       // three-piece bf16 towers (vt_head3.h), same kernel forms by batch size
        const vth3::u32x4* hw3 = reinterpret_cast<const vth3::u32x4*>(m->head3.p);
        if (m->head_fused < 0 ? Bf > 176 : m->head_fused != 0) {
            hipLaunchKernelGGL(vth3::head_fused3_kernel, dim3(B), dim3(768), vth3::FUSED3_LDS_BYTES, st, feat, m->head.p, hw3, m->window.p,
                               score, size, offset, pred, hann, conf, tail ? *tail : TrackTail{}, tail ? 1 : 0);
            HIP_TRY(hipGetLastError());
            return VT_OK;       // (the tracker's tail, if any, ran on the kernel's decoding lane)
        }
        hipLaunchKernelGGL(vth3::head_towers3_kernel, dim3(B, 3), dim3(256), vth3::TOWERS3_LDS_BYTES, st, feat, m->head.p, hw3, score, size,
                           offset);
        HIP_TRY(hipGetLastError());
        return run_decode(m, st, score, size, offset, m->window.p, B, pred, hann, conf, tail);
    }
#endif
    if (m->F == 8 && (m->head_fused < 0 ? Bf > 176 : m->head_fused != 0)) {
        // towers + both decodes in one workgroup per frame
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(B), dim3(768), vth::FusedHeadGeo<8>::LDS_BYTES, st, feat, m->head.p, m->window.p, score,
                               size, offset, pred, hann, conf, m->skip_head, tail ? *tail : TrackTail{}, tail ? 1 : 0);
        };
        if (m->skip_head) go(&vth::head_fused_kernel<8, true>);
        else go(&vth::head_fused_kernel<8, false>);
        HIP_TRY(hipGetLastError());
        return VT_OK;       // (the tracker's tail, if any, ran on the kernel's decoding lane)
    }
    if (m->F == 8) {
        if (m->skip_head)
            hipLaunchKernelGGL((vth::head_towers_kernel<8, 4, true>), dim3(B, 3), dim3(256), vth::Geo<8>::LDS_BYTES, st, feat, m->head.p,
                               score, size, offset, m->skip_head);
        else
            hipLaunchKernelGGL((vth::head_towers_kernel<8, 4, false>), dim3(B, 3), dim3(256), vth::Geo<8>::LDS_BYTES, st, feat, m->head.p,
                               score, size, offset, m->skip_head);
    } else if (m->F == 16 && (m->head_fused < 0 ? Bf > 176 : m->head_fused != 0)) {
        // one workgroup per frame: the three towers in turn on one staged input map, decode from LDS (no decode launch)
#ifndef VT_F16
        if (m->head_bf3 && !m->skip_head) {      // conv1 as three-piece bf16 products (vt_head3.h head_seq3)
            auto go = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(B), dim3(512), vth3::SEQ3_LDS_BYTES, st, feat, m->head.p,
                                   reinterpret_cast<const vth3::u32x4*>(m->head3.p), m->window.p, score, size, offset, pred, hann, conf,
                                   tail ? *tail : TrackTail{}, tail ? 1 : 0, m->dbg_stamps);
            };
            if (m->dbg_stamps) go(&vth3::head_seq3_kernel<8, VT_SEQ3_MAXP, true>);
            else go(&vth3::head_seq3_kernel<8, VT_SEQ3_MAXP, false>);
            HIP_TRY(hipGetLastError());
            return VT_OK;       // (tail: on the kernel's decoding lane)
        }
#endif
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(B), dim3(512), vth::SeqHeadGeo<16>::LDS_BYTES, st, feat, m->head.p, m->window.p, score,
                               size, offset, pred, hann, conf, m->skip_head, tail ? *tail : TrackTail{}, tail ? 1 : 0);
        };
        if (m->skip_head) go(&vth::head_seq_kernel<16, 8, true>);
        else go(&vth::head_seq_kernel<16, 8, false>);
        HIP_TRY(hipGetLastError());
        return VT_OK;       // (tail: on the kernel's decoding lane)
    } else if (m->F == 16 && !m->skip_head && B <= m->head_m1_frames && f0 == 0 &&
               (m->head_split < 0 ? Bf <= HEAD_SPLIT_MAX_B : m->head_split != 0)) {
        // small batches: conv1 of every tower over four row strips (12 workgroups per frame), then the rest of each tower
        hipLaunchKernelGGL(vth::head_conv1_kernel<16>, dim3(4, 3, B), dim3(512), 0, st, feat, m->head.p, m->head_m1.p);
        hipLaunchKernelGGL((vth::head_towers_kernel<16, 8, false, true>), dim3(B, 3), dim3(512), vth::Geo<16>::LDS_BYTES, st,
                           m->head_m1.p, m->head.p, score, size, offset, 0);
    } else if (m->F == 16) {
        // 129 KB of LDS per tower = one workgroup per CU: 8 waves give every SIMD two instruction streams
        if (m->skip_head)
            hipLaunchKernelGGL((vth::head_towers_kernel<16, 8, true>), dim3(B, 3), dim3(512), vth::Geo<16>::LDS_BYTES, st, feat,
                               m->head.p, score, size, offset, m->skip_head);
        else
            hipLaunchKernelGGL((vth::head_towers_kernel<16, 8, false>), dim3(B, 3), dim3(512), vth::Geo<16>::LDS_BYTES, st, feat,
                               m->head.p, score, size, offset, m->skip_head);
    } else {
        return fail(VT_ERR_ARG, "unsupported feat_sz " + std::to_string(m->F));
    }
    HIP_TRY(hipGetLastError());
    return run_decode(m, st, score, size, offset, m->window.p, B, pred, hann, conf, tail);
}

// ViT-Base: towers + conv5 in vitb.hip, then the same decode kernel (first-index argmax, raw and Hann-windowed)
int run_head_vitb(vt_model* m, const float* feat, int B, hipStream_t st, const vt_outputs* o, const vb::Slice* sl = nullptr) {
    const size_t f0 = sl ? sl->f0 : 0, n = (size_t)m->len_x;       // outputs of the slice starting at frame f0
    float* score = ((o && o->score_map) ? o->score_map : m->score.p) + f0 * n;
    float* size = ((o && o->size_map) ? o->size_map : m->size.p) + f0 * 2 * n;
    float* offset = ((o && o->offset_map) ? o->offset_map : m->offset.p) + f0 * 2 * n;
    float* pred = ((o && o->pred_boxes) ? o->pred_boxes : m->pred.p) + f0 * 4;
    float* hann = ((o && o->hann_boxes) ? o->hann_boxes : m->hann.p) + f0 * 4;
    float* conf = ((o && o->conf) ? o->conf : m->conf.p) + f0;
    std::string err;
    int rc = vb::head(m->vb, feat, B, st, score, size, offset, &err, sl);
    if (rc) return fail(rc, err);
    return run_decode(m, st, score, size, offset, m->window.p, B, pred, hann, conf);
}

// Graph chains: a chain may start late (VT_CHAIN_DELAY_US x chain index), so that identical chains do not run in lock step
__global__ void chain_delay_kernel(unsigned long long ticks) {      // 100 MHz ticks
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ------------------------------------------------------------------------------------ self test
__global__ void mfma_selftest_kernel(const float* A, const float* Bm, float* D) {
    // A (16x16 k-chunk as operand image source: A[i][k]), B[k][j]; D[i][j] = sum_k A[i][k] B[k][j]
    const int lane = threadIdx.x, rc = lane & 15, q = lane >> 4;
    f4 a, b;
    for (int r = 0; r < 4; ++r) {
        a[r] = A[rc * 16 + 4 * q + r];
        b[r] = Bm[(4 * q + r) * 16 + rc];
    }
    f4 acc = mfma4(a, b, splat4(0.f));
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + rc] = acc[r];
    // permlane-swap reductions over the 4 lanes sharing (lane & 15): sum must be 15 (rc + 1), max 8 (rc + 1)
    const float v = (float)((1 << q) * (rc + 1));
    D[256 + lane] = quad_sum(v);
    D[320 + lane] = quad_max(v);
}

// Clock / MFMA-rate probe: every wave runs `iters` rounds of 8 independent v_mfma_f32_16x16x4_f32
// and stamps the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime).
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ src, int iters,
                                                    unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
    f4 acc[8];
    const float a0 = src[threadIdx.x], b0 = src[256 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = splat4(0.001f * j);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[j], 0, 0, 0);
    }
    f4 sum = splat4(0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum = sum + acc[j];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = sum.x + sum.y + sum.z + sum.w;   // data dependence on every MFMA
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t k = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
        stamps[k] = t1 - t0;
        stamps[k + 1] = r1 - r0;
    }
}

// ---- crop: which kernel form this device can run
// crop_kernel<false> fetches a bilinear sample's two RGB pixels with ONE 8-byte buffer load at byte offset 3 x: it relies on the
// device serving byte-unaligned dword loads and on out-of-range buffer reads returning zero (tools/src/probe_unaligned.hip).  Neither
// is architectural, so the first vt_create of a process crops a known frame with both forms -- device memory and device-mapped pinned
// host memory (the plugin's zero-copy frames) -- and falls back to the byte-load form on any difference (round 3 advisor).
// The decision is per DEVICE (a process may drive several), taken once under a mutex by whichever vt_create or vt_crop gets there
// first -- ViT-Base models included (round 4 advisor).
constexpr int CROP_MAX_DEVICES = 64;
int g_crop_bytes[CROP_MAX_DEVICES];        // 0: not tested yet, 1: fast form, 2: byte-load form
std::mutex g_crop_mutex;

// u8out: `crops` is a uint8 (B, T, T, 3) patch buffer (sample_target's output; mean3 / std3 unused) instead of the fp32 (B, 3, T, T) crop
void launch_crop(bool bytes, const unsigned char* frames, int H, int W, const double* states, double factor, int T, const float* mean3,
                 const float* std3, int B, hipStream_t st, float* crops, double* rf, bool u8out = false) {
    static const int fast = [] { const char* v = std::getenv("VT_CROP_FAST"); return v && *v ? std::atoi(v) : 1; }();     // groups per workgroup (1, 2, 4); 0: crop_kernel
    static const float none3[3] = {0.f, 1.f, 1.f};
    if (u8out) mean3 = std3 = none3;
    // the tracker's sizes: crop_band_kernel (a workgroup owns a band of VT_CROP_BAND x 256 items; 0: crop_fast_kernel as in round 5)
    static const int band = [] { const char* v = std::getenv("VT_CROP_BAND"); return v && *v ? std::atoi(v) : 4; }();
    bool band_ipt2 = false;
    // ... when its bands fill the chip.  A thread of a band walks its items one after the other, so a few frames are a long dependent chain on a few
    // CUs: one frame at T = 128 takes 12.1 us as four bands of four items, 8.0 as eight bands of two, 5.4 as sixteen workgroups of
    // crop_fast_kernel (tools/gpu_b1prof.sh) -- the form follows the number of workgroups the batch gives each CU
    const long items = (long)B * T * (T / 4);
    const bool bands_fill = band < 0 || items >= 2L * 256 * 256;      // at least one two-item band (512 items) per CU; VT_CROP_BAND=-4 / -2 force a band form (tests)
    if (!bytes && fast > 0 && band != 0 && bands_fill && (T == 64 || T == 128 || T == 256)) {
        if (band > 0 && items < 4L * 256 * 256) band_ipt2 = true;
        auto go = [&](auto kernel, int ipt) {
            hipLaunchKernelGGL(kernel, dim3(T * (T / 4) / (256 * ipt), B), dim3(256), 0, st, frames, H, W, states, factor, mean3[0], mean3[1], mean3[2],
                               std3[0], std3[1], std3[2], crops, rf);
        };
        static const int aligned = [] { const char* v = std::getenv("VT_CROP_ALIGNED"); return v && *v ? std::atoi(v) : 1; }();     // 0: byte-aligned 8-byte windows
        int ipt = ((band >= 4 || band <= -4) && !band_ipt2) ? 4 : 2;
        // the fp32 form (template crops, VT_TRACK_U8=0, ViT-Base) keeps a normalised float4 per channel and item: two items per thread measure
        // 17.3 against 18.0 us (T = 128) and 51.7 against 55.5 (T = 256) for 256 frames; an explicit VT_CROP_BAND decides for both forms
        static const bool band_set = [] { const char* v = std::getenv("VT_CROP_BAND"); return v && *v; }();
        if (!u8out && !band_set) ipt = 2;
        auto pick = [&](auto u8c, auto lgc) {
            constexpr bool U = decltype(u8c)::value;
            constexpr int LG = decltype(lgc)::value;
            if (aligned) ipt == 4 ? go(&vtt::crop_band_kernel<U, LG, 4, true>, 4) : go(&vtt::crop_band_kernel<U, LG, 2, true>, 2);
            else ipt == 4 ? go(&vtt::crop_band_kernel<U, LG, 4, false>, 4) : go(&vtt::crop_band_kernel<U, LG, 2, false>, 2);
        };
        auto by_size = [&](auto u8c) {
            if (T == 64) pick(u8c, std::integral_constant<int, 4>{});
            else if (T == 128) pick(u8c, std::integral_constant<int, 5>{});
            else pick(u8c, std::integral_constant<int, 6>{});
        };
        if (u8out) by_size(std::true_type{});
        else by_size(std::false_type{});
        return;
    }
    if (!bytes && fast > 0 && (T & 3) == 0 && T <= vtt::CROP_FAST_MAX_T) {
        const int ngroups = (T * (T / 4) + 255) / 256;
        if (u8out) {
            hipLaunchKernelGGL((vtt::crop_fast_kernel<1, true>), dim3(ngroups, B), dim3(256), 0, st, frames, H, W, states, factor, T, 0.f, 0.f, 0.f, 1.f,
                               1.f, 1.f, crops, rf);
            return;
        }
        auto go = [&](auto g) {
            constexpr int G = decltype(g)::value;
            hipLaunchKernelGGL((vtt::crop_fast_kernel<G, false>), dim3((ngroups + G - 1) / G, B), dim3(256), 0, st, frames, H, W, states, factor, T, mean3[0],
                               mean3[1], mean3[2], std3[0], std3[1], std3[2], crops, rf);
        };
        if (fast >= 4 && ngroups >= 4) go(std::integral_constant<int, 4>{});
        else if (fast >= 2 && ngroups >= 2) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 1>{});
        return;
    }
    dim3 grid((T * ((T + 3) / 4) + 255) / 256, B);
    if (u8out) {
        if (bytes)
            hipLaunchKernelGGL((vtt::crop_kernel<true, true>), grid, dim3(256), 0, st, frames, H, W, states, factor, T, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, crops, rf);
        else
            hipLaunchKernelGGL((vtt::crop_kernel<false, true>), grid, dim3(256), 0, st, frames, H, W, states, factor, T, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, crops, rf);
        return;
    }
    if (bytes)
        hipLaunchKernelGGL(vtt::crop_kernel<true>, grid, dim3(256), 0, st, frames, H, W, states, factor, T, mean3[0], mean3[1], mean3[2],
                           std3[0], std3[1], std3[2], crops, rf);
    else
        hipLaunchKernelGGL(vtt::crop_kernel<false>, grid, dim3(256), 0, st, frames, H, W, states, factor, T, mean3[0], mean3[1], mean3[2],
                           std3[0], std3[1], std3[2], crops, rf);
}

// device buffers of the self test, released on every path
struct CropProbe {
    unsigned char *dfr = nullptr, *hfr = nullptr;
    double *dst = nullptr, *drf = nullptr;
    float* dout = nullptr;
    ~CropProbe() {
        if (dfr) (void)hipFree(dfr);
        if (hfr) (void)hipHostFree(hfr);
        if (dst) (void)hipFree(dst);
        if (drf) (void)hipFree(drf);
        if (dout) (void)hipFree(dout);
    }
};

// *bytes_form = whether the current device needs the byte-load form
int crop_selftest(bool* bytes_form = nullptr) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= CROP_MAX_DEVICES) return fail(VT_ERR_ARG, "device index beyond the crop self test's table");
    std::lock_guard<std::mutex> lock(g_crop_mutex);
    if (g_crop_bytes[dev] == 0) {
        constexpr int H = 13, W = 17, T = 20, B = 2;          // odd sizes: windows at every byte alignment, crops across all four borders
        std::vector<unsigned char> fr((size_t)B * H * W * 3);
        for (size_t i = 0; i < fr.size(); ++i) fr[i] = (unsigned char)((i * 131u + (i >> 3) * 17u + 7u) & 0xffu);
        const double st[B * 4] = {-2.5, -1.5, 9.0, 8.0, 9.5, 6.25, 9.0, 8.5};      // one box over the top-left corner, one over the bottom-right
        const float mean3[3] = {0.485f, 0.456f, 0.406f}, std3[3] = {0.229f, 0.224f, 0.225f};
        CropProbe b;
        const size_t nout = (size_t)B * 3 * T * T;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b.dfr), fr.size()));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b.hfr), fr.size(), hipHostMallocMapped));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b.dst), sizeof(st)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b.drf), B * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b.dout), 3 * nout * sizeof(float)));
        HIP_TRY(hipMemcpy(b.dfr, fr.data(), fr.size(), hipMemcpyHostToDevice));
        std::memcpy(b.hfr, fr.data(), fr.size());
        HIP_TRY(hipMemcpy(b.dst, st, sizeof(st), hipMemcpyHostToDevice));
        launch_crop(true, b.dfr, H, W, b.dst, 2.0, T, mean3, std3, B, nullptr, b.dout, b.drf);                  // the reference form
        launch_crop(false, b.dfr, H, W, b.dst, 2.0, T, mean3, std3, B, nullptr, b.dout + nout, b.drf);          // fast form, device memory
        launch_crop(false, b.hfr, H, W, b.dst, 2.0, T, mean3, std3, B, nullptr, b.dout + 2 * nout, b.drf);      // fast form, pinned host memory
        HIP_TRY(hipGetLastError());
        std::vector<float> out(3 * nout);
        HIP_TRY(hipMemcpy(out.data(), b.dout, out.size() * sizeof(float), hipMemcpyDeviceToHost));
        const bool same = std::memcmp(out.data(), out.data() + nout, nout * sizeof(float)) == 0 &&
                          std::memcmp(out.data(), out.data() + 2 * nout, nout * sizeof(float)) == 0;
        int form = same ? 1 : 2;
        if (const char* v = std::getenv("VT_CROP_BYTES")) if (*v) form = std::atoi(v) != 0 ? 2 : 1;     // force a form (tests)
        g_crop_bytes[dev] = form;
    }
    if (bytes_form) *bytes_form = g_crop_bytes[dev] == 2;
    return VT_OK;
}

#ifdef VT_F16
// a conv weight image in place: every 16-byte slot's float4 becomes h4 in its first 8 bytes (vt_conv.h load_weights)
__global__ void opnd_inplace_kernel(float* __restrict__ img, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const opnd v = to_opnd(ld4(img + 4 * i));
        *reinterpret_cast<opnd*>(img + 4 * i) = v;
    }
}
int opnd_inplace(float* img, size_t nfloats) {
    hipLaunchKernelGGL(opnd_inplace_kernel, dim3((unsigned)((nfloats / 4 + 255) / 256)), dim3(256), 0, nullptr, img, nfloats / 4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return VT_OK;
}
// float4 -> h4 (the MFMA operand conversion of vt_common.h), n4 quads
__global__ void f32_to_opnd_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<opnd*>(dst)[i] = to_opnd(ld4(src + 4 * i));
}
#endif

// ViT-Base model: the shared part of vt_model is the output scratch, the window and the capture stream
int create_vitb(const vt_config* cfg, vt_model** out) {
    std::string err;
    VbModel* vbm = nullptr;
    int rc = vb::create(cfg, &vbm, &err);
    if (rc) return fail(rc, err);
    if ((rc = crop_selftest())) { vb::destroy(vbm); return rc; }      // every model kind can be handed to vt_crop (after the argument checks: they need no device)
    vt_model* m = new vt_model();
    m->vb = vbm;
    m->cfg = *cfg;
    m->F = cfg->search_size / 16;
    m->Fz = cfg->template_size / 16;
    m->len_x = m->F * m->F;
    m->len_z = m->Fz * m->Fz;
    m->L = m->len_x + m->len_z;
    const size_t B = (size_t)cfg->max_batch;
    auto A = [&](DevBuf& d, size_t n) { if (!rc) rc = d.alloc(n); };
    A(m->score, B * m->len_x); A(m->size, B * 2 * m->len_x); A(m->offset, B * 2 * m->len_x);
    A(m->pred, B * 4); A(m->hann, B * 4); A(m->conf, B);
    if (!rc) rc = upload(m->window, hann2d(m->F));
    if (!rc && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(VT_ERR_HIP, "hipStreamCreate failed");
    // graph chains (vt_graph_capture_steps): frame slices of one step as concurrent chains.  Default (round 5; 0 = auto): a captured step
    // of >= 64 frames runs as TWO chains of half the frames whose persistent GEMMs each launch a workgroup per CU -- the chains' kernels
    // then fill each other's last, partly empty tile rounds (3.75 of 4, 7.5 of 8 at B = 256) and ramps: 17.63 -> 16.88 ms per step at
    // B = 256 (tools/gpu_vbchains.sh; each chain on HALF the CUs instead: 17.66, i.e. nothing -- NOTES R5-6).  Frames are independent:
    // outputs are bit-identical to the one-chain step (tests/test_gpu_variants.py).
    m->graph_chains = env_int("VT_GRAPH_CHAINS", 0);
    m->chain_cus = env_int("VT_CHAIN_CUS", 0);
    m->chain_delay_us = env_int("VT_CHAIN_DELAY_US", 0);
    for (int i = 0; i < 3 && !rc; ++i)
        if (hipStreamCreateWithFlags(&m->side_stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->join_ev[i], hipEventDisableTiming) != hipSuccess)
            rc = fail(VT_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
    if (!rc && hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess) rc = fail(VT_ERR_HIP, "hipEventCreate failed");
    if (rc) { vt_destroy(m); return rc; }
    *out = m;
    return VT_OK;
}

}  // namespace

// =========================================================================================== ABI
// dynamic-LDS limits of every instantiation of the two one-workgroup-per-frame stem kernels
static hipError_t allow_stem_lds() {
    hipError_t e = hipSuccess;
    auto allow = [&](auto kernel, int bytes) {
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    };
    constexpr int lp = (int)vts::PipeGeo<256, 128>::LDS_BYTES, lf = vts::FusedGeo::LDS_BYTES_P;
    allow(&vts::stem_pipe_kernel<256, 128, 0, false>, lp);
    allow(&vts::stem_pipe_kernel<256, 128, 1, false>, lp);
    allow(&vts::stem_pipe_kernel<256, 128, 2, false>, lp);
    allow(&vts::stem_pipe_kernel<256, 128, 0, true>, lp);
    allow(&vts::stem_pipe_kernel<256, 128, 1, false, true>, lp);
    allow(&vts::stem_fused_kernel<0, false>, lf);
    allow(&vts::stem_fused_kernel<1, false>, lf);
    allow(&vts::stem_fused_kernel<2, false>, lf);
    allow(&vts::stem_fused_kernel<0, true>, lf);
    allow(&vts::stem_fused_kernel<0, false, false>, lf);
    allow(&vts::stem_fused_kernel<1, false, false>, lf);
    allow(&vts::stem_fused_kernel<2, false, false>, lf);
    allow(&vts::stem_fused_kernel<1, false, true, true>, lf);
    allow(&vts::stem_fused_kernel<1, false, false, true>, lf);
#ifndef VT_F16
    allow(&vts::stem_stream_kernel<256, 128, 1, true>, (int)vts::StreamGeo<256, 128>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<128, 64, 1, true>, (int)vts::StreamGeo<128, 64>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<256, 128, 0>, (int)vts::StreamGeo<256, 128>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<256, 128, 1>, (int)vts::StreamGeo<256, 128>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<256, 128, 2>, (int)vts::StreamGeo<256, 128>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<128, 64, 0>, (int)vts::StreamGeo<128, 64>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<128, 64, 1>, (int)vts::StreamGeo<128, 64>::LDS_BYTES);
    allow(&vts::stem_stream_kernel<128, 64, 2>, (int)vts::StreamGeo<128, 64>::LDS_BYTES);
#endif
    return e;
}

// vt_load_weights of the shape-generic path (vt_generic.h): plain row-major weights at run-time widths -- BatchNorm folded into the convs
// (Conv2d_BN.fuse, vit_dist.py:22-33) and LayerNorm-1 / -2's affine part folded into qkv / fc1, both in fp64, as on the tuned path.
static int load_weights_generic(vt_model* m, const TensorMap& tm) {
    const vtg::Dims d = m->gd;
    const int C = d.C, HID = d.hid();
    int rc;
    const int sch[5] = {3, C / 8, C / 4, C / 2, C};
    for (int i = 0; i < 4; ++i) {
        const std::string p = "patch_embed.net." + std::to_string(2 * i);
        std::vector<double> w, b;
        if ((rc = fold_conv_bn(tm, p + ".c", p + ".bn", false, sch[i + 1], sch[i], w, b))) return rc;
        if ((rc = upload(m->g_stem_w[i], std::vector<float>(w.begin(), w.end())))) return rc;
        if ((rc = upload(m->g_stem_b[i], std::vector<float>(b.begin(), b.end())))) return rc;
    }
    const float* p;
    if ((rc = need(tm, "pos_embed_z", (int64_t)m->len_z * C, &p))) return rc;
    if ((rc = upload(m->pos_z, std::vector<float>(p, p + (size_t)m->len_z * C)))) return rc;
    if ((rc = need(tm, "pos_embed_x", (int64_t)m->len_x * C, &p))) return rc;
    if ((rc = upload(m->pos_x, std::vector<float>(p, p + (size_t)m->len_x * C)))) return rc;
    std::vector<float> gbp((size_t)m->cfg.depth * d.block_stride() + 2 * C);
    for (int b = 0; b < m->cfg.depth; ++b) {
        const std::string pre = "blocks." + std::to_string(b) + ".";
        float* g = gbp.data() + (size_t)b * d.block_stride();
        // y = W (gamma * n + beta) + b = (W diag gamma) n + (b + W beta)
        auto fold_ln = [&](const char* ln, const char* lin, int out, int o_w, int o_b) -> int {
            const float *G, *Be, *W, *Bi;
            int r2;
            if ((r2 = need(tm, pre + ln + ".weight", C, &G)) || (r2 = need(tm, pre + ln + ".bias", C, &Be)) ||
                (r2 = need(tm, pre + lin + ".weight", (int64_t)out * C, &W)) || (r2 = need(tm, pre + lin + ".bias", out, &Bi)))
                return r2;
            for (int o = 0; o < out; ++o) {
                double acc = (double)Bi[o];
                for (int i = 0; i < C; ++i) {
                    g[o_w + (size_t)o * C + i] = (float)((double)W[(size_t)o * C + i] * (double)G[i]);
                    acc += (double)W[(size_t)o * C + i] * (double)Be[i];
                }
                g[o_b + o] = (float)acc;
            }
            return VT_OK;
        };
        auto plain = [&](const char* lin, int out, int in, int o_w, int o_b) -> int {
            const float *W, *Bi;
            int r2;
            if ((r2 = need(tm, pre + lin + ".weight", (int64_t)out * in, &W)) || (r2 = need(tm, pre + lin + ".bias", out, &Bi))) return r2;
            std::memcpy(g + o_w, W, (size_t)out * in * sizeof(float));
            std::memcpy(g + o_b, Bi, (size_t)out * sizeof(float));
            return VT_OK;
        };
        if ((rc = fold_ln("norm1", "attn.qkv", 3 * C, d.o_wqkv(), d.o_bqkv()))) return rc;
        if ((rc = plain("attn.proj", C, C, d.o_wproj(), d.o_bproj()))) return rc;
        if ((rc = fold_ln("norm2", "mlp.fc1", HID, d.o_w1(), d.o_b1()))) return rc;
        if ((rc = plain("mlp.fc2", C, HID, d.o_w2(), d.o_b2()))) return rc;
    }
    {
        float* g = gbp.data() + (size_t)m->cfg.depth * d.block_stride();
        if ((rc = need(tm, "norm.weight", C, &p))) return rc;
        std::memcpy(g, p, C * sizeof(float));
        if ((rc = need(tm, "norm.bias", C, &p))) return rc;
        std::memcpy(g + C, p, C * sizeof(float));
    }
    if ((rc = upload(m->g_blocks, gbp))) return rc;
    std::vector<float> ghp((size_t)3 * d.tower_stride(), 0.f);
    const char* towers[3] = {"ctr", "offset", "size"};
    for (int t = 0; t < 3; ++t) {
        float* g = ghp.data() + (size_t)t * d.tower_stride();
        for (int i = 0; i < 4; ++i) {
            const std::string cn = std::string("box_head.conv") + std::to_string(i + 1) + "_" + towers[t];
            std::vector<double> w, b;
            if ((rc = fold_conv_bn(tm, cn + ".0", cn + ".1", true, d.hch(i + 1), d.hch(i), w, b))) return rc;
            for (size_t k = 0; k < w.size(); ++k) g[d.ho_w(i) + k] = (float)w[k];
            for (int o = 0; o < d.hch(i + 1); ++o) g[d.ho_b(i) + o] = (float)b[o];
        }
        const int nout = t == 0 ? 1 : 2, c4 = d.hch(4);
        const std::string c5 = std::string("box_head.conv5_") + towers[t];
        if ((rc = need(tm, c5 + ".weight", (int64_t)nout * c4, &p))) return rc;
        std::memcpy(g + d.ho_w5(), p, (size_t)nout * c4 * sizeof(float));
        if ((rc = need(tm, c5 + ".bias", nout, &p))) return rc;
        std::memcpy(g + d.ho_b5(), p, nout * sizeof(float));
    }
    if ((rc = upload(m->g_head, ghp))) return rc;
    m->weights_loaded = true;
    return VT_OK;
}

extern "C" {

const char* vt_last_error(void) { return g_err.c_str(); }
const char* vt_version(void) { return "vittrack-hip 0.2 (gfx950, " VT_PRECISION_NAME " contractions)"; }

int vt_create(const vt_config* cfg, vt_model** out) {
    if (!cfg || !out) return fail(VT_ERR_ARG, "null argument");
    if (cfg->channels == 768) return create_vitb(cfg, out);
    // build_ostrack_dist takes embed_dim / num_heads / the head width from the YAML (vit_dist.py:159-164; head.py:352-359): the shipped widths
    // (48, 1, 32) at the two geometries of the repository's configs run the tuned kernels, everything else the shape-generic ones
    const bool shipped_widths = cfg->channels == 48 && cfg->heads == 1 && cfg->head_channels == 32;
    if (cfg->stride != 16 || cfg->channels < 8 || cfg->channels > 1024 || cfg->channels % 8 != 0 || cfg->heads < 1 || cfg->channels % cfg->heads != 0 ||
        cfg->channels / cfg->heads > vtg::MAXHD || cfg->head_channels < 8 || cfg->head_channels > 1024 || cfg->head_channels % 8 != 0)
        return fail(VT_ERR_ARG,
                    "unsupported model: STRIDE must be 16, CHANNELS a multiple of 8 (the stem's widths are C/8, C/4, C/2, C) divisible by HEADS with a head "
                    "dimension of at most " + std::to_string(vtg::MAXHD) + ", HEAD.NUM_CHANNELS a multiple of 8 (the towers' widths are W, W/2, W/4, W/8); "
                    "got channels=" + std::to_string(cfg->channels) + " heads=" + std::to_string(cfg->heads) + " head_channels=" + std::to_string(cfg->head_channels) +
                    " stride=" + std::to_string(cfg->stride));
    const bool g128 = shipped_widths && cfg->template_size == 64 && cfg->search_size == 128;
    const bool g256 = shipped_widths && cfg->template_size == 128 && cfg->search_size == 256;
    const bool generic = !g128 && !g256;
    if (generic && (cfg->template_size % 16 != 0 || cfg->search_size % 16 != 0 || cfg->template_size < 16 || cfg->search_size < 16 ||
                    cfg->template_size > 512 || cfg->search_size > 512))
        return fail(VT_ERR_ARG, "unsupported geometry (template,search)=(" + std::to_string(cfg->template_size) + "," +
                                    std::to_string(cfg->search_size) + "): sizes must be multiples of 16 in [16, 512]; tuned kernels exist for (64,128) and "
                                    "(128,256), every other size runs the shape-generic kernels");
    if (cfg->depth < 1 || cfg->depth > 12 || cfg->max_batch < 1) return fail(VT_ERR_ARG, "bad depth / max_batch");
    if (!generic) {   // the block kernel keeps every block's LayerNorm vectors and biases in LDS next to the K/V images
        const size_t need = g128 ? blocks_lds_bytes(5, true, true, cfg->depth) : blocks_lds_bytes(20, false, false, cfg->depth);
        if (need > LDS_PER_CU)
            return fail(VT_ERR_ARG, "depth " + std::to_string(cfg->depth) + " needs " + std::to_string(need) +
                                        " B of LDS per workgroup at this geometry (limit " + std::to_string(LDS_PER_CU) + ")");
    }

    if (int rcs = crop_selftest()) return rcs;
    vt_model* m = new vt_model();
    m->cfg = *cfg;
    m->F = cfg->search_size / 16;
    m->Fz = cfg->template_size / 16;
    m->len_x = m->F * m->F;
    m->len_z = m->Fz * m->Fz;
    m->L = m->len_x + m->len_z;
    const size_t B = (size_t)cfg->max_batch;
    int rc = VT_OK;
    auto A = [&](DevBuf& d, size_t n) { if (!rc) rc = d.alloc(n); };
    m->generic = generic;
    m->gd = vtg::Dims{cfg->channels, cfg->heads, cfg->head_channels};
    const size_t C = (size_t)cfg->channels, HW = (size_t)cfg->head_channels;
    A(m->tokens, B * m->L * C);
    A(m->feat, B * m->len_x * C);
    A(m->tokens_c, B * m->L * C);
    if (generic) {
        const size_t T = (size_t)std::max(cfg->search_size, cfg->template_size);
        // ping-pong maps: stem layers 1 / 3 and head convs 1 / 3 in g_a, layers 2 and convs 2 / 4 in g_b
        A(m->g_a, std::max({B * (C / 8) * (T / 2) * (T / 2), B * (C / 2) * (T / 8) * (T / 8), 3 * B * (size_t)m->len_x * HW}));
        A(m->g_b, std::max(B * (C / 4) * (T / 4) * (T / 4), 3 * B * (size_t)m->len_x * (HW / 2)));
        A(m->g_qkv, B * m->L * 3 * C);
        A(m->g_ao, B * m->L * C);
        A(m->g_hid, B * m->L * 4 * C);
        A(m->g_x, B * m->L * C);
    } else {
    A(m->act_x, B * (size_t)(cfg->search_size / 4) * (cfg->search_size / 4) * 12);
    A(m->act_z, B * (size_t)(cfg->template_size / 4) * (cfg->template_size / 4) * 12);
    A(m->zcache, B * (size_t)(m->len_z / 16) * 9 * 256);
    if (m->L / 16 == 20 && !VT_IS_F16) A(m->vlscr, B * vlscr_floats_per_frame(m));      // (the f16 build's block kernels keep V^T as one f16 image)
    m->tile_frames = (int)std::min<size_t>(B, 128);
    A(m->tile_q, 2 * (size_t)m->tile_frames * m->L * 48);      // two sets each (launch_blocks_tile)
    A(m->tile_k, 2 * (size_t)m->tile_frames * m->L * 48);
    A(m->tile_v, 2 * (size_t)m->tile_frames * m->L * 48);
    A(m->tile_x, (size_t)m->tile_frames * m->L * 48);
    }
    if (!generic && m->F == 16) {
        m->head_m1_frames = (int)std::min<size_t>(B, 176);
        A(m->head_m1, (size_t)m->head_m1_frames * 3 * 8 * vth::Geo<16>::NPIX * 4);
        if (!rc && hipMemset(m->head_m1.p, 0, m->head_m1.n * sizeof(float)) != hipSuccess) rc = fail(VT_ERR_HIP, "hipMemset(head_m1) failed");
    }
    A(m->score, B * m->len_x);
    A(m->size, B * 2 * m->len_x);
    A(m->offset, B * 2 * m->len_x);
    A(m->pred, B * 4);
    A(m->hann, B * 4);
    A(m->conf, B);
    if (!rc && hipMemset(m->tokens_c.p, 0, m->tokens_c.n * sizeof(float)) != hipSuccess) rc = fail(VT_ERR_HIP, "hipMemset(tokens) failed");
    m->skip_stem_a = env_int("VT_SKIP_STEM_A", 0);
    m->skip_stem_b = env_int("VT_SKIP_STEM_B", 0);
    m->skip_head = env_int("VT_SKIP_HEAD", 0);
    m->dbg_skip_tile = env_int("VT_DBG_SKIP_TILE", -1);
    m->graph_chains = env_int("VT_GRAPH_CHAINS", 1);
    m->blocks_wlds = env_int("VT_BLOCKS_WLDS", 1);
    m->blocks_bal = env_int("VT_BLOCKS_BAL", 1);
    m->blocks_bf3 = env_int("VT_BLOCKS_BF3", 2);
    m->blocks_bf3_g256 = m->blocks_bf3;
    if (m->blocks_bf3_g256 >= 2 && blocks_lds_bytes(20, false, false, cfg->depth, true, true) > LDS_PER_CU) m->blocks_bf3_g256 = 1;
    if (m->blocks_bf3 >= 2 && blocks_lds_bytes(5, true, true, cfg->depth, true, true) > LDS_PER_CU) m->blocks_bf3 = 1;
    // the BF3 form's staging buffers are 18 KiB larger: beyond depth 8 its small parameters no longer fit beside them -> fp32 form
    if (blocks_lds_bytes(5, true, true, cfg->depth, true) > LDS_PER_CU) m->blocks_bf3 = 0;
    m->stem_fused = env_int("VT_STEM_FUSED", -1);
    m->stem_pipe = env_int("VT_STEM_PIPE", -1);
    m->stem_stream = env_int("VT_STEM_STREAM", -1);
    m->head_fused = env_int("VT_HEAD_FUSED", -1);
    m->head_bf3 = env_int("VT_HEAD_BF3", 1);      // default since the sustained A/B (tools/power_probe.py): 93.5 -> 86.4 us per step at equal clocks
    m->blocks_tile = env_int("VT_BLOCKS_TILE", -1);
    m->head_split = env_int("VT_HEAD_SPLIT", -1);
    m->stem_fuse = env_int("VT_STEM_FUSE", cfg->search_size == 128 ? 1 : 0);
    m->stem_bf3 = env_int("VT_STEM_BF3", 1);
    m->track_u8 = env_int("VT_TRACK_U8", 1);
    if (!generic) {
        const StemPlan sx = stem_plan(cfg->search_size), sz = stem_plan(cfg->template_size);
        m->plan_r2[0] = sx.r2; m->plan_r4[0] = sx.r4; m->plan_r2[1] = sz.r2; m->plan_r4[1] = sz.r4;
        const char* v = std::getenv("VT_STEM_R4_128");
        m->r4_128_forced = v && *v;
    }
    if (!rc && env_int("VT_DBG_STAMPS", 0)) {
        if (hipMalloc(reinterpret_cast<void**>(&m->dbg_stamps), B * 8 * 64 * sizeof(unsigned long long)) != hipSuccess)
            rc = fail(VT_ERR_HIP, "hipMalloc(stamps) failed");
    }
    if (!rc) rc = upload(m->window, hann2d(m->F));
    if (!rc && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(VT_ERR_HIP, "hipStreamCreate failed");
    for (int i = 0; i < 3 && !rc; ++i)
        if (hipStreamCreateWithFlags(&m->side_stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->join_ev[i], hipEventDisableTiming) != hipSuccess)
            rc = fail(VT_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
    if (!rc && hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess) rc = fail(VT_ERR_HIP, "hipEventCreate failed");
    if (!rc) {
        // > 64 KiB of dynamic LDS needs an explicit opt-in; the limits are the exact sizes launch_blocks computes
        const int small_bytes = vtb::small_floats(cfg->depth) * (int)sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 4, 5, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 5, 1, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, false, cfg->depth));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 5, 1, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, false, false, cfg->depth));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 8, 3, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 8, 1, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, true, cfg->depth));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 8, 1, true, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, true, cfg->depth));
#ifndef VT_F16
        if (e == hipSuccess && m->blocks_bf3)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 8, 1, true, true, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, true, cfg->depth, true));
        if (e == hipSuccess && m->blocks_bf3)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 8, 1, true, true, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, true, cfg->depth, true));
        if (e == hipSuccess && m->blocks_bf3 >= 2)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 8, 1, true, true, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, true, cfg->depth, true, true));
        if (e == hipSuccess && m->blocks_bf3 >= 2)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<5, 8, 1, true, true, true, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(5, true, true, cfg->depth, true, true));
#endif
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 8, 3, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth));
#ifndef VT_F16
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 8, 3, false, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 8, 3, false, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth));
        if (e == hipSuccess && m->blocks_bf3_g256 >= 2)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 8, 3, false, false, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth, true, true));
        if (e == hipSuccess && m->blocks_bf3_g256 >= 2)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vtb::blocks_kernel<20, 8, 3, false, false, true, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)blocks_lds_bytes(20, false, false, cfg->depth, true, true));
#endif

        (void)small_bytes;
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_fused_kernel<8, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, vth::FusedHeadGeo<8>::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_fused_kernel<8, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, vth::FusedHeadGeo<8>::LDS_BYTES);
#ifndef VT_F16
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth3::head_fused3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    vth3::FUSED3_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth3::head_towers3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    vth3::TOWERS3_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth3::head_seq3_kernel<8, VT_SEQ3_MAXP, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, vth3::SEQ3_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth3::head_seq3_kernel<8, VT_SEQ3_MAXP, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, vth3::SEQ3_LDS_BYTES);
#endif
        if (e == hipSuccess)
            e = allow_stem_lds();
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_seq_kernel<16, 8, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(vth::SeqHeadGeo<16>::LDS_BYTES));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_seq_kernel<16, 8, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(vth::SeqHeadGeo<16>::LDS_BYTES));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_towers_kernel<16, 8, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(vth::Geo<16>::LDS_BYTES));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_towers_kernel<16, 8, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(vth::Geo<16>::LDS_BYTES));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vth::head_towers_kernel<16, 8, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(vth::Geo<16>::LDS_BYTES));
        if (e != hipSuccess) rc = fail(VT_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    if (rc) {
        vt_destroy(m);
        return rc;
    }
    *out = m;
    return VT_OK;
}

void vt_destroy(vt_model* m) {
    if (!m) return;
    for (vt_graph* g : m->graphs) g->owner = nullptr;      // graphs that outlive their model must not reach into it (they may still be destroyed)
    m->graphs.clear();
    if (m->vb) vb::destroy(m->vb);
    for (int i = 0; i < 4; ++i) { m->stem_w[i].release(); m->stem_b[i].release(); }
    m->stem_w2k.release();
    m->stem_w1u.release();
    m->stem_w3b.release();
    m->stem_w4b.release();
    m->act_x.release(); m->act_z.release();
    DevBuf* all[] = {&m->pos_z, &m->pos_x, &m->blocks, &m->blocks3, &m->head, &m->head3, &m->window, &m->tokens, &m->feat, &m->zcache, &m->vlscr, &m->tokens_c,
                     &m->tile_q, &m->tile_k, &m->tile_v, &m->tile_x, &m->head_m1,
                     &m->score, &m->size, &m->offset, &m->pred, &m->hann, &m->conf};
    for (DevBuf* d : all) d->release();
    for (int i = 0; i < 4; ++i) { m->g_stem_w[i].release(); m->g_stem_b[i].release(); }
    DevBuf* gen[] = {&m->g_blocks, &m->g_head, &m->g_a, &m->g_b, &m->g_qkv, &m->g_ao, &m->g_hid, &m->g_x};
    for (DevBuf* d : gen) d->release();
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    for (int i = 0; i < 3; ++i) {
        if (m->side_stream[i]) (void)hipStreamDestroy(m->side_stream[i]);
        if (m->join_ev[i]) (void)hipEventDestroy(m->join_ev[i]);
    }
    if (m->fork_ev) (void)hipEventDestroy(m->fork_ev);
    if (m->dbg_stamps) (void)hipFree(m->dbg_stamps);
    delete m;
}

int vt_load_weights(vt_model* m, const vt_tensor* tensors, int32_t n) {
    if (!m || !tensors || n < 0) return fail(VT_ERR_ARG, "null argument");
    TensorMap tm;
    for (int i = 0; i < n; ++i)
        if (tensors[i].name && tensors[i].data) tm[tensors[i].name] = {tensors[i].data, tensors[i].numel};
    int rc;
    if (m->vb) {
        std::string err;
        if ((rc = vb::load_weights(m->vb, tm, &err))) return fail(rc, err);
        m->weights_loaded = true;
        return VT_OK;
    }
    if (m->generic) return load_weights_generic(m, tm);
    const int C = 48;
    // ---- stem (patch_embed.net.{0,2,4,6}.{c,bn})
    for (int i = 0; i < 4; ++i) {
        const std::string p = "patch_embed.net." + std::to_string(2 * i);
        std::vector<double> w, b;
        if ((rc = fold_conv_bn(tm, p + ".c", p + ".bn", false, STEM_CH[i + 1], STEM_CH[i], w, b))) return rc;
        if (i < 1) {   // VALU layer: [r][cin][s][cout] sections, weights become scalar operands
            if ((rc = upload(m->stem_w[i], pack_conv_sections(w, STEM_CH[i + 1], STEM_CH[i])))) return rc;
            if ((rc = upload(m->stem_b[i], std::vector<float>(b.begin(), b.end())))) return rc;
            m->stem_w1_f64 = w;
            m->stem_b1_f64 = b;
            if ((rc = fold_w1u(m, m->norm_mean, m->norm_std))) return rc;      // the uint8-patch form of layer 1
        } else {       // MFMA layers: A-operand images, bias padded to whole 16-channel tiles
            const int tiles = (STEM_CH[i + 1] + 15) / 16, nch = (9 * ((STEM_CH[i] + 3) / 4) + 3) / 4;
            std::vector<float> img((size_t)tiles * nch * 256), bias((size_t)tiles * 16, 0.f);
            pack_conv_image(w, STEM_CH[i + 1], STEM_CH[i], img.data());
            for (int o = 0; o < STEM_CH[i + 1]; ++o) bias[o] = (float)b[o];
            if ((rc = upload(m->stem_w[i], img))) return rc;
            if ((rc = upload(m->stem_b[i], bias))) return rc;
#ifdef VT_F16
            if (i >= 2 && (rc = opnd_inplace(m->stem_w[i].p, img.size()))) return rc;      // layers 3 / 4: stored operands (vt_conv.h); layer 2's image stays float4
#endif
            if (i == 1) {   // [tap][ic / 4][16 oc][ic % 4]: element = w[oc][ic][tap], zero beyond 12 x 6
                std::vector<float> k((size_t)9 * 2 * 16 * 4, 0.f);
                for (int tap = 0; tap < 9; ++tap)
                    for (int oc = 0; oc < STEM_CH[2]; ++oc)
                        for (int ic = 0; ic < STEM_CH[1]; ++ic)
                            k[(((size_t)tap * 2 + ic / 4) * 16 + oc) * 4 + ic % 4] = (float)w[((size_t)oc * STEM_CH[1] + ic) * 9 + tap];
                if ((rc = upload(m->stem_w2k, k))) return rc;
            }
            if (i == 2) {   // layer 3 as three-piece bf16 images (vt_stem_fused.h, fp32 build): [out tile 2][pair 4][piece 3][64][8 bf16]
                std::vector<uint16_t> img3((size_t)2 * 4 * 3 * 64 * 8, 0);
                pack_conv_image3(w, STEM_CH[3], STEM_CH[2], img3.data());
                std::vector<float> as_f(img3.size() / 2);
                std::memcpy(as_f.data(), img3.data(), img3.size() * 2);
                if ((rc = upload(m->stem_w3b, as_f))) return rc;
            }
            if (i == 3) {   // layer 4 as three-piece bf16 images (stem_fused with VT_STEM_BF3): [out tile 3][pair 7][piece 3][64][8 bf16]
                std::vector<uint16_t> img4((size_t)3 * 7 * 3 * 64 * 8, 0);
                pack_conv_image3(w, STEM_CH[4], STEM_CH[3], img4.data());
                std::vector<float> as_f(img4.size() / 2);
                std::memcpy(as_f.data(), img4.data(), img4.size() * 2);
                if ((rc = upload(m->stem_w4b, as_f))) return rc;
            }
        }
    }
    const float* p;
    if ((rc = need(tm, "pos_embed_z", (int64_t)m->len_z * C, &p))) return rc;
    if ((rc = upload(m->pos_z, std::vector<float>(p, p + (size_t)m->len_z * C)))) return rc;
    if ((rc = need(tm, "pos_embed_x", (int64_t)m->len_x * C, &p))) return rc;
    if ((rc = upload(m->pos_x, std::vector<float>(p, p + (size_t)m->len_x * C)))) return rc;
    // ---- transformer blocks + final norm
    std::vector<float> bp((size_t)m->cfg.depth * vtb::BLOCK_STRIDE + 2 * C);
    std::vector<uint16_t> bp3((size_t)m->cfg.depth * vtb::BLOCK3_STRIDE * 2, 0);
    for (int b = 0; b < m->cfg.depth; ++b) {
        const std::string pre = "blocks." + std::to_string(b) + ".";
        float* dst = bp.data() + (size_t)b * vtb::BLOCK_STRIDE;
        struct V { const char* name; int off; int n; };
        const V vecs[] = {{"norm1.weight", vtb::O_LN1G, C}, {"norm1.bias", vtb::O_LN1B, C},
                          {"attn.qkv.bias", vtb::O_BQKV, 3 * C}, {"attn.proj.bias", vtb::O_BPROJ, C},
                          {"norm2.weight", vtb::O_LN2G, C}, {"norm2.bias", vtb::O_LN2B, C},
                          {"mlp.fc1.bias", vtb::O_B1, 4 * C}, {"mlp.fc2.bias", vtb::O_B2, C}};
        for (const V& v : vecs) {
            if ((rc = need(tm, pre + v.name, v.n, &p))) return rc;
            std::memcpy(dst + v.off, p, v.n * sizeof(float));
        }
        // norm1 -> qkv and norm2 -> fc1: the LayerNorm's affine part is folded into the linear layer that consumes it, in
        // double (y = W (gamma * n + beta) + b = (W diag gamma) n + (b + W beta)); the kernels normalise only (vt_blocks.h).
        auto fold_ln = [&](const char* wname, int out, int o_ln_g, int o_ln_b, int o_bias, int o_w) -> int {
            const float* W;
            int rc2 = need(tm, pre + wname, (int64_t)out * C, &W);
            if (rc2) return rc2;
            std::vector<float> wf((size_t)out * C);
            for (int o = 0; o < out; ++o) {
                double acc = (double)dst[o_bias + o];
                for (int i = 0; i < C; ++i) {
                    wf[(size_t)o * C + i] = (float)((double)W[(size_t)o * C + i] * (double)dst[o_ln_g + i]);
                    acc += (double)W[(size_t)o * C + i] * (double)dst[o_ln_b + i];
                }
                dst[o_bias + o] = (float)acc;
            }
            pack_linear_image(wf.data(), out, C, dst + o_w);
            return VT_OK;
        };
        if ((rc = fold_ln("attn.qkv.weight", 3 * C, vtb::O_LN1G, vtb::O_LN1B, vtb::O_BQKV, vtb::O_WQKV))) return rc;
        if ((rc = need(tm, pre + "attn.proj.weight", C * C, &p))) return rc;
        pack_linear_image(p, C, C, dst + vtb::O_WPROJ);
        if ((rc = fold_ln("mlp.fc1.weight", 4 * C, vtb::O_LN2G, vtb::O_LN2B, vtb::O_B1, vtb::O_W1))) return rc;
        if ((rc = need(tm, pre + "mlp.fc2.weight", 4 * C * C, &p))) return rc;
        pack_linear_image(p, C, 4 * C, dst + vtb::O_W2);
        pack_mlp_images3(dst + vtb::O_W1, dst + vtb::O_W2, dst + vtb::O_WQKV, dst + vtb::O_WPROJ, bp3.data() + (size_t)b * vtb::BLOCK3_STRIDE * 2);
    }
    {
        float* dst = bp.data() + (size_t)m->cfg.depth * vtb::BLOCK_STRIDE;
        if ((rc = need(tm, "norm.weight", C, &p))) return rc;
        std::memcpy(dst, p, C * sizeof(float));
        if ((rc = need(tm, "norm.bias", C, &p))) return rc;
        std::memcpy(dst + C, p, C * sizeof(float));
    }
    if ((rc = upload(m->blocks, bp))) return rc;
#ifdef VT_F16
    {   // f16 build: the block kernels read their weight images as stored operands (vt_common.h `opnd` = h4) -- converted ONCE here,
        // on the device, by the conversion the kernels applied at every MFMA call before (bit-identical results); BLOCK_STRIDE halves
        // per block at the float layout's offsets, + 1 KiB of slack behind the last image (the staging DMA moves whole KiB)
        const size_t nflt = (size_t)m->cfg.depth * vtb::BLOCK_STRIDE;
        m->blocks3.release();
        if ((rc = m->blocks3.alloc(nflt / 2 + 256 + 256))) return rc;
        hipLaunchKernelGGL(f32_to_opnd_kernel, dim3((unsigned)((nflt / 4 + 255) / 256)), dim3(256), 0, nullptr, m->blocks.p,
                           reinterpret_cast<_Float16*>(m->blocks3.p), nflt / 4);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
#else
    {
        std::vector<float> as_f(bp3.size() / 2);
        std::memcpy(as_f.data(), bp3.data(), bp3.size() * 2);
        if ((rc = upload(m->blocks3, as_f))) return rc;
    }
#endif
    // ---- head (box_head.conv{1..4}_{ctr,offset,size}.{0,1}, conv5_*)
    std::vector<float> hp((size_t)3 * vth::TOWER_STRIDE, 0.f);
#ifndef VT_F16
    std::vector<uint16_t> hp3((size_t)3 * vth3::TOWER3_STRIDE * 8, 0);
#endif
    const char* towers[3] = {"ctr", "offset", "size"};
    const int chans[5] = {48, 32, 16, 8, 4};
    const int woff[4] = {vth::O_W1, vth::O_W2, vth::O_W3, vth::O_W4};
    const int boff[4] = {vth::O_B1, vth::O_B2, vth::O_B3, vth::O_B4};
    for (int t = 0; t < 3; ++t) {
        float* dst = hp.data() + (size_t)t * vth::TOWER_STRIDE;
        for (int i = 0; i < 4; ++i) {
            const std::string cn = std::string("box_head.conv") + std::to_string(i + 1) + "_" + towers[t];
            std::vector<double> w, b;
            if ((rc = fold_conv_bn(tm, cn + ".0", cn + ".1", true, chans[i + 1], chans[i], w, b))) return rc;
            pack_conv_image(w, chans[i + 1], chans[i], dst + woff[i]);
            if (i == 2) pack_conv_quads(w, 8, 16, dst + vth::O_W3Q);
            if (i == 3) pack_conv_quads(w, 4, 8, dst + vth::O_W4Q);
#ifndef VT_F16
            {
                const int woff3[4] = {vth3::O3_W1, vth3::O3_W2, vth3::O3_W3, vth3::O3_W4};
                pack_conv_image3(w, chans[i + 1], chans[i], hp3.data() + ((size_t)t * vth3::TOWER3_STRIDE + woff3[i]) * 8);
            }
#endif
            for (int o = 0; o < chans[i + 1]; ++o) dst[boff[i] + o] = (float)b[o];
        }
        const int nout = t == 0 ? 1 : 2;
        const std::string c5 = std::string("box_head.conv5_") + towers[t];
        if ((rc = need(tm, c5 + ".weight", nout * 4, &p))) return rc;
        std::memcpy(dst + vth::O_W5, p, nout * 4 * sizeof(float));
        if ((rc = need(tm, c5 + ".bias", nout, &p))) return rc;
        std::memcpy(dst + vth::O_B5, p, nout * sizeof(float));
    }
    if ((rc = upload(m->head, hp))) return rc;
#ifdef VT_F16
    for (int t = 0; t < 3; ++t)      // the towers' conv images as stored operands, in place (biases and conv5 stay float)
        for (int i = 0; i < 4; ++i) {
            const int sizes[4] = {vth::O_B1 - vth::O_W1, vth::O_B2 - vth::O_W2, vth::O_B3 - vth::O_W3, vth::O_B4 - vth::O_W4};
            if ((rc = opnd_inplace(m->head.p + (size_t)t * vth::TOWER_STRIDE + woff[i], (size_t)sizes[i]))) return rc;
        }
#endif
#ifndef VT_F16
    {     // the three-piece bf16 images of vt_head3.h (as floats: 16-byte units x 4); F = 16 reads conv1's only
        std::vector<float> as_f(hp3.size() / 2);
        std::memcpy(as_f.data(), hp3.data(), hp3.size() * 2);
        if ((rc = upload(m->head3, as_f))) return rc;
    }
#endif
    m->weights_loaded = true;
    return VT_OK;
}

int vt_set_window(vt_model* m, const float* host_window) {
    if (!m || !host_window) return fail(VT_ERR_ARG, "null argument");
    return upload(m->window, std::vector<float>(host_window, host_window + (size_t)m->F * m->F));
}

int vt_set_form_batch(vt_model* m, int32_t n) {
    if (!m) return fail(VT_ERR_ARG, "null model");
    if (n < 0) return fail(VT_ERR_ARG, "vt_set_form_batch: n must be >= 0 (0 = choose the kernel forms by each call's own batch)");
    if (n != m->form_batch && m->graphs_captured > 0)
        return fail(VT_ERR_STATE, "vt_set_form_batch after vt_graph_capture: the captured graphs keep the forms of their capture -- set the form "
                                  "batch before capturing (or use a fresh model)");
    m->form_batch = n;
    return VT_OK;
}

int vt_query(const vt_model* m, int32_t* len_z, int32_t* len_x, int32_t* feat_sz, int32_t* channels) {
    if (!m) return fail(VT_ERR_ARG, "null model");
    if (len_z) *len_z = m->len_z;
    if (len_x) *len_x = m->len_x;
    if (feat_sz) *feat_sz = m->F;
    if (channels) *channels = m->cfg.channels;
    return VT_OK;
}

int vt_stem(vt_model* m, const float* z_dev, const float* x_dev, int32_t B, void* stream, float* tokens_dev) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!z_dev || !x_dev || !tokens_dev) return fail(VT_ERR_ARG, "null device pointer");
    if (m->vb) {
        std::string err;
        rc = vb::stem(m->vb, z_dev, x_dev, B, static_cast<hipStream_t>(stream), tokens_dev, &err);
        return rc ? fail(rc, err) : VT_OK;
    }
    return run_stem(m, z_dev, x_dev, B, static_cast<hipStream_t>(stream), tokens_dev);
}

int vt_blocks(vt_model* m, const float* tokens_dev, int32_t B, int32_t nblocks, void* stream, float* feat_dev,
              float* resid_dev) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!tokens_dev) return fail(VT_ERR_ARG, "null device pointer");
    if (m->vb) {
        std::string err;
        rc = vb::blocks(m->vb, tokens_dev, B, nblocks, static_cast<hipStream_t>(stream), feat_dev, resid_dev, &err);
        return rc ? fail(rc, err) : VT_OK;
    }
    return run_blocks(m, tokens_dev, B, nblocks, static_cast<hipStream_t>(stream), feat_dev ? feat_dev : m->feat.p,
                      resid_dev);
}

int vt_head(vt_model* m, const float* feat_dev, int32_t B, void* stream, const vt_outputs* out) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!feat_dev) return fail(VT_ERR_ARG, "null device pointer");
    if (m->vb) return run_head_vitb(m, feat_dev, B, static_cast<hipStream_t>(stream), out);
    return run_head(m, feat_dev, B, static_cast<hipStream_t>(stream), out);
}

int vt_forward(vt_model* m, const float* z_dev, const float* x_dev, int32_t B, void* stream, const vt_outputs* out) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!x_dev) return fail(VT_ERR_ARG, "null device pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!z_dev) {   // cached template (vt_set_template): stem on the search crop only, block 0 loads the template's q / k / v
        if (m->vb) return fail(VT_ERR_ARG, "the template cache is implemented for the vit_48 path only");
        if (m->tmpl_frames < B)
            return fail(VT_ERR_STATE, "vt_forward with a null template needs vt_set_template for at least " + std::to_string(B) + " frames first");
        if (m->tmpl_form_batch != m->form_batch)
            return fail(VT_ERR_STATE, "the template cache was written under another form batch: call vt_set_template again after vt_set_form_batch");
        if ((rc = run_stem(m, nullptr, x_dev, B, st, m->tokens_c.p, 0, 1))) return rc;
        if ((rc = run_blocks(m, m->tokens_c.p, B, -1, st, m->feat.p, nullptr, 2))) return rc;
        return run_head(m, m->feat.p, B, st, out);
    }
    if (m->vb) {
        std::string err;
        if ((rc = vb::stem(m->vb, z_dev, x_dev, B, st, nullptr, &err))) return fail(rc, err);
        if ((rc = vb::blocks(m->vb, nullptr, B, -1, st, nullptr, nullptr, &err))) return fail(rc, err);
        return run_head_vitb(m, nullptr, B, st, out);
    }
    if ((rc = run_stem(m, z_dev, x_dev, B, st, m->tokens.p))) return rc;
    if ((rc = run_blocks(m, m->tokens.p, B, -1, st, m->feat.p, nullptr))) return rc;
    return run_head(m, m->feat.p, B, st, out);
}

int vt_set_template(vt_model* m, const float* z_dev, int32_t B, void* stream) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!z_dev) return fail(VT_ERR_ARG, "null device pointer");
    if (m->vb) return fail(VT_ERR_ARG, "the template cache is implemented for the vit_48 path only");
    hipStream_t st = static_cast<hipStream_t>(stream);
    m->tmpl_frames = 0;
    // template token rows (stem(z) + pos_embed_z) stay in the cached step's own token matrix (uncached steps use another) ...
    if ((rc = run_stem(m, z_dev, nullptr, B, st, m->tokens_c.p, 0, 2))) return rc;
    // ... and block 0's LN1 + qkv of those rows goes to the cache.  One block over the whole token matrix: the search
    // rows hold whatever the last cached frame left (per-token work, nothing of theirs is stored); the outputs are scratch.
    if ((rc = run_blocks(m, m->tokens_c.p, B, 1, st, m->feat.p, nullptr, 1))) return rc;
    m->tmpl_frames = B;
    m->tmpl_form_batch = m->form_batch;
    return VT_OK;
}

int vt_cal_bbox(vt_model* m, const float* score_dev, const float* size_dev, const float* offset_dev, int32_t B,
                void* stream, float* bbox_dev, float* max_score_dev) {
    if (!m || !score_dev || !size_dev || !offset_dev || !bbox_dev || B < 1) return fail(VT_ERR_ARG, "bad argument");
    return run_decode(m, static_cast<hipStream_t>(stream), score_dev, size_dev, offset_dev, nullptr, B, bbox_dev, nullptr,
                      max_score_dev);
}

int vt_crop(vt_model* m, const uint8_t* frames_dev, int32_t H, int32_t W, const double* states_dev, double factor,
            int32_t out_size, const float* mean3, const float* std3, int32_t B, void* stream, float* crops_dev,
            double* resize_factor_dev) {
    if (!m || !frames_dev || !states_dev || !crops_dev || !resize_factor_dev || !mean3 || !std3)
        return fail(VT_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1 || out_size < 1 || !(factor > 0.0)) return fail(VT_ERR_ARG, "bad crop arguments");
    bool crop_bytes = false;
    if (int rcs = crop_selftest(&crop_bytes)) return rcs;      // a table look-up after the device's first call
    launch_crop(crop_bytes, frames_dev, H, W, states_dev, factor, out_size, mean3, std3, B, static_cast<hipStream_t>(stream), crops_dev,
                resize_factor_dev);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int vt_crop_u8(vt_model* m, const uint8_t* frames_dev, int32_t H, int32_t W, const double* states_dev, double factor,
               int32_t out_size, int32_t B, void* stream, uint8_t* patch_dev, double* resize_factor_dev) {
    if (!m || !frames_dev || !states_dev || !patch_dev || !resize_factor_dev) return fail(VT_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1 || out_size < 1 || !(factor > 0.0)) return fail(VT_ERR_ARG, "bad crop arguments");
    bool crop_bytes = false;
    if (int rcs = crop_selftest(&crop_bytes)) return rcs;
    launch_crop(crop_bytes, frames_dev, H, W, states_dev, factor, out_size, nullptr, nullptr, B, static_cast<hipStream_t>(stream),
                reinterpret_cast<float*>(patch_dev), resize_factor_dev, true);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int vt_set_normalization(vt_model* m, const float* mean3, const float* std3) {
    if (!m || !mean3 || !std3) return fail(VT_ERR_ARG, "null argument");
    if (m->vb) return fail(VT_ERR_ARG, "uint8 patches are implemented for the vit_48 path only");
    for (int c = 0; c < 3; ++c)
        if (!(std3[c] > 0.f) || !std::isfinite(mean3[c]) || !std::isfinite(std3[c])) return fail(VT_ERR_ARG, "bad mean / std");
    if (m->graphs_captured > 0 && (std::memcmp(mean3, m->norm_mean, 12) != 0 || std::memcmp(std3, m->norm_std, 12) != 0))
        return fail(VT_ERR_STATE, "captured graphs read the folded layer-1 weights: set the normalisation before capturing");
    if (!m->weights_loaded || m->generic) {      // remembered: vt_load_weights folds with these; the shape-generic stem takes them as kernel arguments (not in captured graphs: see above)
        std::memcpy(m->norm_mean, mean3, 12);
        std::memcpy(m->norm_std, std3, 12);
        return VT_OK;
    }
    HIP_TRY(hipDeviceSynchronize());      // no step may be reading the image that is about to be replaced
    return fold_w1u(m, mean3, std3);
}

int vt_patch_u8_supported(const vt_model* m, int32_t B) {
    if (!m) return 0;
    return stem_takes_u8(m, B) ? 1 : 0;
}

int vt_set_open_loop(vt_model* m, int32_t on) {
    if (!m) return fail(VT_ERR_ARG, "null model");
    m->open_loop = on ? 1 : 0;
    return VT_OK;
}

int vt_crop_form(void) {
    bool bytes = false;
    if (crop_selftest(&bytes)) return -1;
    return bytes ? 2 : 1;
}

// Does (mean3, std3) equal the normalisation folded into the uint8 form of layer 1?
static bool same_norm(const vt_model* m, const float* mean3, const float* std3) {
    return std::memcmp(mean3, m->norm_mean, 12) == 0 && std::memcmp(std3, m->norm_std, 12) == 0;
}

int vt_stem_u8(vt_model* m, const uint8_t* x_patch_dev, int32_t B, void* stream, float* tokens_dev) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!x_patch_dev || !tokens_dev) return fail(VT_ERR_ARG, "null device pointer");
    if (m->vb) return fail(VT_ERR_ARG, "uint8 patches are implemented for the vit_48 path only");
    return run_stem(m, nullptr, reinterpret_cast<const float*>(x_patch_dev), B, static_cast<hipStream_t>(stream), tokens_dev, 0, 1, true);
}

int vt_forward_u8(vt_model* m, const float* z_dev, const uint8_t* x_patch_dev, int32_t B, void* stream, const vt_outputs* out) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!x_patch_dev) return fail(VT_ERR_ARG, "null device pointer");
    if (m->vb) return fail(VT_ERR_ARG, "uint8 patches are implemented for the vit_48 path only");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* const xu = reinterpret_cast<const float*>(x_patch_dev);
    if (!z_dev) {     // cached template: the tracker step's network part
        if (m->tmpl_frames < B)
            return fail(VT_ERR_STATE, "vt_forward_u8 with a null template needs vt_set_template for at least " + std::to_string(B) + " frames first");
        if (m->tmpl_form_batch != m->form_batch)
            return fail(VT_ERR_STATE, "the template cache was written under another form batch: call vt_set_template again after vt_set_form_batch");
        if ((rc = run_stem(m, nullptr, xu, B, st, m->tokens_c.p, 0, 1, true))) return rc;
        if ((rc = run_blocks(m, m->tokens_c.p, B, -1, st, m->feat.p, nullptr, 2))) return rc;
        return run_head(m, m->feat.p, B, st, out);
    }
    // a template given with the call: its rows from the fp32 crop, the search rows from the patch, then the uncached blocks
    if ((rc = run_stem(m, z_dev, nullptr, B, st, m->tokens.p, 0, 2))) return rc;
    if ((rc = run_stem(m, nullptr, xu, B, st, m->tokens.p, 0, 1, true))) return rc;
    if ((rc = run_blocks(m, m->tokens.p, B, -1, st, m->feat.p, nullptr))) return rc;
    return run_head(m, m->feat.p, B, st, out);
}

int vt_update_state(vt_model* m, const float* hann_boxes_dev, const double* resize_factor_dev, int32_t search_size,
                    int32_t H, int32_t W, int32_t margin, int32_t B, void* stream, double* states_dev) {
    if (!m || !hann_boxes_dev || !resize_factor_dev || !states_dev || B < 1) return fail(VT_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(vtt::update_state_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                       hann_boxes_dev, resize_factor_dev, search_size, H, W, margin, B, states_dev, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int vt_update_state_record(vt_model* m, const float* hann_boxes_dev, const float* conf_dev, const double* resize_factor_dev,
                           int32_t search_size, int32_t H, int32_t W, int32_t margin, int32_t B, void* stream, double* states_dev,
                           double* record) {
    if (!m || !hann_boxes_dev || !resize_factor_dev || !states_dev || !record || B < 1) return fail(VT_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(vtt::update_state_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                       hann_boxes_dev, resize_factor_dev, search_size, H, W, margin, B, states_dev, conf_dev, record);
    HIP_TRY(hipGetLastError());
    return VT_OK;
}

int vt_track_step(vt_model* m, const uint8_t* frames, int32_t H, int32_t W, double* states_dev, double factor, const float* mean3,
                  const float* std3, int32_t B, void* stream, float* crops_dev, double* resize_factor_dev, const vt_outputs* out,
                  int32_t margin, double* record) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (m->vb) return fail(VT_ERR_ARG, "vt_track_step is implemented for the vit_48 path only");
    if (m->tmpl_frames < B)
        return fail(VT_ERR_STATE, "vt_track_step needs vt_set_template for at least " + std::to_string(B) + " frames first");
    if (m->tmpl_form_batch != m->form_batch)
        return fail(VT_ERR_STATE, "the template cache was written under another form batch: call vt_set_template again after vt_set_form_batch");
    if (!states_dev || !mean3 || !std3 || !crops_dev) return fail(VT_ERR_ARG, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // The crop reaches the stem as the uint8 patch sample_target returns (a quarter of the fp32 crop's bytes, written and read once)
    // whenever the stem form of this batch reads patches and (mean3, std3) is the normalisation folded into its layer 1; else as the
    // fp32 crop of vt_crop.  Either way crops_dev is the workspace: the patch occupies its first B * S * S * 3 bytes.
    const bool u8 = m->track_u8 != 0 && stem_takes_u8(m, B) && same_norm(m, mean3, std3);
    if (u8) {
        if ((rc = vt_crop_u8(m, frames, H, W, states_dev, factor, m->cfg.search_size, B, stream, reinterpret_cast<uint8_t*>(crops_dev), resize_factor_dev))) return rc;
    } else if ((rc = vt_crop(m, frames, H, W, states_dev, factor, m->cfg.search_size, mean3, std3, B, stream, crops_dev, resize_factor_dev))) return rc;
    if ((rc = run_stem(m, nullptr, crops_dev, B, st, m->tokens_c.p, 0, 1, u8))) return rc;
    if ((rc = run_blocks(m, m->tokens_c.p, B, -1, st, m->feat.p, nullptr, 2))) return rc;
    const TrackTail tail{resize_factor_dev, states_dev, record, m->cfg.search_size, H, W, margin, m->open_loop};
    return run_head(m, m->feat.p, B, st, out, 0, &tail);
}

// One slice [f0, f0 + nb) of a batch through the whole step, on stream st.
static int forward_slice(vt_model* m, const float* z, const float* x, size_t f0, int nb, hipStream_t st,
                         const vt_outputs* out, int Btot, int nch) {
    const size_t Tz = m->cfg.template_size, Tx = m->cfg.search_size;
    if (m->chain_delay_us > 0 && f0 > 0) {
        const int c = (int)((f0 * (size_t)nch + (size_t)Btot - 1) / (size_t)Btot);      // chain index of this slice
        hipLaunchKernelGGL(chain_delay_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)c * m->chain_delay_us * 100ull);
    }
    if (m->vb) {   // ViT-Base: the chains' persistent GEMMs split the CUs between them
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
        const vb::Slice sl{f0, Btot, m->chain_cus > 0 ? m->chain_cus : (m->graph_chains == 0 ? ncu : ncu / nch)};
        std::string err;
        int rc;
        if ((rc = vb::stem(m->vb, z + f0 * 3 * Tz * Tz, x + f0 * 3 * Tx * Tx, nb, st, nullptr, &err, &sl))) return fail(rc, err);
        if ((rc = vb::blocks(m->vb, nullptr, nb, -1, st, nullptr, nullptr, &err, &sl))) return fail(rc, err);
        return run_head_vitb(m, nullptr, nb, st, out, &sl);
    }
    float* tok = m->tokens.p + f0 * m->L * 48;
    float* feat = m->feat.p + f0 * m->len_x * 48;
    int rc;
    if ((rc = run_stem(m, z + f0 * 3 * Tz * Tz, x + f0 * 3 * Tx * Tx, nb, st, tok, f0))) return rc;
    if ((rc = run_blocks(m, tok, nb, -1, st, feat, nullptr, 0, f0))) return rc;
    return run_head(m, feat, nb, st, out, f0);
}

int vt_graph_capture_steps(vt_model* m, int32_t nsteps, const float* const* z_dev, const float* const* x_dev, int32_t B,
                           const vt_outputs* out, vt_graph** g) {
    int rc = check_ready(m, B);
    if (rc) return rc;
    if (!g) return fail(VT_ERR_ARG, "null graph out");
    if (nsteps < 1 || nsteps > 64 || !x_dev) return fail(VT_ERR_ARG, "vt_graph_capture_steps: 1..64 steps, x_dev must not be null");
    // One step may be captured as NCH independent chains over frame slices (fork / join with events):
    // the kernels of one slice can then overlap the kernels of the others.  Measured slower with the
    // one-workgroup-per-frame kernels (large LDS: no two workgroups share a CU): 107.7 -> 132 us with 2 chains.
    int nch = m->graph_chains;   // vit_48: default 1; ViT-Base: 0 = auto (two chains from 64 frames up, create_vitb)
    if (nch == 0) nch = (m->vb && B >= 64) ? 2 : 1;
    nch = (nsteps > 1 || !z_dev || !z_dev[0]) ? 1 : std::max(1, std::min({nch, 4, (int)B}));
    if (m->generic) nch = 1;      // the shape-generic kernels share ONE set of scratch buffers (g_a, g_b, g_x, ...): concurrent chains would race on them
    vt_graph* vg = new vt_graph();
    hipError_t e = hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { delete vg; return fail(VT_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e)); }
    rc = VT_OK;
    if (nch == 1) {
        for (int i = 0; i < nsteps && !rc; ++i)
            rc = vt_forward(m, z_dev ? z_dev[i] : nullptr, x_dev[i], B, m->cap_stream, out ? &out[i] : nullptr);
    } else {
        if (hipEventRecord(m->fork_ev, m->cap_stream) != hipSuccess) rc = fail(VT_ERR_HIP, "hipEventRecord(fork)");
        for (int c = 1; c < nch && !rc; ++c)
            if (hipStreamWaitEvent(m->side_stream[c - 1], m->fork_ev, 0) != hipSuccess) rc = fail(VT_ERR_HIP, "hipStreamWaitEvent(fork)");
        for (int c = 0; c < nch && !rc; ++c) {
            const size_t f0 = (size_t)B * c / nch, f1 = (size_t)B * (c + 1) / nch;
            rc = forward_slice(m, z_dev[0], x_dev[0], f0, (int)(f1 - f0), c == 0 ? m->cap_stream : m->side_stream[c - 1], out, B, nch);
        }
        for (int c = 1; c < nch; ++c) {   // always join, even after an error, so the capture can end
            (void)hipEventRecord(m->join_ev[c - 1], m->side_stream[c - 1]);
            (void)hipStreamWaitEvent(m->cap_stream, m->join_ev[c - 1], 0);
        }
    }
    e = hipStreamEndCapture(m->cap_stream, &vg->graph);
    if (rc) { if (vg->graph) (void)hipGraphDestroy(vg->graph); delete vg; return rc; }
    if (e != hipSuccess) { delete vg; return fail(VT_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
    e = hipGraphInstantiate(&vg->exec, vg->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(vg->graph); delete vg; return fail(VT_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    vg->owner = m;
    m->graphs.push_back(vg);
    *g = vg;
    ++m->graphs_captured;
    return VT_OK;
}

int vt_graph_capture(vt_model* m, const float* z_dev, const float* x_dev, int32_t B, const vt_outputs* out, vt_graph** g) {
    return vt_graph_capture_steps(m, 1, &z_dev, &x_dev, B, out, g);
}

int vt_graph_launch(vt_graph* g, void* stream) {
    if (!g || !g->exec) return fail(VT_ERR_ARG, "null graph");
    HIP_TRY(hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
    return VT_OK;
}

void vt_graph_destroy(vt_graph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->owner) {      // the last graph gone: vt_set_form_batch / vt_set_normalization are free again
        auto& v = g->owner->graphs;
        v.erase(std::remove(v.begin(), v.end(), g), v.end());
        if (g->owner->graphs_captured > 0) --g->owner->graphs_captured;
    }
    delete g;
}

int vt_debug_stamps(vt_model* m, int32_t B, unsigned long long* host_out) {
    // Development aid: copies the in-kernel s_memtime stamps of the last launch (block kernel: [B][waves][64];
    // stem_fused / stem_pipe: [B][16][32]); the buffer holds B * 8 * 64 values.
    if (!m || !m->dbg_stamps || !host_out) return fail(VT_ERR_STATE, "stamps are off (set VT_DBG_STAMPS=1 before vt_create)");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, m->dbg_stamps, (size_t)B * 8 * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return VT_OK;
}

int vt_probe_clock(int32_t iters, int32_t waves_per_simd, double* mhz, double* cycles_per_mfma, double* wall_us) {
    // Development probe (not on the hot path; synchronises): sustained shader clock under a dense
    // f32-MFMA loop and the cycles one SIMD spends per v_mfma_f32_16x16x4_f32.
    if (iters < 1 || waves_per_simd < 1 || waves_per_simd > 4) return fail(VT_ERR_ARG, "bad probe arguments");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount * waves_per_simd;
    float *src = nullptr, *sink = nullptr;
    unsigned long long* st = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&src), 512 * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sink), (size_t)nwg * 256 * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st), (size_t)nwg * 8 * sizeof(unsigned long long)));
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0.25f + 0.001f * (float)((i * 37) % 101);
    HIP_TRY(hipMemcpy(src, h.data(), 512 * sizeof(float), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe_kernel, dim3(nwg), dim3(256), 0, nullptr, src, iters, st, sink);   // warm-up
    HIP_TRY(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(probe_kernel, dim3(nwg), dim3(256), 0, nullptr, src, iters, st, sink);
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hs((size_t)nwg * 8);
    HIP_TRY(hipMemcpy(hs.data(), st, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> f, c;
    for (size_t k = 0; k < hs.size(); k += 2) {
        f.push_back((double)hs[k] / (double)hs[k + 1] * 100.0);
        c.push_back((double)hs[k] / ((double)iters * 8.0) / waves_per_simd);
    }
    std::sort(f.begin(), f.end());
    std::sort(c.begin(), c.end());
    if (mhz) *mhz = f[f.size() / 2];
    if (cycles_per_mfma) *cycles_per_mfma = c[c.size() / 2];
    if (wall_us) *wall_us = ms * 1e3;
    (void)hipFree(src); (void)hipFree(sink); (void)hipFree(st);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return VT_OK;
}

int vt_selftest_mfma(void* stream) {
    // exact small integers; B is asymmetric so a transposed read or write cannot pass
    std::vector<float> A(256), Bm(256), D(384, -1.f), ref(256, 0.f);
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 16; ++k) {
            A[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);
            Bm[i * 16 + k] = (float)((i * 11 + k * 2) % 9 - 4);   // B[k=i][j=k]
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 16; ++k) ref[i * 16 + j] += A[i * 16 + k] * Bm[k * 16 + j];
    float *dA = nullptr, *dB = nullptr, *dD = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dA), 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dB), 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dD), 1536));
    HIP_TRY(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB, Bm.data(), 1024, hipMemcpyHostToDevice));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, st, dA, dB, dD);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(D.data(), dD, 1536, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += (D[i] != ref[i]);
    if (bad) return fail(VT_ERR_STATE, "MFMA lane map differs from the assumed one in " + std::to_string(bad) + " / 256 elements");
    for (int l = 0; l < 64; ++l) bad += (D[256 + l] != 15.f * ((l & 15) + 1)) + (D[320 + l] != 8.f * ((l & 15) + 1));
    if (bad) return fail(VT_ERR_STATE, "v_permlane16/32_swap lane map differs from the assumed one (" + std::to_string(bad) + " / 128)");
    return VT_OK;
}

}  // extern "C"
