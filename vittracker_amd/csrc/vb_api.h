// vb_api.h -- internal interface between the C-ABI layer (vittrack.hip) and the ViT-Base runtime (vitb.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <utility>

#include "../../include/vittrack.h"

struct VbModel;

namespace vb {
using TensorMap = std::map<std::string, std::pair<const float*, int64_t>>;

// A frame slice [f0, f0 + B) of a batch of Btot frames, for graph chains that run slices concurrently (vt_graph_capture_steps,
// VT_GRAPH_CHAINS): every workspace is addressed from frame f0, tower-major buffers keep the whole batch's stride, and the
// persistent GEMMs launch at most `cus` workgroups (0 = one per CU) so that two chains share the chip.  z / x and the output
// pointers handed to stem / head are the slice's own (already offset by the caller).
struct Slice {
    size_t f0 = 0;
    int Btot = 0;
    int cus = 0;
};

// Every function returns VT_OK or a VT_ERR_* code and fills *err.
int create(const vt_config* cfg, VbModel** out, std::string* err);
void destroy(VbModel* m);
int load_weights(VbModel* m, const TensorMap& tm, std::string* err);
// patch_embed(z), patch_embed(x), += pos_embed, cat((z, x))  -> the model's f32 residual stream; tokens_out (optional): a copy
int stem(VbModel* m, const float* z, const float* x, int B, hipStream_t st, float* tokens_out, std::string* err, const Slice* sl = nullptr);
// blocks[0..nblocks) on the residual stream (tokens_in: optional replacement, copied in first), then the final norm:
// the search rows go to the head's input map (and to feat_out as f32 (B,Lx,C), optional); resid_out optional copy
int blocks(VbModel* m, const float* tokens_in, int B, int nblocks, hipStream_t st, float* feat_out, float* resid_out, std::string* err,
           const Slice* sl = nullptr);
// CenterPredictor towers + conv5 + sigmoid/clamp on the head's input map (feat_in: optional f32 (B,Lx,C) replacement)
int head(VbModel* m, const float* feat_in, int B, hipStream_t st, float* score, float* size, float* offset, std::string* err,
         const Slice* sl = nullptr);
}  // namespace vb
