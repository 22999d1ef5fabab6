/* vittrack.h -- C ABI of the MI355X-native vit_dist inference path (libvittrack_hip.so).
 *
 * Drop-in boundary for ONE hot path of lpylpy0514/VitTracker: the per-frame forward of the
 * `vit_dist` tracker.  Every entry point names the reference interface it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only: no torch / Python
 * types cross this boundary.  Device pointers are HIP device addresses (fp32, contiguous);
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *
 * Conventions
 *   - Every function returns 0 on success or a negative VT_ERR_* code; vt_last_error() gives a
 *     human-readable message for the calling thread (the reference raises Python exceptions,
 *     lib/test/evaluation/running.py:138-142 swallows them per sequence).
 *   - No entry point that takes a `stream` synchronises with the host or allocates memory, so all
 *     of them may be captured into a hipGraph (the reference syncs once per frame in `.tolist()`,
 *     lib/test/tracker/vit_dist.py:108-109; here the caller decides when to read results back).
 *   - A model is not thread-safe and owns ONE set of workspaces: calls on the same model must not overlap each other (issue
 *     them on one stream, or order the streams).  Independent models -- each with its own workspaces, weights replica and graphs --
 *     may run concurrently on different streams of the same GPU: two models stepped alternately on two streams are how many
 *     independent sequences are served fastest (the reference runs one tracker instance per worker process,
 *     lib/test/evaluation/running.py:105-112).
 */
#ifndef VITTRACK_H
#define VITTRACK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_OK 0
#define VT_ERR_ARG (-1)         /* bad argument / unsupported geometry */
#define VT_ERR_HIP (-2)         /* a HIP runtime call failed */
#define VT_ERR_STATE (-3)       /* weights not loaded, batch larger than max_batch, ... */
#define VT_ERR_MISSING_KEY (-4) /* vt_load_weights: a required tensor is absent or mis-shaped */

typedef struct vt_model vt_model;
typedef struct vt_graph vt_graph;

/* Replaces the cfg fields read by build_ostrack_dist (lib/models/vit_dist/vit_dist.py:159-164)
 * and build_box_head CENTER (lib/models/layers/head.py:352-359).
 *
 * SUPPORTED SHAPES.  The reference builds from any cfg (lib/models/vit_dist/vit_dist.py:159-198; lib/utils/ce_utils.py:22-32 lists
 * template feature sizes 8 / 12 / 7 / 14).  vt_create accepts:
 *     stride 16, depth 1..12, (template, search) = ANY multiples of 16 in [16, 512], and (round 6) ANY widths:
 *         channels a multiple of 8 (the stem's widths are C/8, C/4, C/2, C), heads dividing channels (head dimension <= 256),
 *         head_channels a multiple of 8 (the towers' widths are W, W/2, W/4, W/8)
 *     The shipped widths (channels 48, heads 1, head_channels 32) at (64, 128) and (128, 256) run the tuned kernels (every form of
 *     MFMA, LDS-resident maps, hipGraph-sized forms).  Every other geometry -- e.g. (112, 224), (192, 384); token
 *     counts need not be multiples of 16 -- and every other width -- e.g. 64 / 2 / 64 -- runs the shape-generic kernels of vt_generic.h:
 *     plain fp32, one thread per output value, reference-implementation speed, the same outputs (held to the reference's own outputs at
 *     two other width triples: tests/golden/make_golden_cfg.py) and the whole ABI (stages, template cache, graphs, vt_crop / vt_track_step,
 *     uint8 patches).
 *     channels 768, heads 12, head_channels 256, stride 16, (template, search) = (128, 256), depth 12                   -- ViT-Base
 * and rejects everything else with VT_ERR_ARG and a message naming these.  The tuned kernels are specialised on their tile
 * counts (token tiles per frame, feature chunks, map sides are template parameters -- that is where their register blocking comes
 * from), so widths and crop sizes are NOT run-time parameters of THOSE kernels. */
typedef struct vt_config {
    int32_t template_size; /* DATA.TEMPLATE.SIZE  (128; 64 for G128) */
    int32_t search_size;   /* DATA.SEARCH.SIZE    (256; 128 for G128) */
    int32_t channels;      /* MODEL.BACKBONE.CHANNELS (48)  */
    int32_t heads;         /* MODEL.BACKBONE.HEADS    (1)   */
    int32_t depth;         /* build_ostrack_dist(depth=3)   */
    int32_t head_channels; /* MODEL.HEAD.NUM_CHANNELS (32)  */
    int32_t stride;        /* MODEL.BACKBONE.STRIDE   (16)  */
    int32_t max_batch;     /* workspace is sized for this many frames per call */
} vt_config;

/* One named host tensor of the reference's ckpt['net'] state dict. */
typedef struct vt_tensor {
    const char* name;  /* e.g. "blocks.0.attn.qkv.weight" */
    const float* data; /* host pointer, fp32, contiguous, PyTorch layout */
    int64_t numel;
} vt_tensor;

/* Device output buffers of one forward; any pointer may be NULL (that output is then kept in
 * the model's own scratch).  Shapes follow OstrackDist.forward_head's dict
 * (lib/models/vit_dist/vit_dist.py:149-152) plus the tracker's Hann-windowed decode. */
typedef struct vt_outputs {
    float* score_map;  /* (B,1,F,F)  clamp(sigmoid)            head.py:201 */
    float* size_map;   /* (B,2,F,F)  clamp(sigmoid)            head.py:201 */
    float* offset_map; /* (B,2,F,F)  raw                       head.py:201 */
    float* pred_boxes; /* (B,4) cx,cy,w,h from the raw score   head.py:136,142-160 */
    float* hann_boxes; /* (B,4) cx,cy,w,h from hann2d*score    lib/test/tracker/vit_dist.py:104-105 */
    float* conf;       /* (B,)  max of the raw score map       lib/test/tracker/vit_dist.py:148 */
} vt_outputs;

const char* vt_last_error(void);
const char* vt_version(void);

/* build_ostrack_dist(cfg) + .cuda() on the current device (vit_dist.py:159-164;
 * lib/test/tracker/vit_dist.py:24-28).  Supported: see SUPPORTED SHAPES above (the shipped vit_48_h32 on tuned kernels -- fp32, or
 * f16 contractions in libvittrack_hip_f16.so --, any other widths / stride-16 geometry on the shape-generic fp32 kernels), and
 * channels 768, heads 12, head_channels 256, (128,256): the OSTrack-256 ViT-Base
 * (lib/models/ostrack/ostrack.py:164-286 with lib/models/ostrack/vit.py:94-139; bf16 contractions). */
int vt_create(const vt_config* cfg, vt_model** out);
void vt_destroy(vt_model* m);

/* network.load_state_dict(ckpt['net'], strict=False) (lib/test/tracker/vit_dist.py:25):
 * unknown names (e.g. training-only convs.*) are ignored, a missing required name is an error.
 * BatchNorm is folded into the convs here exactly as Conv2d_BN.fuse does
 * (lib/models/vit_dist/vit_dist.py:22-33), weights are packed into MFMA operand images and
 * uploaded. */
int vt_load_weights(vt_model* m, const vt_tensor* tensors, int32_t n);

/* Overrides the motion window (default: hann2d(F), lib/test/utils/hann.py:6-16, computed at
 * vt_create).  host_window: F*F floats. */
int vt_set_window(vt_model* m, const float* host_window);

/* OstrackDist.forward(z, x) + CenterPredictor.forward + the tracker's windowed cal_bbox
 * (vit_dist.py:77-100,122-153; head.py:130-160; lib/test/tracker/vit_dist.py:103-105).
 * z_dev (B,3,Tz,Tz), x_dev (B,3,Tx,Tx) NCHW fp32 on the device.
 * z_dev may be NULL after vt_set_template: the cached template is used (same result, bit for bit). */
int vt_forward(vt_model* m, const float* z_dev, const float* x_dev, int32_t B, void* stream,
               const vt_outputs* out);

/* Exact template cache (BASELINE config 5).  The tracker stores the template crop at initialize() and feeds the SAME
 * tensor to every forward (lib/test/tracker/vit_dist.py:57-60,87-88); patch_embed(z) + pos_embed_z and block 0's
 * LayerNorm-1 + qkv of those rows are per-token functions of it (vit_dist.py:78-89), hence frame-invariant.  This
 * computes them once for B sequences and keeps them in the model; later vt_forward / vt_graph_capture calls with
 * z_dev == NULL skip that work.  Deeper layers are NOT cached: from block 0's attention on, template tokens depend on
 * the current search tokens. */
int vt_set_template(vt_model* m, const float* z_dev, int32_t B, void* stream);

/* --- the three stages of vt_forward, individually (parity tests, profiling) ----------------- */
/* patch_embed(z), patch_embed(x), += pos_embed, cat (vit_dist.py:78-84) -> tokens (B,L,C). */
int vt_stem(vt_model* m, const float* z_dev, const float* x_dev, int32_t B, void* stream,
            float* tokens_dev);
/* blocks[0..nblocks) then self.norm (vit_dist.py:88-94).  nblocks<0 = all.  feat_dev (B,Lx,C) =
 * normalised search tokens (the head's input, vit_dist.py:126); resid_dev (B,L,C), optional =
 * the un-normalised residual stream after the last executed block. */
int vt_blocks(vt_model* m, const float* tokens_dev, int32_t B, int32_t nblocks, void* stream,
              float* feat_dev, float* resid_dev);
/* forward_head + CenterPredictor (vit_dist.py:122-153; head.py:130-201) on (B,Lx,C) tokens. */
int vt_head(vt_model* m, const float* feat_dev, int32_t B, void* stream, const vt_outputs* out);

/* box_head.cal_bbox(score, size, offset, return_score=True) (head.py:142-160) on arbitrary
 * device maps: bbox_dev (B,4), max_score_dev (B,) optional. First maximum wins ties. */
int vt_cal_bbox(vt_model* m, const float* score_dev, const float* size_dev, const float* offset_dev,
                int32_t B, void* stream, float* bbox_dev, float* max_score_dev);

/* --- the steps either side of the network in track(), on the device ---- */
/* sample_target(image, state, factor, output_sz) + Preprocessor.process
 * (lib/train/data/processing_utils.py:12-79; lib/test/tracker/data_utils.py:11-17) for B sequences:
 * frames_dev (B,H,W,3) uint8, states_dev (B,4) double [x,y,w,h] -> crops_dev (B,3,T,T) fp32
 * normalised with mean3/std3 (host, 3 floats each) and resize_factor_dev (B) double = T / crop_sz.
 * Crop geometry in double with Python's round-half-even; resize = OpenCV INTER_LINEAR uint8
 * fixed-point scheme. */
int vt_crop(vt_model* m, const uint8_t* frames_dev, int32_t H, int32_t W, const double* states_dev, double factor,
            int32_t out_size, const float* mean3, const float* std3, int32_t B, void* stream, float* crops_dev,
            double* resize_factor_dev);
/* sample_target ALONE (lib/train/data/processing_utils.py:12-79): patch_dev (B,T,T,3) uint8 is the array sample_target returns --
 * HWC, before Preprocessor.process -- byte for byte (same geometry and fixed-point resize as vt_crop); resize_factor_dev as vt_crop.
 * A too-small box (processing_utils.py:33-34 raises) writes zeros and a NaN resize factor. */
int vt_crop_u8(vt_model* m, const uint8_t* frames_dev, int32_t H, int32_t W, const double* states_dev, double factor,
               int32_t out_size, int32_t B, void* stream, uint8_t* patch_dev, double* resize_factor_dev);
/* Preprocessor.__init__'s mean / std (lib/test/tracker/data_utils.py:8-9) for the uint8 entry points below (default: the ImageNet
 * values the reference hard-codes).  Preprocessor.process is affine per channel and the stem's first conv is linear, so the
 * normalisation is folded into that layer's weights in fp64 (here and at vt_load_weights); the conv's zero padding becomes the
 * byte value that normalises to zero.  Synchronises the device; not capturable; VT_ERR_STATE once graphs have been captured. */
int vt_set_normalization(vt_model* m, const float* mean3, const float* std3);
/* Preprocessor.process + OstrackDist.forward on a uint8 search patch (lib/test/tracker/data_utils.py:11-17 +
 * lib/models/vit_dist/vit_dist.py:77-100): x_patch_dev (B,S,S,3) uint8 as vt_crop_u8 writes it; z_dev (B,3,Tz,Tz) fp32 as in
 * vt_forward, or NULL after vt_set_template.  Results agree with vt_forward on the normalised fp32 crop to fp32 rounding (the
 * reference rounds three times per input value, the folded layer once; tests: maps within 1e-5), not bit for bit.
 * vit_48 path (every stem form of the tuned geometries; the shape-generic kernels normalise per tap with the reference's own three
 * rounded operations and are bit-identical to vt_forward); VT_ERR_STATE under the diagnostic switches (vt_patch_u8_supported). */
int vt_forward_u8(vt_model* m, const float* z_dev, const uint8_t* x_patch_dev, int32_t B, void* stream, const vt_outputs* out);
/* The search rows of vt_stem from a uint8 patch: tokens_dev (B,L,C), rows [len_z, L) written, template rows untouched. */
int vt_stem_u8(vt_model* m, const uint8_t* x_patch_dev, int32_t B, void* stream, float* tokens_dev);
/* 1 when a batch of B runs a stem form that reads uint8 patches (every vit_48 form outside the diagnostic builds), else 0 (ViT-Base). */
int vt_patch_u8_supported(const vt_model* m, int32_t B);
/* Which crop kernel form the current device runs (decided once per device by a self test against a byte-load twin):
 * 1 = the 8-byte unaligned-window form, 2 = the byte-load form, negative = the self test could not run. */
int vt_crop_form(void);

/* The tail of Vit_dist.track (lib/test/tracker/vit_dist.py:107-111,150-156; clip_box,
 * lib/utils/box_ops.py:97-106): scale the windowed box back to image pixels, map it to the frame,
 * clip with `margin`, and overwrite states_dev (B,4) double in place.  No host sync: a sequence can
 * run as back-to-back graph replays with its state on the device. */
int vt_update_state(vt_model* m, const float* hann_boxes_dev, const double* resize_factor_dev, int32_t search_size,
                    int32_t H, int32_t W, int32_t margin, int32_t B, void* stream, double* states_dev);
/* The same, and what `track()` returns ({"target_bbox", "confidence"}, lib/test/tracker/vit_dist.py:146-148) written as a (B,5)
 * double record [x, y, w, h, max score] to `record`: device memory, or device-mapped pinned host memory -- then the frame step
 * ends without a copy kernel or a device -> host copy, the host reads the record after one stream synchronisation.
 * conf_dev may be NULL (confidence 0). */
int vt_update_state_record(vt_model* m, const float* hann_boxes_dev, const float* conf_dev, const double* resize_factor_dev,
                           int32_t search_size, int32_t H, int32_t W, int32_t margin, int32_t B, void* stream, double* states_dev,
                           double* record);

/* The whole per-frame step of Vit_dist.track() (lib/test/tracker/vit_dist.py:87-148) in ONE call: the crop of the search region
 * (search_size of the model's config; crops_dev = a (B,3,S,S) float workspace of the caller) -> the network on the cached
 * template (vt_set_template first) -> vt_update_state_record; with the small-batch head form the decode kernel runs the tail
 * itself (one launch less per step).  record may be NULL.  vit_48 path only.
 * Round 6: the crop reaches the stem as sample_target's uint8 patch -- the step is vt_crop_u8 -> vt_forward_u8(z = NULL) ->
 * vt_update_state_record, bit for bit, and crops_dev holds the (B,S,S,3) uint8 patch in its first bytes -- whenever
 * (mean3, std3) equal the model's normalisation (vt_set_normalization) and vt_patch_u8_supported(m, B); otherwise (and with
 * VT_TRACK_U8=0 in the environment at vt_create) it is vt_crop -> vt_forward(z = NULL) -> vt_update_state_record with the fp32 crop. */
int vt_track_step(vt_model* m, const uint8_t* frames, int32_t H, int32_t W, double* states_dev, double factor, const float* mean3,
                  const float* std3, int32_t B, void* stream, float* crops_dev, double* resize_factor_dev, const vt_outputs* out,
                  int32_t margin, double* record);
/* Open loop (on != 0): later vt_track_step calls -- and graphs captured from then on -- crop around states_dev, write the step's
 * box and confidence to `record`, and leave states_dev as it is: every frame is searched around boxes the CALLER provides (the
 * ground truth of an accuracy study; the held 30-90 px boxes of bench.py's tracker-step figures, which synthetic noise frames would
 * otherwise drive to the clip limits within a few frames).  The reference's track() is the closed loop (lib/test/tracker/
 * vit_dist.py:107-111 overwrites self.state), which stays the default; vt_update_state[_record] are not affected. */
int vt_set_open_loop(vt_model* m, int32_t on);

/* --- hipGraph: the whole track() device step captured once, replayed per frame -------------- */
int vt_graph_capture(vt_model* m, const float* z_dev, const float* x_dev, int32_t B,
                     const vt_outputs* out, vt_graph** g);
/* nsteps consecutive frames in ONE graph: step i is vt_forward(z_dev[i], x_dev[i], out[i]) (z_dev, or any z_dev[i], may be NULL
 * after vt_set_template; out may be NULL).  Steps run in order, so they may share buffers.  For callers that have the next
 * crops on the device already -- offline evaluation (lib/test/evaluation/running.py:56-102 reads whole sequences), or frames
 * pipelined a few deep: the runtime leaves ~7 us between two graph launches, 4 steps per graph recover ~5 % of a G128 / B=256
 * step (tools/graph_steps.py). */
int vt_graph_capture_steps(vt_model* m, int32_t nsteps, const float* const* z_dev, const float* const* x_dev, int32_t B,
                           const vt_outputs* out, vt_graph** g);
int vt_graph_launch(vt_graph* g, void* stream);
void vt_graph_destroy(vt_graph* g);

/* Kernel forms by group size.  Every stage has several kernel forms and picks one by the batch of the call (the
 * one-workgroup-per-frame forms once the batch fills the chip, multi-workgroup forms below); two forms of a stage agree to fp32
 * rounding (~4e-6), not bit for bit.  A caller that steps a group of N sequences as several models of N / k sequences each (one per
 * stream or per GPU: the reference shards sequences over worker processes, lib/test/evaluation/running.py:105-112) sets n = N on every
 * one of them: each shard then runs the forms the whole group would run, and a sequence's results do not depend on how the group is
 * sharded.  n = 0 (default): forms by each call's own batch.  A call with a batch larger than n uses its own batch.
 * ORDER: the value applies to LATER calls only.  Set it before vt_set_template and before any vt_graph_capture[_steps]: the template
 * cache holds the operands of the form it was written under (vt_forward(z = NULL) / vt_track_step return VT_ERR_STATE until
 * vt_set_template has run again under the new value), and captured graphs keep the forms of their capture (changing the value on a
 * model that has LIVE captured graphs returns VT_ERR_STATE; once every graph of the model has been destroyed the value is free again). */
int vt_set_form_batch(vt_model* m, int32_t n);

/* Geometry / workspace queries (host side of build_box_head: feat_sz etc.). */
int vt_query(const vt_model* m, int32_t* len_z, int32_t* len_x, int32_t* feat_sz, int32_t* channels);

/* MFMA lane-map self test: runs v_mfma_f32_16x16x4_f32 on exact integer data with an
 * asymmetric B and checks the operand / result lane maps the kernels rely on. 0 = as assumed. */
int vt_selftest_mfma(void* stream);

/* Development probe (synchronises; not part of the hot path): sustained shader clock in MHz
 * under a dense f32-MFMA loop with `waves_per_simd` waves on every SIMD, the cycles one SIMD
 * spends per v_mfma_f32_16x16x4_f32, and the wall time of the probe launch.  Used by bench.py to
 * state the clock the roofline fraction was measured at. */
int vt_probe_clock(int32_t iters, int32_t waves_per_simd, double* mhz, double* cycles_per_mfma, double* wall_us);
/* Development aid: with VT_DBG_STAMPS=1 in the environment at vt_create the transformer-block kernel
 * records s_memtime at its phase boundaries; this copies them out ([B][8][64] uint64, wave-major, synchronises). */
int vt_debug_stamps(vt_model* m, int32_t B, unsigned long long* host_out);

#ifdef __cplusplus
}
#endif
#endif /* VITTRACK_H */
