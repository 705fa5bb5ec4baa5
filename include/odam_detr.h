/*
 * odam_detr.h -- C ABI of the MI355X (gfx950) DETR-style monocular 3D detector forward pass.
 *
 * Drop-in boundary for the reference's (likojack/ODAM) detector as OdamProcess.run_detector uses it
 * (src/processor.py:259-289):
 *   src/models/detr.py:49-94      DETR.forward        -> odam_detr_forward
 *   src/models/detr.py:96-140     DETR.postprocess    -> odam_detr_postprocess (arithmetic per query;
 *                                 thresholding + greedy nms_3d :161-205 stay on the host)
 *   src/models/backbone.py:21-94  FrozenBatchNorm2d + torchvision ResNet-50/101 body
 *   src/models/transformer.py     6+6 post-norm encoder/decoder, nn.MultiheadAttention(256, 8)
 *   run_processor.py:32-33        load_state_dict      -> odam_detr_set_weight per state_dict entry,
 *                                 with the reference's key names
 *
 * Conventions as in odam_sq.h: int return codes (0 = OK), odam_last_error(), [dev]/[host] pointers,
 * caller's hipStream_t passed as void*, no allocation or synchronisation inside forward/postprocess
 * (the workspace is allocated by odam_detr_create for max_batch frames of img_h x img_w).
 */
#ifndef ODAM_DETR_H
#define ODAM_DETR_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct odam_detr odam_detr;

typedef struct {
    int resnet_blocks[4];  /* {3,4,6,3} = resnet50, {3,4,23,3} = resnet101 (backbone.py:90) */
    int hidden_dim;        /* 256  (configs/detr_scan_net.yaml) */
    int nheads;            /* 8    head dim must be 32 */
    int dim_feedforward;   /* 2048 */
    int enc_layers;        /* 6 */
    int dec_layers;        /* 6 */
    int num_queries;       /* 100 */
    int num_classes1;      /* 18 + 1 "no object" (detr.py:531-532) */
    int angle_bins;        /* 30 */
    int max_batch;         /* frames per forward call */
    int img_h, img_w;      /* network input size, e.g. 800 x 1066 for a 640x480 frame (transforms.py:78-96) */
    int dtype;             /* 0: fp32 everywhere (parity mode).  1: bf16 weights + activations in memory, bf16 MFMA
                              with fp32 accumulation, fp32 softmax / LayerNorm / head outputs (BASELINE config 4) */
    int pre_norm;          /* 0: post-norm layers (the shipped configuration).  1: `normalize_before` -- every encoder / decoder
                              sub-block normalises its input and adds its output to the stream, the encoder ends in its own
                              LayerNorm "transformer.encoder.norm.*" (src/models/transformer.py:169-188, 240-262, 26-28) */
    int dilation;          /* 1: the DC5 backbone (src/models/backbone.py:89-91): layer4 keeps layer3's resolution -- stride 1, its
                              3x3 filters dilated by 2 from the second block on -- so the token grid is ceil(H/16) x ceil(W/16) */
} odam_detr_cfg;

int odam_detr_create(const odam_detr_cfg* cfg, odam_detr** out);
int odam_detr_destroy(odam_detr* m);

/* One call per state_dict entry, `name` = the reference's key (e.g.
 * "backbone.0.body.layer1.0.conv1.weight", "transformer.encoder.layers.0.self_attn.in_proj_weight",
 * "class_embed.bias"); data [host] float32, contiguous, PyTorch layout.  Unknown names are ignored
 * (e.g. "...num_batches_tracked", as FrozenBatchNorm2d._load_from_state_dict does, backbone.py:36-44).
 * The sine position embedding of the token grid is passed under the name "pos_embed"
 * ([h*w, hidden_dim], position_encoding.py:26-46 -- a constant of the input size). */
int odam_detr_set_weight(odam_detr* m, const char* name, const float* data, const long long* shape, int ndim);

/* Packs weights for the kernels (NHWC k-major filters, FrozenBN folded to scale/bias exactly as
 * backbone.py:46-56 computes them, fused attention projections); fails if any tensor is missing. */
int odam_detr_finalize(odam_detr* m);

/* token grid of the configured input: h = ceil(img_h/32), w = ceil(img_w/32) */
int odam_detr_feature_hw(const odam_detr* m, int* h, int* w);

/*
 * img [dev] [B,3,img_h,img_w] float32 (normalised, as get_transforms() produces).
 * Outputs [dev], the last decoder layer's predictions (detr.py:80-88):
 *   logits [B,Q,num_classes1], boxes [B,Q,4] (sigmoid applied), angle [B,Q,angle_bins], offset [B,Q,2],
 *   size [B,Q,3], depth [B,Q,1], obj_features [B,Q,hidden] (nullable).
 */
int odam_detr_forward(odam_detr* m, const float* img, int B, float* logits, float* boxes, float* angle,
                      float* offset, float* size, float* depth, float* obj_features, void* stream);

/*
 * The same forward over a batch of images of DIFFERENT sizes, zero-padded to the handle's img_h x img_w at the top-left
 * as nested_tensor_from_tensor_list does (src/utils/misc.py:303-320):
 *   key_mask [dev] [B][h*w] bytes, 1 = token lies in the padding (the image mask reduced to the feature grid by nearest
 *            interpolation, backbone.py:79) -- such keys are excluded in encoder self-attention and decoder
 *            cross-attention (key_padding_mask, transformer.py:157-160, 224-228);
 *   pos      [dev] [B][h*w][hidden] float32: each image's own sine embedding (position_encoding.py:26-46 depends on the
 *            mask through the cumulative sums and their normalisation).
 * odam_amd/detector.py::Detector.forward_nested builds both exactly as the reference does.
 */
int odam_detr_forward_masked(odam_detr* m, const float* img, int B, const unsigned char* key_mask, const float* pos,
                             float* logits, float* boxes, float* angle, float* offset, float* size, float* depth,
                             float* obj_features, void* stream);

/* optional taps for parity tests: layer4 feature map as NCHW [B,2048,h,w] and encoder memory [B,h*w,hidden]
 * of the most recent forward (either may be null) */
int odam_detr_debug_read(odam_detr* m, int B, float* layer4_nchw, float* memory, void* stream);

/*
 * Per-query post-processing arithmetic (detr.py:111-140) on device:
 * rows [dev] [B,Q,16] = score, class, x0,y0,x1,y1 (pixels of img_w x img_h), cx3d, cy3d, depth,
 *                       angle_bin, d0,d1,d2, 0,0,0.   K9 [host] row-major 3x3 intrinsics.
 */
int odam_detr_postprocess(odam_detr* m, const float* logits, const float* boxes, const float* angle,
                          const float* offset, const float* size, const float* depth, int B, const float* K9,
                          float img_w, float img_h, float* rows, void* stream);

/* Threshold + greedy nms_3d (detr.py:124-125, 161-205) on ONE frame's rows [host][Q,16] as written by
 * odam_detr_postprocess; keep_idx [host][Q] receives the kept query indices in descending-score order. */
int odam_detr_select(const float* rows, int Q, float threshold, int nms_2d, int* keep_idx, int* n_keep);

/* The detector's input transform on the device (reference: src/datasets/transforms.py:281-290 = resize :75-105 via
 * torchvision F.resize -> PIL Image.resize(BILINEAR), ToTensor :222-224, Normalize :236-243; called per frame from
 * src/processor.py:263): rgb [dev][B,h,w,3] uint8 -> out [dev][B,3,img_h,img_w] float32 at the handle's size,
 * bit-identical to the host transform (Pillow's two-pass 22-bit fixed-point resampling, then float32 /255, -mean, /std).
 * mean, std: host float[3].  The resampling tables of each (h, w) are built and uploaded at first use (synchronous). */
int odam_detr_preprocess_u8(odam_detr* m, const unsigned char* rgb, int B, int h, int w, const float* mean,
                            const float* std_dev, float* out, void* stream);

/* Per-launch timing of the contraction kernel (all conv / linear launches of a forward are bracketed by
 * event pairs while enabled); read returns the totals of the most recent forward. */
int odam_detr_profile_enable(odam_detr* m, int on);
int odam_detr_profile_read(odam_detr* m, int* n_launches, double* total_ms, double* total_flops);
int odam_detr_profile_read_attention(odam_detr* m, int* n_launches, double* total_ms, double* total_flops);

/* ---- single-operator entry points (the same kernels the forward uses; for parity tests and reuse) ---- */
/* NHWC convolution / linear:  x [dev][B,H,W,Cin] (Cin power of two >= 4), w_packed [dev][Cout][Kpad], zero padded
 * to a multiple of the k-tile (32 fp32 / 64 bf16), scale/bias/residual nullable, y [dev][B,Ho,Wo,Cout].
 * k_order 0: k = (ky*KW+kx)*Cin + ci.  k_order 1 (Cin % k-tile == 0, KH*KW <= 32; what the detector uses for its 3x3
 * layers): k = ((ci / kt)*KH*KW + ky*KW+kx)*kt + ci % kt -- all taps of one channel chunk in consecutive k-tiles. */
int odam_op_conv2d_nhwc(const float* x, const float* w_packed, const float* scale, const float* bias,
                        const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                        int stride, int pad, int Kpad, int relu, int k_order, void* stream);
/* same kernel in bf16 mode: x, w_packed, residual, y are raw bfloat16 (Cin a power of two >= 8, Kpad % 64 == 0);
 * out_f32 != 0 writes y as fp32 */
int odam_op_conv2d_nhwc_bf16(const void* x, const void* w_packed, const float* scale, const float* bias,
                             const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                             int stride, int pad, int Kpad, int relu, int out_f32, int k_order, void* stream);
/* bf16 bottleneck tail as one launch (BASELINE config 4; replaces what torchvision's Bottleneck.forward runs after conv1 --
 * conv2, bn2, relu, conv3, bn3, += identity, relu -- as instantiated at src/models/backbone.py:90-92, plus optionally the next
 * block's conv1 / bn1 / relu): x [dev][B,H,W,P] bf16 (P = 64 / 128 / 256), w2 [P][9 P] packed k_order 1, w3 [4 P][P], residual / y
 * [B,Ho,Wo,4 P] bf16, w1n [PN][4 P] + y_next [B,Ho,Wo,PN] optional (PN = 0: none).  Returns 4 where the fused kernel does not
 * apply (shape / size); bit-identical to the layers run one by one through odam_op_conv2d_nhwc_bf16. */
int odam_op_bottleneck_bf16(const void* x, const void* w2, const float* s2, const float* b2, const void* w3, const float* s3,
                            const float* b3, const void* residual, void* y, const void* w1n, const float* s1n, const float* b1n,
                            void* y_next, int B, int H, int W, int P, int stride, int PN, void* stream);
/* which layers the bf16-native 256-row schedule of the contraction kernel takes: 0 none, 1 those large enough to fill
 * the device (default), 2 every eligible layer (parity tests on small shapes).  Process-wide; also ODAM_CG_BIG. */
int odam_op_conv_bf16_mode(int mode);
/* how fp32 layers large enough for the 256-row schedule are multiplied: 0 = v_mfma_f32_32x32x2_f32 on 128x128 tiles
 * (the k-ordered fp32 FMA chain), 1 = the same instruction in the ring
 * kernel, 2 (default) = every fp32 operand split exactly into three bf16 values and the six significant cross products taken on
 * v_mfma_f32_32x32x16_bf16 (fp32-class accuracy -- 0.8-1.4e-7 of sum |a b| against float64, the fp32 instruction
 * 1.1-1.9e-7 -- at 2.7x the matrix rate; last bits differ from mode 0).  Process-wide; also ODAM_CG_BIG_F32. */
int odam_op_conv_f32_mode(int mode);
/* diagnostics: launches of conv1 with the max-pool on its tile (odam_config stem.pool) since the library was loaded -- lets a
 * test see that the fused path, not the conv1 + max-pool pair, produced what it compares */
long long odam_op_pooled_stem_launches(void);
int odam_op_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                      int B, int H, int Lq, int Lk, void* stream);
/* the same op on bf16 tensors (config 4: nn.MultiheadAttention's scaled-dot-product core, src/models/transformer.py:154-167, on
 * v_mfma_f32_32x32x16_bf16): Q / K / V / O [dev] bf16 with the given row pitches (multiples of 8 / 8 / 8 / 4 elements), head h at columns
 * 32 h .. 32 h + 31; standalone entry for tests and timing -- the detector calls the same launcher */
int odam_op_attention_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                           int B, int H, int Lq, int Lk, void* stream);
int odam_op_add_layernorm(const float* x, const float* r, const float* gamma, const float* beta, float* y, int M,
                          void* stream);
int odam_op_maxpool3x3s2_nhwc(const float* x, float* y, int B, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
