// detr_kernels.h -- launchers of the non-GEMM detector kernels (attention, normalisation, layout,
// pooling, post-processing).  Row-major, batch-major ([B, L, C] / NHWC) tensors; `dtype` 0 = fp32 activations,
// 1 = bf16 activations in memory (arithmetic stays fp32; BASELINE config 4).  Tables (pos, gamma, beta) are fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace odam_dk {

// Fused multi-head attention, head dim 32 (hidden 256 / 8 heads, transformer.py:138-139):
//   O[b, q, h*32:(h+1)*32] = softmax_k( (Q[b,q,h] / sqrt(32)) . K[b,k,h] ) V[b,k,h]
// Q/K/V/O are addressed as base + (b*L + row) * ld + h*32, so fused projection buffers work.
// key_mask (nullable) [B][Lk]: non-zero = padded key, left out of the softmax (key_padding_mask, transformer.py:157-160).
int launch_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                     int B, int H, int Lq, int Lk, int dtype, hipStream_t stream, const unsigned char* key_mask = nullptr);

// same kernel with head dimension 64, fp32 (associator GNN, 4 heads of 64; scores scaled by 1/8)
int launch_attention_d64(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         int B, int H, int Lq, int Lk, hipStream_t stream);

// y = LayerNorm(x + r) * gamma + beta over C = 256 (eps 1e-5); r may be null.
// If y_pos != null also writes y_pos = y + pos[row % L]  (the next layer's q/k input).
int launch_add_layernorm(const void* x, const void* r, const float* gamma, const float* beta, void* y,
                         const float* pos, int L, void* y_pos, int M, int dtype, hipStream_t stream);

// out[m] = (x ? x[m] : 0) + pos[m % L], rows of 256 floats
int launch_add_pos(const void* x, const float* pos, int L, void* out, int M, int dtype, hipStream_t stream);

// fp32 [B,3,H,W] -> [B,H,W,4] fp32 (4th channel zero) or [B,H,W,8] bf16 (one 16-byte chunk per pixel)
int launch_nchw_to_nhwc4(const float* in, void* out, int B, int H, int W, int dtype, hipStream_t stream);
// NHWC4 (fp32 or bf16) with a zero frame of 3 rows / 3 + 5 columns, [B][H + 6][W + 8][4] (the stem as a 7x1 convolution over 8-pixel rows)
int launch_nchw_to_nhwc4_framed(const float* in, void* out, int B, int H, int W, int dtype, hipStream_t stream);

// NHWC max-pool 3x3 stride 2 pad 1 (torchvision ResNet stem)
int launch_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, int Ho, int Wo, int dtype,
                        hipStream_t stream);

// NHWC [B,H,W,C] -> NCHW [B,C,H,W]  (only for exposing intermediate features to tests)
int launch_nhwc_to_nchw(const void* in, float* out, int B, int H, int W, int C, int dtype, hipStream_t stream);

// activation buffer (n elements, n % 4 == 0) -> fp32 copy
int launch_to_f32(const void* in, float* out, size_t n, int dtype, hipStream_t stream);

int launch_sigmoid(float* x, int n, hipStream_t stream);

// DETR.postprocess arithmetic up to (not including) thresholding / NMS  (detr.py:111-140), per query:
// rows[b][q][16] = score, class, x0, y0, x1, y1 (pixels), cx3d, cy3d, depth, angle_bin, d0, d1, d2, 0, 0, 0
int launch_postprocess(const float* logits, const float* boxes, const float* angle, const float* offset,
                       const float* size, const float* depth, int B, int Q, int n_cls1, int n_bins, float img_w,
                       float img_h, float fx, float fy, float cx, float cy, float* rows, hipStream_t stream);

// Input transform on the device: uint8 [B,h,w,3] -> float32 [B,3,H,W] = Normalize(ToTensor(PIL-bilinear resize)),
// bit-identical to the host transform (resampling tables: detr_model.hip resample_table, 22-bit fixed point)
int launch_preprocess_u8(const unsigned char* rgb, int B, int h, int w, const int* tx_xmin, const int* tx_cnt,
                         const int* tx_K, int tx_ksize, const int* ty_xmin, const int* ty_cnt, const int* ty_K,
                         int ty_ksize, float* out, int H, int W, const float mean[3], const float stdv[3],
                         hipStream_t stream);

}  // namespace odam_dk
