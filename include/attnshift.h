/* libattnshift_hip.so -- C ABI of the MI355X (gfx950) hot path of AttentionShift.
 *
 * The reference is Python on PyTorch; its only native boundary on this path is
 *   cc_torch.connected_components_labeling(uint8 [H,W]) -> int32 [H,W]
 *   (mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py:23,68  -- source absent upstream)
 * Every other entry point below replaces a run of ATen calls inside one reference function
 * (cited per function, paths relative to the reference root, `stdroi` = the file above).
 *
 * Conventions
 *   - return 0 (AS_OK) on success, a negative AS_E_* otherwise; as_last_error() gives the message
 *     (thread-local).  Never throws, never exits.
 *   - all pointers are DEVICE pointers (HBM) unless stated; contiguous row-major layouts only.
 *   - no ownership transfer: inputs, outputs and workspaces are caller-allocated
 *     (sizes through as_*_workspace_bytes); workspaces need no initialisation.
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream); re-entrant;
 *     no global mutable state.  as_sdpa_bwd (bf16) runs one of its kernels on a library-owned helper stream
 *     (per host thread and device, created on first use and freed at thread exit) that is forked from and joined
 *     back to `stream` with events: to the caller everything is ordered on `stream`, but the first call on a thread
 *     creates a stream (not legal inside a stream capture -- warm up before capturing), and a per-stream allocator
 *     must treat the workspace and dqkv as in use until work queued on `stream` after the call has run.
 *   - dtype: AS_F32 = exact fp32 MFMA path (parity), AS_BF16 = bf16 operands with fp32 accumulate.
 *     Biases, log-sum-exp, roll-out matrices and all of Part B are always fp32; indices int32.
 */
#ifndef ATTNSHIFT_H
#define ATTNSHIFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AS_VERSION 100
#define AS_OK 0
#define AS_E_BADARG (-1)
#define AS_E_UNSUPPORTED (-2)
#define AS_E_LAUNCH (-3)
#define AS_E_WORKSPACE (-4)

#define AS_F32 0
#define AS_BF16 1

#define AS_HEAD_DIM 64 /* every ViT in the reference has D/h = 64 (models/vision_transformer.py:266-288) */

typedef void* as_stream_t;

int as_version(void);
const char* as_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Part A -- backbone attention (models/vision_transformer.py:62-124)
 * ------------------------------------------------------------------------------------------- */

/* padded token count used by the q/k/v workspaces: N rounded up to a multiple of 64 */
int as_npad(int N);

/* out[M,Nout] = act(x[M,K] . W[Nout,K]^T + bias[Nout])           nn.Linear (vision_transformer.py:47-59,
 * 84).  x,W,out in `dtype`; bias fp32 or NULL; act 0 = none, 1 = exact (erf) GELU, 4 = ReLU (the point head's FFN,
 * mmdet/models/backbones/visual_transformer_det.py:26-38).  K % 32 == 0. */
int as_linear_fwd(const void* x, const void* W, const float* bias, void* out, int M, int Nout, int K,
                  int dtype, int act, as_stream_t stream);

/* out[M,Nout] = act(x[M,K] . W[Nout,K]^T + bias) in fp32 for SMALL M (a few hundred rows): the point head of
 * VisionTransformerDet -- class_embed / bbox_embed, two 3-layer FFNs over the B * T = 200 point tokens
 * (mmdet/models/backbones/visual_transformer_det.py:26-38, 145-146, 262-267).  32 x 64 output tiles whose four waves split
 * K, so 200 rows still fill the chip; exact fp32 arithmetic (f32 MFMA chains), deterministic summation order.  Row strides
 * ldx / ldo (elements) let a column slice of a packed activation be read / written in place.  act: 0 none, 1 exact GELU,
 * 4 ReLU, 5 sigmoid (the coordinate head, :267).  K % 16 == 0, K >= 64, ldx % 4 == 0, 16-byte aligned x / W; bias or NULL. */
int as_linear_small_fwd(const float* x, int ldx, const float* W, const float* bias, float* out, int ldo, int M, int Nout, int K,
                        int act, as_stream_t stream);

/* as_linear_fwd with a stream-K schedule for shapes whose tile count leaves the last round of the plain grid mostly empty
 * (ViT-B at 2 x 4197 tokens: fc1 = 396 tiles of 256 x 256 on 256 CUs, fc2 = 198 of 256 x 128): every workgroup contracts
 * the same number of K steps of the flattened (tile, K step) space; a tile cut into pieces is finished by its last-arriving
 * piece from fp32 partials in `ws` (fixed summation order: bitwise reproducible).  as_linear_sk_workspace_bytes returns 0
 * when the plain grid is the better schedule for (M, Nout, K) -- the call then IS as_linear_fwd and ws may be NULL.  bf16.
 * Round-5 measurement: with the fence-based hand-off it is SLOWER than the plain grid on every ViT-B / ViT-L shape (fc1
 * 60 -> 150 us), so the query returns 0 unless AS_GEMM_SK=1 is set in the environment (csrc/gemm.hip sk_plan). */
size_t as_linear_sk_workspace_bytes(int M, int Nout, int K);
int as_linear_sk_fwd(const void* x, const void* W, const float* bias, void* out, int M, int Nout, int K, int dtype, int act,
                     void* ws, size_t ws_bytes, as_stream_t stream);

/* nn.ConvTranspose2d(cin, cout, kernel 2, stride 2) on a channels-last map as one GEMM over its pixels
 * (mmdet/models/backbones/visual_transformer_det.py:107-117, the FPN taps): x [M = B*h*w, cin] (pixel-major, grid w wide),
 * W4 [4*cout, cin] with row (di*2 + dj)*cout + co = weight[ci, co, di, dj], bias4 [4*cout] fp32 or NULL ->
 * out NHWC [B, 2h, 2w, cout]: out[b, 2i+di, 2j+dj, co] = act(sum_ci x[b,i,j,ci] W[ci,co,di,dj] + bias[co]); the epilogue
 * scatters straight to the interleaved pixel (no layout copy).  act as as_linear_fwd (an eval-mode BatchNorm that follows
 * is folded into W4 / bias4 by the caller).  bf16 only; cin % 32 == 0, cout % 8 == 0. */
int as_deconv2x2_fwd(const void* x, const void* W4, const float* bias4, void* out, int M, int w, int cin, int cout,
                     int dtype, int act, as_stream_t stream);

/* QKV projection with the head split fused into the epilogue (vision_transformer.py:75-77):
 *   k : [B,h,Npad,64]   vt : [B,h,64,Npad]  (V transposed so the P.V MFMA reads keys contiguously)
 *   q : [B,h,Npad,64] elements, FRAGMENT-MAJOR inside every 32-row x 64-d tile: element (r, d) of a tile sits at
 *       ((d/16)*64 + r + 32*((d%16)/8))*8 + d%8, so the MFMA operand loads of sdpa / roll-out are coalesced.
 *       VALUES are pre-scaled: q holds (x Wq^T + bq) * log2(e) / sqrt(64), rounded once from the fp32 accumulator, so
 *       that q . k is the base-2 logit every consumer (as_sdpa_fwd / _bwd, as_rollout_*, as_attn_mean_rows) feeds to
 *       exp2 without a multiply per score.  A caller that fills q by hand applies the same factor.
 * rows/cols >= N of the padded layouts are never read unmasked, so they need no initialisation. */
int as_qkv_fwd(const void* x /*[B,N,D]*/, const void* Wqkv /*[3D,D]*/, const float* bqkv /*[3D] or NULL*/,
               void* q, void* k, void* vt, int B, int N, int D, int h, int dtype, as_stream_t stream);

/* softmax(q k^T / sqrt(64)) v without materialising [h,N,N] (vision_transformer.py:79-83).
 *   q is the PRE-SCALED fragment-major workspace of as_qkv_fwd (see there): softmax(q' k^T ln 2) = softmax(q k^T / 8).
 *   o   : [B,N,h*64] (heads concatenated, ready for proj)        lse : [B,h,N] fp32, natural log
 *   workspace : optional, as_sdpa_fwd_workspace_bytes(...) bytes (0 for most shapes): lets the launch split the keys
 *               of the last q-tile over extra workgroups when the plain grid would leave a short second round
 *               (ViT-B/1024^2/B=2: 792 workgroups for 768 resident slots); NULL = plain grid.  Same results. */
size_t as_sdpa_fwd_workspace_bytes(int B, int N, int h, int dtype);
int as_sdpa_fwd(const void* q, const void* k, const void* vt, void* o, float* lse, void* workspace,
                size_t workspace_bytes, int B, int N, int h, int dtype, as_stream_t stream);

/* Backward of as_sdpa_fwd (autograd of vision_transformer.py:79-83; the reference trains the backbone, and with
 * use_checkpoint recomputes each block's forward in backward, visual_transformer_det.py:232-236).  Softmax tiles are
 * recomputed from q, k and lse; no [h,N,N] buffer, no atomics, fixed summation order.
 *   q,k,vt,lse : as written by as_qkv_fwd / as_sdpa_fwd           o, d_o : [B,N,h*64] (forward output, its gradient)
 *   dqkv       : [B,N,3,h,64] = gradient of the QKV projection output in the reference's reshape order (:76)
 *   workspace  : as_sdpa_bwd_workspace_bytes(B,N,h,dtype) bytes, caller-owned
 * Rows [N, Npad) of q / k / vt may hold anything (NaN included): nothing outside [0, N) reaches a gradient. */
size_t as_sdpa_bwd_workspace_bytes(int B, int N, int h, int dtype);
int as_sdpa_bwd(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse,
                void* dqkv, void* workspace, size_t workspace_bytes, int B, int N, int h, int dtype,
                as_stream_t stream);

/* Attention.forward = qkv + sdpa + proj (vision_transformer.py:74-86).  q/k/vt/o are workspaces the
 * caller keeps alive when the roll-out (below) needs this layer; `o` is [B,N,D]. */
int as_attn_fwd(const void* x, const void* Wqkv, const float* bqkv, const void* Wproj, const float* bproj,
                void* out /*[B,N,D]*/, float* lse, void* q, void* k, void* vt, void* o, void* workspace,
                size_t workspace_bytes /* as_sdpa_fwd_workspace_bytes, may be NULL/0 */, int B, int N, int D, int h,
                int dtype, as_stream_t stream);

/* Backward of as_attn_fwd (autograd of Attention.forward, vision_transformer.py:74-86): given dout = dL/d out and the
 * tensors as_attn_fwd saved (q,k,vt,o,lse) returns dL/dx and the parameter gradients.
 *   x, dout, dx : [B,N,D]     dWqkv : [3D,D]   dWproj : [D,D] (element dtype = `dtype`)   dbqkv [3D], dbproj [D] fp32 or NULL
 *   workspace   : as_attn_bwd_workspace_bytes(B,N,D,h,dtype) bytes, caller-owned */
size_t as_attn_bwd_workspace_bytes(int B, int N, int D, int h, int dtype);
int as_attn_bwd(const void* x, const void* Wqkv, const void* Wproj, const void* dout, const void* q, const void* k,
                const void* vt, const void* o, const float* lse, void* dx, void* dWqkv, float* dbqkv, void* dWproj,
                float* dbproj, void* workspace, size_t workspace_bytes, int B, int N, int D, int h, int dtype,
                as_stream_t stream);

/* Swin window attention for one SwinTransformerBlock (models/swin_transformer.py): fuses the pad (:271-276), cyclic
 * shift (:279-280), window_partition (:292-293), the WindowAttention core (:131-153 incl. the relative-position-bias
 * gather :141-144 and the shift mask :233-256), window_reverse (:299-300), reverse shift (:303-304) and un-pad (:308).
 *   qkv      : [B,H,W,3C] = x_normed . Wqkv^T WITHOUT bias, on the original token grid, channel order (3,h,32)
 *   bqkv     : fp32 [3C] or NULL (added here; padded tokens are exactly the bias, as in the reference)
 *   table    : fp32 [(2*ws-1)^2, h] relative_position_bias_table
 *   out      : [B,H,W,C] attention output at the original token positions (input of the proj Linear)
 *   attn_out : fp32 [B*nW, h, ws*ws, ws*ws] softmax (the reference returns it) or NULL
 * ws must be 7 and C == 32*h (all Swin variants of the reference, swin_transformer.py:844-870). */
int as_window_attn_fwd(const void* qkv, const float* bqkv, const float* table, void* out, float* attn_out, int B, int H,
                       int W, int C, int h, int ws, int shift, int dtype, as_stream_t stream);

/* Fused residual add + LayerNorm of the pre-LN block (vision_transformer.py:109-124: x = x + sublayer(norm(x))):
 *   x_out = x_in + delta (fp32 residual stream; delta [M,D] in `dtype`, or NULL),  y_out = LN(x_out)*gamma + beta (`dtype`)
 * x_out may alias x_in; either output may be NULL (x_out NULL: LN only; y_out NULL: add only).  D % 4 == 0, D <= 3072. */
int as_add_layernorm(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps,
                     float* x_out, void* y_out, int M, int D, int dtype, as_stream_t stream);
/* The same with the sub-layer output scaled per SAMPLE: x_out = x_in + delta_scale[row / rows_per_scale] * delta -- the
 * DropPath of the training path (models/vision_transformer.py:21-40, applied at :114-118: x + drop_path(f(norm(x)))),
 * with delta_scale[b] = mask_b / keep_prob drawn by the caller (fp32, one value per image; NULL = 1). */
int as_add_layernorm_scaled(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps,
                            float* x_out, void* y_out, int M, int D, int dtype, const float* delta_scale,
                            int rows_per_scale, as_stream_t stream);

/* out[M,Nout] = x[M,K] . W[Nout,K]^T (no bias, no activation) with the CONTRACTION split over workgroups: fp32 partial
 * products of K ranges, summed in range order (deterministic), written as bf16 (out_f32 = 0) or fp32 (1).  The form of an
 * nn.Linear WEIGHT gradient dW = dy^T . x (models/vision_transformer.py:75-77,84 under autograd): few output tiles, the
 * tokens as K.  bf16 operands, K % 32 == 0, Nout % 4 == 0; workspace: as_linear_splitk_workspace_bytes(M, Nout, K). */
/* The MLP's first linear under autograd (vision_transformer.py:47-59: fc1 -> GELU): pre[M,Nout] = x . W^T + bias rounded
 * to bf16, out = GELU(pre) -- both written by the GEMM's epilogue (no separate activation pass; `pre` is what the
 * backward needs).  bf16 only, K % 32 == 0, Nout % 8 == 0. */
int as_linear_gelu_fwd(const void* x, const void* W, const float* bias, void* out, void* pre, int M, int Nout, int K,
                       int dtype, as_stream_t stream);
/* out[M,Nout] = (x[M,K] . W[Nout,K]^T) * GELU'(pre[M,Nout]): a linear's input gradient times the derivative of the GELU that
 * produced that input, in the epilogue (the GeluBackward pass of autograd).  bf16 only, K % 32 == 0, Nout % 8 == 0. */
int as_linear_dgelu_fwd(const void* x, const void* W, const void* pre, void* out, int M, int Nout, int K, int dtype,
                        as_stream_t stream);

size_t as_linear_splitk_workspace_bytes(int M, int Nout, int K);
int as_linear_splitk_fwd(const void* x, const void* W, void* out, int M, int Nout, int K, int dtype, int out_f32,
                         void* workspace, size_t workspace_bytes, as_stream_t stream);

/* Backward of one nn.Linear y = x W^T + b under autograd (models/vision_transformer.py:47-59 the MLP, :75-77,84 and the
 * MAE-decoder heads' layers, mae_bbox_head_rec.py:148-168), x [M,K], W [Nout,K], dy [M,Nout] bf16:
 *   dx [M,K] bf16   = dy . W           (NULL: skipped)   on the forward kernel against a transposed copy of W
 *   dW [Nout,K]     = dy^T . x         (NULL: skipped)   split-K over the rows (as_linear_splitk_fwd), bf16 or fp32 (dw_f32)
 *   db [Nout] fp32  = column sums of dy (NULL: skipped)  two fixed-order stages, no atomics
 * Nout % 32 == 0, K % 4 == 0; workspace: as_linear_bwd_workspace_bytes(M, Nout, K). */
size_t as_linear_bwd_workspace_bytes(int M, int Nout, int K);
int as_linear_bwd(const void* x, const void* W, const void* dy, void* dx, void* dW, float* db, int M, int Nout, int K,
                  int dtype, int dw_f32, void* workspace, size_t workspace_bytes, as_stream_t stream);
/* as_linear_bwd for the linear that FOLLOWS a GELU (fc2 of the Mlp): x = GELU(pre); dx [M,K] comes out already multiplied
 * by GELU'(pre[M,K]) (as_linear_dgelu_fwd), i.e. it is the gradient of the pre-activation.  pre NULL = as_linear_bwd. */
int as_linear_bwd_dgelu(const void* x, const void* W, const void* dy, const void* pre, void* dx, void* dW, float* db, int M,
                        int Nout, int K, int dtype, int dw_f32, void* workspace, size_t workspace_bytes, as_stream_t stream);

/* k x k / stride-k max pooling of a token-major (NHWC) fp32 map [B,H,W,C] -> [B,H/k,W/k,C] (contiguous): the FPN's
 * coarsest tap, nn.MaxPool2d(k, k) (mmdet/models/backbones/visual_transformer_det.py:120,129,133) on the layout the taps
 * live in here.  x_batch_stride = elements between consecutive images of x (>= H*W*C, multiple of 4): the tap is the
 * patch-token slice of the [B, 1+Np+T, C] token tensor, so its images are not adjacent.  C % 4 == 0, H and W multiples
 * of k; NaNs propagate as in ATen. */
int as_maxpool_nhwc(const float* x, float* out, int B, int H, int W, int C, int k, long long x_batch_stride,
                    as_stream_t stream);

/* Token assembly of prepare_tokens (mmdet/models/backbones/visual_transformer_det.py:192-214: patch embedding + position
 * embedding, class token in front, point tokens + their position embedding behind): out [B,N,D] fp32,
 *   out[b,n] = table[n] + emb[b,n-1]  for 1 <= n <= Np,   out[b,n] = table[n]  otherwise,
 * emb [B,Np,D] (`dtype`), table [N,D] fp32 = [cls + pos_0 ; pos_1..Np ; point_token + point_pos_embed] (the caller's
 * constant).  D % 4 == 0. */
int as_assemble_tokens(const void* emb, const float* table, float* out, int B, int Np, int N, int D, int dtype,
                       as_stream_t stream);

/* Backward of as_add_layernorm for the trainable path.  x [M,D] fp32 = the x_out the forward wrote (x_in + delta); dy
 * [M,D] in `dtype` = gradient of y_out (NULL: the call was add-only); dx_res [M,D] fp32 = gradient of x_out from the
 * residual stream (NULL: none); gamma fp32 [D].  Writes dx_out [M,D] fp32 (gradient of x_in) and / or ddelta_out [M,D]
 * in `dtype` (gradient of delta: the same values), dgamma / dbeta fp32 [D] (either may be NULL).  Column sums go through
 * per-workgroup partials added in workgroup order (deterministic).  D % 4 == 0, D <= 2048. */
size_t as_add_layernorm_bwd_workspace_bytes(int M, int D);
int as_add_layernorm_bwd(const float* x, const void* dy, const float* dx_res, const float* gamma, float eps, float* dx_out,
                         void* ddelta_out, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int M, int D,
                         int dtype, as_stream_t stream);
/* Backward of as_add_layernorm_scaled: ddelta_out = delta_scale[row / rows_per_scale] * (gradient of x_out). */
int as_add_layernorm_bwd_scaled(const float* x, const void* dy, const float* dx_res, const float* gamma, float eps,
                                float* dx_out, void* ddelta_out, float* dgamma, float* dbeta, void* workspace,
                                size_t workspace_bytes, int M, int D, int dtype, const float* delta_scale,
                                int rows_per_scale, as_stream_t stream);

/* Backward of as_window_attn_fwd (autograd of WindowAttention's core + the index maps around it):
 *   d_out     : [B,H,W,C] gradient of the attention output        dqkv : [B,H,W,3C] gradient w.r.t. the bias-free qkv
 *   dtable    : fp32 [(2*ws-1)^2, h] gradient of relative_position_bias_table
 *   dbqkv_pad : fp32 [3C] bias gradient contributed by PADDED tokens (whose qkv is the bias itself); the full qkv-bias
 *               gradient is the column sum of dqkv over real tokens plus this vector
 * No atomics: window partials are reduced in window order by a second launch. */
size_t as_window_attn_bwd_workspace_bytes(int B, int H, int W, int h, int ws);
int as_window_attn_bwd(const void* qkv, const float* bqkv, const float* table, const void* d_out, void* dqkv,
                       float* dtable, float* dbqkv_pad, void* workspace, size_t workspace_bytes, int B, int H, int W,
                       int C, int h, int ws, int shift, int dtype, as_stream_t stream);

/* Head-mean attention rows, recomputed from q,k,lse (visual_transformer_det.py:236,242 keeps
 * attn.mean(1) of every layer; only row slices are ever consumed, stdroi:2272):
 *   out[b,i,:] = (1/h) sum_h softmax_row(row0 + i)          out : [B,nrows,N] fp32 */
int as_attn_mean_rows(const void* q, const void* k, const float* lse, float* out, int B, int N, int h,
                      int row0, int nrows, int dtype, as_stream_t stream);

/* Row-sliced attention roll-out (stdroi:1257-1272 attns_project_to_feature, rows [-T:] only, T <= 128):
 *   A_hat = (mean_h P + I) / rowsum (rowsum == 2),   R_out = R_in . A_hat        R : [B,T,N] fp32
 * with the tiles of mean_h P recomputed from q,k,lse of that layer.  as_rollout_top gives the top layer's
 * R = rows [N-T, N) of A_hat.  Next to the fp32 row-major R every call also emits a FRAGMENT-MAJOR copy `rf`
 * (as_rollout_rfrag_bytes; element dtype = `dtype`) that the next step reads as its MFMA A operand with fully
 * coalesced loads; rf_out may be NULL on the last step.  as_rollout_step splits the contraction (and, for bf16 with
 * h % 4 == 0, the heads in groups of four: the streamed-operand kernel) over several workgroups when given a workspace
 * of as_rollout_step_workspace_bytes (room for 16 partial products, summed in a fixed order by a second tiny launch:
 * deterministic); with workspace == NULL it runs unsplit. */
size_t as_rollout_rfrag_bytes(int B, int N, int dtype);
/* rf = the fragment-major copy of a row-major R [B,T,N] fp32: lets a caller continue the roll-out from a row SUBSET of a
 * previous result (the RoI head only consumes the rows of the matched point tokens, stdroi:2272 + the pos_inds gather). */
int as_rollout_pack(const float* R, void* rf, int B, int N, int T, int dtype, as_stream_t stream);
size_t as_rollout_step_workspace_bytes(int B, int N, int T);
int as_rollout_top(const void* q, const void* k, const float* lse, float* R_out, void* rf_out, int B, int N, int h,
                   int T, int dtype, as_stream_t stream);
int as_rollout_step(const void* q, const void* k, const float* lse, const float* R_in, const void* rf_in,
                    float* R_out, void* rf_out, void* workspace, size_t workspace_bytes, int B, int N, int h, int T,
                    int dtype, as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Part B -- attention-shift pseudo-label generator (stdroi:2209-2415), fp32 + int32
 * ------------------------------------------------------------------------------------------- */

/* 8-connectivity connected components of M binary images; label = 1 + min raster index of the
 * component, background 0.  Replaces cc_torch.connected_components_labeling (stdroi:68). */
int as_ccl_2d(const uint8_t* img /*[M,H,W]*/, int32_t* labels /*[M,H,W]*/, int M, int H, int W,
              as_stream_t stream);

/* CAM -> box (stdroi:2272-2294 + get_bbox_from_cam_fast :60-116), batched over M = Lc*G maps:
 * bilinear x`up` upsample (align_corners=False), min-max normalise (clamp 1e-6), binarise at
 * cam_thr, CCL, keep components with area >= area_ratio * max area, tight box, 'expand' about the
 * point, clip to the image.  boxes [M,4] fp32 (x0,y0,x1,y1); status[m] = number of kept pixels
 * (0 = the reference would have raised; -1 = a row held more foreground runs than the workspace
 * provides, result invalid).  cams_up, if not NULL, receives the upsampled maps
 * [M,H,W] fp32 (the reference keeps them as attn_maps_dealed, stdroi:2282); minmax, if not NULL,
 * their per-map (min, max) [M,2] (reused by norm_attns, stdroi:329). */
size_t as_cam_boxes_workspace_bytes(int M, int Hp, int Wp, int up);
int as_cam_boxes(const float* cams /*[M,Hp,Wp]*/, const float* points /*[M,2] (x,y)*/, float cam_thr,
                 float area_ratio, int M, int Hp, int Wp, int up, float* boxes, int32_t* status,
                 float* cams_up, float* minmax, void* ws, size_t ws_bytes, as_stream_t stream);

/* Candidate masks of the seed sampling (stdroi:1003-1007 -> sample_point_grid :343-371 on norm_attns
 * :329-333) for the G selected maps of one image, straight from the low-resolution CAMs:
 *   nm_g = (up(cams[map_idx[g]]) - min) / (max - min)   with (min, max) = minmax[map_idx[g]] of as_cam_boxes
 *   masks [2G+1, H*W] uint8: rows 0..G-1  nm_g < thr_bg,  rows G..2G-1  nm_g >= thr_fg,
 *                            row 2G       mean_g(nm_g) < thr_bg
 *   counts [2G+1] int32: set pixels per row (written by the call); may be NULL (no counting, no fill launch).
 * The upsampled maps are never materialised (same bilinear arithmetic as as_cam_boxes). */
size_t as_cam_sample_masks_workspace_bytes(int G, int Hp, int Wp, int up);     /* 0 for G <= 8 */
int as_cam_sample_masks(const float* cams /*[M,Hp,Wp]*/, const int32_t* map_idx /*[G]*/,
                        const float* minmax /*[M,2]*/, int G, int Hp, int Wp, int up, float thr_bg, float thr_fg,
                        uint8_t* masks, int32_t* counts, void* ws, size_t ws_bytes, as_stream_t stream);

/* The three candidate masks of one image's G objects in one call (stdroi:433-461, :2356-2358); crops [G,4] int32
 * half-open (x0,y0,x1,y1):
 *   pos    [G,H,W] uint8 = erode_k(in_crop && map_fg > max_crop(map_fg) * pos_thr)
 *   neg    [G,H,W] uint8 =          in_crop && map_bg > max_crop(map_bg) * neg_thr
 *   pseudo [G,H,W] uint8 =                     map_fg > max(map_fg)      * mask_thr
 *   counts [3,G] int32 set pixels of pos / neg / pseudo (written by the call). */
size_t as_mask_candidates_workspace_bytes(int G, int H, int W);
int as_mask_candidates(const float* map_fg, const float* map_bg, const int32_t* crops, float pos_thr, float neg_thr,
                       float mask_thr, int k, uint8_t* pos, uint8_t* neg, uint8_t* pseudo, int32_t* counts, void* ws,
                       size_t ws_bytes, int G, int H, int W, as_stream_t stream);

/* Per-part statistics of get_center_coord_with_feat (stdroi:222-262) for M part maps [M, Hp*Wp]; rois [G,4],
 * owner [M] = object of each part: out_c [M,2] = ((x, y) of the peak centroid + 0.5) * stride, out_yx [M,2] = its
 * integer (y, x), out_area [M] = pixels > 0.9, out_inside [M] = centre inside the owner's box. */
int as_part_stats(const float* maps, const float* rois, const int32_t* owner, float stride, float* out_c,
                  int32_t* out_yx, int32_t* out_area, uint8_t* out_inside, int M, int Hp, int Wp, as_stream_t stream);

/* Part selection of get_center_coord_with_feat (stdroi:222-262) for all G objects x P slots on the device: slots
 * p < ngroups[g] visited in stable descending-area order, taken if inside and position <= num_points; taken slots
 * compacted in (object, visiting) order into rows of coords / coords_org [G*P,2], out_labels / labels_org / corres
 * [G*P] int64, feats [G*P,C] = feat_tok[yx.y * Wp + yx.x]; split [G+1] = per-object counts and their total; sel_slot
 * [G*P] scratch.  area / inside / c / yx as as_part_stats writes them.  P <= 256. */
int as_part_select(const int32_t* area, const uint8_t* inside, const int32_t* ngroups, const float* c, const int32_t* yx,
                   const int64_t* labels, const float* feat_tok, int G, int P, int C, int Wp, int num_points, float* coords,
                   float* coords_org, int64_t* out_labels, int64_t* labels_org, int64_t* corres, float* feats,
                   int32_t* sel_slot, int32_t* split, as_stream_t stream);

/* filter_maps (stdroi:263-271) for G*P prototypes: keep[g,p] = share of sim[g,p] > sim_thr lying on fg_inter[g] >= pos_thr. */
int as_filter_parts(const float* sim /*[G,P,Np]*/, const float* fg_inter /*[G,Np]*/, float sim_thr, float pos_thr,
                    uint8_t* keep /*[G,P]*/, int G, int P, int Np, as_stream_t stream);

/* Device-side draw of the mask points (fast-RNG mode; the head of a random permutation, stdroi:447): the first K
 * distinct values of floor(u[g,:] * n_g), n_g = n_pos + n_neg of object g read at counts[g * count_stride_g] and
 * counts[g * count_stride_g + count_stride_k] (so both a [G,2] table and the [3,G] table of as_mask_candidates are read in
 * place), split into ranks among the positive / negative candidates.  *flag (int32, zeroed by the caller) is OR-ed with 1
 * when some object needs the host path (n_g < 4K or too few distinct draws). */
int as_draw_distinct(const int32_t* counts, int count_stride_g, int count_stride_k, const float* u /*[G,M]*/,
                     int32_t* rank_pos /*[G,K]*/, int32_t* rank_neg, uint8_t* is_pos, int32_t* flag, int G, int M, int K,
                     as_stream_t stream);

/* The reference-RNG mode's draws made on the device (csrc/mt19937.hip): `state` = torch's CPU mt19937 engine as int32[626]
 * (624 state words, left, next: attentionshift_amd/mt19937.py), advanced in place exactly as the reference's host calls
 * advance the global generator.  as_mt_sample_ranks: for S candidate sets with counts[s] members,
 * ranks[s, 0:K] = (torch.randint(n, (len(range(0, n, n // K)),)) % n)[:K]  (sample_point_grid, stdroi:343-371).
 * as_mt_perm_ranks: for G objects with n = counts2[g,0] + counts2[g,1] candidates, ranks[g, 0:K] = torch.randperm(n)[:K]
 * (get_mask_points_single_instance, stdroi:447).  flag (int32, written) != 0: a set the host path must handle (fewer
 * than K candidates -- the reference's refill / fill-in / empty branches -- or a range beyond the one-word draws). */
int as_mt_sample_ranks(int32_t* state /*[626]*/, const int32_t* counts /*[S]*/, int32_t* ranks /*[S,K]*/, int32_t* flag,
                       int S, int K, as_stream_t stream);
int as_mt_perm_ranks(int32_t* state /*[626]*/, const int32_t* counts2 /*[G,2]*/, int32_t* ranks /*[G,K]*/, int32_t* flag,
                     int G, int K, as_stream_t stream);

/* Greedy grouping of merge_maps (stdroi:278-294) for G objects: keep [G,P] uint8, link [G,P,P] uint8 (cos >= thr)
 * -> groups [G,P] int32 bit sets over the prototype ids, in the reference's emission order (0 = unused), ngroups [G]. */
int as_merge_plan(const uint8_t* keep, const uint8_t* link, int32_t* groups, int32_t* ngroups, int G, int P,
                  as_stream_t stream);

/* Patch-grid foreground of get_semantic_centers (stdroi:2011-2012, 2020), one launch:
 *   fg_inter [G,Hp*Wp] = bilinear x(1/up) of erode_k(map_fg > thr)   (map_fg [G, Hp*up, Wp*up])
 *   mask     [G,Hp*Wp] uint8 = fg_inter > thr,  counts [G] = set entries per object (the grid-seed candidates, :1784);
 *   counts may be NULL (no counting, no fill launch) */
int as_semantic_prestage(const float* map_fg, float thr, int k, int G, int Hp, int Wp, int up, float* fg_inter,
                         uint8_t* mask, int32_t* counts, as_stream_t stream);

/* Mean-shift token clustering (stdroi:830-854 cosine_shift_batch + :882-908 update_density_batch,
 * with the box masking of :1819-1824 folded in):
 *   feat      [B,Np,C]  token-major ViT features (Np = Hp*Wp)
 *   box_patch [G,4]     inclusive patch-grid box (x0,y0,x1,y1) of each object = rois // 16
 *   obj_img   [G]       image index of each object
 *   prot_in   [G,P,C]   seed prototypes;  prot_out [G,P,C] shifted prototypes (unnormalised, as the reference); the
 *                       two may be the same buffer
 *   sim_out   [G,P,Np]  cos(prot, UNMASKED feat) (not clamped)
 *   assign_out[S,G,Np]  (optional) argmax prototype per patch per iteration, ties -> lowest index
 *   tau_out   [S,G,P]   (optional) per-prototype density after each iteration
 *   tau0, temp          doubles: the first iteration divides by the python float temp*tau0 rounded once to fp32, exactly
 *                       as `sim_map/(temp*tau)` with scalar arguments does (stdroi:834) */
size_t as_cosine_shift_workspace_bytes(int B, int C, int Hp, int Wp, int G, int P);
int as_cosine_shift(const float* feat, const int32_t* box_patch, const int32_t* obj_img, const float* prot_in,
                    float* prot_out, double tau0, double temp, int n_shift, float* sim_out, int32_t* assign_out,
                    float* tau_out, void* ws, size_t ws_bytes, int B, int C, int Hp, int Wp, int G, int P,
                    as_stream_t stream);
/* The same call on a feature tensor whose images are `feat_batch_stride` floats apart (>= Np*C, a multiple of 4): the
 * reference's caller passes vit_feat as a VIEW of last_feat [B, 1 + Np, C] without its cls row
 * (mmdet/models/detectors/two_stage_point_align.py:77), whose per-image token blocks are contiguous but not adjacent. */
int as_cosine_shift_strided(const float* feat, long long feat_batch_stride, const int32_t* box_patch, const int32_t* obj_img,
                            const float* prot_in, float* prot_out, double tau0, double temp, int n_shift, float* sim_out,
                            int32_t* assign_out, float* tau_out, void* ws, size_t ws_bytes, int B, int C, int Hp, int Wp, int G,
                            int P, as_stream_t stream);

/* Cosine-affinity refinement on the patch grid (stdroi:668-707 get_refined_similarity):
 *   feat   [Np,C] one image, seeds [Gp,C] (mean feature of the sampled points, :335-338)
 *   boxes  [G,4] inclusive patch boxes of the first G maps (selection group only)
 *   n_select  the first n_select maps form the selection group of `is_select=True` (box masking of its first G maps,
 *             keep-the-winner among the group, :676-683 / :697-703); 0 = none (`is_select=False`), Gp = all, a value in
 *             between refines the foreground and the background seed sets of an image in ONE call (group first)
 *   maps   [R+1,Gp,Np] ;  seeds_out [Gp,C] = refined seed features of the last round (with refine_times = 0 a copy of
 *   `seeds`, skipped when seeds_out == seeds) */
size_t as_refine_similarity_workspace_bytes(int C, int Np, int Gp);
int as_refine_similarity(const float* feat, const float* seeds, const int32_t* boxes, int G, int Gp,
                         int refine_times, float tau, int n_select, float* maps, float* seeds_out, void* ws,
                         size_t ws_bytes, int C, int Hp, int Wp, as_stream_t stream);

/* Full-resolution instance maps from the patch-grid refinement (stdroi:1010-1019):
 * bilinear x`up` of fg/bg levels, (1-bg)*fg, per-map max normalisation, decouple_instance.
 *   sim_fg [L,Gp,Np] (first G maps used), sim_bg [L,G,Np] -> map_fg, map_bg [L,G,H,W] */
size_t as_instance_maps_workspace_bytes(int L, int G);
int as_instance_maps(const float* sim_fg, const float* sim_bg, int L, int G, int Gp, int Hp, int Wp, int up,
                     float* map_fg, float* map_bg, void* ws, size_t ws_bytes, as_stream_t stream);

/* Thresholded + eroded candidate masks inside per-map crops: the fg candidates of
 * get_mask_points_single_box_cos_map_fg_bg (stdroi:442, erode(map > max*thr, 21) on the box crop), its bg
 * candidates (:443, k = 1) and the full-map erosion of get_semantic_centers (:2011, crops = NULL, absolute thr):
 *   mask[m][y][x] = 1 iff (y,x) in crop_m and min over the k x k window (restricted to the crop) of [map > t_m]
 *   t_m = relative ? thr * max(map over crop_m) : thr;  crops [M,4] int32 (x0,y0,x1,y1), half-open like the
 *   reference's slicing [y0:y1, x0:x1];  counts[m] = number of set pixels. */
size_t as_crop_threshold_erode_workspace_bytes(int M, int H, int W);
int as_crop_threshold_erode(const float* maps /*[M,H,W]*/, const int32_t* crops, float thr, int relative, int k,
                            uint8_t* mask /*[M,H,W]*/, int32_t* counts /*[M]*/, void* ws, size_t ws_bytes, int M, int H,
                            int W, as_stream_t stream);

/* Rank select: out[m][k] = flat index of the ranks[m][k]-th (0-based) set byte of mask[m] in raster order, i.e.
 * `mask[m].nonzero()[ranks[m][k]]` without the compaction (the reference indexes a .nonzero() list with random
 * indices, stdroi:368-369 and :456); -1 when the rank is outside the population.  HW % 16 == 0. */
/* counts[m] = number of non-zero bytes of row m of a [M,HW] byte mask (the candidate counts of stdroi:346-366 that the
 * reference gets from .nonzero().shape); HW % 16 == 0. */
int as_mask_count(const uint8_t* mask, int32_t* counts, int M, int HW, as_stream_t stream);

size_t as_rank_select_workspace_bytes(int M, int HW);
int as_rank_select(const uint8_t* mask /*[M,HW] 0/1*/, const int32_t* ranks /*[M,K]*/, int32_t* out /*[M,K]*/, void* ws,
                   size_t ws_bytes, int M, int HW, int K, as_stream_t stream);
/* The same selection returning grid coordinates: out_xy [M,K,2] int64 = (flat % W, flat / W), or (flat / W, flat % W)
 * with yx_order, of the selected pixel on a W-wide grid; an out-of-range rank gives pixel 0 (the `clamp(min=0)` of the
 * callers, which flag that case separately). */
int as_rank_select_xy(const uint8_t* mask /*[M,HW] 0/1*/, const int32_t* ranks /*[M,K]*/, int64_t* out_xy /*[M,K,2]*/, void* ws,
                      size_t ws_bytes, int M, int HW, int K, int W, int yx_order, as_stream_t stream);

/* merge_maps (stdroi:278-294) of every object in one launch: cosine links between the kept prototypes (>= thr), the greedy
 * upper-triangular grouping of as_merge_plan, and the merged prototypes matmul(weight, prot) / (weight.sum(-1) + 1e-8) of
 * the first `slots` groups per object (unused slots are zero rows).  ngroups [G] = min(groups, slots); *flag (may be NULL)
 * is OR-ed with 1 when an object has more than `slots` groups.  prot [G,P,C] fp32, keep [G,P] 0/1, P <= 32. */
int as_merge_parts(const float* prot, const uint8_t* keep, float thr, float* merged /*[G,slots,C]*/, int32_t* ngroups /*[G]*/,
                   int32_t* flag, int G, int P, int C, int slots, as_stream_t stream);

/* The default layer selector and its index arithmetic in one launch (roi_head.median_area_selector standing in for the MIL
 * head's choice, stdroi:2953-2972; the patch box of stdroi:1812): per object o with meta[o] = (first row of its image in
 * `boxes`, objects in that image, index in the image) and the image's boxes layer-major ([Lc, cnt, 4] from as_cam_boxes),
 *   pick[o]      the layer whose box area max(x1-x0,0)*max(y1-y0,0) has stable ascending rank (Lc-1)/2 -- or, with
 *                `pick_in` [n] (may be NULL), the layer another selector chose (the trained MIL head), clamped to [0, Lc)
 *   chosen[o]    that box;  map_idx[o] its row in `boxes` (= the row of the CAM stack);  box_patch[o] = floor(box / stride);
 *   box_int[o]   (may be NULL) the box truncated to integers, the crop of stdroi:1981
 * `status` (may be NULL) [rows] int32, the per-box status of as_cam_boxes: *bad (zeroed by the caller) is OR-ed with 1 when any
 * row of any listed object has status <= 0 (a CAM without a foreground component -- where stdroi:80 raises). */
int as_select_median_boxes(const float* boxes /*[rows,4]*/, const int32_t* meta /*[n,3]*/, int Lc, int stride,
                           const int64_t* pick_in, int64_t* pick /*[n]*/, float* chosen /*[n,4]*/, int32_t* map_idx /*[n]*/, int32_t* box_patch /*[n,4]*/,
                           int32_t* box_int /*[n,4]*/, const int32_t* status, int32_t* bad, int n, as_stream_t stream);

/* Rank selection whose ranks are derived on the device from each row's population n (known from the same counting pass), so
 * that the sampling chains need no tensor-op glue and no readback (fast-RNG mode of the seed sampling, stdroi:346-369, and the
 * grid seeds of mean_shift_grid_prototype, stdroi:1790-1792):
 *   mode 1  rank[m][k] = min(int(u[m][k] * float(n_m)), max(n_m - 1, 0))     u [M,K] uniform in [0, 1)
 *   mode 2  rank[m][k] = k * max(n_m / K, 1)                                 every (n / K)-th positive
 * out_xy as as_rank_select_xy (may be NULL if patch_out is given); *flag (int32, may be NULL) is OR-ed with 1 when some
 * n_m < K (the reference's refill branches: the caller repeats that image on the host path).
 * patch_out (may be NULL) [M,K] int64: patch_base + (y / patch_div) * patch_w + x / patch_div of the selected pixel (x, y),
 * i.e. its token index on a patch_w-wide grid of patch_div-pixel patches (`point // 16` of seed_features stdroi:335-338, or
 * with patch_div = 1 the flat seed index of stdroi:1793), row m written to row (m + M - row_rot) % M. */
int as_rank_draw_xy(const uint8_t* mask /*[M,HW] 0/1*/, int mode, const float* u /*[M,K] or NULL*/, int32_t* flag,
                    int64_t* out_xy /*[M,K,2]*/, int64_t* patch_out /*[M,K]*/, int patch_div, int patch_w, int64_t patch_base,
                    int row_rot, void* ws, size_t ws_bytes, int M, int HW, int K, int W, int yx_order, as_stream_t stream);

/* Small-N batched multi-head self-attention (the MAE-decoder box / mask heads: thousands of 50- / 197-token problems of
 * head dim 32 per step; models/vision_transformer.py:62-86 as used by mae_bbox_head_rec.py:148-168):
 *   qkv  [Bp, N, 3, h, d]  the packed output of the reference's qkv Linear (fp32 or bf16), d = 32
 *   out  [Bp, N, h*d]      softmax(q k^T d^-0.5) v, same dtype;   lse [Bp, h, N] fp32 (kept for the backward)
 *   dqkv [Bp, N, 3, h, d]  gradient of qkv given d_out (recomputes the probabilities from q, k, lse; no atomics). */
int as_small_attn_fwd(const void* qkv, void* out, float* lse, int Bp, int N, int h, int d, int dtype, as_stream_t stream);
size_t as_small_attn_bwd_workspace_bytes(int Bp, int N, int h);
int as_small_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv, void* workspace,
                      size_t workspace_bytes, int Bp, int N, int h, int d, int dtype, as_stream_t stream);

/* RoIAlign on the stride-16 feature map (mmcv.ops.RoIAlign as configured at
 * configs/mae/attnshift_voc12aug.py:64-68, 123-127; call sites stdroi:2958 and the RoI head's bbox / mask forward):
 * adaptive sampling (sampling_ratio = 0 -> ceil(roi size / out) samples per bin and axis), aligned half-pixel shift,
 * average pooling.  Token-major layouts:
 *   feat [B,H,W,C] fp32, rois [R,5] = (batch index, x1, y1, x2, y2) in image coordinates, out [R, out*out, C] fp32
 *   backward: dfeat [B,H,W,C] is written completely by a deterministic gather (no atomics; C a multiple of 8, maps of up
 *   to 8192 pixels) */
int as_roi_align_fwd(const float* feat, const float* rois, float* out, int B, int H, int W, int C, int R, int out_size,
                     float spatial_scale, int sampling_ratio, int aligned, as_stream_t stream);
int as_roi_align_bwd(const float* dout, const float* rois, float* dfeat, int B, int H, int W, int C, int R, int out_size,
                     float spatial_scale, int sampling_ratio, int aligned, as_stream_t stream);

/* 2-D chamfer distance, the reference's second native op (mmdet/ops/chamfer_2d/src/chamfer_2d.cu:12-161 behind
 * mmdet/ops/chamfer_2d/dist_chamfer_2d.py:11-58; off the hot path): xyz1 [B,n,2], xyz2 [B,m,2] fp32 ->
 * dist1 [B,n] / dist2 [B,m] = squared distance to the nearest point of the other set, idx1 / idx2 int32 its index
 * (lowest on ties); backward: gxyz += 2 * gdist * (p - q) on the point, -= on its neighbour (both outputs written). */
int as_chamfer_2d_fwd(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                      int B, int n, int m, as_stream_t stream);
int as_chamfer_2d_bwd(const float* xyz1, const float* xyz2, const float* gdist1, const float* gdist2, const int32_t* idx1,
                      const int32_t* idx2, float* gxyz1, float* gxyz2, int B, int n, int m, as_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ATTNSHIFT_H */
