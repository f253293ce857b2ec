/* msocr.h — C ABI of libmsocr.so, the MI355X (gfx950) hot path of manuscript-ocr.
 *
 * The reference (olegiy/manuscript-ocr) has no FFI: its boundary is the Python
 * plugin API (Pipeline / EAST / TRBA).  This header is the INNER native boundary
 * that the Python host in manuscript_ocr_amd/ binds with ctypes; every entry
 * point names the reference code it replaces (paths relative to
 * /root/reference/src/manuscript/).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and
 *     stream-ordered, allocates nothing and never synchronises (graph-capturable);
 *   - activations are NHWC; `dtype` selects the storage/MFMA input type
 *     (MSOCR_F32: v_mfma_f32_32x32x2_f32, exact f32; MSOCR_BF16:
 *     v_mfma_f32_32x32x16_bf16, f32 accumulate);
 *   - return value: 0 = ok, negative = MSOCR_E_* (nothing was launched).
 */
#ifndef MSOCR_H
#define MSOCR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSOCR_F32 0
#define MSOCR_BF16 1

#define MSOCR_OK 0
#define MSOCR_E_ARG (-1)     /* bad argument / unsupported shape */
#define MSOCR_E_LAUNCH (-2)  /* hipLaunch failed */
#define MSOCR_E_NOGPU (-3)

/* flags for msocr_conv2d */
#define MSOCR_CONV_RELU 1u
#define MSOCR_CONV_RESIDUAL 2u /* out = act(conv + bias + residual) */
#define MSOCR_CONV_POOL2 4u    /* msocr_conv3x3_winograd42_fused only: out = maxpool2x2/2(act(conv + bias)), out is [N][H/2][W/2] */

typedef struct msocr_conv_desc {
  int32_t dtype;                 /* MSOCR_F32 | MSOCR_BF16 (input, weight, residual, output) */
  int32_t N, H, W, Cin;          /* input extent; Cin = channels reduced per tap (multiple of 32 f32 / 32 bf16) */
  int64_t in_sN, in_sH, in_sW;   /* input strides in ELEMENTS (channel stride is 1) */
  int32_t KH, KW, stride_h, stride_w, pad_h, pad_w;
  int32_t Ho, Wo, Cout;          /* Cout multiple of 32 */
  int64_t out_ld;                /* elements between consecutive output pixels (>= Cout) */
  int64_t res_ld;                /* same for the residual tensor */
  uint32_t flags;
} msocr_conv_desc;

/* Implicit-GEMM convolution on MFMA: out[n,ho,wo,co] = act(sum_{kh,kw,c} in[n,ho*sh-ph+kh,wo*sw-pw+kw,c] *
 * w[co,kh,kw,c] + bias[co] (+ residual)).  weight layout [Cout][KH][KW][Cin] (dtype), bias f32 with BatchNorm
 * folded in by the host.  Replaces every nn.Conv2d+BatchNorm2d(+ReLU)(+add) of
 *   detectors/_east/east.py:13-30,56-67,96-105 (torchvision ResNet-50 Bottlenecks, DecoderBlock) and
 *   recognizers/_trba/model/seresnet31.py:37-45,81-89,129-136,150-155. */
int msocr_conv2d(const msocr_conv_desc* d, const void* in, const void* weight, const float* bias,
                 const void* residual, void* out, void* stream);

/* The same convolution for KH=KW=3, stride 1, pad 1, MSOCR_F32, as Winograd F(2x2,3x3): input transform (HBM-bound) ->
 * 16 GEMMs [tiles x Cin] x [Cin x Cout] in one MFMA launch -> output transform + bias/residual/ReLU (HBM-bound).
 * 2.25x fewer matrix FLOPs than the direct form; all arithmetic f32 (differs from msocr_conv2d by rounding order only).
 * u_weight = [16][Cout][Cin] f32 on the DEVICE, produced by msocr_winograd_weights_host (a HOST function: both of
 * its pointers are host memory; evaluates G g G^T in f64, rounds once) from the BatchNorm-folded [Cout][3][3][Cin]
 * weight.  workspace: msocr_conv3x3_winograd_workspace_bytes(d) bytes, 16-B aligned (-1 = shape not supported:
 * Cin % 16, Cout % 32).  Same reference code as msocr_conv2d (every 3x3/1/1 conv of SE-ResNet31, ResNet-50 and the
 * EAST decoder). */
int64_t msocr_conv3x3_winograd_workspace_bytes(const msocr_conv_desc* d);
int msocr_conv3x3_winograd(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                           const void* residual, void* out, void* workspace, void* stream);
int msocr_winograd_weights_host(const float* w_khwc_host, int Cout, int Cin, float* u_out_host);
/* The three stages of msocr_conv3x3_winograd one by one (identical kernels, identical results when called in this order on one
 * stream with the same workspace): V = B^T d B into the workspace; Mw[p] = V[p] U[p]^T (16 GEMMs, one MFMA launch);
 * out = act(A^T Mw A + bias (+ residual)).  For tests and for the per-kernel rooflines of bench.py. */
int msocr_winograd_input_transform(const msocr_conv_desc* d, const void* in, void* workspace, void* stream);
int msocr_winograd_gemm(const msocr_conv_desc* d, const float* u_weight, void* workspace, void* stream);
int msocr_winograd_output_transform(const msocr_conv_desc* d, const void* workspace, const float* bias, const void* residual,
                                    void* out, void* stream);

/* The same convolution in the TALL Winograd form F(4,3) x F(2,3) (6x4 input tile -> 4x2 output tile, 24 transform points:
 * 3 multiplies per output where F(2x2,3x3) spends 4 and the direct form 9; V / Mw are 3x the layer's arrays instead of 4x).
 * Only the H axis takes the 6-point transform, whose constants grow the layer's f32 rounding error ~2.5x rms over F(2x2)
 * (DESIGN.md section 4).  u_weight = [24][Cout][Cin] f32 on the DEVICE from msocr_winograd42_weights_host (HOST function, f64,
 * rounded once).  Same shape rules, workspace convention and stage entry points as msocr_conv3x3_winograd. */
int64_t msocr_conv3x3_winograd42_workspace_bytes(const msocr_conv_desc* d);
int msocr_conv3x3_winograd42(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                             const void* residual, void* out, void* workspace, void* stream);
int msocr_winograd42_weights_host(const float* w_khwc_host, int Cout, int Cin, float* u_out_host);
int msocr_winograd42_input_transform(const msocr_conv_desc* d, const void* in, void* workspace, void* stream);
int msocr_winograd42_gemm(const msocr_conv_desc* d, const float* u_weight, void* workspace, void* stream);
int msocr_winograd42_output_transform(const msocr_conv_desc* d, const void* workspace, const float* bias, const void* residual,
                                      void* out, void* stream);

/* ---- split-operand f32 ("bf16x3"): the default arithmetic of the f32 1x1 convolutions and Winograd-domain GEMMs --------------
 * gfx950 runs exact-f32 MFMA at 1/16 of the bf16 rate.  An f32 value is the exact sum of three bf16 values (round-to-nearest
 * residual chain); of the nine cross products of two such sums the six largest are accumulated in f32 on the bf16 matrix pipes,
 * the three dropped ones are <= 2^-25 of the product (below the rounding of the f32 accumulation itself).  The activation operand
 * is split in registers inside the kernel; the WEIGHT operand is split once at load time:
 *   msocr_split_bf16x3_host(w, n, planes)   HOST: w [n] f32 -> planes [3][n] bf16 (uint16), w == p0 + p1 + p2 exactly.
 *   msocr_split_bf16x3_ktile_host(w, nb, rows, k, planes)   HOST: the same split of w [nb][rows][k] (k % 32 == 0) written
 *     K-TILE-MAJOR, planes [3][nb][k/32][rows][32]: the 32-element pieces of all rows for one K-tile are contiguous (64 bytes per row),
 *     so a workgroup's weight tile of one K-tile is ONE dense block — whole 128-byte lines, every byte used.  (Row-major planes
 *     gave 64-byte pieces at a stride of 2 k bytes: half of every fetched line belonged to the NEXT K-tile and was fetched again —
 *     TCP -> L2 read requests 2x the algorithmic count, profiles/r04_pp_ablations.txt.)  This is the layout the three GEMM entry
 *     points below take ("weight_planes" / "u_planes"; k = KH * KW * Cin in the weight's own [KH][KW][Cin] order).
 * msocr_conv1x1_split: msocr_conv2d for KH = KW = 1 / stride 1 / no padding / MSOCR_F32 over a dense pixel sequence
 *   (in_sH == W * in_sW, in_sN == H * in_sH), Cin % 32 == 0, Cout % 64 == 0; weight_planes = K-tile-major planes of [1][Cout][Cin] on the device.
 *   Same flags, epilogue and reference layers as msocr_conv2d (torchvision Bottleneck conv1 / conv3 / downsample, DecoderBlock
 *   conv1x1, SEBasicBlock downsample, the BiLSTM input projections and linears).
 * msocr_conv2d_split: msocr_conv2d for MSOCR_F32 with any kernel size / stride / padding (the strided 3x3 and 1x1 convolutions of
 *   the two ResNet trunks, which have no Winograd form), weight_planes = K-tile-major planes of [1][Cout][KH*KW*Cin]; Cin % 32 == 0, Cout % 64 == 0.
 * msocr_winograd42_gemm_split / msocr_conv3x3_winograd42_split: stage 2 of / the whole msocr_conv3x3_winograd42 with
 *   u_planes = K-tile-major planes ([3][24][Cin/32][Cout][32] bf16) of msocr_winograd42_weights_host's [24][Cout][Cin] output
 *   (Cin % 32, Cout % 64).
 * Results differ from the exact-f32 entry points by rounding only (tests/test_gpu_ops.py bounds both against an f64 reference). */
int msocr_split_bf16x3_host(const float* w_host, int64_t n, uint16_t* planes_out_host);
int msocr_split_bf16x3_ktile_host(const float* w_host, int64_t nbatch, int64_t rows, int64_t k, uint16_t* planes_out_host);
int msocr_conv1x1_split(const msocr_conv_desc* d, const void* in, const void* weight_planes, const float* bias,
                        const void* residual, void* out, void* stream);
int msocr_conv2d_split(const msocr_conv_desc* d, const void* in, const void* weight_planes, const float* bias,
                       const void* residual, void* out, void* stream);
int msocr_winograd42_gemm_split(const msocr_conv_desc* d, const void* u_planes, void* workspace, void* stream);
int msocr_conv3x3_winograd42_split(const msocr_conv_desc* d, const void* in, const void* u_planes, const float* bias,
                                   const void* residual, void* out, void* workspace, void* stream);

/* F(4,3) x F(4,3) on the interpolation points {0, +-3/2, +-2/3, inf} (round 4): 36 transform points per 4 x 4 outputs — 2.25
 * multiplies and workspace words per output instead of the tall form's 3 — at the tall form's rounding error (the textbook points
 * {0, +-1, +-2} would cost 4.7x).  Same contract as the msocr_*winograd42* entry points: workspace =
 * msocr_conv3x3_winograd44_workspace_bytes (V [36][tiles][Cin] f32 + Mw [36][tiles][Cout] f32, tiles = N ceil(H/4) ceil(W/4));
 * u_planes = K-tile-major bf16 planes ([3][36][Cin/32][Cout][32], msocr_split_bf16x3_ktile_host) of msocr_winograd44_weights_host's
 * [36][Cout][Cin] f32 output (HOST function, f64, rounded once); Cin % 32 == 0, Cout % 64 == 0; the 36 GEMMs run with split
 * operands on the bf16 matrix pipes.  The three stages are also callable one by one (same workspace). */
int64_t msocr_conv3x3_winograd44_workspace_bytes(const msocr_conv_desc* d);
int msocr_winograd44_weights_host(const float* w_khwc_host, int Cout, int Cin, float* u_out_host);
int msocr_winograd44_input_transform(const msocr_conv_desc* d, const void* in, void* workspace, void* stream);
int msocr_winograd44_gemm_split(const msocr_conv_desc* d, const void* u_planes, void* workspace, void* stream);
int msocr_winograd44_output_transform(const msocr_conv_desc* d, const void* workspace, const float* bias, const void* residual,
                                      void* out, void* stream);
int msocr_conv3x3_winograd44_split(const msocr_conv_desc* d, const void* in, const void* u_planes, const float* bias,
                                   const void* residual, void* out, void* workspace, void* stream);

/* Cin == 64: the tall Winograd form with the 24 transform-domain GEMMs (K = 64) and the output transform fused in one kernel, so
 * Mw never reaches HBM (unfused, a 64-channel layer is HBM-bound on Mw).  workspace holds V only
 * (msocr_conv3x3_winograd42_fused_workspace_bytes; -1 = unsupported: Cin != 64, Cout % 32, or POOL2 with odd H / W or a residual).
 * u_weight as for msocr_conv3x3_winograd42.  With MSOCR_CONV_POOL2 in d->flags the kernel also applies the 2x2 / stride-2 max-pool
 * that closes conv0 of SE-ResNet31 (recognizers/_trba/model/seresnet31.py:81-89: conv3x3 64->128, BN, ReLU, MaxPool2d(2, 2)) and writes the pooled
 * [N][H/2][W/2][Cout] map (d->out_ld = its channel stride).  msocr_winograd42_fused_gemm_output is stage 2 alone (stage 1 =
 * msocr_winograd42_input_transform into the same workspace). */
int64_t msocr_conv3x3_winograd42_fused_workspace_bytes(const msocr_conv_desc* d);
int msocr_conv3x3_winograd42_fused(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                                   const void* residual, void* out, void* workspace, void* stream);
int msocr_winograd42_fused_gemm_output(const msocr_conv_desc* d, const float* u_weight, const void* workspace, const float* bias,
                                       const void* residual, void* out, void* stream);
/* The same two entry points with the 24 K = 64 GEMMs on the bf16 matrix pipes (split-operand arithmetic, see msocr_conv1x1_split):
 * u_planes = [3][24][Cout][64] bf16 = msocr_split_bf16x3_host of msocr_winograd42_weights_host's output. */
int msocr_conv3x3_winograd42_fused_split(const msocr_conv_desc* d, const void* in, const void* u_planes, const float* bias,
                                         const void* residual, void* out, void* workspace, void* stream);
int msocr_winograd42_fused_gemm_output_split(const msocr_conv_desc* d, const void* u_planes, const void* workspace, const float* bias,
                                             const void* residual, void* out, void* stream);

/* u8 RGB images (N x H x W x 3) -> normalised NHWC with C padded 3->cpad (4 or 8) inside a zero canvas
 * out[N][Hp][Wp][cpad], image origin at (pad_t, pad_l); the zero border is the stem convolution's padding.
 *   mode 0: EAST ToTensor+Normalize, detectors/_east/infer.py:127-132,305  -> (x/255 - .5)/.5
 *   mode 1: TRBA A.Normalize(.5,.5,max 255), recognizers/_trba/data/transforms.py:185-193 -> (x-127.5)*f32(1/127.5) */
int msocr_normalize_u8(const uint8_t* src, int N, int H, int W, int pad_t, int pad_l, int Hp, int Wp, int cpad,
                       int mode, int dtype, void* out, void* stream);

/* cv2.resize(img,(dw,dh)) INTER_LINEAR for u8 HxWx3 (OpenCV 11-bit fixed point), batched.
 * Replaces detectors/_east/infer.py:304 (restated from OpenCV, parity unpinned). */
int msocr_resize_linear_u8(const uint8_t* src, int N, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream);

/* MaxPool2d(k, stride=s, padding=p) on NHWC.  (torchvision resnet maxpool 3/2/1; seresnet31.py:88 2/2/0) */
int msocr_maxpool2d(const void* in, int N, int H, int W, int C, int64_t in_ld, int k, int s, int p, int dtype,
                    void* out, int Ho, int Wo, int64_t out_ld, void* stream);

/* F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) written into the first C channels of a
 * concat buffer (out_ld >= C): fuses the torch.cat of detectors/_east/east.py:87-92. */
int msocr_upsample2x_bilinear(const void* in, int N, int H, int W, int C, int64_t in_ld, int dtype, void* out,
                              int64_t out_ld, void* stream);

/* OutputHead (east.py:96-105): score = sigmoid(w_s . x + b_s), geo = W_g x + b_g from the 32-channel h1.
 * w9 = [9][32] f32 (row 0 score, rows 1..8 geo), b9 = [9] f32.  score_out [N][H][W] f32, geo_out [N][H][W][8] f32. */
int msocr_east_head(const void* h1, int64_t npix, int64_t in_ld, int dtype, const float* w9, const float* b9,
                    float* score_out, float* geo_out, void* stream);

/* decode_quads_from_maps (detectors/_east/utils.py:328-381), batched over pages: threshold (strict >),
 * quantise to q x q cells, unique in (y,x) order, decode 4 vertices + score at the cell centre.
 * cand_out [N][max_cand][9] f32, count_out [N] int32 (clamped to max_cand; overflow flag in bit 31). */
int msocr_east_decode(const float* score, const float* geo, int N, int H, int W, float thresh, float scale,
                      int quant, float* cand_out, int32_t* count_out, int max_cand, void* stream);

/* locality_aware_nms (detectors/_east/lanms.py:156-207 incl. standard_nms :133-153 and the fp64 geometry :7-130),
 * batched over pages.  cand [N][max_cand][9] f32 + counts from msocr_east_decode; boxes_out [N][max_cand][9] f32,
 * nbox_out [N] int32.  workspace: msocr_lanms_workspace_bytes(N, max_cand) bytes. */
int64_t msocr_lanms_workspace_bytes(int N, int max_cand);
int msocr_east_lanms(const float* cand, const int32_t* counts, int N, int max_cand, double iou_thr, float* boxes_out,
                     int32_t* nbox_out, void* workspace, void* stream);

/* The box filters of EAST.predict after the NMS (infer.py:340-356): expand_boxes (utils.py:384-422), scale to the original
 * page (infer.py:134-147), removal of fully contained boxes (infer.py:174-214, cv2.pointPolygonTest restated), area anomalies
 * (infer.py:216-233, np.mean / np.std with NumPy's f32 pairwise summation) and axis-aligned conversion (infer.py:149-172), with
 * NumPy's f32 operation order: bit-identical to the host implementation.  expand_w/h, scale_x = orig_w / target_w,
 * scale_y = orig_h / target_h and sigma are the Python floats of the reference (rounded to f32 where NumPy does).
 * msocr_east_box_tail: one workgroup per page on boxes [N][max_cand][9] / nbox [N] from msocr_east_lanms ->
 * out [N][max_cand][9], n_out [N] (-1 for a page with more than min(max_cand, 16384) boxes: use the host path; pages above
 * 2048 boxes keep their per-box arrays in the workspace instead of LDS); workspace:
 * msocr_east_box_tail_workspace_bytes(N, max_cand) bytes.
 * msocr_east_box_tail_host: HOST twin running the same code on the CPU (quads_host [M][9] -> out_host [<=M][9], *n_out_host). */
int64_t msocr_east_box_tail_workspace_bytes(int N, int max_cand);
int msocr_east_box_tail(const float* boxes, const int32_t* nbox, int N, int max_cand, double expand_w, double expand_h,
                        double scale_x, double scale_y, int axis_aligned_output, int remove_anomalies, double sigma, int min_count,
                        float* out, int32_t* n_out, void* workspace, void* stream);
int msocr_east_box_tail_host(const float* quads_host, int M, double expand_w, double expand_h, double scale_x, double scale_y,
                             int axis_aligned_output, int remove_anomalies, double sigma, int min_count, float* out_host,
                             int32_t* n_out_host);

/* ---- TRBA ---------------------------------------------------------------------------------------------- */

/* SELayer (recognizers/_trba/model/seresnet31.py:5-20) fused with the residual tail of SEBasicBlock.forward
 * (:61-66): out = relu(x * sigmoid(W2 relu(W1 mean_hw(x))) + identity).  x, identity, out: [N][HW][C] (ld = C).
 * w1 [C/16][C] f32, w2 [C][C/16] f32.  gate_ws: [N][C] f32 scratch. */
int msocr_se_residual(const void* x, const void* identity, int N, int HW, int C, int dtype, const float* w1,
                      const float* w2, float* gate_ws, void* out, void* stream);

/* AdaptiveAvgPool2d((1,None)) + squeeze + permute (recognizers/_trba/model/model.py:388-390):
 * [N][H][W][C] (dtype) -> [N][W][C] f32. */
int msocr_mean_over_h(const void* in, int N, int H, int W, int C, int dtype, float* out, void* stream);

/* One bidirectional LSTM layer + Linear (BidirectionalLSTM.forward, model/model.py:9-21).
 * xproj [B][T][2][4H] f32 = x W_ih^T + b_ih + b_hh for both directions (an msocr_conv2d 1x1 / GEMM);
 * w_hh_t [2][H][H][4] f32 = weight_hh^T with the 4 gates (i,f,g,o) of unit j interleaved: [dir][k][j][gate];
 * hcat_out [B][T][2H] f32. */
int msocr_bilstm_recurrent(const float* xproj, const float* w_hh_t, int B, int T, int H, float* hcat_out,
                           void* stream);
/* The same recurrence (H == 256) with the step's product h W_hh^T on the bf16 matrix pipes in the split-operand form, 32 crops per
 * workgroup (csrc/bilstm_mfma.hip).  whh_planes (device): two blocks (forward, reverse) of msocr_attn_pack_split_elems(4 H)
 * uint16 each, packed on the host by msocr_attn_pack_split_host(w_hh_t + dir * H * 4 H, 4 H, 1, .). */
int msocr_bilstm_recurrent_split(const float* xproj, const uint16_t* whh_planes, int B, int T, int H, float* hcat_out,
                                 void* stream);

typedef struct msocr_attn_weights {
  const float* h2h_wt;   /* [H][H]   h2h.weight^T */
  const float* h2h_b;    /* [H] */
  const float* score_w;  /* [H] */
  const float* wih_ctx_t;/* [H][H][4]  rnn.weight_ih[:, :H]^T, gates of unit j interleaved: [k][j][gate] */
  const float* wih_tok;  /* [V][H][4]  rnn.weight_ih[:, H:]^T (one-hot matmul == row gather): [token][j][gate] */
  const float* whh_t;    /* [H][H][4]  rnn.weight_hh^T: [k][j][gate] */
  const float* b_gates;  /* [H][4]     b_ih + b_hh: [j][gate] */
  const float* gen_wt;   /* [H][V]  generator.weight^T */
  const float* gen_b;    /* [V] */
} msocr_attn_weights;

/* Attention decode (model/model.py:34-46 cell, :227-259 greedy, :92-225 beam): one workgroup per batch row,
 * the whole step loop in ONE launch (rows are independent).  batch_H, proj_H: [B][T][H] f32 with
 * proj_H = i2h(batch_H) hoisted out of the loop (the reference recomputes it every step); H == 256, T <= 48,
 * V <= 256, steps <= 64.
 * greedy: steps = max_len+1; logits_out [B][steps][V] f32, ids_out [B][steps] i32 for ALL steps.
 * beam  : steps = max_len; per step the kernel stores every beam's temperature-scaled logits, back-pointers,
 *         tokens and the arg-max beam into `workspace`, and fin_step_out[b] = number of steps after which every
 *         beam of row b is finished (or steps).  lp_dev[steps] = f32 length-penalty factors
 *         ((5+t+1)^alpha / 6^alpha, computed by the host exactly as model.py:160) or NULL when alpha <= 0.
 *         Optional early exit (all three pointers non-NULL): chunk_id_dev[B] = index of the reference chunk (the slice of
 *         batch_size crops one model call sees) of every row, chunk_size_dev[nchunks] = rows per chunk, all of them inside
 *         this call; chunk_state_dev[2*nchunks] int32 zeroed by the caller.  A workgroup then leaves the step loop once every
 *         chunk its rows belong to is completely finished (model.py:215), so steps >= the chunk's run length are skipped.
 *         The default kernel runs the three matrix products of a step on the f32 matrix cores, 4 rows x 8 beams per
 *         workgroup (csrc/attn_beam_mfma.hip); MSOCR_BEAM_MFMA=0 selects the VALU kernel (one row per workgroup).
 * The reference stops the loop for the whole batch chunk (model.py:215,254); the host derives each row's run length
 * t_run from ids/fin_step and msocr_attn_beam_finalize walks the back-pointers from (t_run-1, best beam at t_run-1):
 * logits_out [B][steps][V] (rows t < t_run valid), ids_out [B][steps] (-1 beyond t_run). */
int msocr_attn_greedy(const float* batch_H, const float* proj_H, const msocr_attn_weights* w, int B, int T, int H,
                      int V, int steps, int sos_id, int eos_id, int blank_id, float* logits_out, int32_t* ids_out,
                      void* stream);
int64_t msocr_attn_beam_workspace_bytes(int B, int steps, int beam, int V);
int msocr_attn_beam(const float* batch_H, const float* proj_H, const msocr_attn_weights* w, int B, int T, int H,
                    int V, int steps, int beam, const float* lp_dev, float temperature, int sos_id, int eos_id,
                    int blank_id, int32_t* fin_step_out, void* workspace, const int32_t* chunk_id_dev,
                    const int32_t* chunk_size_dev, int32_t* chunk_state_dev, void* stream);
/* Optional split form of the three per-step weight matrices (device pointers, each packed by msocr_attn_pack_split_host and
 * copied to the device by the caller): with it the matrix-core kernel forms every f32 product from three bf16 terms per
 * operand on the bf16 matrix pipe (six partial products, f32 accumulation: f32 result up to 2^-25 relative per product and the
 * order of summation) instead of the 1/16-rate exact-f32 MFMA.  NULL = exact-f32 products. */
typedef struct msocr_attn_split_weights {
  const uint16_t* h2h_p;  /* from h2h_wt, N = H */
  const uint16_t* whh_p;  /* from whh_t,  N = 4 H, gate_interleaved */
  const uint16_t* gen_p;  /* from gen_wt, N = V */
} msocr_attn_split_weights;

/* msocr_attn_beam with the context half of the LSTMCell input product hoisted out of the step loop (matrix-core kernel only):
 * ctx_gates [B][T][H][4] f32 = batch_H x rnn.weight_ih[:, :H]^T with the four gates of a unit adjacent (row j*4+g of the
 * product), computed once per call by a GEMM (msocr_conv1x1_split).  W_ih[:, :H] (sum_t alpha_t batch_H_t) == sum_t alpha_t
 * (W_ih[:, :H] batch_H_t): same result up to f32 summation order, half the matrix work per step. */
int msocr_attn_beam_hoisted(const float* batch_H, const float* proj_H, const float* ctx_gates, const msocr_attn_weights* w,
                            const msocr_attn_split_weights* ws, int B, int T, int H, int V, int steps, int beam, const float* lp_dev,
                            float temperature, int sos_id, int eos_id, int blank_id, int32_t* fin_step_out, void* workspace,
                            const int32_t* chunk_id_dev, const int32_t* chunk_size_dev, int32_t* chunk_state_dev, void* stream);
/* msocr_attn_greedy on the matrix cores (round 4; Attention._greedy_decode, model.py:227-259): 32 crops per workgroup = one MFMA
 * row block, the three per-step products in the split-operand form (ws must be given), the context half of the gate product
 * hoisted as in msocr_attn_beam_hoisted.  Same outputs and argument meaning as msocr_attn_greedy; H == 256, V <= 256, T <= 48
 * (other shapes: MSOCR_E_ARG — call msocr_attn_greedy). */
int msocr_attn_greedy_hoisted(const float* batch_H, const float* proj_H, const float* ctx_gates, const msocr_attn_weights* w,
                              const msocr_attn_split_weights* ws, int B, int T, int H, int V, int steps, int sos_id, int eos_id,
                              int blank_id, float* logits_out, int32_t* ids_out, void* stream);
/* Packing of a decoder weight for the fields of msocr_attn_split_weights; HOST memory in and out.  wt: [256][N] f32 row-major (h2h_wt,
 * gen_wt) or, with gate_interleaved != 0, [256][N/4][4] (whh_t).  out: msocr_attn_pack_split_elems(N) = 3 * 256 * ceil32(N)
 * uint16 (bf16 bit patterns), laid out [plane][k / 16][column][k % 16] with wt == plane0 + plane1 + plane2 exactly. */
int64_t msocr_attn_pack_split_elems(int N);
int msocr_attn_pack_split_host(const float* wt_host, int N, int gate_interleaved, uint16_t* out_host);
int msocr_attn_beam_finalize(const void* workspace, int B, int V, int steps, int beam, const int32_t* trun_dev,
                             float* logits_out, int32_t* ids_out, void* stream);

/* Recognition confidence (recognizers/_trba/__init__.py:413-431): mean over the t_run generated positions of
 * exp(log_softmax(logits)[id]).  logits [B][steps][V] f32, ids [B][steps] i32 (must be valid for t < trun[b]),
 * trun_dev [B] i32 -> conf_out [B] f32 (0 when t_run == 0). */
int msocr_seq_confidence(const float* logits, const int32_t* ids, const int32_t* trun_dev, int B, int V, int steps,
                         float* conf_out, void* stream);

/* Word crops -> recogniser canvases on the device: clamped AABB crop (Pipeline._extract_word_image,
 * _pipeline.py:204-221) + ResizeAndPadA (recognizers/_trba/data/transforms.py:85-120: aspect-preserving resize,
 * INTER_AREA if any axis shrinks else INTER_LINEAR, pasted at x=0 / vertically centred on a 255 canvas).
 * pages [N][H][W][3] u8; descriptor per crop = 8 x int32 {page, x1, y1, x2, y2, new_w, new_h, y0} (the host
 * evaluates Python's banker's rounding of the new size); desc_host is the same array in host memory, used only
 * to validate bounds before the launch, or NULL when the descriptors were produced on the device
 * (msocr_reading_order_crops): the kernel then checks every descriptor itself and emits a white canvas for an
 * invalid one.  canvases [M][img_h][img_w][3] u8.  (cv2 restated: parity unpinned.) */
int msocr_crop_resize_pad(const uint8_t* pages, int N, int H, int W, const int32_t* desc_dev,
                          const int32_t* desc_host, int M, int img_h, int img_w, uint8_t* canvases, void* stream);

/* HOST function (plain C++, all pointers host memory): reading order of a page's word boxes, i.e. the Python glue between
 * detector and recogniser: resolve_intersections + sort_boxes_reading_order(+_with_resolutions)
 * (detectors/_east/utils.py:500-644) and the "first word with an equal box" re-match of _pipeline.py:113-121, with the
 * reference's integer / double arithmetic.  boxes_host [n][4] int32 (x_min, y_min, x_max, y_max);
 * order_out_host [n] int32: entry k = index of the input box at position k of the reading order (duplicates as the
 * reference's dict semantics produce them). */
int msocr_reading_order_host(const int32_t* boxes_host, int n, double y_tol_ratio, double x_gap_ratio,
                             int32_t* order_out_host);

/* The same glue ON THE DEVICE, one workgroup per page, fed by msocr_east_box_tail's output, plus what follows it on the way to the
 * recogniser: word AABBs with np.int32 truncation and the min_text_size filter (_pipeline.py:100-133), the clamped crop window
 * (_pipeline.py:204-221) and ResizeAndPadA's size arithmetic (recognizers/_trba/data/transforms.py:91-95,114-117) -> descriptors in
 * the format of msocr_crop_resize_pad.  boxes [N][max_cand][9] f32, nbox [N] (negative: page skipped, ncrop = -1).
 * order_out [N][max_cand]: entry k = index of the word at reading-order position k; keep_out [N][max_cand]: 1 where position k
 * yields a crop; desc_out [N][max_cand][8]: the page's crop descriptors in order, compacted (page field = page_base + n);
 * ncrop_out [N]: crops of the page, or -1 = take the host path for this page (more than min(max_cand, 16384) boxes, more than
 * 4096 text lines, or more intersecting box pairs than the pair buffer holds).  Bit-identical to msocr_reading_order_host + the
 * host descriptor arithmetic.  workspace: msocr_reading_order_workspace_bytes(N, max_cand) bytes, 16-B aligned. */
int64_t msocr_reading_order_workspace_bytes(int N, int max_cand);
int msocr_reading_order_crops(const float* boxes, const int32_t* nbox, int N, int max_cand, int page_h, int page_w,
                              int min_text_size, int img_h, int img_w, double y_tol_ratio, double x_gap_ratio, int page_base,
                              int32_t* order_out, int32_t* keep_out, int32_t* desc_out, int32_t* ncrop_out, void* workspace,
                              void* stream);

/* ---- image ingest: JPEG -> RGB on the device -------------------------------------------------------------------------
 * Replaces the file decode of read_image (detectors/_east/utils.py:477-497: cv2.imread / PIL = libjpeg-turbo defaults).
 * 8-bit baseline / extended-sequential Huffman JPEG, grayscale or YCbCr 4:4:4 / 4:2:2 / 4:2:0, one interleaved scan, restart
 * markers.  The serial entropy decode of a stream without restart markers runs on the HOST (msocr_jpeg_parse_host fills `info`;
 * msocr_jpeg_entropy_decode_host writes the quantised coefficients, natural order, component after component:
 * [blocks_h][blocks_w][64] int16 at coef_off[c]); a stream WITH a restart interval is decoded on the DEVICE, one thread per
 * interval (msocr_jpeg_scan_prepare_host + msocr_jpeg_entropy_decode_device below);
 * dequantisation + inverse DCT (libjpeg "islow"), fancy chroma upsampling and YCbCr->RGB run on the DEVICE
 * (msocr_jpeg_reconstruct: coef_dev = the same array in device memory, workspace = msocr_jpeg_workspace_bytes(info) bytes,
 * rgb_out [height][width][3] u8).  msocr_jpeg_reconstruct_host is the HOST twin of the device stage (same code; all pointers
 * host memory).  Unsupported or corrupt streams: MSOCR_E_ARG / info.supported = 0 -> use the host decoder. */
typedef struct msocr_jpeg_info {
  int32_t width, height, ncomp;   /* ncomp 1 (grayscale) or 3 (YCbCr) */
  int32_t hs[3], vs[3];           /* sampling factors per component */
  int32_t blocks_w[3], blocks_h[3]; /* component planes in 8x8 blocks, padded to whole MCUs */
  int32_t supported;
  int64_t coef_off[3];            /* int16 elements */
  int64_t coef_total;
  uint16_t quant[3][64];          /* natural order */
} msocr_jpeg_info;
int msocr_jpeg_parse_host(const uint8_t* data_host, int64_t len, msocr_jpeg_info* info_out);
int msocr_jpeg_entropy_decode_host(const uint8_t* data_host, int64_t len, const msocr_jpeg_info* info, int16_t* coef_out_host);
/* Device-side entropy decode of a BATCH of streams with restart intervals (DRI): the intervals between RSTn markers are independent
 * byte-aligned bit streams.  Host, per page (thread-safe; no state): msocr_jpeg_scan_prepare_host walks the markers of a stream
 * msocr_jpeg_parse_host accepted and writes (a) the page's descriptor — msocr_jpeg_scan_desc_bytes() opaque bytes: frame geometry,
 * Huffman tables, bytes_base = where the file's first byte sits in the batch byte buffer — and (b) one (begin, end) pair of uint32
 * byte offsets INTO THE FILE per interval.  Returns the number of intervals, or MSOCR_E_ARG (no restart interval / corrupt marker
 * sequence / more than bounds_cap intervals: use msocr_jpeg_entropy_decode_host).
 * Device: msocr_jpeg_entropy_decode_device(bytes_dev = the files' bytes as they are on disk, descs_dev [n_pages] descriptors (8-byte
 * aligned), max_intervals = the largest interval count of a page, bounds_dev = the pages' pairs one page after the other,
 * page_base_dev [n_pages][2] int64 = {where the page's coefficient array starts in coef_dev (int16 elements), index of the page's
 * first pair in bounds_dev}, coef_dev [coef_total] int16 (zero-filled by this call), status_dev [n_pages]: 0, or 1 = the page's
 * stream is bad (same verdict as msocr_jpeg_entropy_decode_host: take the host reader)).  Bit-identical coefficients to
 * msocr_jpeg_entropy_decode_host.  msocr_jpeg_entropy_decode_intervals_host is the HOST twin of the kernel (same per-interval
 * decoder; all pointers host memory). */
int64_t msocr_jpeg_scan_desc_bytes(void);
int64_t msocr_jpeg_scan_prepare_host(const uint8_t* data_host, int64_t len, const msocr_jpeg_info* info, int64_t bytes_base,
                                     void* desc_out, uint32_t* bounds_out, int64_t bounds_cap);
int msocr_jpeg_entropy_decode_device(const uint8_t* bytes_dev, const void* descs_dev, int32_t n_pages, int32_t max_intervals,
                                     const uint32_t* bounds_dev, const int64_t* page_base_dev, int16_t* coef_dev, int64_t coef_total,
                                     int32_t* status_dev, void* stream);
int msocr_jpeg_entropy_decode_intervals_host(const uint8_t* bytes_host, const void* descs_host, int32_t n_pages,
                                             const uint32_t* bounds_host, const int64_t* page_base_host, int16_t* coef_host,
                                             int64_t coef_total, int32_t* status_host);
int64_t msocr_jpeg_workspace_bytes(const msocr_jpeg_info* info);
int msocr_jpeg_reconstruct(const msocr_jpeg_info* info, const int16_t* coef_dev, void* workspace_dev, uint8_t* rgb_out_dev,
                           void* stream);
int msocr_jpeg_reconstruct_host(const msocr_jpeg_info* info, const int16_t* coef_host, uint8_t* rgb_out_host);

/* f32 <-> bf16 / layout helpers */
int msocr_nchw_f32_to_nhwc(const float* in, int N, int C, int H, int W, int dtype, void* out, int64_t out_ld,
                           void* stream);
int msocr_nhwc_to_nchw_f32(const void* in, int N, int C, int H, int W, int64_t in_ld, int dtype, float* out,
                           void* stream);

const char* msocr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MSOCR_H */
