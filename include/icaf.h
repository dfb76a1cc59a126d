/*
 * icaf.h — C ABI of libicaf.so, the MI355X (gfx950) implementation of ICAFusion's inference hot path.
 *
 * The reference (chanchanchan97/ICAFusion) is pure Python on PyTorch: it has no FFI / plugin layer to mirror
 * (SURVEY.md §8b).  Its "operator interface" for this path is the set of nn.Module forwards in
 * models/common.py and models/yolo_test.py plus utils/general.non_max_suppression; each entry point below
 * replaces the arithmetic of one of those (file:line cited per function, paths relative to the reference root).
 * The host side (the icafusion_amd/models package) keeps the reference's Python class names and constructor signatures
 * and calls these functions through ctypes — see INTEGRATION.md for the binding stub.
 *
 * Conventions
 *   - every pointer named x/y/w/... is a DEVICE pointer unless the comment says "host";
 *   - activations are NHWC ("channels-last"): element (b,h,w,c) lives at ((b*H + h)*W + w)*ld + c, where the
 *     pixel stride `ld` >= C lets a tensor be a channel slice of a wider buffer (this is how Concat, C3's
 *     torch.cat and SPPF's torch.cat are eliminated);
 *   - dtype codes select the storage/compute type of activations and packed weights; accumulation, bias,
 *     normalisation statistics, softmax and the Detect/NMS arithmetic are always fp32;
 *   - kernels are enqueued on the caller's HIP stream (hipStream_t passed as void*), never synchronise, never
 *     allocate;
 *   - return value 0 = ok, negative = error; icaf_last_error() returns a thread-local message.
 */
#ifndef ICAF_H
#define ICAF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* icaf_stream_t; /* hipStream_t */

enum { ICAF_F32 = 0, ICAF_BF16 = 1, ICAF_F16 = 2 };
enum { ICAF_ACT_NONE = 0, ICAF_ACT_SILU = 1, ICAF_ACT_GELU = 2 };
enum { ICAF_OK = 0, ICAF_ERR_ARG = -1, ICAF_ERR_HIP = -2, ICAF_ERR_UNSUPPORTED = -3 };

const char* icaf_last_error(void);
/* Probe knobs of the library, set by the host's one options object (icafusion_amd/options.py) when the library is loaded — the library reads no
 * environment variable: "detect_elementwise" (!= 0: icaf_detect_decode by the one-thread-per-element kernel), "attn_qsplit" (n > 0: query splits per
 * head of icaf_cross_attention), "sppf_vpb" (n > 0: cap on the channel vectors per workgroup of icaf_sppf_pool); 0 = the library's own choice.
 * ICAF_ERR_ARG for an unknown name.  None of them changes a result. */
int icaf_set_option(const char* name, int value);
int icaf_version(void);
/* device facts used by the host for grid sizing / reporting: CU count, LDS bytes per workgroup, gcnArchName */
int icaf_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ---- input staging -------------------------------------------------------------------------------------
 * Reference: detect_twostream.py:70-80 / test.py:116-123 hand the model NCHW float tensors in [0,1].
 * mode 0: NCHW fp32 -> NHWC `dtype`, channels zero-padded from C to Cpad.
 * mode 1: NCHW fp32 -> space-to-depth NHWC: out[b][h/2][w/2][(dy*2+dx)*C + c] = in[b][c][h][w], padded to
 *         Cpad.  A 6x6 / stride-2 / pad-2 convolution over the image (first layer of each stream,
 *         rows 0 and 10 of the Transfusion yaml files) equals a 3x3 / stride-1 / pad-1 convolution over this tensor.
 */
int icaf_preprocess_nchw(const float* img, void* out, int dtype, int B, int C, int H, int W, int Cpad, int mode,
                         icaf_stream_t s);
/* Same staging straight from the dataloader's uint8 batch (reference test.py:116-123: `img.to(device)`, `.float()`,
 * `/= 255.0`, `img[:, :3]` / `img[:, 3:]`).  img: [B][Ctot][H][W] uint8; stream s (0 <= s < nstreams) takes channels
 * [c0 + s*C, c0 + (s+1)*C) and is written to out[s][b] (nstreams*B images), value = (float)u8 / 255.0f. */
int icaf_preprocess_u8(const unsigned char* img, void* out, int dtype, int B, int Ctot, int c0, int C, int nstreams,
                       int H, int W, int Cpad, int mode, icaf_stream_t s);

/* Staging + stem convolution in one persistent kernel: the 6x6 / stride 2 / pad 2 Conv(+BN+SiLU) of yaml rows 0 and 10
 * (models/common.py:48-60) computed straight from the NCHW images (fp32 [nstreams*B][3][H][W], or img_u8 != 0: the
 * dataloader's uint8 [B][ctot][H][W] batch, stream s = channels [3s, 3s+3), value / 255) — no staged copy of the images
 * is written.  w: packed space-to-depth weights [Np][192] per stream (icafusion_amd.ops.s2d_conv_weight), y: NHWC
 * [nstreams][B][H/2][W/2][ldy]; *_gs = per-stream strides in elements / floats.  16-bit types, Cout in {32, 64}. */
int icaf_stem(const void* img, int img_u8, int ctot, const void* w, const float* bias, void* y, int ldy, int dtype,
              int nstreams, int B, int H, int W, int Cout, int Kp, long long w_gs, long long bias_gs, long long y_gs,
              icaf_stream_t s);

/* Yaml rows 0-2 (10-12) up to the C3's first GEMM in ONE persistent kernel — the three layers every image pixel goes
 * through at the highest resolutions (models/common.py:48-60, 216-227; models/transformer/yolov5s_*.yaml rows 0-2):
 *   t0 = SiLU(conv6x6/s2/p2(img) + bias0)          C0 channels at H/2 x W/2   (the stem, as icaf_stem)
 *   t1 = SiLU(conv3x3/s2/p1(t0) + bias1)           C1 channels at H/4 x W/4
 *   y  = SiLU(W2 . t1 + bias2)                     C2 channels (the C3's cv1 | cv2), NHWC [nstreams][B][H/4][W/4][ldy]
 * t0 and t1 live in LDS only (rounded to the storage type exactly where the three-launch form writes them, so the
 * result is bit-identical to icaf_stem -> icaf_conv2d with a chained 1x1).  Weights are the packed matrices of those
 * launches: w0 [Np][192] (space-to-depth), w1 [Np][Kp1] (k = ky, kx, c0), w2 [Np][Kp2]; *_gs = per-stream strides.
 * Built for C0 = 32, C1 = 64, C2 <= 64 (yolov5s), 16-bit types. */
typedef struct icaf_stem2_args {
    const void* img; int img_u8, ctot;             /* as icaf_stem */
    int dtype, nstreams, B, H, W;
    const void* w0; const float* bias0; long long w0_gs, bias0_gs; int Kp0, C0;
    const void* w1; const float* bias1; long long w1_gs, bias1_gs; int Kp1, C1;
    const void* w2; const float* bias2; long long w2_gs, bias2_gs; int Kp2, C2;
    void* y; long long y_gs; int ldy, reserved;
} icaf_stem2_args;
int icaf_stem2(const icaf_stem2_args* a, icaf_stream_t s);

/* ---- implicit-GEMM convolution / linear ------------------------------------------------------------------
 * Replaces Conv.forward / fuseforward (models/common.py:48-60: SiLU(BN(Conv2d))) with BN folded into the
 * weights (utils/torch_utils.py:182-202), nn.Linear (+GELU) inside CrossAttention / CrossTransformerBlock
 * (models/common.py:607-618,704-709), the residual add of Bottleneck (models/common.py:194) and the
 * LearnableCoefficient mixes (models/common.py:746-750), and Detect's 1x1 output convs (models/yolo_test.py:50).
 *
 *   y[m][n] = alpha_res * res[m][n] + alpha_acc * act( sum_k A[m][k] * Wp[n][k] + bias[n] [+ bilinear(pre)[m][n]] )
 *
 * m = (b, ho, wo) output pixel, k = (kh, kw, cin) gathered on the fly from x (zero outside the image),
 * Wp = packed weights [Np][Kp] (K-major, Np = Cout rounded up to 128, Kp = K rounded up to 64 elements, zero
 * padded).  `groups` > 1 batches independent problems (the two modalities of DMFF) in gridDim.z; the *_gs
 * fields are the per-group strides in ELEMENTS (bytes/sizeof for x,w,y,res; floats for bias).
 */
typedef struct icaf_conv_args {
    const void* x;
    const void* w;
    const float* bias; /* may be NULL */
    void* y;
    const void* res; /* may be NULL */
    long long x_gs, w_gs, bias_gs, y_gs, res_gs;
    int groups;
    int B, H, W, Cin, ldx;
    int Ho, Wo, Cout, ldy;
    int kh, kw, sh, sw, ph, pw;
    int ldr;
    int Kp;
    int act;
    int dtype;     /* x, w, res */
    int out_dtype; /* y: same as dtype, or ICAF_F32 */
    float alpha_acc[2];
    float alpha_res[2];
    int tile; /* 0 = auto; otherwise force a launch configuration (tuning / tests; a configuration the layer does not satisfy is an error,
               * never replaced silently).  1-4 (+10 / 20 / 30 per pipeline), 25 / 26 / 28 / 29: igemm.hip tiles; 40 + shape: ctile.hip;
               * 51 / 52: igemm_stream.hip; 61 - 66: igemm_wreg.hip; 71: cstream.hip; 80 + shape (81 - 85): cwide.hip.
               * Every configuration of a layer produces the same bits (same K order, MFMA step, epilogue expressions). */
    /* Optional pre-activation term, bilinearly resized (align_corners=False) from a coarse fp32 map:
     *   y = alpha_res*res + alpha_acc * act( A.W + bias + bilinear(pre)[b][ho][wo][n] )
     * pre: [B][pre_h][pre_w][ldpre] fp32 (NULL = none).  This is how DMFF's tail (models/common.py:827-841:
     * F.interpolate(tokens) + feature, cat, conv1x1_out) runs as ONE GEMM over the untouched features: the 1x1
     * convolution commutes with the (linear) resize, so conv(cat(f + up(t))) = conv(cat(f)) + up(conv(cat(t))). */
    const float* pre;
    int pre_h, pre_w, ldpre;
    int pre_mode; /* 0: bilinear (align_corners=False); 1: nearest, src = floor(dst * pre_h / Ho) — nn.Upsample('nearest') in
                   * front of a Concat feeding a 1x1 conv (head rows 24-26 / 28-30): W.cat(up(a), b) = up(Wa.a) + Wb.b, so
                   * the low-resolution product Wa.a is the `pre` map of the GEMM over b and neither up(a) nor the concat exist */
    /* Optional chained 1x1 convolution + SiLU consuming this layer's output tile in place (w2 != NULL):
     *   y2 = SiLU( W2 . SiLU(A.W + bias) + bias2 )        [this layer must be SiLU, 16-bit, one N tile: Cout <= 256]
     * The intermediate tensor is never written (y is ignored) unless chain_keep is set.  It is how a backbone down-sampling Conv and the fused
     * cv1|cv2 GEMM of the C3 block behind it (models/common.py:56-60 then :226) run as one launch.
     * w2: packed [Np][Kp2] with K = Cout, bias2 may be NULL, y2: NHWC with pixel stride ldy2, Cout2 <= 256 channels. */
    const void* w2;
    const float* bias2;
    void* y2;
    long long w2_gs, bias2_gs, y2_gs;
    int Kp2, Cout2, ldy2;
    int chain_keep; /* != 0: y IS written as well; only then may `res` be set: the chained 1x1 consumes y as stored, residual
                     * included (a Bottleneck's 3x3 + shortcut followed by the next Bottleneck's 1x1, models/common.py:193-194) */
    /* Optional second copy of the packed weights in FRAGMENT-MAJOR order (NULL = none), read by the launch configurations that feed
     * the weight operand from registers (igemm_wreg.hip, cwide.hip: tile ids 61 - 66, 81 - 85): [Np / 32][Kp / 16][64 lanes][8 elements], lane
     * (hi * 32 + r) of block (nb, ks) holding w[nb * 32 + r][ks * 16 + hi * 8 .. + 8] of the K-major matrix above; wf_gs = group
     * stride in elements.  16-bit types (icafusion_amd.ops.frag_weights builds it once per layer). */
    const void* wf;
    long long wf_gs;
    /* Optional second half of the chained 1x1's input (NULL = none; needs w2): the chained layer then reads K = [this layer's output
     * tile (Cout channels) | x2 (Cout more channels of the same pixels)]:
     *   y2 = SiLU( W2 . cat( [alpha_res*res +] SiLU(A.W + bias), x2 ) + bias2 )
     * This is the TAIL of a C3 block (models/common.py:216-227: cv3(cat(m(cv1(x)), cv2(x)))): the last Bottleneck's 3x3 (+ shortcut,
     * `res` allowed without chain_keep) carries cv3, x2 = cv2's output; the Bottleneck chain's result and the concatenation never reach HBM.
     * x2: NHWC, pixel stride ldx2 elements, x2_gs = group stride in elements; W2 packed [Np][Kp2] with K in cv3's own order [m | cv2].
     * Built for 128 -> 128 3x3 / stride 1 layers with Cout2 = 256, Kp2 = 256, 16-bit types, launch configurations 81 / 82 (cwide.hip);
     * bit-identical to the two launches. */
    const void* x2;
    long long x2_gs;
    int ldx2, reserved2;
} icaf_conv_args;

int icaf_conv2d(const icaf_conv_args* a, icaf_stream_t s);

/* Whole Bottleneck in one launch (models/common.py:184-194): y = [x +] SiLU(conv3x3(SiLU(conv1x1(x)))) for c_ -> c_ -> c_
 * channels with c_ in {32, 64}, 16-bit types.  `conv` describes the 3x3 / stride 1 / pad 1 layer (x = block input,
 * w / bias = its packed weights, y = block output — a DIFFERENT buffer than x —, res = x for the shortcut or NULL);
 * w1 / bias1 are the packed 1x1 weights ([Np][Kp1], Kp1 = 128 bytes) applied first.  The 1x1 output never reaches HBM.
 * shape: LDS patch 1 = 8x32 pixels (c_ = 32), 2 = 8x32 (c_ = 64), 3 = 8x16 (c_ = 64).
 * With conv.w2 != NULL (shape 1 only) the C3's cv3 rides on the block (models/common.py:226):
 *   y2 = SiLU( W2 . cat(x2, [x +] SiLU(conv3x3(...))) + bias2 )
 * x2 = the cv2 half of cv3's input (NHWC, c_ channels, pixel stride ldx2), W2 packed [Np][64] with its K columns in the
 * order [cv2 | m]; only y2 (conv.y2 / ldy2, Cout2 = 64, 16-byte aligned rows) is written — conv.y is ignored.  Bit-identical to icaf_bottleneck
 * followed by the 1x1 icaf_conv2d. */
typedef struct icaf_bneck_args {
    icaf_conv_args conv;
    const void* w1;
    const float* bias1;
    long long w1_gs, bias1_gs; /* per-group strides (elements / floats) */
    int Kp1;
    int shape;
    const void* x2;
    long long x2_gs;
    int ldx2, reserved;
} icaf_bneck_args;
int icaf_bottleneck(const icaf_bneck_args* a, icaf_stream_t s);
/* name of the kernel instantiation icaf_conv2d would launch for these args (host string, for profiling) */
int icaf_conv2d_kernel_name(const icaf_conv_args* a, char* buf, int buf_len);

/* ---- SPPF / upsample / copy ------------------------------------------------------------------------------
 * icaf_sppf_pool: the three chained k x k stride-1 max pools of SPPF.forward (models/common.py:262-267);
 *   y1 = mp(x), y2 = mp(y1), y3 = mp(y2) computed in one pass (-inf padding semantics).
 * icaf_upsample_nearest: nn.Upsample(None, scale, 'nearest') rows of the head (yaml rows 24, 28).
 * icaf_copy_channels: generic channel-slice copy (fallback for Concat, models/common.py:313-321).
 */
int icaf_sppf_pool(const void* x, int ldx, void* y1, void* y2, void* y3, int ldy, int dtype, int B, int H, int W,
                   int C, int k, icaf_stream_t s);
int icaf_upsample_nearest(const void* x, int ldx, void* y, int ldy, int dtype, int B, int H, int W, int C,
                          int scale, icaf_stream_t s);
int icaf_copy_channels(const void* x, int ldx, void* y, int ldy, int dtype, long long rows, int C,
                       icaf_stream_t s);
/* icaf_axpby: y = a*x0 + b*x1 over `rows` pixels of C channels — the `Add` fusion block (models/common.py:324-331:
 * x[0]*w + x[1]*(1-w)) of the *_Add_* configs. */
int icaf_axpby(const void* x0, int ld0, const void* x1, int ld1, void* y, int ldy, int dtype, long long rows, int C,
               float a, float b, icaf_stream_t s);

/* ---- DMFF (TransformerFusionBlock, models/common.py:762-865) ---------------------------------------------
 * icaf_dmff_pool_tokens: AdaptivePool2d avg + max (models/common.py:868-891), LearnableWeights mix
 *   (:579-587) and positional embedding add (:817-823) for both modalities.
 *   tokens[g][b][n][c] = w1_g*avg + w2_g*max + pos_g[n][c],  g = 0 (RGB) / 1 (IR),  n = th*W' + tw.
 * icaf_layernorm: nn.LayerNorm(C), eps 1e-5 over the last dim; group g uses (gamma_g, beta_g)
 *   (CrossAttention.LN1/LN2 :646,:651; CrossTransformerBlock.LN2 applied to both groups :749-750).
 * icaf_cross_attention: the two crossed softmax(QK^T/sqrt(dk))V products of CrossAttention.forward (:670-685).
 *   qkv[g][row][3C] holds [q | k | v] of modality g; out[0] = softmax(q_1 k_0^T) v_0, out[1] = softmax(q_0 k_1^T) v_1
 *   with heads laid out as in .view(b, n, h, dk) (:647-649).
 * icaf_dmff_upsample_merge: eval-mode F.interpolate(bilinear, align_corners=False) of the token maps back to
 *   (H, W), residual add of the original features and channel concat (:827-840):
 *   out[b][h][w][g*C + c] = bilinear(tokens[g])[b][h][w][c] + fea_g[b][h][w][c].
 */
int icaf_dmff_pool_tokens(const void* fea_rgb, int ld_rgb, const void* fea_ir, int ld_ir, const float* pos_rgb,
                          const float* pos_ir, void* tokens, int dtype, int B, int H, int W, int C, int th, int tw,
                          int kh, int kw, int sh, int sw, float w1_rgb, float w2_rgb, float w1_ir, float w2_ir,
                          icaf_stream_t s);
int icaf_layernorm(const void* x, void* y, const float* gamma0, const float* beta0, const float* gamma1,
                   const float* beta1, int dtype, long long rows_per_group, int C, int groups, float eps,
                   icaf_stream_t s);
int icaf_cross_attention(const void* qkv, void* out, int dtype, int B, int N, int C, int heads, icaf_stream_t s);
int icaf_dmff_upsample_merge(const void* tokens, const void* fea_rgb, int ld_rgb, const void* fea_ir, int ld_ir,
                             void* out, int ldo, int dtype, int B, int H, int W, int C, int th, int tw,
                             icaf_stream_t s);

/* ---- Detect decode (models/yolo_test.py:43-65, eval branch) -----------------------------------------------
 * p: fp32 output of the level's 1x1 conv, NHWC [B][ny][nx][ldp] with channel = a*no + o.
 * Writes z[b][row_offset + (a*ny + y)*nx + x][o] (sigmoid + grid/anchor decode), logits (raw class scores,
 * may be NULL) and raw[b][a][y][x][o] (the permuted pre-sigmoid map the reference also returns, may be NULL).
 * anchors_px: host pointer to na*2 floats = anchor sizes in pixels (anchor_grid).
 */
int icaf_detect_decode(const float* p, int ldp, float* z, float* logits, float* raw, int B, int ny, int nx, int na,
                       int no, long long rows_total, long long row_offset, float stride, const float* anchors_px,
                       icaf_stream_t s);
/* A Detect level in ONE launch (16-bit feature maps, 3 anchors, no in {6, 8, 14}): the level's 1x1 output convolution
 * (models/yolo_test.py:50; `a` describes it as for icaf_conv2d: 1x1, no activation, groups 1, Cout = na * no; a->y is ignored)
 * with the decode above as the epilogue of the persistent streaming GEMM — the fp32 conv map is never written.  z / logits / raw
 * are bit-identical to icaf_conv2d (fp32 out) followed by icaf_detect_decode. */
int icaf_detect_conv(const icaf_conv_args* a, float* z, float* logits, float* raw, int na, int no, long long rows_total,
                     long long row_offset, float stride, const float* anchors_px, icaf_stream_t s);

/* ---- fused DMFF block (16-bit token types) -------------------------------------------------------------------
 * One CrossTransformerBlock iteration (models/common.py:737-759) in two launches:
 *   icaf_dmff_ln_qkv    qkv[g] = LayerNorm_g(x[g]) W_qkv,g^T + b   — CrossAttention.LN1 / LN2 (:661-662) fused in front of the
 *                       six Linear(C, C) projections (:664-669); qkv[g][row][3C] = [que | key | val] of modality g.
 *   icaf_dmff_attn_mlp  per 64 token rows of one (image, modality g): the crossed attention of those rows over all heads
 *                       (softmax(q_{1-g} k_g^T / sqrt(dk)) v_g, :670-681), out-projection and coefficient mix
 *                       x_att = coef_res_attn[g] * x + coef_acc_attn[g] * (att W_o,g^T + b) (:682-685, :745-746), the block's shared
 *                       LayerNorm LN2 (:749-750), MLP Linear(C, 4C) -> GELU(erf) -> Linear(4C, C) (:704-709) and
 *                       y = coef_res_mlp[g] * x_att + coef_acc_mlp[g] * (mlp + b) (:751-752).  Attention output, x_att, the
 *                       normalised tile and the hidden activations stay in LDS / registers.
 * x: tokens [2][B*N][C] (group stride x_gs elements); y: element (g, row, c) at y + g*y_gs + row*ldy + c (may alias neither x
 * nor qkv).  Weights are packed [2][Np][Kp] (Np multiple of 128, Kp of 64; *_gs = per-modality strides), biases fp32 [2][Np].
 * Requirements: dtype bf16 / f16, C % 64 == 0, head dim % 8 == 0, hidden % 128 == 0; icaf_dmff_attn_mlp additionally C <= 512
 * (icaf_dmff_attn_mlp_lds_bytes returns the LDS bytes a launch needs, or (size_t)-1 when the shape is not covered: callers
 * then run the per-layer entry points above).  Rounding points equal those of the per-layer launches. */
typedef struct icaf_dmff_args {
    const void* x; void* qkv; void* y;
    const void* wqkv; const float* bqkv;
    const void* wo; const float* bo;
    const void* w1; const float* b1;
    const void* w2; const float* b2;
    const float* ln_attn_gamma[2]; const float* ln_attn_beta[2];   /* CrossAttention.LN1 (RGB tokens), LN2 (IR tokens) */
    const float* ln_mlp_gamma; const float* ln_mlp_beta;           /* CrossTransformerBlock.LN2, both modalities */
    long long wqkv_gs, bqkv_gs, wo_gs, bo_gs, w1_gs, b1_gs, w2_gs, b2_gs, x_gs, y_gs;
    int dtype, B, N, C, heads, Kp, Kp4, hidden, ldy, reserved;
    float eps_attn, eps_mlp;
    float coef_res_attn[2], coef_acc_attn[2], coef_res_mlp[2], coef_acc_mlp[2];   /* coefficient1/3, 2/4, 5/7, 6/8 */
    void* debug_clock;   /* NULL, or 8 int64 slots: workgroup (0,0,0) of icaf_dmff_attn_mlp stores the shader clock at its phase boundaries */
    /* reserved: icaf_dmff_wide_ln_qkv reads it as "output-channel passes per workgroup": 1 = one (A/B switch; measured slower), anything else =
     * three; other entry points ignore it */
    /* fp32 RESIDUAL STREAM across the shared-weight iterations (models/common.py:744-752 run `loops` times: x = block(x)); read by
     * icaf_dmff_wide_proj_mlp / _split / icaf_dmff_wide_reduce only.  y32 != NULL: this iteration's tokens are ALSO written in fp32 to
     * y32 [2][B*N][C] (x_att is then kept in fp32 inside the step); x32 != NULL: the residual x of this iteration is read from the previous
     * iteration's fp32 tokens instead of the 16-bit x (LayerNorm + QKV keep reading the 16-bit tokens).  16-bit dtypes, 16-byte aligned. */
    const float* x32; float* y32;
} icaf_dmff_args;
int icaf_dmff_ln_qkv(const icaf_dmff_args* a, icaf_stream_t s);
int icaf_dmff_attn_mlp(const icaf_dmff_args* a, icaf_stream_t s);
/* The block for the WIDE levels (C = 256 or 512, 16-bit types; dmff_wide.hip): one iteration = icaf_dmff_wide_ln_qkv,
 * icaf_cross_attention, icaf_dmff_wide_proj_mlp.
 *   icaf_dmff_wide_ln_qkv    as icaf_dmff_ln_qkv (LayerNorm :661-662 + the six projections :664-669);
 *   icaf_dmff_wide_proj_mlp  out-projection + coefficient mix (:682-685, :745-746), the shared LayerNorm (:749-750), MLP + mix
 *                            (:704-709, :751-752) of 64 token rows per workgroup; att = the attention output tokens [2][B*N][C]
 *                            (contiguous, as icaf_cross_attention writes them); a->qkv / wqkv / bqkv are not read.
 * DIFFERENT WEIGHT LAYOUT: a->wqkv / wo / w1 / w2 point at FRAGMENT-MAJOR copies of the packed [2][Np][Kp] matrices,
 * [2][Np/32][Kp/16][64][8] (lane (hi*32 + r) of block (nb, ks) = w[nb*32 + r][ks*16 + hi*8 .. +8], the layout of icaf_conv_args.wf;
 * same *_gs strides): each wavefront streams its MFMA weight operands straight from L2 into registers.  hidden % 256 == 0. */
int icaf_dmff_wide_ln_qkv(const icaf_dmff_args* a, icaf_stream_t s);
int icaf_dmff_wide_proj_mlp(const icaf_dmff_args* a, const void* att, icaf_stream_t s);
/* The same step with the MLP's hidden columns split over `ksplit` (2 or 4) workgroups per 64-row tile — for levels with fewer tiles
 * than CUs and more weight bytes than an XCD's L2 (P5: 100 tokens x 32 images at C = 512): every workgroup repeats out-projection +
 * LayerNorm (:682-685, :745-750), runs Linear(C, 4C) -> GELU -> Linear(4C, C) (:704-709) over ITS hidden slice, parks x_att in y and
 * writes fp32 partial sums into partial [ksplit][2][B*N][C].  icaf_dmff_wide_reduce (the next launch on the stream) adds them in slice
 * order (deterministic: no atomics, no in-kernel fences) and applies bias + the coefficient mix (:751-752) in place on y.  Same operands
 * and fragment-major weight layout as icaf_dmff_wide_proj_mlp; hidden % (256 * ksplit) == 0. */
int icaf_dmff_wide_proj_mlp_split(const icaf_dmff_args* a, const void* att, float* partial, int ksplit, icaf_stream_t s);
int icaf_dmff_wide_reduce(const icaf_dmff_args* a, const float* partial, int ksplit, icaf_stream_t s);
int icaf_dmff_attn_mlp_lds_bytes(int C, int N, int heads, int dtype, size_t* bytes);

/* ---- NMS (utils/general.py:518-607 + torchvision.ops.nms semantics) ----------------------------------------
 * pred: [B][rows][5+nc] fp32 (cx, cy, w, h, obj, cls...).  Per image: obj > conf filter, conf = obj*cls, best
 * class or multi-label expansion, optional class filter (host int array), top max_nms by score (stable),
 * class-offset boxes, greedy IoU suppression in descending score order (ties: ascending candidate index), first
 * max_det survivors.
 * det: [B][max_det][6] (x1,y1,x2,y2,conf,cls), count: [B], keep_idx: [B][max_det] = indices into the image's
 * candidate list exactly as torchvision.ops.nms would return them (may be NULL: one small launch less).
 * No candidate list and no full sort are materialised: the workspace (icaf_nms_workspace_bytes; 256-byte aligned, contents
 * irrelevant on entry) holds one 32-bit score key per candidate slot, per-image score histograms and per-chunk candidate counts;
 * only the score ranges the greedy walk actually reaches are gathered and sorted, in LDS (nms.hip).  Enqueues a memset and
 * 2-3 kernels on `s`; max_det <= 1024, nc <= 65535, class filter ids 0..255. */
/* Validation statistics of test.py:196-230 on the device, one workgroup per image: the NMS output block det
 * [B][max_det][6] / count[B] (letterboxed pixel space) is mapped to native image space with scale[b] = {gain, pad_x,
 * pad_y, w0, h0} (scale_coords + clip_coords, utils/general.py:386-407; NULL = already native), every detection is paired
 * with the best-IoU label of its class (labels [L][5] = cls, x1, y1, x2, y2 in native space, image b owning rows
 * [label_off[b], label_off[b+1])), and claims it in index order if its IoU exceeds iouv[0]:
 * correct[b][i][t] = claimed && iou > iouv[t] (uint8).  predn (optional) receives the native-space boxes [B][max_det][4]. */
int icaf_match_predictions(const float* det, const int* count, int B, int max_det, const float* labels, const int* label_off,
                           int max_labels_per_image, const float* scale, const float* iouv, int T, unsigned char* correct,
                           float* predn, icaf_stream_t s);
int icaf_nms_workspace_bytes(int B, long long rows, int nc, int multi_label, size_t* bytes);
int icaf_nms(const float* pred, int B, long long rows, int nc, float conf_thres, float iou_thres, int multi_label,
             int agnostic, const int* classes_host, int n_classes, int max_det, int max_nms, float max_wh,
             float* det, int* count, int* keep_idx, void* workspace, size_t workspace_bytes, icaf_stream_t s);

/* ---- HIP graph capture / events (so the Python host never needs a tracing compiler) ------------------------ */
int icaf_graph_begin(icaf_stream_t s);
int icaf_graph_end(icaf_stream_t s, void** graph_exec);
int icaf_graph_launch(void* graph_exec, icaf_stream_t s);
int icaf_graph_destroy(void* graph_exec);
int icaf_event_create(void** ev);
int icaf_event_record(void* ev, icaf_stream_t s);
int icaf_stream_wait_event(icaf_stream_t s, void* ev);         /* hipStreamWaitEvent: fork / join of capture branches */
int icaf_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int icaf_event_destroy(void* ev);
int icaf_stream_sync(icaf_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* ICAF_H */
