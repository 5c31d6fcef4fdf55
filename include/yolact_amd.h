/* yolact_amd.h — C ABI of libyolact_amd.so, the MI355X (gfx950) YOLACT inference hot path.
 *
 * Every entry point: plain pointers + sizes, an explicit hipStream_t (passed as void*), returns
 *   0 = ok, negative = argument/shape error (YMI_E*), positive = hipError_t of the launch.
 * The library never allocates or frees device memory: the caller owns every input, output and
 * workspace buffer (SURVEY §8(b) "ownership").  All tensors are float32 unless stated, device
 * pointers, densely packed in the layout named in the comment.  Activations are NHWC.
 *
 * Each function cites the reference interface (file:line under dbolya/yolact) it replaces.
 */
#ifndef YOLACT_AMD_H
#define YOLACT_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMI_ABI_VERSION 8

/* negative return codes (ymi_strerror) */
#define YMI_EFORMAT (-4)       /* corrupt or truncated input stream (ymi_jpeg_*) */
#define YMI_EUNSUPPORTED (-5)  /* valid input outside the supported subset (ymi_jpeg_*) */

/* activation codes (epilogue) */
enum { YMI_ACT_NONE = 0, YMI_ACT_RELU = 1, YMI_ACT_LEAKY01 = 2, YMI_ACT_TANH = 3, YMI_ACT_SIGMOID = 4 };
/* residual modes */
enum { YMI_RES_NONE = 0, YMI_RES_ADD = 1, YMI_RES_BILINEAR = 2 };

/* One output segment of a convolution: output channels [n0,n1) of pixel (b,pix) go to
 *   ptr + b*batch_stride + pix*row_stride + (n - n0)      (element units)
 * so a single GEMM can scatter to several tensors (the three shared prediction-head convs,
 * yolact.py:169-173, become one GEMM with Cout = A*4 + A*81 + A*32) and can write straight
 * into the level-concatenated [B, P, k] tensors (yolact.py:633-634) without a torch.cat. */
typedef struct {
  int32_t n0, n1;
  int32_t act;          /* YMI_ACT_* applied to this segment */
  int32_t row_stride;   /* elements between consecutive output pixels */
  int64_t batch_stride; /* elements between consecutive images */
  float *ptr;
} ymi_conv_seg;

/* Fused convolution descriptor.
 * Replaces nn.Conv2d [+ BatchNorm2d (eval)] [+ residual add | FPN top-down bilinear add] [+ activation]:
 *   backbone.py:37-57 (Bottleneck), :126-139 (stem), :222-236 (darknet unit), yolact.py:319-361 (FPN),
 *   utils/functions.py:163-213 (make_net convs), yolact.py:146-193 (prediction heads).
 * y[b,oy,ox,n] = act( scale[n] * sum_{ky,kx,c} x[b, oy*s-p+ky, ox*s-p+kx, c] * w[n,ky,kx,c] + bias[n]  (+ res) )
 * fp32 in, fp32 accumulate, fp32 out.  The products are formed by the tile `tile` selects: plain ids = v_mfma_f32_32x32x2_f32
 * (exact fp32 products); `| YMI_TILE_X3` = three-piece bf16 split; `| YMI_TILE_H2` = two-piece fp16 split (see the enum). */
typedef struct {
  const float *x;       /* [B,H,W,ldx] NHWC; channels [0,Cin) of each pixel are used; 16-byte aligned, < 2 GiB */
  const float *w;       /* packed [CoutPad][Kpad], k = (ky*kw+kx)*Cin + c; CoutPad % 128 == 0, Kpad % 32 == 0, zero padded */
  const float *scale;   /* [Cout] or NULL (=1)  — folded BatchNorm gamma/sqrt(var+eps) */
  const float *bias;    /* [Cout] or NULL (=0)  — conv bias or folded BatchNorm shift */
  const float *res;     /* residual source or NULL */
  int32_t B, H, W, Cin, ldx;
  int32_t Ho, Wo, Cout;
  int32_t kh, kw, stride, pad;
  int32_t Kpad;
  int32_t res_mode;     /* YMI_RES_* */
  int32_t res_ld;       /* channel stride of res pixels */
  int32_t res_H, res_W; /* source size for YMI_RES_BILINEAR (res is [B,res_H,res_W,res_ld]) */
  int32_t res_after_act;/* 1: y = act(conv) + res (darknet block, backbone.py:214-215); 0: y = act(conv + res) */
  int32_t nseg;         /* 1..3 */
  int32_t tile;         /* 0 = auto; else YMI_TILE_* override (tests / tuning) */
  int32_t cin_alg;      /* real (un-padded) input channels for FLOP accounting; 0 = Cin */
  int32_t split_k;      /* 0 / 1: off.  S > 1: 1x1 convolutions (and the pipelined DCN tiles, YMI_TILE_DCNP) only — the K = Cin reduction is cut into S ranges computed by S
                         * times as many blocks (small maps, long K: too few output tiles to fill 256 CUs otherwise); the
                         * partial sums go through `split_ws` and are added in a fixed order (deterministic) by a second
                         * launch that applies scale / bias / residual / activation.  Needs (Kpad / 32) % S == 0, one dense
                         * output segment, activation none / ReLU / LeakyReLU, residual none / add. */
  ymi_conv_seg seg[3];
  const void *w_x3;     /* optional, for tile | YMI_TILE_X3: the SAME filters pre-split into three bf16 planes
                         * [3][CoutPad][Kpad] (uint16 bit patterns; plane 0 = top 8 significant bits by truncation, 1 = next 8,
                         * 2 = last 8; plane0 + plane1 + plane2 == w exactly).  With it the kernel splits only the activations
                         * on the fly (a third of the VALU work); NULL: both operands are split on the fly. */
  float *split_ws;      /* split_k > 1: workspace of split_k * B*Ho*Wo * Cout floats, 16-byte aligned */
  int32_t cout_alg;     /* real output channels for FLOP accounting when Cout carries zero-filter padding columns; 0 = Cout */
  int32_t _pad2;
  /* -- fp16x2 tiles (tile | YMI_TILE_H2) ------------------------------------------------------------------------------
   * w_h2: the SAME filters as two fp16 planes [2][CoutPad][Kpad] (uint16 bit patterns): row n is scaled by the power of two
   * sW[n] that maps max|w[n,:]| into [2^13, 2^14); plane 0 = fp16(w * sW) (round to nearest), plane 1 = fp16(w * sW - plane 0).
   * scale_h2[n] = scale[n] / sW[n]  (or 1 / sW[n] without a scale): the epilogue's per-channel factor with the filter scale
   * folded in (exact: a power of two).  winv_h2[n] = 1 / sW[n] alone (partial sums of split-K launches).
   * x_amax: DEVICE magnitude-bound slot of the input tensor: YMI_AMAX_SUB (16) floats, YMI_AMAX_STRIDE (64) floats apart
   * (4 KB per slot), whose MAXIMUM is an upper bound on |x| — raised by the launch that produced x (its y_amax) or by
   * ymi_amax_f32, zeroed by the caller before that launch; the kernel scales x by the power of two sA that maps
   * bound * x_amax_mul into [2^13, 2^14) before the fp16 split and divides the accumulators by sA.  A bound that is too small
   * can overflow fp16.  (Sub-slots: thousands of waves commit to one tensor's bound; spread over 16 cache lines the atomics
   * of a launch's first residency round cost ~1 us instead of ~50.) */
  const void *w_h2;
  const float *scale_h2;
  const float *winv_h2;
  const float *x_amax;
  float *y_amax;        /* any tile: DEVICE slot (same layout as x_amax) or NULL: raised atomically so that its maximum becomes
                         * max|y| over everything this launch writes.  nseg > 1 (ABI 5): nseg CONSECUTIVE slots (YMI_AMAX_SUB *
                         * YMI_AMAX_STRIDE floats apart), segment k raises slot k — the segments of one launch may be different
                         * tensors with different consumers (the merged head0.upfeature + proto_net[0] launch) */
  float x_amax_mul;     /* static factor on *x_amax; 0 = 1 */
  int32_t _pad3;
} ymi_conv_desc;

/* block tile BMxBN; _Kn = the block's 4 waves also split K n ways (partial sums reduced in LDS in a fixed order:
 * deterministic, but a different fp32 summation order than the unsplit tiles) */
enum { YMI_TILE_AUTO = 0, YMI_TILE_128x128 = 1, YMI_TILE_128x64 = 2, YMI_TILE_64x64 = 3, YMI_TILE_128x32 = 4,
       YMI_TILE_64x128 = 5, YMI_TILE_32x32_K4 = 6, YMI_TILE_64x32_K2 = 7, YMI_TILE_32x64_K2 = 8,
       /* _Sn = n-stage LDS pipeline (n-1 K steps of LDS-DMA in flight); Cin % 32 == 0 layers only */
       YMI_TILE_64x64_S3 = 9, YMI_TILE_64x64_S4 = 10, YMI_TILE_64x128_S3 = 11, YMI_TILE_128x64_S3 = 12,
       YMI_TILE_32x32_K4_S4 = 13, YMI_TILE_64x32_K2_S3 = 14, YMI_TILE_32x64_K2_S3 = 15,
       /* _W8 = 512-thread blocks (8 waves): half the global->LDS traffic per FLOP of the 4-wave tile of equal wave tile */
       YMI_TILE_128x128_W8 = 16, YMI_TILE_256x128_W8 = 17, YMI_TILE_128x256_W8 = 18,
       /* one block per CU, deep LDS-DMA pipeline (96 - 144 KB of stages): the regime the bf16x3 variants need, whose K
        * step is too short to hide a DMA round trip behind one or two co-resident blocks */
       YMI_TILE_128x128_S3 = 19, YMI_TILE_128x128_W8_S3 = 20, YMI_TILE_256x128_W8_S3 = 21, YMI_TILE_128x128_W8_S4 = 22,
       /* round 6, ymi_conv3x3_winograd_f32 only (`| YMI_TILE_H2`, v_planes = 1): the grouped GEMM of the Winograd path as ONE persistent
        * producer / consumer launch over (component, 128-row tile, 256-column block) work items whose chunk stream does not stop at item
        * boundaries (csrc/wgemm.hip); M is bit-identical to the 128 x 128 tile's */
       YMI_TILE_WG_128x256 = 23,
       /* tile | YMI_TILE_X3: the same block tile computed as "bf16x3" — every fp32 operand split exactly into three bf16
        * pieces BY TRUNCATION (24 mantissa bits), 6 of the 9 piece products on v_mfma_f32_32x32x16_bf16 with fp32
        * accumulation; the dropped terms (m*l, l*m, l*l) are < 2^-21 |a b| in the worst case (|m| < 2^-7 |a|, |l| < 2^-15 |a|;
        * simulated maximum 6.4 * 2^-24) and, because truncated pieces carry the sign of their operand, always of the sign of
        * a*b: a one-sided error of +0.63 * 2^-24 |a b| on average.  Cin % 32 == 0
        * layers; available for tiles 1, 2, 3, 5, 6, 7, 8, 9, 11, 12, 16, 17, 19 - 22. */
       YMI_TILE_X3 = 32,
       /* tile | YMI_TILE_H2: the same block tile computed as "fp16x2" — x * s = h + l with two fp16 pieces taken by ROUND TO
        * NEAREST (11 + 11 significant bits + two signs: about two thirds of all fp32 values are represented exactly, the rest
        * with an error of one fp32 ulp; unbiased), s a power of two per tensor (activations, from x_amax) or per filter row;
        * 3 of the 4 piece products (h*l, l*h, h*h) on v_mfma_f32_32x32x16_f16 with fp32 accumulation; the dropped l*l term is
        * <= 2^-22 |a b|, unbiased.  Half the matrix-pipe work of X3 and no byte permutes in the split.  Needs w_h2, scale_h2,
        * x_amax.  Same base tiles as X3. */
       YMI_TILE_H2 = 64,
       /* YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_* (ymi_dcn_v2_forward_f32 only): the software-pipelined gather-GEMM of
        * csrc/dcn.hip — corner loads three K chunks ahead in a register ring, every sample formed once per row block and stored
        * to LDS as the two fp16 planes, no operand split in the MFMA loop.  The low five bits select ITS block tile (enum
        * below), not a YMI_TILE_* of the conv engine.  One dense output, activation none / ReLU / LeakyReLU, no residual. */
       YMI_TILE_DCNP = 128 };
/* block tiles of the pipelined DCN kernel (BM x BN, _W8 = 512-thread blocks) */
enum { YMI_DCNP_64x128 = 1, YMI_DCNP_64x128_W8 = 2, YMI_DCNP_64x64 = 3, YMI_DCNP_128x128_W8 = 4, YMI_DCNP_128x64_W8 = 5,
       YMI_DCNP_32x128 = 6,
       /* 6 / 8 / 10 / 12 waves of 32 x 64, corner loads two chunks ahead (one register slot): one block per CU, rows sized to M / 256 */
       YMI_DCNP_96x128_W6 = 7, YMI_DCNP_128x128_W8_R1 = 8, YMI_DCNP_160x128_W10 = 9, YMI_DCNP_192x128_W12 = 10,
       /* 256 output channels per block (every sample gathered once for all of them): the Cout >= 256 layers, normally with
        * ymi_conv_desc.split_k = S (chunk-aligned K ranges, partial sums through split_ws, deterministic second pass) because their
        * maps are small */
       YMI_DCNP_64x256_W8 = 11, YMI_DCNP_96x256_W12 = 12, YMI_DCNP_128x256_W16 = 13,
       /* 64 x 64 wave tiles (ordinary convolutions only: ymi_conv2d_nhwc_f32) */
       YMI_DCNP_128x256_W8T = 14, YMI_DCNP_128x128_W4T = 15, YMI_DCNP_256x128_W8T = 16,
       /* 32 output channels per block (ordinary convolutions only): the 27-channel conv_offset_mask of a DCN layer
        * (dcn_v2.py:107-112) with its filters zero-padded to 32 — the cost of such a layer is staging its input, so a block
        * spends its waves on rows, not on columns nobody needs */
       YMI_DCNP_128x32_W4 = 17, YMI_DCNP_256x32_W8 = 18, YMI_DCNP_64x32_W2 = 19,
       /* the weight-stationary streaming kernel of csrc/wstat.hip (ordinary convolutions with Cout <= 32 / <= 64, no residual):
        * the filters of the block's K range stay in LDS, every wave streams its own rows from global memory straight into MFMA
        * operand registers, no barrier in the loop.  BM x BN, _W4 / _W8 = waves per block (32 or 64 rows per wave).  The K range
        * of a block must fit 64 KB of LDS: ymi_conv_desc.split_k >= Kpad * BN / 16384 (YMI_EARG otherwise) */
       YMI_DCNP_WS_128x32_W4 = 20, YMI_DCNP_WS_256x32_W8 = 21, YMI_DCNP_WS_256x32_W4 = 22, YMI_DCNP_WS_512x32_W8 = 23,
       YMI_DCNP_WS_128x64_W4 = 24, YMI_DCNP_WS_256x64_W8 = 25, YMI_DCNP_WS_256x64_W4 = 26, YMI_DCNP_WS_512x64_W8 = 27,
       /* round 5, csrc/patch.hip: 3x3 / stride 1 / pad 1, 64 -> 64 channels (conv2 of the first ResNet stage, backbone.py:44-46), one
        * dense output, no residual: a persistent block per CU owns 8 x 16 pixel output tiles, the 10 x 18 x 64 input patch lives in
        * LDS as fp16 planes (loaded ONCE instead of once per filter tap) and the filters live in registers */
       YMI_DCNP_PATCH_C64 = 28,
       /* round 6, csrc/pcconv.hip: the same convolutions as the pipelined tiles (3x3 / pad 1 or 1x1 / pad 0, Cin % 32 == 0, one dense
        * output, optional residual, optional split_k) as a PRODUCER / CONSUMER block: four consumer waves (one per SIMD: fragment
        * reads + MFMAs only) and four producer waves (row requests, the fp32 -> fp16-plane split, the filter LDS-DMAs), so that a
        * request stalled on the vector-memory pipe no longer holds MFMAs back.  BM x BN = the block tile; results are bit-identical
        * to the pipelined tiles' (same products, same order per accumulator) */
       YMI_DCNP_PC_128x128 = 29,
       /* round 6, csrc/patch2.hip: 3x3 / stride 1 / pad 1, Cin % 32 == 0, Cout >= 64, no residual, up to three dense output segments
        * with boundaries at multiples of 128 channels: the input patch of a TH x TW pixel tile (256 / 192 pixels; the shape is picked
        * per map size) lives in LDS one 32-channel chunk at a time, only the filters stream per (chunk, tap) step — the nine taps of a
        * 3x3 convolution read the same pixels, so the A operand crosses the global -> LDS path once instead of nine times */
       YMI_DCNP_PATCH2_256 = 30, YMI_DCNP_PATCH2_192 = 31 };

int ymi_abi_version(void);
const char *ymi_strerror(int code);

/* -- convolution engine ---------------------------------------------------------------- */
int ymi_conv2d_nhwc_f32(const ymi_conv_desc *d, void *stream);
/* algorithmic FLOPs (2*MACs) of a descriptor — used by bench.py's roofline accounting */
double ymi_conv_flops(const ymi_conv_desc *d);
/* which tile the auto heuristic picks (YMI_TILE_*) */
int ymi_conv_pick_tile(const ymi_conv_desc *d);

/* Winograd F(2x2,3x3) variant of ymi_conv2d_nhwc_f32 for kh = kw = 3, stride 1, pad 1, Cin % 32 == 0, Cout % 4 == 0, no
 * residual, one dense output, activation none / ReLU / LeakyReLU (nn.Conv2d 3x3 + BN + ReLU of backbone.py:37-57,
 * yolact.py:319-361, utils/functions.py:163-213).  2.25x fewer multiplications; results differ from the direct kernel by
 * fp32 rounding of the transforms only (<= 2e-6 relative).
 * u: transformed filters [G][CoutPad][C] (U = G g G^T, CoutPad % 128 == 0, zero padded), V / M: caller workspaces of
 * G*T*C and G*T*ceil(Cout/4)*4 floats with T = B*ceil(H/m)*ceil(W/m), G = (m+2)^2 = 16 (m = 2) or 36 (m = 4; 4x fewer
 * multiplications than the direct kernel, fp32 rounding ~4x that of m = 2, still <= 1e-5 relative).  With nseg > 0 the output transform scatters
 * to segments (the shared prediction-head conv, yolact.py:169-193). */
typedef struct {
  const float *x;       /* [B,H,W,C] NHWC */
  const float *u;
  const float *scale;   /* [Cout] or NULL */
  const float *bias;    /* [Cout] or NULL */
  float *y;             /* [B,H,W,Cout] */
  float *V, *M;
  int32_t B, H, W, C, Cout;
  int32_t act;          /* YMI_ACT_NONE / RELU / LEAKY01 */
  int32_t tile;         /* tile of the 16-group GEMM (0 = auto) */
  int32_t nseg;         /* 0: dense y (Cout % 4 == 0, act none/ReLU/LeakyReLU).  1..3: scatter to seg[] like ymi_conv_desc
                         * (any Cout, any YMI_ACT_* per segment; `y` and `act` are ignored) */
  int32_t m;            /* output tile edge: 0 or 2 = F(2x2,3x3) (16 GEMMs), 4 = F(4x4,3x3) (36 GEMMs) */
  int32_t _pad0;
  ymi_conv_seg seg[3];
  const void *u_x3;     /* optional, for tile | YMI_TILE_X3: u pre-split into bf16 planes [G][3][CoutPad][C] (see ymi_conv_desc.w_x3) */
  int32_t cout_alg;     /* real output channels for FLOP accounting (see ymi_conv_desc.cout_alg); 0 = Cout */
  int32_t v_planes;     /* tile | YMI_TILE_H2 only.  1: the input transform writes V directly as two fp16 planes [G][2][T][C]
                         * (scaled by the power of two derived from x_amax * the transform's gain bound; same bytes as fp32 V) and
                         * the grouped GEMM runs with NO operand split in its loop; 0: V stays fp32 and is split on the fly */
  /* tile | YMI_TILE_H2: u as fp16x2 planes [G][2][CoutPad][C] with one scale per (component, filter row);
   * uinv_h2 [G][CoutPad] = 1 / that scale; x_amax / y_amax as in ymi_conv_desc (y_amax may be NULL) */
  const void *u_h2;
  const float *uinv_h2;
  const float *x_amax;
  float *y_amax;
  /* ABI 4, m = 4 only.  x_up != NULL: the layer's input is F.interpolate(x_up, scale_factor 2, bilinear, align_corners False)
   * (+ ReLU if up_relu) of x_up [B,H/2,W/2,C] (H, W even), evaluated INSIDE the input transform with the operation order of
   * ymi_bilinear_nhwc_f32 (bit-identical to materialising it; `x` is ignored).  Saves the write and the read of the upsampled
   * tensor: protonet's interpolate -> conv (utils/functions.py:187-206, yolact.py:588-599).  x_amax = the bound of x_up. */
  const float *x_up;
  int32_t up_relu, _pad4;
  /* ABI 6, m = 4, nseg = 0, Cout = 256 only.  proj_w_h2 != NULL: the layer's output is consumed by ONE 1x1 convolution to
   * proj_cout <= 32 channels (protonet's last layer, utils/functions.py:163-213: conv3x3 + ReLU -> conv1x1) and nothing else; the
   * output transform multiplies every 4x4 tile by those filters while it is in registers / LDS and writes
   * proj_y [B,H,W,proj_ldy] = act2(conv1x1(act(conv3x3(x)))) — `y` is not written (may be NULL), y_amax is not updated.
   * proj_w_h2 / proj_scale_h2: the 1x1's fp16x2 filter planes [2][128][256] and scale_h2 [128] (ymi_conv_desc.w_h2 / scale_h2 of
   * the same layer); the tile is scaled by a power of two derived from its own maximum, so no magnitude bound of y is needed. */
  const void *proj_w_h2;
  const float *proj_scale_h2, *proj_bias;   /* proj_bias may be NULL */
  float *proj_y, *proj_y_amax;              /* proj_y_amax: magnitude-bound slot of proj_y, may be NULL */
  int32_t proj_cout, proj_ldy, proj_act, _pad5;
} ymi_wino_desc;
int ymi_conv3x3_winograd_f32(const ymi_wino_desc *d, void *stream);

/* Raises the magnitude-bound slot `out` (ymi_conv_desc.x_amax layout: 16 sub-slots 64 floats apart, 961 floats in all) to
 * max_i |x[i]| over n floats (n % 4 == 0, x 16-byte aligned): the bound of a tensor that no ymi_conv launch produced (the
 * network input).  The caller zeroes the slot beforehand. */
#define YMI_AMAX_SUB 16
#define YMI_AMAX_STRIDE 64
int ymi_amax_f32(const float *x, long n, float *out, void *stream);

/* -- layout / pooling / resize ---------------------------------------------------------- */
/* x [B,C,H,W] (C<=4) -> y [B,H,W,4], zero-filled channels C..3.  Entry of Yolact.forward (yolact.py:564). */
int ymi_nchw_to_nhwc4_f32(const float *x, float *y, int B, int C, int H, int W, void *stream);
/* the same, and raises the magnitude-bound slot `amax` (ymi_amax_f32's layout, zeroed by the caller) to max |x|: one launch
 * for the two passes over the network input */
int ymi_nchw_to_nhwc4_amax_f32(const float *x, float *y, int B, int C, int H, int W, float *amax, void *stream);
/* y [B,H,W,C] NHWC -> x [B,C,H,W] (returning NCHW tensors to callers that expect them) */
int ymi_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W, void *stream);
/* nn.MaxPool2d(3, stride 2, pad 1) on NHWC, C % 4 == 0 (backbone.py:80,131). */
int ymi_maxpool3x3s2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo, void *stream);
/* F.interpolate(mode='bilinear', align_corners=False) on NHWC, C % 4 == 0, optional ReLU
 * (layers/interpolate.py:4-17; utils/functions.py:194). scale_h/scale_w: the 1/scale_factor torch would
 * use (pass 0 to derive in/out like the size= form). */
int ymi_bilinear_nhwc_f32(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                          float scale_h, float scale_w, int relu, void *stream);

/* The FPN top-down sum as its own in-place pass (ABI 7; yolact.py:332-334: x = F.interpolate(x, size, bilinear) + lat_layer(convout)):
 *   y [B,Ho,Wo,C] += bilinear(x [B,Hi,Wi,C] -> Ho x Wo),  C % 4 == 0, both 16-byte aligned.
 * Same interpolation (fp32 coordinates, same expression) as the YMI_RES_BILINEAR epilogue of ymi_conv2d_nhwc_f32: a lateral
 * convolution without residual followed by this pass equals the fused launch (bit for bit when the same GEMM tile runs).  What it buys: the lateral convolutions
 * of the lower levels depend only on their backbone stage, so the engine launches them early on its side stream, beside the later
 * (under-filled) backbone stages, and only this small pass stays on the critical path.  y_amax: magnitude-bound slot of the SUM
 * (ymi_conv_desc.x_amax layout; zeroed by the caller; may be NULL). */
int ymi_bilinear_add_nhwc_f32(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo, float *y_amax, void *stream);

/* Small direct convolution for FastMaskIoUNet (yolact.py:363-375; config maskiou_net, data/config.py:785-791):
 * x [B,H,W,Cin] NHWC, w [kh*kw*Cin][CoutPad4] (k = (ky*kw+kx)*Cin + c, CoutPad4 = ceil(Cout/4)*4, zero padded),
 * y [B,Ho,Wo,Cout]; optional bias and ReLU. */
int ymi_conv2d_direct_nhwc_f32(const float *x, const float *w, const float *bias, float *y, int B, int H, int W,
                               int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int relu,
                               void *stream);
/* y[b,c] = max_p x[b,p,c]  (F.max_pool2d over the whole map, yolact.py:372) */
int ymi_global_maxpool_nhwc_f32(const float *x, float *y, int B, int HW, int C, void *stream);

/* -- Detect: softmax + decode + Fast NMS (layers/functions/detection.py:32-180) ---------- */
typedef struct {
  const float *conf;    /* [B,P,C] raw class logits if conf_is_logits else post-softmax scores */
  const float *loc;     /* [B,P,4] */
  const float *coef;    /* [B,P,D] mask coefficients (post-tanh) */
  const float *priors;  /* [P,4] cx,cy,w,h */
  int32_t B, P, C, D;   /* C includes background class 0; D = mask_dim */
  int32_t conf_is_logits;
  int32_t top_k;        /* cfg.nms_top_k (200), <= 256 */
  int32_t max_det;      /* cfg.max_num_detections (100), <= 256 */
  float conf_thresh;    /* 0.05 */
  float nms_thresh;     /* 0.5 */
  int32_t cross_class;  /* detection.py:111-135 variant */
  int32_t conf_ld;      /* floats between consecutive priors' class rows in `conf`; 0 = C (dense).  The engine pads the rows
                         * to a multiple of 4 (81 -> 84) so that the head GEMM / Winograd output transform can write them
                         * with 16-byte stores */
  int32_t _pad1;
  /* workspaces (caller-allocated) */
  float *scores_t;      /* [B,C-1,P] class-major foreground scores */
  int32_t *keep;        /* [B,P] 1 if max fg score > conf_thresh */
  int32_t *num_keep;    /* [B] K */
  float *maxsc;         /* [B,P] max foreground score per prior */
  int32_t *argmax;      /* [B,P] its class (0-based foreground index) */
  float *cand_score;    /* [B,(C-1)*top_k] per-class survivors, -1 where empty */
  int32_t *cand_prior;  /* [B,(C-1)*top_k] */
  /* outputs, fixed capacity cap = cross_class ? top_k : max_det */
  int32_t *out_count;   /* [B] */
  float *out_box;       /* [B,cap,4] x1,y1,x2,y2 relative */
  float *out_score;     /* [B,cap] */
  int64_t *out_class;   /* [B,cap] */
  float *out_coef;      /* [B,cap,D] */
  int32_t *out_prior;   /* [B,cap] index of the source prior (parity checks) */
  float *out_rec;       /* optional [B, 1 + cap*(6+D)]: the same detections as ONE fixed-size fp32 record per image — count, then per
                         * detection box 4 | score | class | coef D — i.e. the payload of the data-parallel gather
                         * (yolact_amd/parallel.py) written by the selection kernel itself; rows past the count are left untouched */
} ymi_detect_desc;
int ymi_detect_f32(const ymi_detect_desc *d, void *stream);

/* -- postprocess: mask assembly (layers/output_utils.py:69-99, layers/box_utils.py:327-373) */
/* masks_lo[n,y,x] = crop(sigmoid(sum_k proto[y,x,k]*coef[n,k]), box[n]);  proto [ph,pw,D], coef [N,D], box [N,4] */
int ymi_lincomb_crop_f32(const float *proto, const float *coef, const float *box, float *masks_lo,
                         int ph, int pw, int D, int N, int crop, void *stream);
/* out[n,y,x] = bilinear(masks_lo[n], (h,w))[y,x] > thresh ? 1 : 0 (float32) ; thresh<0 -> no binarise */
int ymi_mask_upsample_f32(const float *masks_lo, float *out, int N, int ph, int pw, int h, int w,
                          float thresh, void *stream);
/* The same two steps for a whole fixed-capacity batch in ONE launch each (the per-image Python loop of eval.py:149 /
 * output_utils.py:35 collapsed): proto [B,ph,pw,D], coef [B,cap,D], box [B,cap,4], masks_lo [B,cap,ph,pw], out
 * [B,cap,h,w]; count [B] int32 ON THE DEVICE = live detections per image (null: all cap); rows past count[b] are
 * left untouched. */
int ymi_lincomb_crop_batch_f32(const float *proto, const float *coef, const float *box, const int32_t *count,
                               float *masks_lo, int B, int cap, int ph, int pw, int D, int crop, void *stream);
int ymi_mask_upsample_batch_f32(const float *masks_lo, const int32_t *count, float *out, int B, int cap, int ph, int pw,
                                int h, int w, float thresh, void *stream);
/* boxes [N,4] relative -> int64 absolute pixels via sanitize_coordinates(cast=False) then truncation */
int ymi_boxes_to_pixels(const float *box, int64_t *out, int N, int w, int h, void *stream);

/* -- FastBaseTransform (utils/augmentations.py:616-658): img [N,H,W,3] float32 BGR -> bilinear (align_corners=False)
 * resize to (oh, ow) -> normalisation -> RGB.  mean_bgr / std_bgr: HOST pointers to 3 floats in BGR order
 * (data/config.py:28-29).  mode: 0 (x-mean)/std, 1 x-mean, 2 x/255, 3 none (backbone.transform, data/config.py:181-202).
 * out_nhwc4 = 0: out [N,3,oh,ow] (the reference's return value); 1: out [N,oh,ow,4] RGB0 = the engine's input layout. */
int ymi_fast_base_transform_f32(const float *img, float *out, int N, int H, int W, int oh, int ow,
                                const float *mean_bgr, const float *std_bgr, int mode, int out_nhwc4, void *stream);

/* -- mask_iou (layers/box_utils.py:98-113; consumer: eval.py:376-384,435-440 prep_metrics) --------------------------
 * masks_a [A,n], masks_b [B,n] float32 (n = h*w) -> iou [A,B] = inter / (area_a + area_b - inter), or inter / area_a when
 * iscrowd.  ws: caller-allocated workspace of A*B + A + B floats.  Exact for 0/1 masks (integer partial sums). */
int ymi_mask_iou_f32(const float *masks_a, const float *masks_b, int A, int B, long n, int iscrowd, float *ws, float *iou,
                     void *stream);

/* -- prep_metrics without float masks (eval.py:376-384,416-440; layers/box_utils.py:33-80,98-113) ------------------------ */
/* out[a,b] = IoU of point-form boxes a, b (iscrowd: intersection / area_a), reference op order.  box_a [A,4], box_b [B,4]. */
int ymi_jaccard_f32(const float *box_a, const float *box_b, int A, int B, int iscrowd, float *out, void *stream);
/* Masks as bits: word j of mask m (bits[m * W64 + j], W64 = ceil(n / 64)) holds pixels 64 j .. 64 j + 63 of the flat [h*w] mask
 * (bit i = pixel 64 j + i; padding bits 0).  ymi_mask_bits_f32 packs float masks [N, n] (bit = value > 0.5: ground truth);
 * ymi_mask_upsample_bits does F.interpolate(masks_lo [N,ph,pw], (h,w), bilinear) > thresh (output_utils.py:91-94) straight into
 * bits: every bit equals the pixel ymi_mask_upsample_f32 writes, 1/32 of the bytes. */
int ymi_mask_bits_f32(const float *masks, int N, long n, uint64_t *bits, void *stream);
int ymi_mask_upsample_bits(const float *masks_lo, int N, int ph, int pw, int h, int w, float thresh, uint64_t *bits, void *stream);
/* iou[a,b] from bit masks: popcount(a & b) / (|a| + |b| - popcount(a & b))  (iscrowd: / |a|) — integers below 2^24 evaluated in
 * fp32 exactly as box_utils.py:98-113 evaluates the float masks: bit-identical results. */
int ymi_mask_iou_bits(const uint64_t *bits_a, const uint64_t *bits_b, int A, int B, long W64, int iscrowd, float *iou, void *stream);

/* -- prep_display, GPU half (eval.py:186-209,228): alpha-composite n instance masks onto a frame.
 * img [h,w,3] float32 0..255 (channel order as given), masks [n,h,w] float32, colors [n,3] float32 0..1 (device, same
 * channel order as img), out [h,w,3] uint8.  n = 0 just converts the frame. */
int ymi_composite_masks_u8(const float *img, const float *masks, const float *colors, int n, int h, int w, float alpha,
                           unsigned char *out, void *stream);

/* -- COCO result wire format (eval.py:320-324 Detections.add_mask -> pycocotools.mask.encode; maskApi.c rleEncode,
 * rleToString): run lengths of the COLUMN-major flattening of each mask, starting with a (possibly empty) run of zeros.
 * masks [N,h,w] float32 (nonzero = foreground; the path's masks are exactly {0,1}), w <= 2048.
 * counts [N,cap] uint32, nruns [N] = true number of runs of each mask (a mask with nruns > cap was truncated: retry
 * with a larger cap). */
int ymi_mask_rle_f32(const float *masks, int N, int h, int w, uint32_t *counts, int32_t *nruns, int cap, void *stream);
/* The same counts for the masks `ymi_mask_upsample_f32(masks_lo, ..., thresh)` WOULD write, without writing them: upsample
 * (bilinear, align_corners=False, output_utils.py:91), threshold (`> thresh`, :94) and run-length encoding in one kernel.
 * masks_lo [N,ph,pw] = the cropped sigmoid masks at prototype resolution (ymi_lincomb_crop_f32).  The COCO result path
 * (eval.py:403-429) then reads N*ph*pw*4 bytes instead of writing and re-reading N*h*w*4.  32*w + 12*h <= 65536 (LDS tables). */
int ymi_mask_rle_upsampled_f32(const float *masks_lo, int N, int ph, int pw, int h, int w, float thresh, uint32_t *counts,
                               int32_t *nruns, int cap, void *stream);
/* counts -> the ASCII string pycocotools stores in 'counts' (delta to counts[i-2] for i > 2, 5 bits per character,
 * 0x20 = continuation, + 48).  str [N,cap_chars] bytes, nchars [N] = true length (> cap_chars: truncated). */
int ymi_rle_to_string(const uint32_t *counts, const int32_t *nruns, int N, int cap, uint8_t *str, int32_t *nchars,
                      int cap_chars, void *stream);

/* -- COCODetection.pull_item's image read (data/coco.py:138-141: `img = cv2.imread(path)` -> uint8 BGR [h,w,3]) ----------
 * cv2.imread on a JPEG = libjpeg-turbo with the library defaults (ISLOW IDCT, fancy upsampling) + EXIF orientation.
 * Split: the serial half (markers + Huffman decoding, baseline and progressive) runs on the HOST and yields the quantised
 * coefficient blocks; dequantisation, IDCT, chroma upsampling, YCbCr -> BGR and the orientation run on the GPU.
 * Bit-exact against libjpeg-turbo (pinned through oracle/jpeg_oracle.py).  Errors: YMI_EFORMAT corrupt / truncated
 * stream, YMI_EUNSUPPORTED a valid file outside this subset (arithmetic coding, lossless, 12-bit, CMYK / YCCK,
 * fractional sampling ratios). */
enum { YMI_JPEG_YCBCR = 0, YMI_JPEG_RGB = 1, YMI_JPEG_GRAY = 2 };
typedef struct {
  int32_t width, height;          /* as stored in the file (before the EXIF orientation is applied) */
  int32_t ncomp;                  /* 1 or 3 */
  int32_t progressive;
  int32_t orientation;            /* EXIF tag 0x0112, 1..8 (1 when absent) */
  int32_t color;                  /* YMI_JPEG_* : how the components map to BGR */
  int32_t out_width, out_height;  /* of the decoded image = (height, width) swapped for orientations 5..8 */
  int32_t hs[3], vs[3];           /* sampling factors as declared */
  int32_t hf[3], vf[3];           /* upsampling factors max_h / h, max_v / v */
  int32_t bw[3], bh[3];           /* block grid of each component, padded to whole MCUs */
  int32_t dw[3], dh[3];           /* real (downsampled) size of each component in samples */
  int64_t coef_count;             /* int16 coefficients in total = sum bw*bh*64 */
  int64_t plane_bytes;            /* device workspace for ymi_jpeg_reconstruct_bgr_u8 (uint8 planes) */
} ymi_jpeg_info;
/* HOST.  Header only: sizes for the caller's allocations. */
int ymi_jpeg_parse(const uint8_t *data, size_t n, ymi_jpeg_info *info);
/* HOST.  Entropy-decode every scan into coefs (host memory, coef_capacity int16; layout [component][by][bx][64], natural
 * (row-major) order inside a block, zero where no scan wrote) and latch the quantisation tables: qt [3][64] natural order. */
int ymi_jpeg_decode_coefs(const uint8_t *data, size_t n, int16_t *coefs, int64_t coef_capacity, uint16_t *qt,
                          ymi_jpeg_info *info);
/* DEVICE.  coefs / qt: device copies of the host results; planes_ws: info->plane_bytes; out: [out_height,out_width,3] uint8 BGR. */
int ymi_jpeg_reconstruct_bgr_u8(const ymi_jpeg_info *info, const int16_t *coefs, const uint16_t *qt, uint8_t *planes_ws,
                                uint8_t *out, void *stream);

/* -- COCODetection.pull_item's ground-truth masks (data/coco.py:144-148: `self.coco.annToMask(obj)`), HOST code -----------
 * pycocotools maskApi.c restated (pycocotools is an unpinned pip dependency of the reference, environment.yml:30):
 * rleFrPoly / rleFrString / rleDecode.  Every call ORs its region into mask [h,w] uint8 ROW-major (0/1), so a polygon
 * list (union of polygons, rleMerge intersect = 0) is a loop of calls on one zeroed mask. */
int ymi_coco_poly_fill_u8(const double *xy, int k, int h, int w, uint8_t *mask);          /* k vertices, x0 y0 x1 y1 ... */
int ymi_coco_rle_fill_u8(const uint32_t *counts, long n, int h, int w, uint8_t *mask);    /* uncompressed counts */
int ymi_coco_rle_string_fill_u8(const char *s, long len, int h, int w, uint8_t *mask);    /* compressed 'counts' string */

/* -- DCNv2 forward (external/DCNv2/src/vision.cpp:5, dcn_v2.h:9-39, dcn_v2_cuda.cu:42-172) ---- */
typedef struct {
  ymi_conv_desc conv;    /* main 3x3 conv: x, packed w, bias, epilogue, outputs (kh=kw=3, pad=1) */
  const float *offmask;  /* [B,Ho,Wo,ldo] NHWC output of conv_offset_mask: ch 2k=dh_k, 2k+1=dw_k, 18+k=mask logit (om_layout 0) */
  int32_t ldo;           /* channel stride of offmask pixels (>= 27) */
  int32_t mask_is_prob;  /* 0: channels 18.. are mask LOGITS, the kernel applies the sigmoid (DCN.forward, dcn_v2.py:118-128 — the
                          * engine's plans); 1: they are the modulation itself, already in [0,1] (dcn_v2_conv / DCNv2.forward,
                          * dcn_v2.py:16-33,85-96, whose callers pass torch.sigmoid(mask)).  Occupies what was tail padding. */
  int32_t om_layout;     /* channel order of offmask.  0: the reference's (dcn_v2.py:118-122: 18 offsets, then 9 masks).  1: per tap
                          * [dh_k, dw_k, mask_k] at channels 3k .. 3k+2 — a caller that owns conv_offset_mask's filters (the engine)
                          * permutes their ROWS once on the host, and the gather kernel then fetches a tap's three values with one
                          * 12-byte load instead of three (it is bound by vector-memory instructions) */
  int32_t _pad1;
} ymi_dcn_desc;
int ymi_dcn_v2_forward_f32(const ymi_dcn_desc *d, void *stream);

/* -- ResNet stem in one launch (backbone.py:126-133 + the layout change of yolact.py:564) ---------------------------------
 * x [B,3,H,W] NCHW fp32 (the normalised image) -> conv 7x7 / 2 / pad 3 (3 -> 64) + folded BN + ReLU -> max-pool 3x3 / 2 / pad 1
 * -> y [B,Hp,Wp,64] NHWC fp32, Hp = ((H - 1) / 2 + 1 - 1) / 2 + 1.  The 64-channel stem output stays in LDS (csrc/stem.hip).
 * fp16x2 arithmetic: filters as the two fp16 planes of engine.Packed(cin_pad = 4).h2() (k = (7 ky + kx) * 4 + c, Kpad 224);
 * the input is scaled per tile from the tile's own maximum, so no magnitude bound of x is needed; y_amax as in ymi_conv_desc. */
typedef struct ymi_stem_desc {
  const float *x;        /* [B,3,H,W] */
  float *y;              /* [B,Hp,Wp,64] */
  int32_t B, H, W, cout_pad;
  const void *w_h2;      /* fp16 planes [2][cout_pad][kpad] */
  const float *scale_h2, *bias;
  float *y_amax;         /* magnitude-bound slot of y (may be NULL) */
  int32_t kpad, _pad0;   /* 224 */
} ymi_stem_desc;
int ymi_stem_pool_f32(const ymi_stem_desc *d, void *stream);

/* -- two chained 1x1 convolutions of a ResNet stage in one launch (csrc/chain.hip: 64 planes; csrc/chain2.hip, ABI 8: 128 / 256) --
 *     y = act_a(scale_a * (W_a x) + bias_a + res)      conv3 + bn3 + shortcut + ReLU of Bottleneck b   (backbone.py:49-55)
 *     z = act_b(scale_b * (W_b y) + bias_b)            conv1 + bn1 + ReLU of Bottleneck b + 1          (backbone.py:41-43)
 * x [M,ldx] (k_a = 64 channels), res [M,res_ld] or NULL, y [M,ldy] (n_a = 256), z [M,ldz] (n_b = 64) or NULL (then only y is
 * computed): NHWC rows, M = B*H*W.  Filters as the fp16x2 planes of the two layers (ymi_conv_desc.w_h2 / scale_h2: planes
 * [2][cout_pad][K]); x_amax / y_amax / z_amax as in ymi_conv_desc (x_amax required).  y is written once and never read back:
 * the second GEMM takes it from LDS.
 * ALIASING: y and z must not overlap each other, res or (for y) x.  z MAY be x exactly in place — z == x and ldz == ldx — and in no
 * other overlapping form: a block reads x rows [32 t, 32 t + 32) before it writes the same rows of z and no other block touches
 * them; any other overlap lets one block's z rows land on x rows another block has not read yet (engine.Plan checks this before
 * it fuses two launches into this one; the library does not). */
typedef struct ymi_chain_desc {
  const float *x, *res;
  float *y, *z;
  const void *w_a_h2, *w_b_h2;
  const float *scale_a_h2, *bias_a, *scale_b_h2, *bias_b;   /* biases may be NULL */
  const float *x_amax;
  float *y_amax, *z_amax;                                    /* may be NULL */
  int64_t M;
  int32_t ldx, res_ld, ldy, ldz;
  int32_t k_a, n_a, n_b;                                     /* (P, 4P, P) with P = 64 (csrc/chain.hip), 128 or 256 (csrc/chain2.hip) */
  int32_t cout_pad_a, cout_pad_b;                            /* rows per filter plane (engine.Packed.CoutPad) */
  int32_t act_a, act_b, _pad0;                               /* YMI_ACT_NONE / RELU / LEAKY01 */
  /* ABI 8, P = 128 / 256 only (ignored at P = 64): the y slices of a block feed ONE accumulation of z and therefore share one
   * power-of-two scale, derived when the kernel starts from a rigorous bound of y:
   *     |y| <= amax(x) * gain_a + bias_max_a + amax(res)
   * gain_a = max_n(|folded BN scale_n| * sum_k |w_a[n, k]|) (> 0), bias_max_a = max_n |bias_a[n]| (>= 0), both from the fp32
   * filters (engine.Packed.l1_gain()); res_amax = the magnitude-bound slot of the residual tensor (required when res != NULL). */
  const float *res_amax;
  float gain_a, bias_max_a;
} ymi_chain_desc;
int ymi_pointwise_chain_f32(const ymi_chain_desc *d, void *stream);

/* -- native plan executor (ABI 7; csrc/plan_exec.cpp) ----------------------------------------------------------------------------
 * The engine's execution plan is a flat list of the calls above on two HIP streams (A = the caller's stream, B = a side stream
 * for the small P4..P7 / Detect / projection-shortcut launches) with record / wait markers between them.  ymi_plan_run walks ops
 * [first, last) in ONE call — the reference drives its layers from Python one nn.Module at a time (yolact.py:564-676); ~230
 * interpreter-level calls per batch-8 step is what that shape costs here, and on a slow host it bounds batch 1.
 * Descriptors are caller-owned and may be patched between runs (input pointer, prototype output).  `events`: handles from
 * ymi_event_create, indexed by RECORD / WAIT ops.  overlap = 0: everything on stream_a, markers skipped (serialised kernels, what
 * per-kernel timing needs).  skip_sections: bit k set = ops whose `section` is k are skipped (bit YMI_SEC_PROTO: the protonet, when
 * cfg.eval_mask_branch is False).  On failure returns the failing call's code and, through failed_op, its index. */
enum { YMI_OP_NOP = 0, YMI_OP_CONV = 1, YMI_OP_WINO = 2, YMI_OP_DCN = 3, YMI_OP_CHAIN = 4, YMI_OP_STEM = 5, YMI_OP_INPUT = 6,
       YMI_OP_BILINEAR = 7, YMI_OP_MAXPOOL = 8, YMI_OP_BILINEAR_ADD = 9, YMI_OP_RECORD = 10, YMI_OP_WAIT = 11, YMI_OP_MEMSET = 12 };
enum { YMI_SEC_NONE = 0, YMI_SEC_BACKBONE = 1, YMI_SEC_FPN = 2, YMI_SEC_PROTO = 3, YMI_SEC_HEADS = 4 };
typedef struct ymi_plan_op {
  int32_t kind;        /* YMI_OP_* */
  int32_t stream;      /* 0 = A, 1 = B */
  int32_t section;     /* YMI_SEC_* (the reference's timer sections, yolact.py:570-607) */
  int32_t _pad;
  const void *desc;    /* CONV / WINO / DCN / CHAIN / STEM: the call's descriptor */
  void *p[3];          /* pointer arguments of the descriptor-less calls (see csrc/plan_exec.cpp) */
  int64_t i[8];        /* their integer arguments; RECORD / WAIT: i[0] = event index; MEMSET: i[0] = bytes */
  double f[2];         /* BILINEAR: scale_h, scale_w */
} ymi_plan_op;
int ymi_event_create(void **ev);
int ymi_event_destroy(void *ev);
int ymi_plan_run(const ymi_plan_op *ops, int first, int last, void *stream_a, void *stream_b, void *const *events, int overlap,
                 int skip_sections, int32_t *failed_op);

/* -- workspace sizes (ABI 7) ------------------------------------------------------------------------------------------------
 * The library never allocates device memory (SURVEY 8(b) "ownership": the reference's extension allocates its own outputs and
 * scratch, dcn_v2_cuda.cu:89-91,165-170; here the CALLER does).  A host that is not the Python shim asks this function how many
 * BYTES a workspace needs instead of re-deriving the formulas in the comments above.  `what` selects the workspace, `desc` points
 * at the descriptor of the call it belongs to (only shape fields are read, pointers may be NULL).  Returns the size in bytes
 * (>= 0; 0 = the call needs no such workspace in this configuration) or a negative YMI_E* code. */
enum {
  YMI_WS_WINO_V = 1,        /* desc: ymi_wino_desc      -> ymi_wino_desc.V  = G*T*C floats, T = B*ceil(H/m)*ceil(W/m), G = (m+2)^2 */
  YMI_WS_WINO_M = 2,        /* desc: ymi_wino_desc      -> ymi_wino_desc.M  = G*T*ceil(Cout/4)*4 floats */
  YMI_WS_SPLITK = 3,        /* desc: ymi_conv_desc      -> ymi_conv_desc.split_ws = split_k * B*Ho*Wo * Cout floats (0 when split_k <= 1) */
  YMI_WS_MASK_IOU = 4,      /* desc: ymi_mask_iou_shape -> ymi_mask_iou_f32's ws = A*B + A + B floats */
  YMI_WS_JPEG_COEFS = 5,    /* desc: ymi_jpeg_info filled by ymi_jpeg_parse -> int16 coefficient buffer (host, and its device copy) */
  YMI_WS_JPEG_PLANES = 6,   /* desc: ymi_jpeg_info      -> ymi_jpeg_reconstruct_bgr_u8's planes_ws */
  YMI_WS_DETECT_SCORES_T = 7,   /* desc: ymi_detect_desc -> scores_t  [B,C-1,P] floats */
  YMI_WS_DETECT_PER_PRIOR = 8,  /* desc: ymi_detect_desc -> keep / maxsc / argmax: [B,P] 4-byte elements EACH */
  YMI_WS_DETECT_CAND = 9,       /* desc: ymi_detect_desc -> cand_score / cand_prior: [B,(C-1)*top_k] 4-byte elements EACH */
  YMI_WS_DETECT_REC = 10,       /* desc: ymi_detect_desc -> out_rec [B, 1 + cap*(6+D)] floats, cap = cross_class ? top_k : max_det */
  YMI_WS_AMAX_SLOT = 11,        /* desc: NULL            -> one magnitude-bound slot (x_amax / y_amax): YMI_AMAX_SUB * YMI_AMAX_STRIDE floats */
  YMI_WS_RLE_COUNTS = 12        /* desc: ymi_rle_shape   -> ymi_mask_rle_f32's counts [N,cap] uint32 (cap = h*w + 1 covers every mask) */
};
typedef struct { int32_t A, B; int64_t n; } ymi_mask_iou_shape;
typedef struct { int32_t N, h, w, cap; } ymi_rle_shape;      /* cap <= 0: the safe capacity h*w + 1 */
int64_t ymi_workspace_bytes(int what, const void *desc);

/* -- box calibration (ABI 7; csrc/calib.hip) ------------------------------------------------------------------------------------
 * Two fixed micro-workloads the caller times (events on `stream`) right before a benchmark, so that a throughput number can be
 * attributed to the box or to the code (bench.py `box_calibration`).  Not on the product path.
 * ymi_calib_mfma_f16: `blocks` x 4 waves, each `iters` x 4 register-resident v_mfma_f32_32x32x16_f16 on random-mantissa operands;
 *   *flops = the fp16 FLOPs executed.  ymi_calib_hbm_copy: dst[i] = src[i] over n_floats (n % 4 == 0, 16-byte aligned), float4
 *   lanes; *bytes = read + written. */
int ymi_calib_mfma_f16(float *out, int blocks, int iters, double *flops, void *stream);
int ymi_calib_hbm_copy(const float *src, float *dst, long n_floats, double *bytes, void *stream);
/* every one of `blocks` 256-thread blocks reads src [n_floats] (n % 4096 == 0; 1 MB stays resident in every XCD's L2) `iters` times with
 * 16-byte loads; *bytes = bytes delivered to the CUs.  The path the GEMM tiles are bound by (global -> CU at L2-hit latency). */
int ymi_calib_l2_read(const float *src, long n_floats, int blocks, int iters, float *out, double *bytes, void *stream);
/* one lane follows i = chain[i] from `start` for `hops` dependent loads and stores where it ended in out[0]; the caller lays the
 * permutation (one entry per 128-byte line over the footprint to probe) and divides its event time by hops: load-to-use latency. */
int ymi_calib_latency(const int32_t *chain, long n, int start, int hops, int32_t *out, void *stream);

/* -- profiling hooks -------------------------------------------------------------------- */
/* When enabled, every conv launch is bracketed by hipEvents on its stream; ymi_prof_read returns
 * (after synchronising) per-launch milliseconds, flops and tile ids. Used by bench.py roofline. */
/* Diagnostics: when buf != NULL every conv block writes 8 x uint64 {xcc<<32 | HW_ID, t_start, t_loop, t_epilogue, t_end,
 * t_transposed, t_computed, 0} (shader clock) at buf[blockIdx*8]; cap_blocks = capacity in blocks (larger grids are not traced). NULL disables. */
int ymi_debug_set_trace(void *buf, long cap_blocks);
int ymi_prof_enable(int on);
int ymi_prof_count(void);
int ymi_prof_read(int i, float *ms, double *flops, int32_t *tile, int32_t *kind);
int ymi_prof_reset(void);

#ifdef __cplusplus
}
#endif
#endif
