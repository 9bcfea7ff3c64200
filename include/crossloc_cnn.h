/*
 * crossloc_cnn.h — C ABI of the MI355X-native scene-coordinate CNN kernels (libcrossloc_hip.so).
 *
 * Replaces what PyTorch dispatches for TransPoseNet.forward
 * (/root/reference/networks/networks.py:466-502; encoder :221-256, decoder :319-360,
 * MLR fusion :483-494): nn.Conv2d 3x3/1x1 stride 1/2, nn.GroupNorm(32, C) + ReLU + residual add,
 * and the decoder head epilogue (mean offset, exp(hardtanh)).
 *
 * The host side (crossloc_amd/networks.py) lowers one forward pass to an array of xl_op records and
 * executes it with ONE call to xl_cnn_run on a HIP stream.  All tensors are float32 device memory.
 * Activations are NHWC inside the network ([B, H, W, C], pixel stride `ld` floats so a tensor can
 * be a channel slice of a wider buffer — the MLR concat is free); the module boundary is NCHW:
 * XL_OP_CONV1 reads the NCHW image, XL_OP_HEAD writes the NCHW [B,4,Ho,Wo] prediction.
 */
#ifndef CROSSLOC_CNN_H
#define CROSSLOC_CNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    XL_OP_CONV1 = 0,     /* 3x3 s1 p1 conv, Cin in {1,3} NCHW input -> NHWC, + bias (networks.py:186-189).  Inference form
                            (Cin 3, Cout 32, 32 groups): with `stats` the op only emits the GroupNorm partial sums of its
                            output (nchunks workgroups per image x reserved_i*256 pixels each); with aux2 = {scale, shift}
                            pairs it writes relu(conv*scale+shift) - the raw output never exists in memory.
                            reserved_i = 0 selects the matrix-pipe form of the same two passes: one workgroup per 16 x 64
                            output tile (nchunks = ceil(Hi/16) * ceil(Wi/64)), `w` = the weight fragments
                            [3 planes][3 window rows][64 lanes][8] bf16 (networks.py, _Plan.conv1_fragments) */
    XL_OP_CONV = 1,      /* 3x3 (pad 1) or 1x1 conv, stride 1 or 2, NHWC, implicit GEMM on fp32 MFMA, + bias.
                            With stats != NULL and groups > 0 the epilogue also emits the GroupNorm partial sums of
                            its output ([B][nchunks][groups][2], one entry per 128-row tile overlapping an image;
                            needs Ho*Wo >= 128 and nchunks >= ceil(Ho*Wo/128)+1); the consuming GN_APPLY sets
                            reserved_i = 128 and no GN_STATS pass is needed. */
    XL_OP_GN_STATS = 2,  /* per-(image, chunk, group) partial sums of x and x^2 (fp64) */
    XL_OP_GN_APPLY = 3,  /* GroupNorm affine from the partial sums, fused ReLU / residual add / ReLU */
    XL_OP_HEAD = 4,      /* fc3 1x1 conv to (n_task + n_pos) channels + mean offset + exp(hardtanh), NCHW out */
    /* ---- backward (what autograd + cuDNN dgrad/wgrad did for loss.backward(), train_single_task.py:298) */
    XL_OP_WGRAD = 5,     /* conv weight gradient: split-K implicit GEMM over pixels + fixed-order reduce -> OIHW */
    XL_OP_GNB_STATS = 6, /* GroupNorm(+fused epilogue) backward, pass 1: per-(image, chunk, channel) sums.  All GNB ops:
                            stats = the layer's forward coefficient table written by XL_OP_GN_FINAL with out2 set
                            ([B][C][2] {scale, shift} followed by [B][C][2] {mean, rstd}); stats2 = scratch of
                            B*nchunks2*C*3 + B*C*6 doubles + B*C*3 floats */
    XL_OP_GNB_APPLY = 7, /* pass 3: dx, optional d(residual) */
    XL_OP_GNB_PARAMS = 8,/* d gamma, d beta and the bias gradient of the preceding conv */
    XL_OP_HEAD_BWD = 9,  /* backward of XL_OP_HEAD: d(input) NHWC, d(fc3 weight), d(fc3 bias) */
    XL_OP_CONV1_WGRAD = 10, /* weight + bias gradient of the NCHW-input first conv (Cout 32); reserved_i = image rows per
                               workgroup (default 16), stats2 = scratch of B*ceil(Hi/rows)*28*Cout floats.  aux2 != NULL (round 5):
                               aux is the gradient w.r.t. the layer's GroupNorm(+ReLU) OUTPUT and the backward apply pass runs on
                               load - aux2 = conv1's raw output (ld_in), w = the forward table of XL_OP_GN_FINAL (out, out2),
                               bias = the coefficients [B][Cout][3] XL_OP_GNB_FINAL left, flags = XL_GN_RELU_IN of that layer;
                               no XL_OP_GNB_APPLY is run for it (conv1 has no data gradient: its dx has no other reader) */
    XL_OP_WINO_IN = 12,     /* Winograd input transform: in [B,Hi,Wi,Cin] -> out V [(m+2)^2][B*Ho*Wo][Cin] with Ho x Wo tiles
                               of m x m outputs; ksize = m: 2 = F(2x2,3x3) (Hi, Wi even), 4 = F(4x4,3x3) (Ho = ceil(Hi/4)), 6 = F(6x6,3x3)
                               (Ho = ceil(Hi/6)).
                               m = 4 and 6: aux2 = per-(image, channel) {scale, shift} pairs of a deferred GroupNorm, applied
                               (with ReLU when flags has XL_GN_RELU_IN) to every in-image pixel before the transform.
                               m = 6 with out2 (the fold form): `in` is the raw output of a convolution whose GroupNorm(+ReLU,
                               +residual aux, +ReLU: flags XL_GN_*) apply pass is performed here, and the activation is also
                               written to out2 (pixel stride ld_out); w, optional: {scale, shift} pairs of the RESIDUAL's own
                               deferred GroupNorm + ReLU (aux is then a raw convolution output too) */
    XL_OP_WINO_OUT = 13,    /* Winograd output transform + bias (+ GroupNorm partial sums): in M [(m+2)^2][tiles][Cin] ->
                               out [B,Hi,Wi,Cin]; ksize = m; reserved_i = tiles per workgroup, nchunks = workgroups per image */
    XL_OP_DUC_HEAD = 14,    /* full-size (semantics) head: x8 pixel shuffle of in [B,Hi,Wi,Cout*64] + bilinear resize to
                               Ho x Wo + fc3 (w [Cout][Cout], bias) + mean (aux) / exp(hardtanh) -> out NCHW [B,Cout,Ho,Wo] */
    XL_OP_DUC_HEAD_BWD = 15, /* backward of XL_OP_DUC_HEAD (pure pixel-shuffle case): d activation, d fc3.weight / bias */
    XL_OP_WINO_DY = 16,     /* Winograd weight gradient: in dY [B,Hi,Wi,Cin] -> out dM = A dY A^T [(m+2)^2][tiles][Cin]; ksize = m
                               (6 = F(6x6,3x3); anything else = F(4x4,3x3)) */
    XL_OP_GNB_FINAL = 18,   /* GroupNorm backward, pass 2 (between STATS and APPLY): totals of the chunk sums, per-(image,
                               channel) apply coefficients and the sums XL_OP_GNB_PARAMS turns into d gamma / d beta / d bias */
    XL_OP_WINO_WFINAL = 17, /* in dU [(m+2)^2][Cout][Cin] -> out dg = G^T dU G, OIHW [Cout][Cin][3][3]; ksize = m as above */
    XL_OP_STEM12 = 19,      /* conv1 (3 -> 32, 3x3 s1) + GroupNorm + ReLU evaluated inside the operand stage of conv2 (32 -> 64, 3x3 s2),
                               csrc/xl_stem_fused.hip (networks.py:186-193 of the reference): the 32-channel full-resolution tensor
                               never exists.  in = image [B,3,Hi,Wi] NCHW; w = conv1 weight fragments (as XL_OP_CONV1, reserved_i = 0);
                               bias = conv1 bias[32]; aux2 = {scale, shift} pairs [B][32][2] of conv1's GroupNorm (XL_OP_GN_FINAL on
                               the statistics of a statistics-only XL_OP_CONV1); aux = conv2 weight fragments
                               [18 K-steps][3 planes][2 column blocks][64 lanes][8] bf16 (networks._Plan.conv2_fragments);
                               stats2 = conv2 bias[64] (read only); out2 = two int32, zero (tile queue, zero
                               again after the launch); out = RAW conv2 output [B,Ho,Wo,64] NHWC (ld_out),
                               Ho = (Hi-1)/2+1; flags & XL_GN_RELU_IN: ReLU behind the GroupNorm.
                               stats (optional, groups = 32, nchunks = ceil(Wo/16) * ceil(Ho/4) * 2): GroupNorm partial sums of
                               the output, fp64 [B][nchunks][32][2] {sum, sum of squares}, every entry written - XL_OP_GN_FINAL
                               with reserved_i = 0 sums them.
                               flags & XL_CONV_PAIR_F16 (round 5): conv2 as three fp16 passes - aux = its weight fragments as pairs
                               [18 K-steps][2 planes {hi, lo}][2 column blocks][64 lanes][8] fp16 followed by the inverse weight
                               scale (one float; networks._Plan.conv2_pair_fragments), scale = {s, 1 / s} of conv1's normalised output */
    XL_OP_S2_DGRAD = 20,    /* data gradient of a 3x3 stride-2 stem convolution (conv2 / conv3 of training plans) on the bf16 matrix pipe,
                               csrc/xl_stem_dgrad.hip: in = dY [B,Hi,Wi,Cin] (Cin = the layer's output channels, 64 or 128), w = weight
                               fragments [9 taps][Cin/16][3 planes][Cout/32][64 lanes][8] bf16 (networks._Plan.s2_dgrad_fragments),
                               out = dX [B,Ho,Wo,Cout] (Cout = 32 or 64, overwritten), stats = two int32, zero (tile queue) */
    XL_OP_FILL0 = 21,       /* round 5: `Cin` bytes at `out` set to zero on the stream (the slots of the gradient maxima, zeroed at the
                               head of a backward op list: XL_OP_GNB_APPLY / XL_OP_WINO_DY with `scale` set record max |result| there
                               as float bits; the pair GEMMs that read those results take XL_CONV_PAIR_AMAX) */
    XL_OP_GNB_PARAMS_LIST = 22, /* XL_OP_GNB_PARAMS of a whole backward pass in ONE launch (28 launches of 10 us each per step of the
                               reference's network): in = device array of `Cin` xl_gnb_params_item entries; the XL_OP_GNB_FINAL
                               ops of those layers carry scale = the item's `sums` (a buffer of the layer's own: the shared
                               scratch behind stats2 is reused by the next layer).  Bit-identical to the per-layer ops. */
    XL_OP_GN_FINAL = 11  /* per-(image, channel) GroupNorm scale/shift [B][C][2] from the partial sums (out);
                            GN_APPLY with aux2 = that buffer skips its own finalisation.  out2 (training plans):
                            [B][C][2] {mean, rstd} for the GroupNorm backward ops.  reserved_i = rows per producer tile (0: all
                            nchunks entries are valid; > 0: one entry per tile overlapping the image; < 0: tiles start at
                            image boundaries), stride > 1 with reserved_i != 0: that many entries per tile (the stride-2 stem
                            convolutions write one per row block of waves) */
};

/* xl_op.flags for XL_OP_CONV / XL_OP_WINO_IN (opt-in split-bf16 GEMMs, csrc/xl_gemm_split.hip): the batched GEMM reads
 * its operands as three bf16 planes per fp32 value (a = a1 + a2 + a3, exact) and multiplies six term pairs on the bf16
 * matrix pipe with fp32 accumulation; XL_OP_WINO_IN (ksize 6) with the flag writes V in that form. */
#define XL_CONV_SPLIT_BF16 64
#define XL_CONV_SPLIT_IL 512   /* with XL_CONV_SPLIT_BF16: the planes of a 16-channel chunk are interleaved,
                                  operand layout [Z][rows][C/16][3][16] bf16 (csrc/xl_gemm_split.hip, 256 x 256 tiles;
                                  Cout a multiple of 256).
                                  nchunks2 <= 1 with both flags: a plain 1x1 stride-1 convolution on the same pipe - only
                                  `w` is pre-split ([Cout][Cin/16][3][16] bf16), `in` / `out` are fp32 NHWC (ld_in, ld_out),
                                  `bias` is added, XL_CONV_NORM_IN applies (Cin <= 512) and `stats` receives the GroupNorm
                                  partial sums of the output per 256-row tile (nchunks >= ceil(Ho*Wo / 256) + 1, 16 channels
                                  per group, Ho*Wo >= 256); XL_OP_GN_FINAL / XL_OP_GN_APPLY take reserved_i = 256 for them */
/* ksize 3, stride 2 with both flags (csrc/xl_stem_split.hip: conv2..conv4 of the stem, Cout 64 / 128 / 256): `stats`
 * (optional, groups = 32) receives the GroupNorm partial sums of the output, fp64 [B][nchunks][32][2], one entry per (tile
 * overlapping the image, row block of waves): (rows per tile, entries per tile) = (128, 4) / (128, 2) / (256, 2) for Cout
 * 64 / 128 / 256 and (128, 2) for the 128-column latency form (reserved_i = 128); nchunks >= (ceil(Ho*Wo / rows) + 1) *
 * entries; XL_OP_GN_FINAL takes reserved_i = rows, stride = entries. */
#define XL_CONV_SPLIT_ACT 1024 /* with both flags above and nchunks2 = Z > 1: the Z batched GEMMs of a Winograd layer whose
                                  activation operand V arrives as plain fp32 [Z][rows][Cin] (XL_OP_WINO_IN without the split
                                  flags) and is split into its three bf16 terms inside the GEMM kernel on its way into LDS, like
                                  the activations of a 1x1 layer: V costs 4 bytes per element in HBM (written once, read
                                  once) instead of 6; only `w` is pre-split ([Z][Cout][Cin/16][3][16] bf16) */
#define XL_CONV_M_TILE_MAJOR 2048 /* on the batched GEMM op (with XL_CONV_SPLIT_ACT) and on the XL_OP_WINO_OUT (ksize 6) that
                                  consumes its result: the product M is laid out [tiles][Z][Cout] instead of [Z][tiles][Cout], so
                                  the Z x Cout block the output transform reads per tile is one contiguous piece */
#define XL_CONV_PAIR_F16 8192   /* round 5, with XL_CONV_SPLIT_BF16 | XL_CONV_SPLIT_IL: the same GEMMs with HALF the matrix-pipe work
                                  (csrc/xl_gemm_pair.hip): an activation a (times the power of two *scale, see xl_op.scale) is the fp16 pair
                                  {hi = fp16(a), lo = fp16((a - hi) * 2^11)} - 22 significand bits, lo in fp16's NORMAL range whatever
                                  the magnitude of a - and a weight w (times a per-matrix power of two that brings max|w| into
                                  [2^14, 2^15)) is the pair {hi, lo = fp16(w - hi)}, from whose hi the kernels derive hs = hi * 2^-11 in
                                  registers; the three products hi*hi, hi*lo and lo*hs are exact in fp32
                                  (v_mfma_f32_32x32x16_f16, fp32 accumulate), what is dropped (lo*lo) is below 2^-22 of the
                                  leading product; the epilogue un-scales (exact).  Layouts, 4 bytes per element (what fp32 costs):
                                  activations [Z][rows][C/8][2][8] fp16, weights [Z][rows][C/16][2][16] fp16 followed by 2 Z floats (xl_cnn_pack_wino_weight_pair /
                                  xl_cnn_pair_weight).  On XL_OP_WINO_IN (ksize 6): V is written in the activation layout;
                                  on XL_OP_CONV with nchunks2 = Z > 1 and without XL_CONV_SPLIT_ACT: `in` is that V (both operands
                                  reach LDS by DMA, no conversion in the GEMM); with XL_CONV_SPLIT_ACT or nchunks2 <= 1: `in` is
                                  fp32 and the kernel forms the pairs on the operand's way into LDS (1x1 layers, normalise-on-load).
                                  xl_op.scale must be set; |a * scale| <= 65504 is the caller's contract (crossloc_amd/networks.py
                                  derives the scale from the GroupNorm bound |gn(x)| <= sqrt(N - 1) |gamma| + |beta|) */
#define XL_CONV_PAIR_AMAX 16384 /* with XL_CONV_PAIR_F16 on an XL_OP_CONV whose fp32 activation operand is a GRADIENT (data gradients of
                                  training plans): `scale` points at the float bits of max |activation| recorded by the pass that wrote the
                                  operand (or its untransformed source: Winograd launches, nchunks2 > 1, allow for |B^T d B| <= 225 max|d|)
                                  and the kernel derives the power-of-two scale itself.  On XL_OP_WGRAD (csrc/xl_wgrad_pair.hip) the dY
                                  operand always works this way: out2 = that slot, scale = {s, 1 / s} of the `in` operand */
/* xl_op.flags for XL_OP_CONV */
#define XL_CONV_DGRAD 1        /* data gradient: `in` is dY (Hi x Wi x Cin = forward output), result is dX; weights
                                  packed with xl_cnn_pack_conv_weight_dgrad; `stride` is the forward stride */
#define XL_CONV_ACCUMULATE 2   /* out += result (second writer of a gradient buffer) */
#define XL_CONV_NORM_IN 128    /* 1x1 forward conv: `in` is the RAW output of the producing convolution and the producer's
                                  GroupNorm is applied while the operand is loaded (aux2 = {scale, shift} pairs
                                  [B][Cin][2] written by XL_OP_GN_FINAL): x -> x*scale + shift, no separate apply pass */
#define XL_CONV_NORM_RELU 256  /* ... followed by ReLU */
#define XL_CONV_NORM_ADD 4096  /* 1x1 layers on the split pipe, with XL_CONV_NORM_IN | XL_CONV_NORM_RELU: ... then `aux` (pixel stride
                                  ld_aux) is added and ReLU applied again - the producer's whole GroupNorm + ReLU + residual + ReLU
                                  epilogue happens while the operand is loaded (no apply pass, the activation is never written) */
/* extra xl_op.flags for XL_OP_GNB_* */
#define XL_GN_ACC_AUX 8        /* d(residual) is accumulated into out2 instead of written */
#define XL_GN_NO_CONV_BIAS 16  /* no conv precedes this GroupNorm: skip the bias gradient */

/* xl_op.flags for XL_OP_GN_APPLY:  v = gn(x); RELU_IN: v = max(v,0); ADD: v += aux; RELU_OUT: v = max(v,0) */
#define XL_GN_RELU_IN 1
#define XL_GN_ADD 2
#define XL_GN_RELU_OUT 4

typedef struct xl_op {
    int32_t type;
    int32_t B, Hi, Wi, Cin;        /* input geometry (XL_OP_GN_*: Hi*Wi pixels of Cin channels) */
    int32_t Ho, Wo, Cout;          /* output geometry */
    int32_t ksize, stride;         /* XL_OP_CONV: 3 or 1; 1 or 2 */
    int32_t groups, nchunks;       /* GroupNorm groups; pixel chunks of the stats pass */
    int32_t flags;
    int32_t ld_in, ld_out, ld_aux; /* pixel strides in floats of in / out / aux (NHWC tensors) */
    int32_t n_task, n_pos;         /* XL_OP_HEAD: task channels (mean added) and positive channels */
    int32_t nchunks2, reserved_i;  /* backward: pixel chunks of the GNB stats pass / split-K factor of WGRAD;
                                      GN_APPLY: reserved_i = conv tile rows when the stats came from a conv epilogue;
                                      WGRAD: groups > 1 = that many independent k=1 weight-gradient GEMMs over consecutive
                                      blocks of in / aux / out (Winograd);
                                      forward CONV: nchunks2 > 1 = that many independent GEMMs over consecutive
                                      blocks of in / w / out (Winograd), reserved_i = 64 selects 64-row tiles */
    float eps;                     /* GroupNorm epsilon (1e-5) */
    float clamp_lo, clamp_hi;      /* XL_OP_HEAD hardtanh bounds (-16.10, 13.82), networks.py:355-356 */
    float reserved;
    const void *in;                /* input activations (image for CONV1) */
    const void *w;                 /* CONV: [Cout][Cin/32][k*k][32] (xl_cnn_pack_conv_weight); CONV1: [27 or 9][Cout];
                                      GN_APPLY: gamma[C]; HEAD: [Cout][Cin] */
    const void *bias;              /* CONV*/
    const void *aux;               /* GN_APPLY: residual tensor; HEAD: mean[n_task] */
    void *stats;                   /* GN_*: fp64 partial sums [B][nchunks][groups][2] */
    void *out;
    const void *aux2;              /* GNB_*: forward output of the fused epilogue (ReLU mask); HEAD_BWD: forward out */
    void *out2;                    /* GNB_APPLY: d(residual), pixel stride in Cout (0: ld_out); GNB_PARAMS: d beta;
                                      HEAD_BWD: d weight */
    void *stats2;                  /* GNB_*: fp64 backward sums; WGRAD: fp32 split-K partials; GNB_PARAMS out3 = d bias */
    const void *scale;             /* XL_CONV_PAIR_F16: device float[2] {s, 1 / s}, s a power of two: the activation operand is
                                      multiplied by s before it is split into its fp16 pair, the result by 1 / s (and by the
                                      weights' own inverse scale) in the epilogue */
} xl_op;

/* one layer of XL_OP_GNB_PARAMS_LIST: sums = [B][C][6] doubles written by XL_OP_GNB_FINAL (its `scale`), gamma[C], results
 * d gamma[C], d beta[C], d bias[C] of the preceding convolution (NULL: none) */
typedef struct xl_gnb_params_item { const double *sums; const float *gamma; float *dgamma, *dbeta, *dbias; int32_t B, C, G, HW; } xl_gnb_params_item;

/* Execute ops[0..n_ops) in order on `stream` (hipStream_t; NULL = default). Asynchronous.
 * Returns 0 or a negative xl status (include/crossloc_dsac.h); on error nothing further is launched. */
int xl_cnn_run(const xl_op *ops, int n_ops, void *stream);

/* sizeof(xl_op) as compiled, so a binding can verify its struct layout. */
int xl_cnn_op_size(void);

/* sizeof of the device-table entries a binding builds: which = 0: xl_pair_item, 1: xl_gnb_params_item (-1: unknown). */
int xl_cnn_item_size(int which);

/* The same op list as ONE executable HIP graph: xl_cnn_graph_capture records the launches of ops[0..n_ops) on `stream`
 * (a created stream, not NULL; nothing executes during the capture, and every kernel of the list must have run once
 * before - the first launch of a kernel configures it) and returns an opaque handle; xl_cnn_graph_launch replays it on a
 * stream with the pointers the ops held at capture time (bind the image / result to fixed buffers); _destroy frees it.
 * Returns XL_ERR_UNSUPPORTED from _launch while per-op profiling is on.  For batch-1 test_single_task.py:347-363. */
int xl_cnn_graph_capture(const xl_op *ops, int n_ops, void *stream, void **graph_out);
int xl_cnn_graph_launch(void *graph, void *stream);
int xl_cnn_graph_destroy(void *graph);

/* Weight layout transform used at load time: PyTorch conv weight [Cout][Cin][k][k] (device) ->
 * [Cout][Cin/32][k*k][32] (device): 32-channel chunk major, tap, channel within chunk; Cin % 32 == 0. */
int xl_cnn_pack_conv_weight(const float *w_oihw_dev, float *w_ohwi_dev, int Cout, int Cin, int k, void *stream);

/* Weight layout for XL_CONV_DGRAD: [Cout][Cin][k][k] -> [Cin][Cout/32][k*k][32] (the transposed operand of the
 * same implicit GEMM; taps are NOT flipped, the kernel mirrors the offsets).  Cout % 32 == 0. */
int xl_cnn_pack_conv_weight_dgrad(const float *w_oihw_dev, float *w_dgrad_dev, int Cout, int Cin, int k, void *stream);

/* Winograd weight transform U = G g G^T of a 3x3 convolution for F(m x m, 3x3), m in {2, 4, 6}, evaluated in float64 per
 * (output, input) channel pair and rounded once (Lavin & Gray 2016; the F(6x6) matrices carry the scaling of wincnn).  What
 * a training step re-runs after every optimizer step (train_single_task.py:298-300: the weights change in place).
 *   dgrad = 0: rows = output channels, K = input channels: U[xi][o][c];
 *   dgrad = 1: the data-gradient operand - the flipped kernel with the channel roles swapped: U[xi][c][o].
 *   form 0: fp32 [(m+2)^2][rows][K];
 *   form 1: three bf16 planes, each [(m+2)^2][rows][K] (the exact split u = u1 + u2 + u3), planes (m+2)^2*rows*K apart;
 *   form 2: interleaved planes [(m+2)^2][rows][K/16][3][16] bf16 (K % 16 == 0). */
int xl_cnn_pack_wino_weight(const float *w_oihw_dev, void *dst_dev, int Cout, int Cin, int m, int dgrad, int form, void *stream);

/* fp32 weight matrix -> interleaved bf16 planes [rows][K/16][3][16] of the split-pipe kernels (K % 16 == 0):
 *   taps = 1: src = [rows][K] (a 1x1 convolution's [Cout][Cin]);
 *   taps = 9: src = OIHW [rows][K/9][3][3], K ordered tap-major: k = (3 ky + kx) * Cin + c (csrc/xl_stem_split.hip);
 *   taps = 0: src = [K][rows], i.e. the transpose - the operand of a 1x1 layer's data gradient, dX = dY W. */
int xl_cnn_split_weight(const float *src_dev, void *dst_dev, int rows, int K, int taps, void *stream);

/* ---- fp16 pair / triple operands (XL_CONV_PAIR_F16, csrc/xl_gemm_pair.hip, csrc/xl_pack.hip)
 * xl_cnn_pack_wino_weight_pair: U = G g G^T of F(6x6,3x3) / F(4x4,3x3) (float64 inside) per frequency z, scaled by the power of two
 * 2^e_z with max|U_z| 2^e_z in [2^14, 2^15), as pairs [Z][rows][K/16][2][16] fp16 {hi, lo}, followed by 2 Z floats:
 * [Z] scratch (the maxima, as float bits) and [Z] inverse scales 2^-e_z - dst holds Z*rows*K*4 + 8 Z bytes.  dgrad as
 * xl_cnn_pack_wino_weight.
 * xl_cnn_pair_weight: a plain [rows][K] matrix (taps as xl_cnn_split_weight), one scale: rows*K*4 + 8 bytes. */
int xl_cnn_pack_wino_weight_pair(const float *w_oihw_dev, void *dst_dev, int Cout, int Cin, int m, int dgrad, void *stream);
int xl_cnn_pair_weight(const float *src_dev, void *dst_dev, int rows, int K, int taps, void *stream);

/* The same packs for a LIST of matrices in three launches (a training loop re-packs every layer after every optimizer step):
 * items_dev = device array of n entries; m = 6 / 4: every entry a Winograd layer (src = OIHW weights, rows = Cout, K = Cin,
 * kind = dgrad, dst as xl_cnn_pack_wino_weight_pair); m = 0: plain matrices (rows, K, kind = taps, dst as xl_cnn_pair_weight).
 * max_elements = the largest rows*K (Cout*Cin) of the list.  Results are bit-identical to the per-matrix calls. */
typedef struct xl_pair_item { const float *src; void *dst; int rows, K, kind, pad; } xl_pair_item;
int xl_cnn_repack_pairs(const xl_pair_item *items_dev, int n, int m, long long max_elements, void *stream);
/* fp32 [rows][K] -> activation pairs [rows][K/8][2][8] fp16 of src * scale[0] (tests, tools; K % 16 == 0) */
int xl_cnn_pair_activation(const float *src_dev, void *dst_dev, long long rows, int K, const float *scale_dev, void *stream);
/* The activation scale of a plan: out[0..1] = {s, 1/s} for GroupNorm outputs, out[2..3] = {s/256, 256/s} for their Winograd
 * transforms (|B^T d B| <= 225 max|d| for F(6x6,3x3)), with s the largest power of two such that
 *     s * sum over the n GroupNorm layers of (sqrtN[i] * max|gamma_i| + max|beta_i|)  <=  2^14
 * - a bound on every activation of a network whose convolutions all read GroupNorm outputs and sums of them (residuals):
 * |gn(x)| <= sqrt(N - 1) |gamma| + |beta| for a group of N elements.  gamma / beta: n device pointers each (host arrays),
 * C[i] channels.  out: 6 floats, 8-byte aligned (out[4..5] is scratch).  Re-run whenever the parameters change. */
int xl_cnn_pair_scales(const float *const *gamma_dev, const float *const *beta_dev, const int *C, const float *sqrtN, int n,
                       float *out_dev, void *stream);

/* Per-op HIP-event timing for measurement (bench.py): between prof_begin and prof_end every op launched by
 * xl_cnn_run is bracketed by two events on its own stream (up to max_records ops).  prof_end waits for the
 * recorded events, writes (index in the op list, op type, elapsed ms) per record and returns the record count
 * (or a negative status); prof_pause(0/1) suspends/resumes recording; prof_filter restricts recording to ops of one
 * type (-1: all) whose nchunks2 is at least the given value (2: the batched Winograd GEMM launches only), so that a
 * throughput run pays the two event records only around the launches it reports. */
int xl_cnn_prof_begin(int max_records);
int xl_cnn_prof_pause(int on);
int xl_cnn_prof_filter(int op_type, int min_nchunks2);
int xl_cnn_prof_end(int32_t *op_index, int32_t *op_type, float *ms, int capacity);

/* Text of the last HIP failure reported by an xl_cnn_* call on this thread. */
const char *xl_cnn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
