/* hrfuser_hip.h — C ABI of libhrfuser_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the HRFuser backbone hot path.  The reference has no
 * native layer: its backbone calls PyTorch ATen ops (cuDNN/cuBLAS underneath).  Every entry
 * point below names the ATen call sites in /root/reference it replaces; a host binding
 * (ctypes here, cgo/JNI elsewhere) passes raw device pointers, sizes and a hipStream_t.
 *
 * Conventions
 *   - all activations fp32, NHWC ("NLC" in the reference) unless explicit element strides
 *     (sB,sY,sX,sC) are taken, which also admit NCHW views (network inputs / input grads);
 *   - weights keep the reference's parameter layouts (Conv2d OIHW, Linear (out,in)) so
 *     state-dicts load unchanged;
 *   - every function enqueues on `stream` and returns immediately: 0 = HRF_OK, 1 = bad argument,
 *     2 = launch failure.  Nothing throws, allocates, or synchronises.  Re-entrant per device.
 *   - "stats" buffers are double[2*C] = (sum, sum of squares | sum, sum*x) accumulated with
 *     atomics: zero them (hipMemsetAsync) before the producing launch.
 *   - transform-on-load modes (tf_mode): 0 none, 1 affine, 2 affine+ReLU, 3 affine+GELU(erf),
 *     4 LayerNorm (rowstat = (rows,2) mean,rstd).  They replace materialised BatchNorm /
 *     LayerNorm / activation tensors of the reference graph.
 */
#ifndef HRFUSER_HIP_H_
#define HRFUSER_HIP_H_

/* Cross-block reductions (BatchNorm moments, LayerNorm / depthwise parameter gradients).
 * Same-address global atomics serialise at ~25 ns each on MI355X (measured: 512 blocks -> 13.5 us),
 * so every accumulator that many blocks add into is REPLICATED: a block adds into copy
 * (block index % HRF_STAT_COPIES) and the (tiny) consumer kernel sums the copies.
 *   - every `stats` / `gstats` argument points to [HRF_STAT_COPIES][2*C] doubles (zeroed by the caller);
 *   - hrf_ln_bwd / hrf_dwconv_bwd_weight take `copy_stride`: element distance between the copies of
 *     their fp32 parameter-gradient accumulators (0 = one copy, plain accumulation into the grads);
 *     hrf_fold_copies adds the summed copies into the gradient arena.
 * 4 copies: every block of a CONSUMER launch re-reads all copies in its finalize-on-load prologue (below), and the request
 * count on those hot cache lines is what that prologue costs (tools/bench_fin.py: +0.5 us per consumer launch at 4 copies, +1.0 at
 * 8, +2.0 at 16, +4.5 ... 11 at 32).  Same-box A/B of the captured training steps (tools/ab_copies.sh, ab_copies_models.sh):
 * round 2: 16 copies 17.20 ms, 8: 16.70, 4: 16.64, 2: 17.1, 1: 18.3; round 5 (kernels 30 % shorter, the prologue a larger share):
 * HRFuser-T 32 copies 16.14 ms, 16: 12.58, 8: 11.61, 4: **11.47**, 2: 11.54; STF 8: 23.63, 4: 23.39; HRFuser-B 8: 42.98, 4: 42.98. */
#ifndef HRF_STAT_COPIES      /* (a build may override it: tools/ab_copies.sh) */
#define HRF_STAT_COPIES 4
#endif

/* BatchNorm finalize ON LOAD (consumer side).  A train-mode BatchNorm needs its batch moments complete before anything
 * can be normalised, i.e. a grid-wide dependency between the producing convolution and its consumer; the kernel boundary
 * between the two already is that dependency.  A consumer kernel that is handed one of these for an input tensor derives
 * the scale/shift of that BatchNorm itself, in its prologue, from the producer's replicated moments `stats`
 * ([HRF_STAT_COPIES][2*C] doubles) - the arithmetic of hrf_bn_finalize, evaluated redundantly by every block (C <=
 * HRF_FIN_MAXC) - so the ~330 separate hrf_bn_finalize launches of a training step (and their launch boundaries on the
 * dependency chain) disappear.  `write` != 0: block 0 of the launch also stores scale/shift/mean/invstd (read by the
 * backward kernels) and updates the running statistics; exactly one consumer launch per BatchNorm and step sets it.
 * hrf_bn_bfin_t is the backward counterpart: the data-gradient kernel of the PRODUCING convolution derives the
 * BatchNorm-backward coefficients (dy = cA*du + cB*y + cC) from `gstats` = replicated (sum du, sum du*y); `write` != 0:
 * block 0 stores cA/cB/cC (read by the weight-gradient kernel) and adds dgamma / dbeta. */
#define HRF_FIN_MAXC 576
typedef struct hrf_bn_fin {
  const double* stats;
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* scale; float* shift; float* mean; float* invstd;
  double count; float eps; float momentum; int update_running; int write; int C;
  int copies;                    /* replication of `stats`: 0 = HRF_STAT_COPIES; 1 = already folded (SyncBN: the packed,
                                    all-reduced moments of hrf_bn_pack) */
  const double* count_ptr;       /* nullable: the sample count read from the DEVICE instead of `count` - SyncBN all-reduces
                                    the ranks' row counts with the moments (hrf_bn_pack `rows`), so ranks with unequal
                                    batches normalise by the true global count, as torch.nn.SyncBatchNorm does */
} hrf_bn_fin_t;
typedef struct hrf_bn_bfin {
  const double* gstats;
  const float* gamma; const float* mean; const float* invstd;
  float* dgamma; float* dbeta; float* cA; float* cB; float* cC;
  double count; int train; int write; int C;
  int copies;                    /* as in hrf_bn_fin_t */
  const double* gstats_local;    /* SyncBN: this rank's folded moments (parameter gradients); NULL = gstats */
  float pgrad_scale;             /* gstats_local == NULL: dgamma / dbeta += pgrad_scale * (value from gstats); 0 means 1.
                                    SyncBN without a rank-local copy: the all-reduced sums give the GLOBAL dgamma / dbeta,
                                    every rank adds 1/world of it and the gradient all-reduce restores the sum */
  const double* count_ptr;       /* as in hrf_bn_fin_t */
} hrf_bn_bfin_t;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense convolution / Linear engine (fp32 MFMA implicit GEMM) -------------------------
 * Replaces F.conv2d(k=1|3, groups=1) and F.linear:
 *   stems hrnet.py:341-358, hrfuser_hrformer_based.py:380-396; Bottleneck resnet.py:166-205;
 *   transitions hrnet.py:430-459; CrossFFN 1x1 hrformer.py:268,280; fuse 1x1
 *   hrformer.py:511-517,544-551; qkv/out_proj hrformer.py:84,86; q/k/v/out_proj
 *   hrfuser_hrformer_based.py:92-96.  Linear = 1x1 conv over a (1,1,rows,C) view.            */
int hrf_conv_fwd(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                 const float* w, const float* bias, int KH, int stride, int Cout,
                 float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                 int tf_mode, const float* tf_scale, const float* tf_shift,
                 const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat, float ln_eps, void* stream);
/* The same forward with a SPLIT OVER K for deep contractions that leave the chip empty (3x3, K = 9*Cin >= 1024, <= 128 row blocks:
 * the 256 -> 36 stride-2 transition, hrnet.py:430-459): four K slices write partial tiles (slice 0 into y, the others into
 * `scratch`), a second launch adds them in a fixed order (bit-reproducible) and takes the moments.
 * hrf_conv_fwd_split_scratch -> floats of scratch this problem wants (0: it would not be split - call hrf_conv_fwd);
 * hrf_conv_fwd_split with scratch == NULL is hrf_conv_fwd. */
long hrf_conv_fwd_split_scratch(int sB, int sY, int sX, int sC, int B, int H, int W, int Cin, int KH, int stride, int Cout,
                                int ldY, int yoff);
int hrf_conv_fwd_split(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                       const float* w, const float* bias, int KH, int stride, int Cout,
                       float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                       int tf_mode, const float* tf_scale, const float* tf_shift,
                       const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat, float ln_eps,
                       float* scratch, void* stream);
/* dX (or, epi=1, dU = dX*act'(scale*xraw+shift) plus (sum dU, sum dU*xraw) for the producer BN).
 * (cA,cB,cC) != NULL applies the BatchNorm backward on load: dy = cA*du + cB*yraw + cC; with `bfin` (nullable, Cout <=
 * HRF_FIN_MAXC) the coefficients are derived in the kernel prologue from bfin->gstats instead of being read (see
 * hrf_bn_bfin_t).  `tf_fin` of hrf_conv_fwd (nullable, tf_mode 1..3, Cin <= HRF_FIN_MAXC) is the forward analogue.    */
int hrf_conv_bwd_data(const float* dy, int ldD, int doff, const float* yraw,
                      const float* cA, const float* cB, const float* cC, const hrf_bn_bfin_t* bfin,
                      const float* w, int KH, int stride, int Cout,
                      int B, int H, int W, int Cin,
                      float* dx, int sB, int sY, int sX, int sC, int accumulate,
                      int epi, const float* xraw, int ldXr, const float* tf_scale,
                      const float* tf_shift, int act, double* stats, void* stream);
/* dW += , dbias += (split-K over pixels, fp32 atomics: zero or pre-load the targets).          */
int hrf_conv_bwd_weight(const float* dy, int ldD, int doff, const float* yraw,
                        const float* cA, const float* cB, const float* cC,
                        const float* x, int sB, int sY, int sX, int sC,
                        int B, int H, int W, int Cin, int KH, int stride, int Cout,
                        int tf_mode, const float* tf_scale, const float* tf_shift,
                        const float* tf_rowstat, float* dw, float* dbias, void* stream);

/* ---- the front-end 3x3 convolutions on tap-major PACKED weights (csrc/conv3x_engine.hip) ----------------------------------
 * The stems' second convolution, the Bottleneck conv2 of layer1 and transition1 (hrnet.py:341-358,417-459;
 * resnet.py:263-302; hrfuser_hrformer_based.py:380-396) are the FLOP carriers of the backbone.  Their kernels want the weights
 * as wp[tap][n][k] (a K step = one contiguous run per output channel; backward-data = the same kernel on the transposed pack):
 *   hrf_conv3x_pack_size  floats of one pack: 9 * Np * Kp, N rounded up to 64, K to 32 (zero rows / columns past the tensor);
 *   hrf_conv3x_pack       ONE launch (per 16 jobs) packs any number of OIHW tensors w = [Cout][Cin][3][3]:
 *                         dir 0 (forward operand, N = Cout, K = Cin):        wp[tap][n][k] = w[n][k][tap]
 *                         dir 1 (backward-data operand, N = Cin, K = Cout):  wp[tap][n][k] = w[k][n][tap]
 *                         `jobs` is a HOST array; the packs are caller-owned scratch, refreshed whenever w changes (once per step);
 *   hrf_conv3x_supported  1 when hrf_conv_fwd_packed (dir 0) / hrf_conv_bwd_data_packed (dir 1) take the shape:
 *                         KH = 3, NHWC rows; dir 0: stride 1, Cout > 32; dir 1: stride 1, Cin > 32, or stride 2, Cin > 32 and
 *                         Cout <= 64; on-load BatchNorm <= 256 channels.
 *   hrf_conv_fwd_packed / hrf_conv_bwd_data_packed: hrf_conv_fwd / hrf_conv_bwd_data (same arguments, same results) with
 *                         `wp` = the pack of `w` in the matching direction; HRF_ERR_ARG for unsupported shapes. */
typedef struct hrf_conv3x_pack_job {
  const float* w; float* wp;
  int Cout, Cin, dir;
} hrf_conv3x_pack_job_t;
long hrf_conv3x_pack_size(int Cout, int Cin, int dir);
int hrf_conv3x_pack(const hrf_conv3x_pack_job_t* jobs, int n, void* stream);
int hrf_conv3x_supported(int Cin, int Cout, int KH, int stride, int dir);
int hrf_conv_fwd_packed(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                        const float* w, const float* bias, int KH, int stride, int Cout,
                        float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                        int tf_mode, const float* tf_scale, const float* tf_shift,
                        const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat, float ln_eps,
                        const float* wp, void* stream);
int hrf_conv_bwd_data_packed(const float* dy, int ldD, int doff, const float* yraw,
                             const float* cA, const float* cB, const float* cC, const hrf_bn_bfin_t* bfin,
                             const float* w, int KH, int stride, int Cout,
                             int B, int H, int W, int Cin,
                             float* dx, int sB, int sY, int sX, int sC, int accumulate,
                             int epi, const float* xraw, int ldXr, const float* tf_scale,
                             const float* tf_shift, int act, double* stats, const float* wp, void* stream);

/* The weight gradient of the LARGE 3x3 problems (the stems' 64 -> 64 convolutions, transition1: >= 16 384 output pixels, >= 32 input
 * channels, NHWC rows, no bias gradient) on the LDS-staged 32x32x2 kernel of csrc/wgrad3x_engine.hip: the pixel splits leave their
 * sums as plain stores in caller-owned scratch, a second small launch of the same call folds them into dw (no atomics: bit-
 * reproducible).  hrf_conv_bwd_weight_scratch -> floats of scratch the problem wants (0: it would not use any - call
 * hrf_conv_bwd_weight); hrf_conv_bwd_weight_s = hrf_conv_bwd_weight with that scratch (contents irrelevant; NULL, or a problem
 * that wants none: plain hrf_conv_bwd_weight).  Launches at once, also between hrf_wgrad_group_begin / _end. */
long hrf_conv_bwd_weight_scratch(int sB, int sY, int sX, int sC, int B, int H, int W, int Cin, int KH, int stride, int Cout,
                                 int tf_mode, int has_bias);
int hrf_conv_bwd_weight_s(const float* dy, int ldD, int doff, const float* yraw,
                          const float* cA, const float* cB, const float* cC,
                          const float* x, int sB, int sY, int sX, int sC,
                          int B, int H, int W, int Cin, int KH, int stride, int Cout,
                          int tf_mode, const float* tf_scale, const float* tf_shift,
                          const float* tf_rowstat, float* dw, float* dbias, float* scratch, void* stream);

/* ---- depthwise 3x3 convolution, pad 1, stride 1|2, NHWC (F.conv2d groups=C) -----------------
 * CrossFFN hrformer.py:271-277 (bias, stride 1, input = GELU(BN(h1)) applied on load) and the
 * fuse-down chains hrformer.py:532-541 (stride 2, no bias).  w is (C,1,3,3).                    */
int hrf_dwconv_fwd(const float* x, int B, int H, int W, int C, const float* w, const float* bias,
                   int stride, int tf_mode, const float* tf_scale, const float* tf_shift, float* y,
                   double* stats, const hrf_bn_fin_t* tf_fin, void* stream);
int hrf_dwconv_bwd_data(const float* dy, const float* yraw, const float* cA, const float* cB,
                        const float* cC, const hrf_bn_bfin_t* bfin, const float* w, int stride, int B, int H, int W, int C,
                        float* dx, int accumulate, int epi, const float* xraw, const float* tf_scale,
                        const float* tf_shift, int act, double* stats, void* stream);
/* hrf_dwconv_bwd_data (stride 1, epi = 1: the input was act(tf_scale*xraw+tf_shift)) that ALSO accumulates the weight /
 * bias gradient of the same convolution (what hrf_dwconv_bwd_weight computes from dy, yraw, x = act(.)): dW[ky][kx] =
 * sum_p x[p]*dy'[p-(ky-1,kx-1)] uses the staged dy' element dx[p] needs anyway, and x[p] is a by-product of act'.
 * dw / dbias (nullable bias): [HRF_STAT_COPIES] replicated fp32 accumulators, `copy_stride` apart (see above). */
int hrf_dwconv_bwd_data_weight(const float* dy, const float* yraw, const float* cA, const float* cB,
                               const float* cC, const hrf_bn_bfin_t* bfin, const float* w, int B, int H, int W, int C,
                               float* dx, const float* xraw, const float* tf_scale, const float* tf_shift, int act,
                               double* stats, float* dw, float* dbias, long copy_stride, void* stream);
int hrf_dwconv_bwd_weight(const float* dy, const float* yraw, const float* cA, const float* cB,
                          const float* cC, const float* x, int B, int H, int W, int C, int stride,
                          int tf_mode, const float* tf_scale, const float* tf_shift, float* dw,
                          float* dbias, long copy_stride, void* stream);

/* ---- 7x7 windowed attention core (per window x head: q k^T*d^-1/2 + RPB, softmax, @v) --------
 * WindowMSA hrformer.py:103-128 / WindowMCA hrfuser_hrformer_based.py:115-148 incl. the centre
 * zero-pad, window partition/merge and de-pad of hrformer.py:196-236 / hrfuser...:202-248 as
 * index math.  q/k/v/o are (B*H*W, ld) NHWC projections with column offsets (packed qkv = one
 * buffer, three offsets).  kpad/vpad (C) = key/value of a padded token (the projection biases);
 * padded keys are attended, not masked (with_pad_mask=False).  head_dim in {8,16,18,32,39}.
 * bwd: dq/dk/dv written for every real token; dkpad/dvpad/drpb accumulate (+=, atomics).        */
int hrf_window_attn_fwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                        const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                        const float* rpb, float* o, int ldo, int B, int H, int W, int C, int heads,
                        void* stream);
int hrf_window_attn_bwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                        const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                        const float* rpb, const float* dout, int lddo,
                        float* dq, int lddq, int dqoff, float* dk, int lddk, int dkoff,
                        float* dv, int lddv, int dvoff, float* dkpad, float* dvpad, float* drpb, long copy_stride,
                        int B, int H, int W, int C, int heads, void* stream);

/* ---- fused window-attention block (csrc/attn_block.hip): the whole token-local half of an HRFormerBlock
 * (hrformer.py:365-373: norm1 -> LocalWindowSelfAttention :184-236 / WindowMSA :96-131 -> residual -> norm2 -> CrossFFN
 * layers[0] :268) or of one modality of a fusion block (hrfuser_hrformer_based.py:305-317: norm1[k] / norm2[k] ->
 * MultiWindowCrossAttention :189-248 / WindowMCA :106-151 incl. Dropout -> DropPath + residuals; after the last modality
 * norm3 + the CrossFFN head) in ONE launch per direction, one workgroup per 7x7 window; head_dim 18 (HRFuser-T / STF
 * widths 18 / 36 / 72 / 144: hrf_attn_block_supported).  Rows are NHWC (B*H*W, C); weights keep the reference layouts
 * (Linear (out, in); for the packed qkv Linear pass three pointers into the same tensor).
 *   forward : out = res (+ res2) + mask*mscale*rowscale[b] * out_proj(attn(LN_q(xq), LN_kv(xkv)));  xkv == xq: self-
 *             attention (LN_kv unused).  mask / rowscale / res2 nullable; mscale = 1 when unused.
 *             out_rowstat (nullable): LayerNorm (mean, rstd) of the out rows with out_eps.
 *             w1 != NULL: h1 = LN_2(out) w1^T + b1 (hidden = 4*C rows, pre-BatchNorm) and, stats1 != NULL, its
 *             replicated moments [HRF_STAT_COPIES][2*hidden] (+=).
 *   backward: gout = dL/d out rows; du1 (+ cA1/cB1/cC1, or bfin1 to derive them on load like hrf_conv_bwd_data) = the
 *             gradient reaching h1 through its BatchNorm (dy1 = cA*du1 + cB*h1 + cC).  With gx = gout + (gradient through
 *             LN_2 / w1):  dres (+)= gx;  dq (+)= gradient through LN_q (+ gx when dq_add_res: self-attention, where the
 *             residual row IS the query row);  dkv (+)= gradient through LN_kv (+ gx when dkv_add_res: the modality row
 *             is both residual and key/value source).  Each *_acc flag selects += over =; NULL outputs are skipped.
 *             Parameter gradients are NOT accumulated with atomics: every workgroup writes the complete partial sums of
 *             its window to its own slot pslot[blockIdx * slot_stride + off_*] (plain stores, deterministic);
 *             hrf_fold_slots adds the slots into the gradient arena.  Slot layout (floats; off_* < 0: not produced):
 *             w1 [4C][C], b1 [4C], ln2 gamma / beta [C]; wo [C][C], bo [C]; wq / wk / wv [C][C], bq / bk / bv [C];
 *             LN_q gamma / beta, LN_kv gamma / beta [C].  The relative-position-bias gradient is a leaf: the kernel leaves
 *             dS[key][query] of every (window, head) in ds_plane [windows][heads][49][49]; hrf_rpb_grad (any time later)
 *             gathers it into the replicated accumulator drpb [HRF_STAT_COPIES][169][heads] (copy_stride apart).
 *             hrf_attn_block_bwd_supported: widths 18 / 36.
 * The last five fields are derived by the library.                                                                     */
typedef struct hrf_attn_block {
  int B, H, W, C, heads;
  const float* xq; const float* xkv;
  const float* lnq_g; const float* lnq_b; const float* lnkv_g; const float* lnkv_b; float ln_eps;
  const float* wq; const float* bq; const float* wk; const float* bk; const float* wv; const float* bv;
  const float* rpb; const float* wo; const float* bo;
  const float* res; const float* res2;
  const float* mask; float mscale; const float* rowscale; int rows_per_sample;
  float* out; float* out_rowstat; float out_eps;
  const float* ln2_g; const float* ln2_b; const float* w1; const float* b1; float* h1; double* stats1; int hidden;
  const float* gout; const float* du1; const float* cA1; const float* cB1; const float* cC1; const hrf_bn_bfin_t* bfin1;
  float* dres; int dres_acc; float* dq; int dq_acc; int dq_add_res; float* dkv; int dkv_acc; int dkv_add_res;
  float* pslot; long slot_stride; float* ds_plane;
  float* gx_park;      /* [windows][64][ceil(C/16)*16] floats of step-lifetime scratch (18-channel backward, 8-wave form; else unused) */
  /* CrossFFN tail of the PRECEDING block formed on load (hrformer.py:371-372 x = x' + DropPath(GELU(BN3(fc3(.)))), self-
   * attention only, tail_raw != NULL): the block input rows are x = tail_res + tail_rowscale[b] * GELU(tail_scale *
   * tail_raw + tail_shift) (tail_fin: scale / shift derived on load from the producer's moments, as hrf_conv_fwd's fin).
   * forward : every workgroup forms the rows of its window and WRITES them to x_out (== xq == xkv == res: the block's
   *           residual and, in the backward, its recomputation source).
   * backward: with dx = the block-input gradient (what dq receives; dq_add_res = 1): tail_du = dx * rowscale * GELU'(u),
   *           tail_gstats [HRF_STAT_COPIES][2*C] += (sum tail_du, sum tail_du * tail_raw): the hrf_act_bwd launch of the
   *           preceding block's BatchNorm is gone; tail_scale / tail_shift must be in memory (published by the forward). */
  const float* tail_res; const float* tail_raw; const float* tail_scale; const float* tail_shift; const float* tail_rowscale;
  const hrf_bn_fin_t* tail_fin; float* x_out; float* tail_du; double* tail_gstats;
  int off_w1, off_b1, off_g2, off_bt2, off_wo, off_bo, off_wq, off_bq, off_wk, off_bk, off_wv, off_bv;
  int off_gq, off_btq, off_gkv, off_btkv, off_rpb;
  int nWh, nWw, pt, pl; float scale;
} hrf_attn_block_t;
int hrf_attn_block_supported(int C, int heads);
int hrf_attn_block_bwd_supported(int C, int heads);
int hrf_attn_block_fwd(const hrf_attn_block_t* p, void* stream);
int hrf_attn_block_bwd(const hrf_attn_block_t* p, void* stream);
int hrf_rpb_grad(const float* ds_plane, int nwin, int heads, float* drpb, long copy_stride, void* stream);

/* ---- eval-mode CrossFFN in one launch (csrc/ffn_eval.hip) ---------------------------------------------------------------
 * out = x + GELU(BN3(fc3( GELU(BN2(dw3x3( GELU(BN1(fc1( LN(x) ))) ))) ))) with FROZEN BatchNorm statistics: the second half
 * of an HRFormerBlock / HRFuserFusionBlock in eval mode - hrformer.py:351 (norm2), :267-295 (CrossFFN.forward), :371-372
 * (residual; DropPath is the identity) resp. hrfuser_hrformer_based.py:291,315-316.  x, out: (B,H,W,C) NHWC fp32; w1 [hidden][C]
 * + b1, wd [hidden][3][3] + bd, w3 [C][hidden] + b3 in their Conv2d layouts; (s_k, t_k) = the frozen-statistics affine of
 * BatchNorm k (scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean * scale).  hidden = 4 C;
 * hrf_ffn_eval_supported: C in {18, 36} (the two finest branches of HRFuser-T / STF, where a launch has hundreds of tiles);
 * other widths keep the per-op kernels.  The 4C-wide hidden tensor never leaves the chip.                                   */
typedef struct hrf_ffn_eval {
  int B, H, W, C, hidden;
  const float* x;
  const float* ln_g; const float* ln_b; float ln_eps;
  const float* w1; const float* b1; const float* s1; const float* t1;
  const float* wd; const float* bd; const float* s2; const float* t2;
  const float* w3; const float* b3; const float* s3; const float* t3;
  float* out;
} hrf_ffn_eval_t;
int hrf_ffn_eval_supported(int C, int hidden);
int hrf_ffn_eval(const hrf_ffn_eval_t* p, void* stream);
/* the same gather for EVERY fused layer of a step in one launch: seg = nseg rows of 5 longs on the device {offset (floats) of the
 * layer's ds_plane in `planes`, windows, heads, address of its drpb accumulator, copy_stride}; max_nwin / max_heads = the
 * largest of the rows (launch geometry).                                                                               */
int hrf_rpb_grad_all(const float* planes, const long* seg, int nseg, int max_nwin, int max_heads, void* stream);
/* dst[map[i]] += sum_{s < nslots} slots[s*slot_stride + i]  (i < n; map[i] < 0: skipped).  One launch folds the slots of
 * every fused layer of a step: seg = nseg rows of 5 longs {slot offset (floats) into `slots`, nslots, slot_stride, n,
 * offset into `map`} on the device.                                                                                  */
int hrf_fold_slots(const float* slots, const long* seg, int nseg, const int* map, float* dst, long max_n, void* stream);

/* ---- BatchNorm bookkeeping (F.batch_norm, 329 call sites: every build_norm_layer(norm_cfg)) ---
 * stats = (sum y, sum y^2) from the producing conv -> scale/shift used by consumers' loaders,
 * saved mean/invstd, running-stat update (momentum, unbiased var).  With SyncBN the host
 * all-reduces `stats` (RCCL) between the producer and this call.                               */
int hrf_bn_finalize(const double* stats, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, double count, float eps, float momentum, int update_running,
                    float* scale, float* shift, float* mean_out, float* invstd_out, int C, void* stream);
/* gstats = (sum du, sum du*yraw): dgamma += , dbeta += , and the on-load backward coefficients
 * dy = cA*du + cB*yraw + cC.  train=0: frozen statistics (eval / norm_eval).  gstats_local
 * (nullable) = this rank's moments for the parameter grads when gstats was all-reduced (SyncBN). */
int hrf_bn_bwd_finalize(const double* gstats, const double* gstats_local, const float* gamma,
                        const float* mean, const float* invstd, double count, int train, float* dgamma, float* dbeta, float* cA, float* cB,
                        float* cC, int C, void* stream);

/* SyncBN over RCCL, several mutually independent BatchNorms per collective (the three sensor streams at equal depth, the
 * branches of one HRModule, the fuse layers of one exchange): hrf_bn_pack folds the replicated moments of n layers into
 * `packed` (2*C doubles per layer, back to back) -> ONE all-reduce of `packed` by the host -> hrf_bn_finalize_packed /
 * hrf_bn_bwd_finalize_packed finalise all n layers from the packed sums (the `stats` / `gstats` / `write` fields of the
 * structs are ignored; packed_local = this rank's copy of `packed` taken before the all-reduce: parameter gradients use the
 * rank-local moments).  stats / C / fins / bfins are HOST arrays of n entries.  `rows` (nullable HOST array of n doubles):
 * this rank's sample count of every layer, stored behind the sums at packed[sum_i 2*C_i + i] so that the SAME all-reduce
 * yields the global counts (hrf_bn_fin_t.count_ptr points there): `packed` then holds sum_i 2*C_i + n doubles.            */
int hrf_bn_pack(const double* const* stats, const int* C, int n, const double* rows, double* packed, void* stream);
int hrf_bn_finalize_packed(const hrf_bn_fin_t* fins, int n, const double* packed, void* stream);
int hrf_bn_bwd_finalize_packed(const hrf_bn_bfin_t* bfins, int n, const double* packed, const double* packed_local /* nullable: see pgrad_scale */, void* stream);

/* ---- peer-to-peer SyncBN exchange over xGMI (csrc/p2p_exchange.hip): what torch.nn.SyncBatchNorm's all-gather / all-reduce
 * of the per-layer moments does in the reference (norm_cfg type SyncBN, configs/_base_/models/cascade_rcnn_hrfuser_fpn_nus_clr_
 * fusion.py:2), without a communicator: every rank exposes an INBOX (fine-grained device memory, mapped by its peers through
 * IPC handles) of `world` source regions x 2 generation parities x slot_doubles doubles plus world x 2 x nslots flags; every
 * BatchNorm layer and direction owns a static slot (slot_off doubles into a region, flag index slot_id; 2*C + 1 doubles).
 * hrf_p2p_exchange = ONE launch on the lane that needs the result: fold this rank's replicated moments of n layers (the job of
 * hrf_bn_pack; rows as there), store them into the slot of every PEER's inbox, release flag = *gen at system scope, spin until
 * all peers show *gen (time-out: *err = ((source+1) << 32) | (slot_id+1), the launch completes with garbage and the
 * host raises), add the contributions in rank order into `packed` (layout of hrf_bn_pack: sums back to back, then n counts).
 * phase: 0 / 3 = all of it (the product); 1 = fold + push only, 2 = wait + reduce only (single-thread emulator tests).
 * hrf_p2p_tick: *gen += 1, once per training step before the first exchange (inside the captured graph).
 * hrf_p2p_alloc / open / close / free: inbox memory and its 64-byte IPC handle (hipExtMallocWithFlags fine-grained,
 * hipIpcGetMemHandle / hipIpcOpenMemHandle).  inbox[p] / flags[p]: rank p's data / flag area AS MAPPED INTO THIS PROCESS.     */
#define HRF_P2P_MAX_RANKS 8
typedef struct hrf_p2p {
  int world, rank;
  double* inbox[HRF_P2P_MAX_RANKS];
  long* flags[HRF_P2P_MAX_RANKS];
  long slot_doubles, nslots;
  const long* gen;
  long* err;
  long timeout_ticks;            /* 100 MHz ticks; <= 0: wait for ever */
} hrf_p2p_t;
int hrf_p2p_exchange(const hrf_p2p_t* ctx, const double* const* stats, const int* C, int n, const double* rows,
                     const long* slot_off, const int* slot_id, double* packed, int phase, void* stream);
int hrf_p2p_tick(long* gen, void* stream);
int hrf_p2p_alloc(long bytes, void** ptr, void* handle64);
int hrf_p2p_open(const void* handle64, void** ptr);
int hrf_p2p_close(void* ptr);
int hrf_p2p_free(void* ptr);

/* ---- GroupNorm (norm_cfg = dict(type='GN', num_groups=G): build_norm_layer at hrnet.py:338-339,438,459,476,
 * resnet.py:34-35,161-164, hrformer.py:269,278,281,518,542,552 -> nn.GroupNorm / F.group_norm).  Per-(sample, group)
 * statistics, no exchange between samples or ranks.  Three launches per direction:
 *   hrf_gn_moments  out[b][0][c] += sum_p v, out[b][1][c] += sum_p v*w (w NULL: v*v) over the rows_per_sample pixels of
 *                   sample b; `out` ([B][2][C] doubles) must be zero.  Forward: v = raw conv output; backward: v = du, w = raw.
 *   hrf_gn_apply    y = gamma*(raw - mean)*rstd + beta with (mean, rstd) of every (sample, group) folded from `mom`,
 *                   also stored in stat[b][g][2] for the backward.
 *   hrf_gn_bwd      draw = rstd*(du*gamma - mean_group(du*gamma) - xhat*mean_group(du*gamma*xhat)); dgamma += , dbeta += . */
int hrf_gn_moments(const float* v, const float* w, int B, long rows_per_sample, int C, double* out, void* stream);
int hrf_gn_apply(const float* raw, const double* mom, const float* gamma, const float* beta, float eps, int B,
                 long rows_per_sample, int C, int G, float* y, float* stat, void* stream);
int hrf_gn_bwd(const float* du, const float* raw, const float* stat, const double* gmom, const float* gamma, int B,
               long rows_per_sample, int C, int G, float* draw, float* dgamma, float* dbeta, void* stream);

/* ---- LayerNorm over channels (F.layer_norm: hrformer.py:343,351; hrfuser_hrformer_based.py:279-291) */
int hrf_ln_stats(const float* x, int rows, int C, float eps, float* rowstat, void* stream);
int hrf_ln_bwd(const float* da, const float* x, const float* rowstat, const float* gamma, int rows, int C,
               float* dx, int accumulate, float* dgamma, float* dbeta, long copy_stride, void* stream);

/* `ln_rowstat` (hrf_conv_fwd, hrf_affine_act_res; nullable): also emit the LayerNorm row statistics
 * (mean, rstd with ln_eps) of the OUTPUT rows - the consumer's LayerNorm needs no hrf_ln_stats launch. */
/* ---- BN-apply + activation + residual materialisation and its adjoint ---------------------
 * act: 0 none, 1 ReLU, 2 GELU.  act_first=1: out = res + rowscale[b]*act(sc1*y1+sh1)
 * (CrossFFN tail hrformer.py:371 / DropPath); act_first=0: out = act(sc1*y1+sh1 + res + sc2*y2+sh2)
 * (Bottleneck tail resnet.py:282-300, transition ReLU hrnet.py:438-440).                        */
int hrf_affine_act_res(const float* y1, const float* sc1, const float* sh1, const float* y2,
                       const float* sc2, const float* sh2, const float* res, const float* rowscale,
                       int rows_per_sample, int act, int act_first, float* out, long rows, int C, float* ln_rowstat, float ln_eps,
                       const hrf_bn_fin_t* fin1, const hrf_bn_fin_t* fin2, void* stream);
/* out = res + res2 + y*mask*mscale*rowscale[b]: nn.Dropout(proj_drop) hrfuser_hrformer_based.py:97
 * and mmcv DropPath hrfuser_hrformer_based.py:301-315 arithmetic (mask / rowscale nullable).    */
int hrf_scale_add(const float* y, const float* mask, float mscale, const float* rowscale,
                  int rows_per_sample, const float* res, const float* res2, float* out, long rows, int C,
                  void* stream);
/* mode 0: g = dout*(out>0); 1: g = dout*rowscale*gelu'(sc*y1+sh); 2: g = dout.  st_k (nullable)
 * accumulate (sum g, sum g*y_k) for the BatchNorms whose output fed the activation.            */
int hrf_act_bwd(const float* dout, const float* out, const float* y1, const float* sc, const float* sh,
                const float* rowscale, int rows_per_sample, int mode, float* g, const float* y2,
                const float* y3, double* st1, double* st2, double* st3, long rows, int C, void* stream);

/* ---- HRModule cross-resolution exchange (hrnet.py:184-207; fuse layers hrformer.py:498-561) ---
 * out = ReLU(sum of up to four terms); term type 0 unused, 1 identity, 2 BN-affine of a same-
 * resolution raw conv output, 3 BN-affine of a bilinearly up-sampled (align_corners=False)
 * low-resolution raw conv output (Hs x Ws), 4 the same through nn.Upsample(mode='nearest') by the integer factor
 * H / Hs (the convolutional HRModule, hrnet.py:135-146; H % Hs == 0 and W % Ws == 0).              */
int hrf_fuse_sum(int type0, const float* p0, const float* sc0, const float* sh0, int Hs0, int Ws0,
                 int type1, const float* p1, const float* sc1, const float* sh1, int Hs1, int Ws1,
                 int type2, const float* p2, const float* sc2, const float* sh2, int Hs2, int Ws2,
                 int type3, const float* p3, const float* sc3, const float* sh3, int Hs3, int Ws3,
                 float* out, int B, int H, int W, int C, const hrf_bn_fin_t* fins, void* stream);
/* `fin1` / `fin2` of hrf_affine_act_res and `fins` of hrf_fuse_sum (an array of FOUR hrf_bn_fin_t, one per term; entries
 * with stats == NULL are unused; C <= HRF_FIN_MAXC / 2) finalise the BatchNorm of the respective operand on load.  */
/* adjoint of the bilinear up-sampling (gather form) + (sum du, sum du*ylow) moments              */
int hrf_bilinear_up_bwd(const float* g, int ldG, int goff, int B, int H, int W, int C, const float* ylow, int Hs, int Ws,
                        float* du, double* stats, void* stream);
/* adjoint of the nearest up-sampling of term type 4 (block sums) + the same moments                       */
int hrf_nearest_up_bwd(const float* g, int ldG, int goff, int B, int H, int W, int C, const float* ylow, int Hs, int Ws,
                       float* du, double* stats, void* stream);

/* ---- HRFPN neck pieces (mmdet/models/necks/hrfpn.py:77-100) --------------------
 * hrf_bilinear_up_into: out[pix][off + c] = F.interpolate(x, size=(H,W), mode='bilinear')[pix][c]
 *   (align_corners=False; a plain copy when Hs == H): the up-sample + channel concat in one pass;
 *   its adjoint is hrf_bilinear_up_bwd reading the concat gradient through (ldG, goff).
 * hrf_avg_pool / hrf_avg_pool_bwd: F.avg_pool2d(kernel_size = stride = k) on NHWC rows.
 * hrf_slice_cols: dst[row][c] (+)= src[row][off + c] - the concat adjoint of the branch that is not up-sampled. */
int hrf_bilinear_up_into(const float* x, int Hs, int Ws, int C, float* out, int ldOut, int off, int B, int H, int W,
                         void* stream);
int hrf_avg_pool(const float* x, int B, int H, int W, int C, int k, float* out, void* stream);
/* The 3x3 / pad-1 patches of a FEW-channel input as rows (the stem's first convolution, 3 -> 64 stride 2 on the NCHW network input:
 * hrnet.py:341-347, hrfuser_hrformer_based.py:380-386):  cols[(b, yo, xo)][ci * 9 + tap] = x(b, ci, yo * stride - 1 + tap / 3,
 * xo * stride - 1 + tap % 3), zero outside the image; x through element strides (NCHW or channels-last), rows `ld` >= 9 Cin floats apart.
 * The column order is the OIHW weight's memory order: F.conv2d(x, w, stride, 1) = cols . w.view(Cout, 9 Cin)^T (hrf_conv_fwd with KH = 1
 * on the rows) and grad_weight = dY^T . cols (hrf_conv_bwd_weight with KH = 1) - both on the channel-contiguous row kernels
 * instead of 27 strided gathers per output pixel, and the patches are formed once per step for both. */
int hrf_im2col3x3(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin, int stride,
                  float* cols, int ld, void* stream);
int hrf_slice_cols(const float* src, int ld, int off, long rows, int C, float* dst, int accumulate, void* stream);
int hrf_avg_pool_bwd(const float* g, int B, int H, int W, int C, int k, float* dx, int accumulate, void* stream);

/* Wide 3x3 / stride-1 / pad-1 convolution on tap-major packed weights (the HRFPN 256->256 output convolutions,
 * hrfpn.py:60-70,92-100, and their backward-data): K % 32 == 0 input channels, N % 64 == 0 output channels.
 *   hrf_conv3_pack   wp[tap][n][k] = w[n][k][tap]        (dir 0: forward operand, N = Cout, K = Cin)
 *                    wp[tap][n][k] = w[k][n][8 - tap]    (dir 1: backward-data operand, N = Cin, K = Cout)
 *                    w is the OIHW tensor [Cout][Cin][3][3]; wp holds 9*Cout*Cin floats (caller-owned scratch,
 *                    refreshed whenever w changes - once per step).
 *   hrf_conv3_packed y[pix][n] (+)= bias[n] + sum_{tap,k} x[pix + tap][k] * wp[tap][n][k]   (zero padding);
 *                    forward: x = input rows, wp = pack(dir 0); backward-data (aten convolution_backward's
 *                    grad_input): x = dY rows, wp = pack(dir 1), bias = null, accumulate = 1 to add into a gradient. */
int hrf_conv3_pack(const float* w, int Cout, int Cin, int dir, float* wp, void* stream);
int hrf_conv3_packed(const float* x, int ldX, const float* wp, const float* bias, float* y, int ldY, int accumulate,
                     int B, int H, int W, int K, int N, void* stream);
/*   hrf_conv3_wgrad_wide  dW[co][ci][3][3] += sum_pix dY[pix][co] * x[pix + tap][ci],  dbias[co] += sum_pix dY[pix][co]
 *                    (aten convolution_backward's grad_weight / grad_bias of the same convolution; OIHW dW;
 *                    Cout % 128 == 0, Cin % 64 == 0; dbias nullable).  The pixel-split partial sums pass through
 *                    `scratch`: hrf_conv3_wgrad_wide_scratch(...) floats, caller-owned, contents irrelevant. */
long hrf_conv3_wgrad_wide_scratch(int B, int H, int W, int Cin, int Cout);
int hrf_conv3_wgrad_wide(const float* dy, int ldD, const float* x, int ldX, int B, int H, int W, int Cin, int Cout,
                         float* dw, float* dbias, float* scratch, void* stream);

/* Wide 1x1 convolution / Linear as an LDS-tiled MFMA row GEMM (the HRFPN reduction convolution, hrfpn.py:53-58,85):
 *   hrf_rowgemm_pack  wp[n][k] (n < Np, k < Kp) = w[n][k] (dir 0: forward operand, w = [Cout][Cin]) or w[k][n]
 *                     (dir 1: backward-data operand), zero outside the tensor: Np / Kp are the padded sizes.
 *   hrf_rowgemm       y[m][n] (+)= bias[n] + sum_k x[m][k] * wp[n][k];  K % 16 == 0, N % 16 == 0 (pad columns of x must
 *                     hold zeros or finite values matched by zero weights; bias, if given, has N entries). */
int hrf_rowgemm_pack(const float* w, int Cout, int Cin, int dir, int Np, int Kp, float* wp, void* stream);
int hrf_rowgemm(const float* x, int ldX, const float* wp, const float* bias, float* y, int ldY, int accumulate,
                long M, int K, int N, void* stream);

/* Grouped weight gradients.  Weight gradients are leaves of the backward graph; between hrf_wgrad_group_begin() and
 * hrf_wgrad_group_end(stream) every hrf_conv_bwd_weight call that maps to the pixel-major kernel is queued instead of
 * launched (its `stream` argument is ignored), and _end issues the queue on `stream`, up to 16 problems of one kernel
 * variant per launch (results identical to separate launches).  Calls that use other kernels launch immediately. */
int hrf_wgrad_group_begin(void);
int hrf_wgrad_group_end(void* stream);

/* Multi-problem launches for the EQUAL-SHAPE layers of sibling sensor streams.  The reference runs the camera stream's
 * finest branch and the M modality streams through layers of identical shape with different weights
 * (hrfuser_hrformer_based.py:536-544 stems, :564-565 stage 2 || LidarStageB, :585-586 stage 3 || LidarStageC), each as its own
 * sequence of ATen launches.  Here the hot kernels take their arguments as an array of up to 4 problems (blockIdx.z selects
 * the problem; csrc/hrf_group.h): between hrf_group_begin() and hrf_group_end(stream) the launches of the calls to
 *   hrf_conv_fwd, hrf_conv_bwd_data, hrf_dwconv_fwd, hrf_dwconv_bwd_data(_weight), hrf_attn_block_fwd / _bwd,
 *   hrf_window_attn_fwd / _bwd, hrf_affine_act_res, hrf_act_bwd, hrf_scale_add, hrf_ln_stats, hrf_ln_bwd, hrf_fuse_sum,
 *   hrf_bilinear_up_bwd
 * are queued (their `stream` argument is ignored) and hrf_group_end issues them on `stream`: launches of the same kernel
 * instantiation and launch geometry that sit at the same position of DIFFERENT calls become ONE launch; everything else is
 * issued as it would have been, the launches of one call in their order.  The caller brackets MUTUALLY INDEPENDENT calls
 * only.  Results are identical to separate launches.  Per-thread state; begin / end must pair on one thread.
 * hrf_group_count(what): process-wide totals since load - 0 launches issued by hrf_group_end, 1 calls' launches they
 * carried, 2 hrf_group_end calls (measurement aid); 3: the number of problems per launch the library was COMPILED for
 * (HRF_GROUP_MAX).  The shipped build uses 1 - begin / end then only queue and re-issue - because on MI355X the merged
 * launches measured slower than one HIP stream per sensor (csrc/hrf_group.h); -DHRF_GROUP_MAX=4 builds the
 * multi-problem kernels (the CPU-emulator test build does). */
int hrf_group_begin(void);
int hrf_group_end(void* stream);
long hrf_group_count(int what);

/* Device-side input pipeline: the image side of Normalize -> RandomFlip -> Pad(size_divisor) -> RandomDrop ->
 * DefaultFormatBundle (mmdet/datasets/pipelines/transforms.py:706-753,440-466,649-664,487-514; formating.py:212-227) for ONE
 * sensor of a batch in one pass: in = [B][H0][W0][C] HWC images (float32, or uint8 when is_u8), out = [B][Hp][Wp][C]
 * float32 (channels-last storage of the logical (B,C,Hp,Wp) tensor the backbone takes);
 *   out = drop[b] ? 0 : inside ? (in(b, y, flip[b] ? W0-1-x : x, to_rgb ? C-1-c : c) - mean[c]) * stdinv[c] : 0.
 * mean / stdinv: C floats on the device (stdinv = float32(1 / float64(std)), as mmcv.imnormalize computes it);
 * flip / drop: B bytes each (nullable): the random decisions stay with the host's RNG. */
int hrf_pack_input(const void* in, int is_u8, int B, int H0, int W0, int C, const float* mean, const float* stdinv,
                   int to_rgb, const unsigned char* flip, const unsigned char* drop, float* out, int Hp, int Wp,
                   void* stream);

/* ---- fused flat-buffer AdamW (configs/hrfuser: AdamW lr 3e-4, wd 0.01, decay_mult 0 masks) ---
 * state = float[4] on device: {1-b1^t, 1-b2^t, t, lr}; hrf_adamw_tick advances t on device so a
 * captured hipGraph replays correct bias corrections; lr < 0 in the call = take the learning rate from state[3]
 * (the host updates that one float between replays: warm-up / step schedules work under hipGraph replay).
 * wd_mask[i]: weight-decay multiplier of element i (0 for `norm` / `relative_position_bias_table` keys); < 0 =
 * the element belongs to a parameter that never receives a gradient: skipped entirely, as torch.optim skips
 * parameters whose .grad is None (mmdet runs DDP with find_unused_parameters=True for transition1.0.1.*).        */
int hrf_adamw_tick(float* state, float beta1, float beta2, void* stream);
/* dst[map[i]] += sum_k scratch[k*copy_stride + i], k < HRF_STAT_COPIES (see top of file)          */
int hrf_fold_copies(const float* scratch, long copy_stride, const int* map, float* dst, long n,
                    void* stream);
int hrf_adamw(float* p, const float* g, float* m, float* v, const float* wd_mask, long n, float lr,
              float beta1, float beta2, float eps, float weight_decay, const float* state,
              float grad_scale, void* stream);

/* The digest (sha256, 64 hex characters) of the sources, headers, flags and target this library was built from
 * (hrfuser_amd/build_ext.py compares it with the tree on every build(): a source change is never paired with an old binary). */
const char* hrf_build_digest(void);

/* Measurement and tuning entry points (GPU time stamps inside a captured graph, the critical-lane probe, kernel tuning knobs,
 * the in-situ timing report of the grouped weight gradients) are NOT part of this interface: include/hrfuser_hip_debug.h. */

/* hipMemsetAsync on `stream` (the per-step zeroing of the replicated accumulators). */
int hrf_memset(void* ptr, int value, long bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HRFUSER_HIP_H_ */
