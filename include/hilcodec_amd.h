/* hilcodec_amd.h — C ABI of the MI355X-native HILCodec encode -> RVQ -> decode hot path.
 *
 * The reference (aask1357/hilcodec) is pure Python/PyTorch and has NO FFI: its hot path sits behind
 * nn.Module classes and bottoms out in ATen calls.  This header therefore declares the entry points
 * a binding of that path would need: one per arithmetic step of the folded graph (SURVEY.md
 * Appendix A).  Each entry cites the reference code it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the ctypes stub that binds them from the reference's modules.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator in this repo);
 *    the library allocates nothing, keeps no mutable global state (only a per-device cache of immutable
 *    occupancy facts), and is re-entrant;
 *  - tensors are dense fp32, channel-major `[B][C][T]` (time contiguous) unless stated;
 *  - `stream` is a `hipStream_t` passed as `void*`; all work is enqueued on it, nothing syncs;
 *  - return value: HILC_OK (0) or a negative HILC_ERR_* code; nothing is launched on error;
 *  - "hist" arguments are the streaming caches of `models/hilcodec/causal_layers.py:147-188`
 *    (the samples *before* t = 0 of the layer's input); NULL means zero history, which is the
 *    causal zero padding of the offline model (`models/hilcodec/modules/conv.py:222-236`).
 *  - prologue  pro(v) = in_elu ? ELU(v * in_scale) : v * in_scale   (Scale + ELU modules,
 *    `models/hilcodec/modules/seanet.py:165-178`, torch.nn.ELU(alpha=1));
 *    it is applied to `x` only — histories hold already-activated samples, as the reference's
 *    caches do.
 */
#ifndef HILCODEC_AMD_H
#define HILCODEC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HILC_OK 0
#define HILC_ERR_SHAPE (-1)       /* a dimension is <= 0 or inconsistent            */
#define HILC_ERR_NULL (-2)        /* a required pointer is NULL                     */
#define HILC_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after launch   */
#define HILC_ERR_UNSUPPORTED (-4) /* configuration outside what the kernels cover   */
#define HILC_ERR_RANGE (-5)       /* n outside 1..Nq (reference: AssertionError)    */

#define HILC_ABI_VERSION 15   /* 2: packed residual-block weights; 3: hilc_spec_block; 4: hilc_spec_block_conv_pre; 5: waveform history in both; 6-7: *_x3 (experimental; REMOVED in 14); 8: hilc_dws_conv_wave_row; 9: hilc_resblock_stream_supported (wide blocks in hilc_resblock_stream); 10: hilc_resblock_chain; 11: hilc_encoder_stage; 12: batched cache updates (REMOVED in 14); 13: hilc_decoder_stage; 14: the entry points that only served rejected experiments are gone (split-bf16 decoder GEMMs, batched cache updates); hilc_decoder_stage_post, hilc_encoder_stage0; 15: hilc_rvq_encode[_mixed] take `flags` (HILC_RVQ_VALU_ONLY replaces the HILC_RVQ_VALU environment variable) */

int hilc_abi_version(void);
const char* hilc_error_string(int code);
/* text of the hipError_t behind the calling thread's most recent HILC_ERR_LAUNCH */
const char* hilc_last_hip_error(void);

/* ---- pointwise (1x1) convolution: fp32-MFMA GEMM with fused prologue / epilogue ----------------
 * y[b,m,t] = (sum_k wt[k][m] * pro(x[b,k,t]) + bias[m]) * out_scale + res[b,m,t]
 * Replaces: nn.Conv1d(k=1) inside SConv1d/NormConv1d (`models/hilcodec/modules/conv.py:115-134,
 * 202-236`) plus the ELU / Scale in front of it (`seanet.py:26-52,322-340`) and, with `res`, the
 * `x.add_(y.mul_(scale))` of SpecBlock (`seanet.py:241-246`).
 * wt is the folded weight TRANSPOSED to `[K][M]` (k-major).  bias, res may be NULL.  res may alias y. */
int hilc_pw_conv(const float* x, const float* wt, const float* bias, const float* res, float* y,
                 int B, int K, int M, int T, float in_scale, int in_elu, float out_scale, void* stream);

/* ---- fused depthwise-separable block: pointwise conv -> depthwise causal conv through LDS ----------
 * h[b,m,t] = sum_k wt[k][m] * pro(x[b,k,t])                       (no bias: seanet.py:33-37)
 * y[b,m,o] = post((sum_j dw_w[m][j] * h[b,m,o*stride - pad + j] + dw_b[m]) * out_scale + res[b,m,o])
 * with h(t<0) = h(t>=T) = 0, pad = (ksize-1)-(stride-1), T_out = ceil(T/stride).  Supported: ksize 5 /
 * stride 1 (residual-block halves, `seanet.py:26-52,129-148`; decoder/encoder pre/post pairs) and
 * ksize = 2*stride (encoder down-sampling, `seanet.py:322-340`; res/out_elu/out_scale unused there).
 * The [M x T] intermediate h lives only in LDS: one HBM read of x and one write of y per block half. */
int hilc_dws_conv(const float* x, const float* wt, const float* dw_w, const float* dw_b, const float* res,
                  float* y, int B, int K, int M, int T, int ksize, int stride, float in_scale, int in_elu,
                  float out_scale, int out_elu, void* stream);

/* 1 if hilc_dws_conv (ksize 5, stride 1) runs this shape in the wave-row tile form (depthwise taps on the accumulator
 * registers), 0 for the column-block form with the LDS epilogue.  Both give the same bits; a pure function of the shape and the
 * device's CU count (diagnostics and tests). */
int hilc_dws_conv_wave_row(int B, int M, int T, int has_res);

/* ---- fused up-sampling stage: [Scale, ELU,] depthwise transposed conv (k = 2*stride) -> pointwise conv ---
 * u[b,k,q*stride+p] = tr_w[k][p] * pro(x[b,k,q]) + tr_w[k][p+stride] * pro(x[b,k,q-1])      (x[-1] = 0)
 * y[b,m,t] = sum_k wt[k][m] * u[b,k,t] + bias[m],   t < Tin*stride
 * The up-sampled tensor u only exists as the GEMM's B operand (computed in the loader).
 * Replaces: `seanet.py:424-441` (Scale, ELU, SConvTranspose1d, SConv1d k=1 with bias). Requires
 * (Tin*stride) % 4 == 0 and M % 4 == 0 (else HILC_ERR_UNSUPPORTED: use hilc_dw_convtr + hilc_pw_conv). */
int hilc_up_conv(const float* x, const float* tr_w, const float* wt, const float* bias, float* y,
                 int B, int K, int M, int Tin, int stride, float in_scale, int in_elu, void* stream);

/* Streaming hop of the same stage (`streaming.py:520-648` Decoder.forward, `causal_layers.py:168-188`): hist `[B][K]` =
 * pro(x[b,k,-1]) of the previous hop (the transposed conv's cache holds ACTIVATED samples, like hilc_dw_convtr's),
 * hist_out receives pro(x[b,k,Tin-1]).  Both optional; must not alias. */
int hilc_up_conv_stream(const float* x, const float* hist, float* hist_out, const float* tr_w, const float* wt,
                        const float* bias, float* y, int B, int K, int M, int Tin, int stride, float in_scale,
                        int in_elu, void* stream);

/* Strides other than 2 / 4 / 8 (the codec's 5): with `tr_w_expanded` = the table written by hilc_up_conv_expand_taps
 * (`[K][stride][8]` floats, 16-B aligned: for each phase p0 = t mod stride of a 4-column group its eight taps as two
 * 16-B words) the loader issues two vector tap loads per row instead of eight scalar ones.  hist / hist_out /
 * tr_w_expanded may be NULL (then this is hilc_up_conv_stream). */
int hilc_up_conv_expand_taps(const float* tr_w, float* expanded, int K, int stride, void* stream);
int hilc_up_conv_expanded(const float* x, const float* hist, float* hist_out, const float* tr_w,
                          const float* tr_w_expanded, const float* wt, const float* bias, float* y, int B, int K,
                          int M, int Tin, int stride, float in_scale, int in_elu, void* stream);

/* Streaming hop of hilc_dws_conv for the wide layers (DWSBlock.forward `streaming.py:160-192`, CausalConv1d
 * `causal_layers.py:147-165`): T <= 128 samples per stream and call, T % stride == 0, any ksize >= stride; longer
 * hops (T % 4 == 0; per-clip tiles with a recomputed halo) for ksize == 2 * stride.
 * hist `[B][M][ksize-stride]` = the last pointwise outputs of the previous hop (NULL = zeros), hist_out receives
 * the new cache (must not alias hist).  A tile holds whole clips, so nothing is recomputed. */
int hilc_dws_conv_stream(const float* x, const float* wt, const float* dw_w, const float* dw_b, const float* hist,
                         float* hist_out, const float* res, float* y, int B, int K, int M, int T, int ksize,
                         int stride, float in_scale, int in_elu, float out_scale, int out_elu, void* stream);

/* ---- fully fused residual block (C in {64, 96, 128, 192} and — narrow-tile shapes — {256, 384, 512, 768}; T % 4 == 0) --------
 * y = x + out_scale * (dw2(pw2(ELU(dw1(pw1(ELU(pre_scale * x))) + dw1_b))) + dw2_b)
 * One HBM read of x and one write of y per block; both pointwise outputs and the mid activation
 * stay in LDS.  w1t / w2t are the two `[C][C]` pointwise matrices in the PACKED layout written by
 * hilc_resblock_pack_weights (below), dw*_w `[C][5]`.  y must not alias x.
 * Replaces: SEANetResnetBlock.forward (`seanet.py:129-148`) with skip='identity', kernel 5.
 * hilc_resblock_supported(C, T) tells the caller whether this specialisation exists. */
int hilc_resblock(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b, const float* w2t,
                  const float* dw2_w, const float* dw2_b, float* y, int B, int C, int T, float pre_scale,
                  float out_scale, void* stream);
int hilc_resblock_supported(int C, int T);

/* One-off (per checkpoint) re-layout of a k-major `[C][C]` pointwise matrix (wt[k][m], the layout hilc_pw_conv
 * takes) into "MFMA lane order": the operands one lane feeds to the matrix pipe for a 16-deep K slice become
 * consecutive 16-B words, so the fused block streams its weights with a quarter of the load instructions.
 * packed: `C*C` floats, must not alias wt.  C in {64, 96, 128, 192}, and {256, 384, 512, 768} for the streaming form. */
int hilc_resblock_pack_weights(const float* wt, float* packed, int C, void* stream);

/* Streaming form of the same block (`streaming.py:195-276` ResBlock with two DWSBlock caches,
 * `causal_layers.py:147-167`): hist1 / hist2 `[B][C][4]` = the last 4 samples of the two depthwise convs'
 * inputs (the pointwise outputs) from the previous hop, hist*_out receive the new caches.  All four are
 * optional (NULL = zero history / no cache written); outputs must not alias inputs.
 * Besides the offline widths the streaming form takes the WIDE blocks of a hop, where a stream contributes 8 or 40
 * frames: C = 256 / 384 (64-column tiles over the flat stream-major column space) and C = 512 / 768 (32-column tiles of
 * whole streams, T in {4, 8, 16, 32}: no halo) — both GEMMs and the activation between them in one launch instead of two
 * hilc_dws_conv_stream launches; same products in the same order, bit-identical.  hilc_resblock_stream_supported(C, T)
 * tells the caller whether the specialisation exists (else HILC_ERR_UNSUPPORTED: fall back to two launches). */
int hilc_resblock_stream_supported(int C, int T);
int hilc_resblock_stream(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                         const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1,
                         const float* hist2, float* hist1_out, float* hist2_out, float* y, int B, int C, int T,
                         float pre_scale, float out_scale, void* stream);

/* Same kernels with a caller-owned dynamic tile scheduler: `sched` = two ints in device memory that are ZERO at
 * launch; the kernel leaves them zero again (the last workgroup re-arms them), so one buffer per stream can be
 * reused by every call on that stream.  Workgroups take tiles by ticket instead of from static lists — the
 * workgroups that share a CU are not served equally, and static lists leave a third of the kernel at half
 * occupancy.  streaming = 0: hilc_resblock semantics (hist* ignored and may be NULL); 1: hilc_resblock_stream.
 * sched == NULL: static lists. */
int hilc_resblock_balanced(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                           const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1,
                           const float* hist2, float* hist1_out, float* hist2_out, float* y, int* sched,
                           int streaming, int B, int C, int T, float pre_scale, float out_scale, void* stream);

/* ---- the residual blocks of ONE STAGE in one launch (ABI 10) --------------------------------------------------------
 * The reference runs a stage's blocks one after the other: encoder `seanet.py:316-330` (`self.blocks[i]`, 2 per stage),
 * decoder `seanet.py:437-452` (3 after each up-sampling layer); streaming `streaming.py:497-503` / `:633-639` with two
 * caches per block.  hilc_resblock_chain(x, y, blocks, nblk, ...) == nblk calls of hilc_resblock_stream with the output of
 * one as the input of the next, bit for bit — but per tile the blocks run back to back and a block's output stays in
 * registers as the next block's input and shortcut: the activations between the blocks never reach HBM and the stage is one
 * launch instead of nblk (a streaming hop of 1024 streams is 5-10 tiles per workgroup and launch).
 * blocks[i]: that block's parameters; w1t / w2t PACKED by hilc_resblock_pack_weights_rc(.., C, hilc_resblock_chain_row_classes(C))
 * (the chain's 8-wave shapes split the rows in two classes also below C = 192, so the layout differs from hilc_resblock's);
 * hist* as in hilc_resblock_stream (each optional; ignored with streaming = 0: the offline causal model, hilc_resblock).
 * hilc_resblock_chain_supported tells whether the specialisation exists: C in {64, 96, 128, 192} any T % 4 == 0;
 * streaming: C in {512, 768} with whole streams tiling 32 columns, and C = 256 (32-column tiles, four waves, runs of whole streams:
 * measured slower inside a 1024-stream hop than one launch per block, so the engine leaves it off); offline: C in {256, 384, 512} (C = 768: one block per launch,
 * the carry slots of a second do not fit LDS); nblk = 2, or 3 at the decoder's widths (96, 192, streaming 768, offline 384).
 * Else HILC_ERR_UNSUPPORTED: launch the blocks one by one. */
typedef struct hilc_resblock_params {
  const float* w1t; const float* dw1_w; const float* dw1_b;
  const float* w2t; const float* dw2_w; const float* dw2_b;
  const float* hist1; const float* hist2; float* hist1_out; float* hist2_out;
  float pre_scale, out_scale;
} hilc_resblock_params;
int hilc_resblock_chain_supported(int C, int T, int nblk, int streaming);
int hilc_resblock_chain_row_classes(int C);           /* streaming form */
int hilc_resblock_chain_row_classes_offline(int C);   /* offline form (streaming = 0: hilc_resblock semantics, C in {64 ... 768}) */
int hilc_resblock_pack_weights_rc(const float* wt, float* packed, int C, int row_classes, void* stream);
int hilc_resblock_chain(const float* x, float* y, const hilc_resblock_params* blocks, int nblk, int streaming,
                        int B, int C, int T, void* stream);

/* ---- a DECODER STAGE of a streaming hop in one launch (ABI 13): its up-sampling layer and its residual blocks -----------------
 * `seanet.py:431-452` ([Scale, ELU, depthwise SConvTranspose1d k = 2r stride r, 1x1 conv 2C -> C + bias], then the stage's
 * SEANetResnetBlocks); streaming `streaming.py:629-639`, the transposed conv's cache `[B][2C][1]` = its last ACTIVATED input frame
 * (`causal_layers.py:168-188`).  Equals hilc_up_conv_stream followed by hilc_resblock_chain bit for bit; the `[B][C][T]` tensor
 * between them never exists.  x `[B][2C][T/r]`, tr_w `[2C][2r]`, w_lo / w_hi: rows [0, C) / [C, 2C) of the k-major `[2C][C]`
 * pointwise weight, each packed with hilc_resblock_pack_weights_rc(.., C, hilc_resblock_chain_row_classes(C)).
 * Stages: C = 768 with r = 8 (streaming: whole streams per 32-column tile, T in {8, 16, 32}, nblk 1..3; offline: nblk = 1 — the
 * up-sampling layer and the stage's FIRST block, the carry slots of a second do not fit LDS), C = 384 with r = 5 (tr_w = the EXPANDED
 * tap table `[2C][r][8]` of hilc_up_conv_expand_taps; nblk 1..3; a streaming hop (ABI 15) on 32-column carry tiles, runs of whole
 * streams, twelve waves — rounds 4-5: nblk = 1 on 64-column halo tiles),
 * C = 192 with r = 4 and C = 96 with r = 2 (streaming hops and, with streaming = 0, the offline model: hist* ignored); nblk 1..3. */
typedef struct hilc_up_params {
  const float* x; const float* tr_w; const float* w_lo; const float* w_hi; const float* bias;
  const float* hist; float* hist_out;
  float in_scale; int stride;
} hilc_up_params;
int hilc_decoder_stage_supported(int C, int T, int nblk, int stride, int streaming);
int hilc_decoder_stage(const hilc_up_params* up, const hilc_resblock_params* blocks, int nblk, float* y, int streaming,
                       int B, int C, int T, void* stream);

/* ---- the decoder's LAST stage AND its closing layer in one launch (ABI 14) --------------------------------------------------
 * `seanet.py:453-476`: `[Scale, ELU, SConv1d(C, 1, k = 5, bias)]`, then the model's final scale / tanh — hilc_conv_post — as the closing
 * phase of the hilc_decoder_stage launch of the offline model's last stage (C = 96, r = 2, three blocks): the last block leaves
 * ELU(in_scale * y) in the LDS tile, the launch stores `wav` `[B][1][T]`; the stage's `[B][C][T]` output never reaches HBM.  Equals
 * hilc_decoder_stage followed by hilc_conv_post bit for bit (same row classes c mod 8, same order of the partial sums).  streaming = 0:
 * the offline model (`up->hist`, the blocks' caches and `post->hist` are ignored).  streaming = 1 (ABI 15): a hop (`streaming.py:639-648`) on the
 * carry form's runs of whole streams, with every cache of hilc_decoder_stage plus the closing conv's — equal, bit for bit, to
 * hilc_decoder_stage(streaming) followed by hilc_conv_post with the same caches.  hilc_decoder_stage_post_supported names the shapes;
 * everything else: HILC_ERR_UNSUPPORTED. */
typedef struct hilc_post_params {
  const float* w;      /* [C][ksize] */
  const float* bias;   /* [1] or NULL */
  float* wav;          /* [B][1][T] */
  const float* hist;   /* streaming (ABI 15): [B][C][ksize-1] the conv's cache = hilc_conv_post's (activated samples; NULL = zeros) */
  float* hist_out;     /* streaming: receives the next hop's cache (may be NULL) */
  float in_scale, out_scale;
  int do_tanh, ksize;
} hilc_post_params;
int hilc_decoder_stage_post_supported(int C, int T, int nblk, int stride, int ksize);
int hilc_decoder_stage_post(const hilc_up_params* up, const hilc_resblock_params* blocks, int nblk, const hilc_post_params* post,
                            int streaming, int B, int C, int T, void* stream);

/* ---- an ENCODER STAGE in one launch (ABI 11): its residual blocks and its down-sampling layer ------------------------------
 * `seanet.py:316-339` (`self.blocks[i]`, then `self.downsample[i]` = [Scale, ELU, 1x1 conv C -> 2C without bias, depthwise conv
 * k = 2r stride r with bias]); streaming `streaming.py:497-511` with the layer's cache `[B][2C][r]`.  Equals hilc_resblock_chain
 * followed by hilc_dws_conv / hilc_dws_conv_stream (stride r, in_scale, in_elu = 1, `res`) bit for bit; the stage's output
 * `[B][C][T]` never reaches HBM.  w_lo / w_hi: columns [0, C) / [C, 2C) of the k-major `[C][2C]` pointwise weight, each packed
 * like a block's matrix (hilc_resblock_pack_weights_rc with the row classes of the chain form in use).  res (optional): added to
 * the output, e.g. the next stage's SpecBlock branch.  Stages: C = 64 with r = 2, C = 128 with r = 4, and the wide stages C = 256 with
 * r = 5 and C = 512 with r = 8 — offline in the carry form of the narrow-tile shapes; a streaming hop (ABI 15): C = 256 on 32-column carry
 * tiles (runs of whole streams; T % 40 == 0, i.e. whole frames of the hop), C = 512 on whole-stream tiles (T in {8, 16, 32}); nblk 1..2;
 * T % 4 == 0, T % r == 0.  hilc_encoder_stage_supported names the shapes; everything else HILC_ERR_UNSUPPORTED (callers launch the
 * blocks and hilc_dws_conv[_stream] instead). */
typedef struct hilc_down_params {
  const float* w_lo; const float* w_hi; const float* dw_w; const float* dw_b;
  const float* hist; float* hist_out; const float* res; float* y;
  float in_scale; int stride;
} hilc_down_params;
int hilc_encoder_stage_supported(int C, int T, int nblk, int stride, int streaming);
int hilc_encoder_stage(const float* x, const hilc_resblock_params* blocks, int nblk, const hilc_down_params* down, int streaming,
                       int B, int C, int T, void* stream);

/* ---- the encoder's FIRST stage with its input computed in the launch (ABI 14; a streaming hop: ABI 15) ------------------------
 * `seanet.py:280-286` (first conv k = 5, 1 -> 64), `:220-246` (stage 0's SpecBlock: STFT n_fft 64 hop 1 -> log-magnitude -> 1x1 conv),
 * `:316-339` (the stage's residual blocks and down-sampling layer): hilc_spec_block_conv_pre's arithmetic as the opening phase of the
 * hilc_encoder_stage launch for C = 64, r = 2 — equal to the two launches bit for bit; the `[B][64][T]` tensor between them never
 * reaches HBM.  `spec`: the arguments of hilc_spec_block_conv_pre (packed tables from hilc_spec_block_pack).  streaming = 0: the offline
 * model (`spec->hist`, the blocks' caches and `down->hist` are ignored).  streaming = 1 (ABI 15, `streaming.py:490-511`): a hop on runs of whole
 * streams, T >= 128 (a 128-column tile then holds at most one stream start: its waveform segment is staged as two pieces, each with the 63
 * samples in front of it — a stream's history at t = 0), with every cache of hilc_encoder_stage(streaming). */
typedef struct hilc_spec0_params {
  const float* wav;         /* [B][T] */
  const float* dft_packed; const float* nyq_sin; const float* pw_packed; const float* bias;
  const float* pre_w;       /* [64][5] */
  const float* pre_b;       /* [64] or NULL */
  const float* hist;        /* streaming (ABI 15): [B][hist_len] waveform history, as hilc_spec_block_conv_pre's (NULL = zeros) */
  int hist_len;
  float pre_in_scale, mean, std, out_scale;
  int normalize, n_fft, hop, pre_ksize;
} hilc_spec0_params;
int hilc_encoder_stage0_supported(int T, int nblk, int stride, int n_fft, int hop, int pre_ksize, int streaming);
int hilc_encoder_stage0(const hilc_spec0_params* spec, const hilc_resblock_params* blocks, int nblk, const hilc_down_params* down,
                        int streaming, int B, int T, void* stream);

/* ---- depthwise causal convolution, kernel `ksize`, stride `stride` ----------------------------
 * pad = (ksize-1) - (stride-1);  T_out = ceil(T / stride)
 * y[b,c,o] = post((sum_j w[c][j] * xe[b,c,o*stride - pad + j] + bias[c]) * out_scale + res[b,c,o])
 * xe(t) = t < 0 ? hist[b,c,pad+t] (0 if hist NULL) : t < T ? pro(x[b,c,t]) : 0 ; post = ELU if out_elu.
 * hist_out (optional) receives the last `pad` samples of [hist | pro(x)]  -> next call's hist.
 * Replaces: depthwise SConv1d (`conv.py:202-236`, groups=C: `seanet.py:42-51,330-339,352-355,
 * 417-419`), CausalConv1d (`causal_layers.py:147-165`) and the residual tail of
 * SEANetResnetBlock.forward (`seanet.py:144-148`). res may alias y. */
int hilc_dw_conv(const float* x, const float* hist, const float* w, const float* bias, const float* res,
                 float* y, float* hist_out, int B, int C, int T, int ksize, int stride,
                 float in_scale, int in_elu, float out_scale, int out_elu, void* stream);

/* ---- depthwise causal transposed convolution, kernel 2*stride, stride `stride` -----------------
 * y[b,c,q*stride+p] = w[c][p] * xe[q] + w[c][p+stride] * xe[q-1],  T_out = T*stride,
 * xe(-1) = hist[b,c,0] (0 if NULL).  hist_out (optional) receives pro(x[b,c,T-1]).
 * Replaces: SConvTranspose1d (`conv.py:239-282`, right-trim k-s) and CausalConvTranspose1d
 * (`causal_layers.py:168-188`) for groups=C, k=2s, plus the Scale+ELU in front (`seanet.py:424-441`). */
int hilc_dw_convtr(const float* x, const float* hist, const float* w, float* y, float* hist_out,
                   int B, int C, int T, int stride, float in_scale, int in_elu, void* stream);

/* ---- first encoder conv: Conv1d(1 -> C, ksize, causal, bias) on the waveform --------------------
 * y[b,c,t] = sum_j w[c][j] * (in_scale * we[b, t-(ksize-1)+j]) + bias[c];  we(t<0) = hist[b, hist_len+t] or 0.
 * Replaces: `seanet.py:280-286` (Scale(1/wav_std) + SConv1d) / `streaming.py:490`. */
int hilc_conv_pre(const float* wav, const float* hist, int hist_len, const float* w, const float* bias,
                  float* y, int B, int C, int T, int ksize, float in_scale, void* stream);

/* ---- last decoder conv: [Scale, ELU,] Conv1d(C -> 1, ksize, causal, bias), * out_scale, tanh -----
 * y[b,0,t] = act((sum_c sum_j w[c][j] * xe[b,c,t-(ksize-1)+j] + bias[0]) * out_scale), act = tanh if do_tanh.
 * Replaces: `seanet.py:457-473` / `streaming.py:643-647`. hist/hist_out as in hilc_dw_conv.
 * Order of the channel sum (ksize 5, T % 4 == 0, 16-B aligned x): eight classes c mod 8, each an fmaf chain over (c, tap) ascending, the
 * classes added in ascending order — the order of hilc_decoder_stage_post's closing phase. */
int hilc_conv_post(const float* x, const float* hist, const float* w, const float* bias, float* y,
                   float* hist_out, int B, int C, int T, int ksize, float in_scale, int in_elu,
                   float out_scale, int do_tanh, void* stream);

/* ---- causal STFT magnitude -> log -> normalise (the front half of a SpecBlock) ------------------
 * frame f covers we[b, f*hop-(n_fft-1) .. f*hop];  T_f = (T-1)/hop + 1;
 * mag = sqrt(max(re^2+im^2,1e-12));  spec[b,k,f] = normalize==2 ? mag : log(max(mag,1e-5)), then
 * (. - mean)/std if normalize==1 (0: plain log-magnitude, streaming model with merged normalisation)
 * basis_t: `[n_fft][m_pad]` fp32, row n holds (cos_0, sin_0, cos_1, sin_1, ...)*hann for sample n,
 *          m_pad = round_up(n_fft+2, 32), zero padded (host-built from the reference's basis).
 * Replaces: CausalSTFT (`conv.py:285-358`, `causal_layers.py:72-144`) + `seanet.py:224-236`. */
int hilc_stft_logmag(const float* wav, const float* hist, int hist_len, const float* basis_t, float* spec,
                     int B, int T, int n_fft, int hop, float mean, float std, int normalize, void* stream);

/* ---- one-launch SpecBlock (long encoder stages: n_fft 64 / 128 / 256 with the codec's hops 1 / 2 / 8, C == n_fft) ---
 * y[b,m,f] = x[b,m,f] + out_scale * (sum_k pw[k][m] * spec[b,k,f] + bias[m]),  spec as in hilc_stft_logmag: `hist`
 * `[B][hist_len]` (hist_len >= n_fft-1) = the waveform before t = 0 for a streaming hop (`streaming.py:482-490`), NULL = zeros
 * i.e. SpecBlock.forward (`models/hilcodec/modules/seanet.py:220-246`: CausalSTFT `conv.py:329-358`, log / normalise,
 * 1x1 conv, `x.add_(y.mul_(scale))`) without the [n_fft/2+1 x T_f] tensor ever reaching HBM.  Bit-identical to
 * hilc_stft_logmag + hilc_pw_conv(res = x).  T_f = (T-1)/hop + 1 must be a multiple of 4; x, y 16-B aligned, y != x.
 * x == NULL: the branch alone, y = out_scale * (W spec + bias) — it depends on the waveform only, so a streaming hop computes it
 * beside the previous stage and the down-sampling layer in front adds it as its `res` (same two roundings, same result).
 * dft_packed / pw_packed: hilc_spec_block_pack of
 *   which = 0: the k-major `[n_fft][n_fft]` DFT matrix whose columns are (cos_0, cos_{N/2}, cos_1, sin_1, cos_2, sin_2, ...,
 *              cos_{N/2-1}, sin_{N/2-1}) * hann — the reference basis without the all-zero sin_0 row and without sin_{N/2};
 *   which = 1: the k-major `[n_fft/2+1][C]` conv weight (hilc_pw_conv's layout);
 * nyq_sin `[n_fft]`: the sin_{N/2} row of the reference basis (|.| <= 1.4e-4, evaluated as a scalar chain).
 * hilc_spec_block_packed_floats(n_fft, which) = size of a packed operand. */
int hilc_spec_block_supported(int n_fft, int hop, int C, int T);
int hilc_spec_block_packed_floats(int n_fft, int which);
int hilc_spec_block_pack(const float* w, float* packed, int K, int n_fft, int which, void* stream);
int hilc_spec_block(const float* wav, const float* hist, int hist_len, const float* dft_packed, const float* nyq_sin,
                    const float* pw_packed, const float* bias, const float* x, float* y, int B, int T, int n_fft, int hop,
                    float mean, float std, int normalize, float out_scale, void* stream);
/* First encoder stage: the same with x = the first conv of the same waveform, computed in the kernel instead of read:
 * x[b,m,t] = sum_j pre_w[m][j] * (pre_in_scale * wav[b, t-(ksize-1)+j]) + pre_b[m]   (hilc_conv_pre; `seanet.py:280-286`,
 * `:368-372`).  n_fft = 64, hop = 1, pre_ksize = 5 only (HILC_ERR_UNSUPPORTED otherwise); bit-identical to
 * hilc_conv_pre + hilc_spec_block. */
int hilc_spec_block_conv_pre(const float* wav, const float* hist, int hist_len, const float* dft_packed,
                             const float* nyq_sin, const float* pw_packed, const float* bias, const float* pre_w,
                             const float* pre_b, float pre_in_scale, float* y, int B, int T, int n_fft, int hop, int pre_ksize,
                             float mean, float std, int normalize, float out_scale, void* stream);

/* ---- streaming cache update: out[row][i] = last `pad` samples of [hist[row][0..hist_len) | x[row][0..T)] ----
 * Replaces: `cache = x[:, :, -causal_padding:]` after `torch.cat((cache, x), dim=2)`
 * (`causal_layers.py:160-162`, waveform cache `streaming.py:486-488`). hist may be NULL (zeros). */
int hilc_tail(const float* x, const float* hist, float* out, long rows, int T, int pad, int hist_len,
              void* stream);

/* ---- L2 normalisation over channels: y = x / max(||x||_2, eps) * scale ---------------------------
 * x `[B][C][T]`; y `[B][C][T]` or, if channel_last_out, `[B][T][C]` (streaming encoder output).
 * Replaces: L2Norm (`seanet.py:151-162`, `streaming.py:279-286`). */
int hilc_l2norm(const float* x, float* y, int B, int C, int T, float eps, float scale,
                int channel_last_out, void* stream);

/* ---- residual VQ encode ------------------------------------------------------------------------
 * for i < n:  idx_i = argmin_k(norms[i][k] - 2 * <r, E_i[k]>) (first minimum), r -= E_i[idx_i], q += E_i[idx_i]
 * z: `[B][C][T]` (channel_last=0) or `[B][T][C]` (1).  codebooks `[Nq][K][C]`, codebooks_t `[Nq][C][K]`
 * (the same tables transposed, for coalesced scoring), norms `[Nq][K]` = |E|^2 (host: embed.pow(2).sum).
 * indices int64: `[B][n][T]` (stage_major=0, offline return_indices) or `[n][B][T]` (1, streaming).
 * q (optional) same layout as z.  frame_err (optional) `[B*T]` receives sum_c (z-q)^2 per frame.
 * Replaces: EuclideanCodebook.forward + ResidualVQ.forward eval branch
 * (`models/hilcodec/vector_quantize.py:132-176,199-243`, `modules/vector_quantize.py:141-195,490-516`,
 * `models/hilcodec/streaming.py:51-68,89-100`).  Returns HILC_ERR_RANGE unless 1 <= n <= Nq.
 * Every score is one fp32 fmaf chain over the channels in ascending order, whatever the batch size: small batches on the VALU
 * (4 or 16 frames per workgroup), 8 192 frames and more on the matrix pipe (v_mfma_f32_32x32x2_f32, 32 frames per workgroup,
 * the A operand read straight from codebooks_t) — same bits, same indices.  flags: 0, or HILC_RVQ_VALU_ONLY = keep large batches
 * on the VALU form too (16 frames per workgroup; the caller's switch should an fp32 MFMA ever stop being a sequential fmaf chain
 * over ascending k); any other bit: HILC_ERR_UNSUPPORTED. */
#define HILC_RVQ_VALU_ONLY 1
int hilc_rvq_encode(const float* z, const float* codebooks, const float* codebooks_t, const float* norms,
                    int64_t* indices, float* q, float* frame_err, int B, int C, int T, int K, int Nq, int n,
                    int channel_last, int stage_major, int flags, void* stream);

/* Mixed-bitrate batch (`n` drawn per request from `dropout_index`, configs/hilcodec_*.yaml:38,
 * `infer_n` :117,130): clip b uses stages [0, n_per_clip[b]) (int32 `[B]`, device; values clamped to [1, n]);
 * rows >= n_per_clip[b] of `indices` are written as -1 and leave q / the residual untouched, so clip b's
 * results equal a uniform call with n = n_per_clip[b].  `n` = rows of `indices` (>= every entry).
 * n_per_clip == NULL is hilc_rvq_encode. */
int hilc_rvq_encode_mixed(const float* z, const float* codebooks, const float* codebooks_t, const float* norms,
                          const int* n_per_clip, int64_t* indices, float* q, float* frame_err, int B, int C,
                          int T, int K, int Nq, int n, int channel_last, int stage_major, int flags, void* stream);

/* mean over `count` of frame_err[0..frames) in a fixed order -> loss[0]  (F.mse_loss, `vector_quantize.py:233`) */
int hilc_mse_finalize(const float* frame_err, float* loss, int frames, double count, void* stream);

/* ---- residual VQ decode (Dequantizer): q = sum_{i<n} E_i[idx_i] ---------------------------------
 * Replaces: Dequantizer.forward (`streaming.py:148-157`) / F.embedding sums in ResidualVQ. */
int hilc_rvq_decode(const int64_t* indices, const float* codebooks, float* q, int B, int C, int T,
                    int K, int Nq, int n, int channel_last, int stage_major, void* stream);

/* mixed-bitrate decode: clip b sums stages [0, n_per_clip[b]) only (rows beyond are ignored, e.g. the -1 rows
 * hilc_rvq_encode_mixed writes); n_per_clip == NULL is hilc_rvq_decode. */
int hilc_rvq_decode_mixed(const int64_t* indices, const float* codebooks, const int* n_per_clip, float* q, int B,
                          int C, int T, int K, int Nq, int n, int channel_last, int stage_major, void* stream);

/* ---- RVQ training side: EMA cluster statistics and codebook update -----------------
 * hilc_rvq_ema_stats: bucket `[n][K + K*C]`, per stage s: K counts (#frames with code k) then `[K][C]` sums of
 * the stage-s input residuals of those frames — `torch.cat([embed_onehot.sum(0), (embed_onehot.t() @ flatten).view(-1)])`
 * (`models/hilcodec/vector_quantize.py:155-162`) for all stages at once, so that the data-parallel reduction is ONE
 * all-reduce of n*(K + K*C) floats instead of n (`:163`).  `indices` as written by hilc_rvq_encode (rows =
 * index_rows >= n), `codebooks` = the tables the indices were computed with.  Deterministic (no atomics).
 * hilc_rvq_ema_update: ema_num = ema_num*decay + counts*(1-decay); ema_embed likewise; embed = ema_embed/ema_num
 * (`ema_inplace` `:16-17`, `:165-169`), in place on `[n][K]` / `[n][K][C]` tensors. */
int hilc_rvq_ema_stats(const float* z, const float* codebooks, const int64_t* indices, float* bucket, int B, int C,
                       int T, int K, int n, int index_rows, int channel_last, int stage_major, void* stream);
int hilc_rvq_ema_update(float* embed, float* ema_num, float* ema_embed, const float* bucket, double decay, int K,
                        int C, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HILCODEC_AMD_H */
