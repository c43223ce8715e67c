/*
 * cocr.h -- C ABI of libcocr_hip.so: the MI355X (gfx950) implementation of the
 * recognition hot path of mittagessen/conformer_ocr:
 *
 *     line batch (N,1,H,W) -> conformer encoder -> linear decoder -> CTC decode
 *
 * The reference has no FFI for this path: it is the Python class
 * `PytorchRecognitionModel` (reference conformer_ocr/pred.py:50-210).  The host
 * class `conformer_ocr_amd.pred.PytorchRecognitionModel` keeps that class's
 * surface and binds these entry points through ctypes; each entry point cites
 * the reference lines whose work it takes over.
 *
 * Conventions: plain pointers and sizes only (no torch types); every function
 * returns 0 on success or a negative COCR_E* code, with a thread-local message
 * behind cocr_last_error(); the caller owns every buffer it passes in; the
 * library owns weights and workspace; a cocr_model is bound to one device and
 * must not be used from two host threads at once.  `stream` is a hipStream_t
 * (NULL = the default stream); all device work of a call is enqueued on it and
 * the call returns without synchronising unless stated.
 */
#ifndef COCR_H
#define COCR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cocr_model cocr_model;

/* element types of buffers crossing the boundary */
enum { COCR_F32 = 0, COCR_BF16 = 1, COCR_U8 = 2, COCR_I64 = 3 };

enum {
    COCR_OK = 0,
    COCR_EINVAL = -1,      /* bad argument / shape / name   -> ValueError   */
    COCR_ESTATE = -2,      /* call out of order             -> RuntimeError */
    COCR_EHIP = -3,        /* HIP runtime error             -> RuntimeError */
    COCR_EUNSUPPORTED = -4 /* hyper-parameters outside what the kernels cover */
};

/* The shape-determining constructor arguments of PytorchRecognitionModel
 * (pred.py:51-66); dropout probabilities are inference no-ops and not passed. */
typedef struct {
    int32_t num_classes;
    int32_t height;
    int32_t encoder_dim;
    int32_t num_encoder_layers;
    int32_t num_attention_heads;
    int32_t feed_forward_expansion_factor;
    int32_t conv_expansion_factor;
    int32_t conv_kernel_size;
    int32_t half_step_residual;
    int32_t subsampling_conv_channels;
    int32_t subsampling_factor;
} cocr_hparams;

const char *cocr_last_error(void);
const char *cocr_version(void);

/* Replaces the module construction of pred.py:70-91 (ConformerEncoder + nn.Linear decoder).
 * Validates like the reference constructors do (attention.py:53, convolution.py:132-133,177-178). */
int cocr_create(const cocr_hparams *hp, int device, cocr_model **out);
void cocr_destroy(cocr_model *m);

/* Replaces nn.Module.load_state_dict (pred.py:194,209): hands one state-dict entry to the model.
 * `name` is the reference key below `nn.` ("encoder.layers.0...", "decoder.weight"); `host` is
 * read during the call (copied).  dtype COCR_F32, or COCR_I64 for num_batches_tracked (ignored). */
int cocr_set_tensor(cocr_model *m, const char *name, const void *host, int dtype, int ndim, const int64_t *shape);

/* Names of tensors still missing, '\n'-separated, into buf (for strict loading); returns the count. */
int cocr_missing_tensors(cocr_model *m, char *buf, size_t buflen);

/* Packs the weights for the kernels and uploads them: BatchNorm folded into the depthwise taps
 * (convolution.py:140-141, eval mode), positional projection P = PE Wpos^T precomputed for the full
 * 9999-row table (embedding.py:35-66, attention.py:62,85,146), GLU / flatten permutations.
 * compute_dtype: COCR_BF16 (bf16 operands, fp32 accumulate, fp32 residual stream) or COCR_F32.
 * Synchronises the device. */
int cocr_finalize(cocr_model *m, int compute_dtype);

/* Multi-GPU start-up: a rank that receives its weights by broadcast allocates the packed blob
 * without filling it, hands (ptr, bytes) to its collective library (RCCL broadcast from the rank
 * that ran cocr_finalize), and is then ready.  The blob layout depends only on (hparams, dtype). */
int cocr_finalize_empty(cocr_model *m, int compute_dtype);
/* Kernel-specific re-layouts of some matrices (MFMA-fragment order) are DERIVED from the blob on the first forward after
 * cocr_finalize, cocr_blob_import or a call of cocr_weight_blob: a caller that writes through the pointer later must call
 * cocr_weight_blob again (or use cocr_blob_import) before the next forward, or those copies stay stale. */
int cocr_weight_blob(cocr_model *m, void **device_ptr, size_t *bytes);
/* The same through a caller-owned device buffer of exactly `bytes` = blob size (stream-ordered device-to-device copies):
 * export on the root, broadcast the caller's buffer, import on the other ranks. */
int cocr_blob_export(cocr_model *m, void *dst_device, size_t bytes, void *stream);
int cocr_blob_import(cocr_model *m, const void *src_device, size_t bytes, void *stream);

/* calc_length (convolution.py:240-247) with k=3, s=2, p=1 repeated log2(subsampling_factor) times. */
int32_t cocr_out_len(int32_t in_len, int32_t subsampling_factor);

/* Several packed copies of ONE model (one per stream of a caller that keeps several batches in flight) need one set of weights: after
 * this call `m` -- same hyper-parameters, same device -- reads `owner`'s packed weights, fragment-major copies and positional tables
 * and keeps only a workspace and captured launch sequences of its own.  (Four private copies of the cfg2 model are 4 x ~100 MB, more than
 * the 256 MB Infinity Cache: every forward then streamed its weights from HBM, cycling four copies on one stream ran 30 % slower than one.)
 * `owner` must outlive `m` and must not itself share; weights are changed through the owner (set_tensor + finalize, blob import, the
 * training entry points), `m` sees them at its next forward; finalizing `m` gives it weights of its own again. */
int cocr_share_weights(cocr_model *m, cocr_model *owner);

/* Host-side collation of a line batch: what kraken's `collate_sequences` does for the reference's loaders (cli/test.py:186-189, the
 * `DataLoader(..., collate_fn=collate_sequences)`): N lines of H rows each, line i `widths[i]` elements wide and C-contiguous, are
 * copied left-aligned into the (N,H,W) batch `dst` and the rest of every row is zeroed.  Plain host memory on both sides (dst is
 * typically a pinned staging buffer the caller uploads with one copy); elem_size 1 (uint8 lines) or 4 (float32); the rows are split
 * over `threads` host threads (<= 1: the calling thread alone).  No GPU call is made. */
int cocr_collate_lines(const void *const *lines, const int32_t *widths, int N, int H, int elem_size, void *dst, int W, int threads);

/* Pre-allocates workspace for batches up to N lines of width W (otherwise cocr_forward grows it on
 * demand, which synchronises and must not happen inside a stream capture). */
int cocr_reserve(cocr_model *m, int N, int W);

/* PytorchRecognitionModel.forward (pred.py:101-122): lines (N,H,W) [the (N,1,H,W) batch with the
 * singleton channel dropped; W fastest], DEVICE memory of type line_dtype (COCR_F32 in [0,1] or
 * COCR_U8, read as u8/255).  in_lens (N) HOST int32 pixel widths.  logits: DEVICE float32
 * (N,T,num_classes) with T = cocr_out_len(W).  out_lens: HOST int32 (N), written before return.
 * The padded region is processed like the reference does: no masking. */
int cocr_forward(cocr_model *m, const void *lines, int line_dtype, int N, int H, int W,
                 const int32_t *in_lens, float *logits, int32_t *out_lens, void *stream);

/* Replaces the per-line host loop `self.ctc_decoder(seq[:, :seq_len])` with
 * kraken.lib.ctc_decoder.greedy_decoder (pred.py:138-145,158-164,173-178; model.py:163-168):
 * per line argmax over classes (first index on ties), merge runs, drop blank (0).
 * logits DEVICE float32 (N,T,ncls); out_lens HOST int32 (N).  Outputs DEVICE, caller-owned:
 * labels/starts/ends int32 (N,max_per_line), conf float32 (N,max_per_line) = max over the run of the
 * label's logit, counts int32 (N).  A line emits at most ceil(len/1) runs; max_per_line >= T is safe;
 * excess runs are counted but not stored.
 * When `logits` is the buffer the model's LAST cocr_forward wrote (same pointer, same N*T; models of up to 128
 * classes), the per-frame argmax was already taken in the decoder product's epilogue and only the run merge is
 * launched.  A caller that edits those logits in place before decoding calls cocr_forget_argmax first. */
int cocr_ctc_greedy(cocr_model *m, const float *logits, int N, int T, int ncls, const int32_t *out_lens,
                    int32_t *labels, int32_t *starts, int32_t *ends, float *conf, int32_t *counts,
                    int max_per_line, void *stream);
int cocr_forget_argmax(cocr_model *m);

/* CTC prefix beam search (kraken.lib.ctc_decoder.beam_decoder's algorithm on log-softmax(logits);
 * semantics fixed in oracle/ctc_ref.py::beam_decoder).  Same buffers as cocr_ctc_greedy; conf is a
 * softmax probability.  beam <= 32. */
int cocr_ctc_beam(cocr_model *m, const float *logits, int N, int T, int ncls, const int32_t *out_lens,
                  int32_t *labels, int32_t *starts, int32_t *ends, float *conf, int32_t *counts,
                  int max_per_line, int beam, void *stream);

/* The loss of the reference's training / validation step, RecognitionModel._step (model.py:119,136-142):
 *     nn.CTCLoss(reduction='sum', zero_infinity=True)(log_softmax(probits, -1).transpose(0, 1), target, encoder_lens, label_lens)
 * and its gradient with respect to `probits` (what autograd hands to the decoder's backward).  probits DEVICE float32 (N,T,ncls), the
 * output of cocr_forward; out_lens HOST (N) valid frames; targets HOST int32, the batch's labels concatenated (the reference's 1-D
 * `target`), label_lens HOST (N); blank = 0, labels in [1, ncls), at most 255 per line.  nll (N) float32, DEVICE or pinned host:
 * per-line negative log-likelihoods, 0 where no alignment exists (zero_infinity) -- the reference's scalar is their plain sum.
 * grad (N,T,ncls) DEVICE float32 or NULL (validation: loss only): d sum(nll) / d probits, zero for frames >= out_lens[n].
 * Deterministic (no floating-point atomics).  Stream-ordered, does not synchronise (workspace growth does). */
int cocr_ctc_loss(cocr_model *m, const float *probits, int N, int T, int ncls, const int32_t *out_lens, const int32_t *targets,
                  const int32_t *label_lens, float *nll, float *grad, void *stream);

/* Training step of the output layer: the part of the reference's `training_step` (model.py:147-152, autograd through
 * `nn['decoder'] = nn.Linear(encoder_dim, num_classes)`, model.py:115) between the encoder output and the criterion, and the
 * reference's default optimizer (`torch.optim.AdamW`, model.py:47,283-284).  The encoder's backward is not in this library yet;
 * with these calls the output layer trains on a frozen backbone (`freeze_backbone`, default_specs.py:47).
 * cocr_decoder_backward: grad_probits DEVICE float32 (N,T,ncls) = d loss / d probits (cocr_ctc_loss) for the LAST cocr_forward on this
 * model (N, T must match it: the encoder output it multiplied is still in the workspace) ->
 *   grad_weight (ncls, encoder_dim), grad_bias (ncls): DEVICE float32, the `.grad` of decoder.weight / decoder.bias;
 *   grad_output (N,T,encoder_dim) DEVICE float32 or NULL: d loss / d encoder output (what the encoder's backward would receive).
 * Deterministic (fixed-order reductions).  Stream-ordered.
 * cocr_decoder_adamw: one torch.optim.AdamW step (decoupled weight decay, no amsgrad) on decoder.weight / decoder.bias with the given
 * gradients (after the caller's all-reduce across ranks, if any); keeps an fp32 master copy and the two moment buffers inside the
 * model (created on the first call: step counter 1), and writes the updated values where the next cocr_forward reads them.
 * cocr_finalize / cocr_blob_import discard that state.
 * cocr_get_tensor: decoder.weight / decoder.bias as float32 into HOST memory (the trained master copy if training has started, else
 * the tensor given to cocr_set_tensor) -- for writing checkpoints.  Synchronises `stream`. */
int cocr_decoder_backward(cocr_model *m, const float *grad_probits, int N, int T, float *grad_weight, float *grad_bias, float *grad_output,
                          void *stream);
int cocr_decoder_adamw(cocr_model *m, const float *grad_weight, const float *grad_bias, float lr, float beta1, float beta2, float eps,
                       float weight_decay, void *stream);
int cocr_get_tensor(cocr_model *m, const char *name, float *host_out, int64_t max_elems, void *stream);

/* Line pre-processing in front of the path -- the step the reference delegates to kraken's
 * ImageInputTransforms(1, 96, 0, 1, (16, 0), valid_norm=False) (reference dataset.py:89, cli/test.py:156): grayscale, scale to
 * height `out_h` keeping the aspect ratio (Pillow's 8-bit LANCZOS resampler, bit for bit), `pad` zero columns left and right,
 * right-zero-filled to the batch width.  Input: N raw 8-bit line crops packed in one DEVICE buffer `pixels` (line i starts at
 * byte offsets[i], heights[i] rows of widths[i] pixels of channels[i] = 1 (L) or 3 (RGB, converted like Pillow) bytes;
 * channels == NULL: all 1).  Output: `out` (N, out_h, out_w) uint8 on the device -- the COCR_U8 line batch of cocr_forward
 * (pixel / 255 are the reference's [0, 1] floats) -- and out_widths[i] (HOST) = scaled width + 2 pad, the batch's `seq_lens`.
 * Errors: COCR_EINVAL if a scaled line does not fit out_w.  Synchronises `stream` once (tap tables are built on the host).
 * cocr_preproc_width: the width line (h, w) will have, for choosing out_w / bucketing before the call. */
int32_t cocr_preproc_width(int32_t h, int32_t w, int32_t out_h, int32_t pad);
int cocr_preproc_lines(cocr_model *m, const uint8_t *pixels, const int64_t *offsets, const int32_t *heights, const int32_t *widths,
                       const int32_t *channels, int N, int out_h, int pad, int out_w, uint8_t *out, int32_t *out_widths, void *stream);

/* Launch-overhead control: with graph replay on, cocr_forward captures its ~40 kernel launches into a hipGraph and
 * replays it.  A caller that reuses (lines, logits, N, W, dtype, stream) gets a graph on its own buffers (second
 * identical call captures, later ones replay; contents may change, addresses not).  A caller with fresh buffers per
 * call gets a graph keyed by (N, W, dtype, stream) on library-owned staging buffers: one device-to-device copy of the
 * lines in and of the logits out per call.  Off by default. */
int cocr_set_graph(cocr_model *m, int on);

/* Rows of the (N*T, D) activation one workgroup of the row-chain kernels owns (bf16 mode, encoder_dim 256 / 512).
 * 0 (default): by the batch's row count -- the largest block (96 rows at D = 256, 64 at D = 512: every weight byte is
 * streamed once per block, the cheapest form when several batches are in flight).  A caller with ONE batch in flight
 * gets a shorter forward from smaller blocks (e.g. 48: twice the workgroups, 32 lines x 300 frames = 200 workgroups).
 * Same results bit for bit (a row's arithmetic does not depend on the block it is in). */
int cocr_set_chain_rows(cocr_model *m, int rows);

/* Test taps: with debug on, cocr_forward keeps a float32 copy of every stage output
 * ("front.z2", "front.z3", "front.y", "l<i>.ffn1|mhsa|conv|ffn2|out", "l<i>.q|k|v|ctx|glu|dw").
 * cocr_debug_tap copies one to HOST memory; *n_elems receives its element count. */
int cocr_set_debug(cocr_model *m, int on);
int cocr_debug_tap(cocr_model *m, const char *name, float *host_out, int64_t max_elems, int64_t *n_elems);

/* Kernel timing hook for bench.py's roofline leg: average device time (ms) per launch of each
 * kernel family over the forwards run since cocr_profile(m,1), measured with HIP events on the
 * forward's own stream.  Names are '\n'-separated in `names`; ms[i], launches[i] per family. */
int cocr_profile(cocr_model *m, int on);
int cocr_profile_read(cocr_model *m, char *names, size_t names_len, double *ms, int64_t *launches, int max_entries);

/* ---- Training step of the whole network: RecognitionModel.training_step (model.py:129-152) + torch.optim.AdamW (model.py:283-284).
 * fp32.  cocr_train_begin copies every parameter / buffer given through cocr_set_tensor to the device (AdamW state zeroed);
 * cocr_train_step runs the TRAIN-mode forward (BatchNorm1d batch statistics over all positions of the padded batch + running-statistics
 * update, dropout at the reference's six sites with probabilities dropout_p = {input, feed_forward, attention, conv}, masks from
 * (seed, site, index)), the criterion nn.CTCLoss(reduction='sum', zero_infinity=True) on log_softmax(probits), and the backward through
 * decoder and encoder: afterwards cocr_train_get(name, kind = 1) returns d loss / d parameter for every reference state-dict name
 * (kind = 0: the current value of a parameter / buffer).  lines as for cocr_forward; in_lens, targets (concatenated labels),
 * label_lens HOST int32; *loss_out HOST.  cocr_train_adamw applies one AdamW step to all parameters; cocr_train_end copies the trained
 * values back into the model's state (call cocr_finalize again to serve them) and frees the training state. */
int cocr_train_begin(cocr_model *m);
int cocr_train_step(cocr_model *m, const void *lines, int line_dtype, int N, int H, int W, const int32_t *in_lens, const int32_t *targets,
                    const int32_t *label_lens, const float *dropout_p, uint64_t seed, float *loss_out, void *stream);
int cocr_train_get(cocr_model *m, const char *name, int kind, float *host_out, int64_t n_elems, void *stream);
int cocr_train_adamw(cocr_model *m, float lr, float beta1, float beta2, float eps, float weight_decay, void *stream);
int cocr_train_end(cocr_model *m);
/* Matmul precision of the training step's Linear / pointwise-conv products (forward, input and weight gradients): 0 (default) exact fp32
 * on the matrix cores; 1 = both operands rounded to bf16, fp32 accumulation -- what the reference trains under
 * (torch.set_float32_matmul_precision('medium'), cli/train.py:252).  Parameters, activations, gradients and the optimizer stay fp32. */
int cocr_train_set_matmul(cocr_model *m, int bf16_operands);
/* The flat device gradient vector (float32, all parameters): a data-parallel job all-reduces it between cocr_train_step and cocr_train_adamw. */
int cocr_train_grad_buffer(cocr_model *m, void **device_ptr, size_t *n_floats);
/* The flat device VALUE vector in the same layout (parameters [0, *n_params), then the BatchNorm running statistics up to *n_total) and
 * the place of a reference state-dict name in both vectors.  This is what puts the step behind torch autograd (reference
 * model.py:129-152: `loss.backward()`, then any torch optimizer, model.py:283-289): the host copies `net.nn.parameters()` in, runs
 * cocr_train_step, hands slices of the gradient vector to autograd and copies the running statistics out -- conformer_ocr_amd/autograd.py. */
int cocr_train_param_buffer(cocr_model *m, void **device_ptr, size_t *n_total, size_t *n_params);
int cocr_train_layout(cocr_model *m, const char *name, int64_t *offset, int64_t *n_elems, int *is_param);

#ifdef __cplusplus
}
#endif
#endif /* COCR_H */
