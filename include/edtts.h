/*
 * edtts.h -- C ABI of libedtts_hip.so: the MI355X (gfx950) implementation of the DDIM few-step sampler path
 * of Krabbens/edge-diffusion-tts.
 *
 * The reference has no FFI of its own (pure Python/PyTorch); the interface this library
 * sits behind is the reference's Python class API.  Each entry point below names the reference code whose
 * arithmetic it replaces (paths relative to /root/reference/edge_diffusion_tts/).  The Python host package
 * (edge-diffusion-tts_amd/edge_diffusion_tts_amd/native.py) binds these symbols with ctypes; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer unless the comment says "host".
 *   - all tensors are dense fp32, channel-last [B, T, n_mels] / [B, T, hidden]; indices are int64.
 *   - the caller owns every buffer (inputs, outputs, packed weights, workspace); the library never
 *     allocates device memory, never synchronises the host, and only enqueues work on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream).  All calls are graph-capturable.
 *   - return value: 0 on success, a negative EDTTS_ERR_* otherwise; edtts_last_error() returns a
 *     thread-local message for the last failing call.
 *   - environment switches, read once per process: EDTTS_SUBSTREAMS=1..8 (see edtts_set_substreams, default 4);
 *     EDTTS_DSCONV_UNFUSED=1 forces the three-kernel conv path; EDTTS_DSCONV_NOGROUP=1 / EDTTS_DSCONV_WAVES8=1 select the older
 *     one-kernel forms (A/B hooks: same results within the layer's 1e-5).
 */
#ifndef EDTTS_H_
#define EDTTS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDTTS_VERSION 400 /* 0.4.0: sub-batches on several streams (edtts_set_substreams); the workspace grew accordingly; cooperative kernels */

enum {
  EDTTS_OK = 0,
  EDTTS_ERR_UNSUPPORTED = -1, /* dims have no compiled kernel instance            */
  EDTTS_ERR_ARG = -2,         /* null pointer / size out of range (IndexError/RuntimeError analogue) */
  EDTTS_ERR_HIP = -3          /* a HIP runtime call or launch failed              */
};

/* Bits of the index-error word (edtts_index_errors): an index the reference would have raised IndexError for was
 * clamped into range by a kernel (the kernels never fault on bad indices). */
enum {
  EDTTS_IDX_SEM = 1, /* sem_idx outside [0, codebook_size)   (nn.Embedding, models/decoder.py:88) */
  EDTTS_IDX_STEP = 2 /* step_idx outside [0, n_step_emb)     (models/decoder.py:79-80) */
};

/* Decoder hyper-parameters: the CFG fields read by models/decoder.py:17-64 plus table sizes. */
typedef struct EdttsDims {
  int32_t hidden;        /* CFG.hidden            (config.py:103) */
  int32_t layers;        /* CFG.layers                            */
  int32_t heads;         /* CFG.heads                             */
  int32_t n_mels;        /* CFG.n_mels                            */
  int32_t ffn_mult;      /* CFG.ffn_mult: 1 .. 4 (a run-time tile count of the FFN phases) */
  int32_t codebook_size; /* CFG.codebook_size: rows of token_emb  */
  int32_t semantic_dim;  /* CFG.semantic_dim: sem_proj input      */
  int32_t window;        /* CFG.attn_window_size; < 0 = full self-attention (window_size=None) */
  int32_t max_pos;       /* rows of pos_emb.pe         (decoder.py:38: 1000) */
  int32_t max_ctx_pos;   /* rows of context_pos_emb.pe (decoder.py:41: 512)  */
  int32_t n_step_emb;    /* rows of step_emb           (decoder.py:32: 16)   */
  int32_t compute_dtype; /* EDTTS_F32: everything fp32 (the reference's arithmetic).  EDTTS_BF16: contractions on bf16 MFMA with
                            fp32 accumulation; residual stream, norms, softmax and sampler updates stay fp32 (the reference's AMP
                            precedent: utils/speed_utils.py:70, train_v2.py:290).  head_dim 32 shapes only (hidden = 32 * heads, hidden % 64
                            == 0): built in 256/8/80 (BASELINE config 3) and 64/2/80, more through EDTTS_INSTANCES_BF16 at build time. */
} EdttsDims;

enum { EDTTS_F32 = 0, EDTTS_BF16 = 1 };

int edtts_version(void);
const char* edtts_last_error(void);

/* ---- weights ------------------------------------------------------------------------------------------
 * The decoder state-dict (key names = SURVEY.md section 8a row 5 = decoder.state_dict() of
 * models/decoder.py:17-64) is handed over as an array of device pointers, one per "slot".  Slots
 * [0, n_global) are the non-layer tensors, then n_layer slots per transformer layer.  The slot -> key-name
 * mapping is queried, so the host never hard-codes the order.  Name "time_freqs" is the one derived slot:
 * the fp32 frequency row of SinusoidalTimeEmb (layers/embeddings.py:38-41), computed by the host exactly as
 * the reference does. */
int edtts_num_global_slots(void);
int edtts_num_layer_slots(void);
const char* edtts_global_slot_name(int i); /* e.g. "token_emb.weight" */
const char* edtts_layer_slot_name(int i);  /* e.g. "attn.qkv.weight" (prefix "layers.<l>." added by the host) */

/* Size of the packed-weight blob (MFMA-fragment order GEMM operands + tables) for these dims. */
int edtts_packed_bytes(const EdttsDims* dims, size_t* out_bytes);
/* Re-pack the state-dict into `packed` (once per weight load). slots: host array of device pointers. */
int edtts_pack_weights(const EdttsDims* dims, const void* const* slots, int n_slots, void* packed, void* stream);

/* ---- workspace ----------------------------------------------------------------------------------------
 * Scratch for activations (h, two ping-pong sets of q, k, v^T), the per-call cross-attention K/V cache and the AdaLN rows.
 * cond_rows = number of (t, step_idx) rows the conditioning kernel is run for (B for a plain forward,
 * num_steps for the fused sampler).  The workspace must be ZERO-FILLED once after allocation (padding
 * lanes are read but never written) and may then be reused for any number of calls of the same shape. */
int edtts_workspace_bytes(const EdttsDims* dims, int B, int T, int S, int cond_rows, size_t* out_bytes);

/* Out-of-range indices: where the reference raises IndexError (token ids >= codebook_size, step_idx >= 16), the kernels clamp
 * the index -- they never fault -- and OR an EDTTS_IDX_* bit into the first word of the workspace.  This call copies that word
 * to *flags_host, clears it, and SYNCHRONISES `stream` (the one call of this library that does; not graph-capturable).  The
 * Python host calls it after every decoder / sampler call when EDTTS_CHECK_INDICES=1 and raises IndexError, as the reference
 * does (a deviation from the reference otherwise: a corrupted token stream yields a plausible mel, silently).
 * (t / t_prev of edtts_ddim_step / edtts_ddpm_step are clamped into the table too; those calls have no workspace, the host
 * checks them with a device reduction in the same debug mode.) */
int edtts_index_errors(void* workspace, int* flags_host, void* stream);

/* ---- decoder forward  (models/decoder.py:66-109, EdgeDiffusionDecoder.forward) --------------------------
 * x [B,T,n_mels], t [B] int64, step_idx [B] int64 or NULL, exactly one of sem_idx [B,S] int64 /
 * sem_features [B,S,semantic_dim] non-NULL (both NULL -> EDTTS_ERR_ARG, the reference's ValueError,
 * decoder.py:90).  Writes eps [B,T,n_mels].  T <= max_pos, S <= max_ctx_pos, step_idx < n_step_emb are the
 * reference's limits; the first two are checked here, index contents are the caller's. */
int edtts_decoder_forward(const EdttsDims* dims, const void* packed, void* workspace, int B, int T, int S,
                          const float* x, const int64_t* t, const int64_t* step_idx, const int64_t* sem_idx,
                          const float* sem_features, float* eps, void* stream);

/* ---- DDIM update  (schedule.py:157-202, DiffusionSchedule.get_ddim_step) --------------------------------
 * alpha_bar [n_table] fp32 table; t, t_prev [B] int64 (t_prev < 0 -> alpha_bar_prev = 1); n_per_batch =
 * T*n_mels; eta >= 0; noise [B,T,n_mels] or NULL (required when eta > 0).  Writes x_prev and x0 (clamped
 * to [-3,3]).  Same operation order as the reference: division by sqrt(ab), direction from the raw eps.
 * Alignment: with n_per_batch % 4 == 0 and all tensors 16-byte aligned the kernel moves float4s; any other combination takes a
 * scalar path with identical results (no alignment requirement on the caller). */
int edtts_ddim_step(const float* alpha_bar, int n_table, const float* x, const float* eps, const int64_t* t,
                    const int64_t* t_prev, int B, size_t n_per_batch, float eta, const float* noise,
                    float* x_prev, float* x0, void* stream);

/* ---- DDPM update  (schedule.py:204-238, DiffusionSchedule.ddpm_step) ------------------------------------
 * tables: alphas, alpha_bar, betas, posterior_variance [n_table]; noise [B,T,n_mels] is the draw the
 * reference takes from torch.randn_like (the host supplies it).  Alignment: as edtts_ddim_step. */
int edtts_ddpm_step(const float* alphas, const float* alpha_bar, const float* betas, const float* post_var,
                    int n_table, const float* x, const float* eps, const int64_t* t, int B, size_t n_per_batch,
                    const float* noise, float* x_prev, void* stream);

/* ---- whole sampler loop  (inference.py:23-53, EdgeInference.generate_mel) -------------------------------
 * sem_idx [B,S] int64; x_T [B,2S,n_mels] initial noise (already multiplied by temperature);
 * num_steps in [1, n_step_emb]; timesteps (host) int64[num_steps] as produced by
 * range(diff_steps-1, 0, -stride)[:num_steps]; coef (host) float[num_steps*4] =
 * {sqrt(1-ab_t), sqrt(ab_t), sqrt(ab_prev), sqrt(1-ab_prev)} per step (eta = 0), computed in fp32 by the
 * host with the reference's own expressions.  x_work [B,2S,n_mels] is scratch for x_t; x0_out receives the
 * last step's clamped x0 prediction (what generate_mel returns, inference.py:53).
 * The cross-attention K/V and all AdaLN rows are computed once per call; each step's final transformer
 * layer fuses final_norm + out_proj + the DDIM update. */
int edtts_generate(const EdttsDims* dims, const void* packed, void* workspace, int B, int S,
                   const int64_t* sem_idx, const float* x_T, int num_steps, const int64_t* timesteps_host,
                   const float* coef_host, float* x_work, float* x0_out, void* stream);

/* ---- full-schedule ancestral sampler  (BASELINE config 5; schedule.py:204-238 applied num_steps times) -----------
 * The loop the reference implies but never wrote: for i = 0 .. num_steps-1, t = t_first - i:
 *     eps = decoder(x, t, sem_idx, step_idx=None);   x = ddpm_step(x, t, eps)
 * with the conditioning rows of all steps and the cross-attention K/V computed once, and the DDPM update
 * fused into the last transformer layer of every step.  x_T [B,2S,n_mels]; t_all (device) int64[num_steps] =
 * the timesteps in the order they are visited; coef (host) float[num_steps*3] = {1/sqrt(alpha_t),
 * beta_t/sqrt(1-alpha_bar_t), [t>0]*sqrt(posterior_variance_t)} per step (schedule.py:227-237).
 * Noise: noise_all [num_steps,B,2S,n_mels] if non-NULL (parity runs inject the draws the oracle used),
 * otherwise standard normals from an in-kernel Philox4x32-10 generator keyed by (seed, step, GLOBAL element index), where
 * global element = batch_offset*2S*n_mels + local element: a rank that samples rows [lo, hi) of a batch passes batch_offset = lo
 * and draws what a single GPU would have drawn for those rows.  seed and step are by-value kernel arguments: a captured
 * hipGraph replays the SAME noise; to advance it, re-capture or update the kernel node parameters with a new seed.
 * The workspace must have been sized with cond_rows = num_steps.  x_out receives x after the last step. */
int edtts_sample_ddpm(const EdttsDims* dims, const void* packed, void* workspace, int B, int S,
                      const int64_t* sem_idx, const float* x_T, int num_steps, const int64_t* t_all,
                      const float* coef_host, const float* noise_all, uint64_t seed, int64_t batch_offset, float* x_out,
                      void* stream);

/* ---- start noise  (inference.py:33: torch.randn(B, T_out, n_mels) * temperature) ---------------------------------
 * out[i] = scale * N(0,1) drawn from the Philox4x32-10 stream (seed, stream_id) at GLOBAL element index elem_offset + i, for
 * i in [0, n): a rank that owns rows [lo, hi) of a batch passes elem_offset = lo*T*n_mels and gets exactly the values a
 * single GPU would draw for those rows (shard-count-invariant, no global draw).  n and elem_offset multiples of 4; n = 0 is a
 * no-op (out may be NULL then: the empty shard of a rank that has no utterances).
 *
 * Philox stream ids.  Every draw of the library is keyed (seed, stream id, global element).  The id space is split so that two
 * draws made with one seed can never coincide:
 *     [0x00000, 0x10000)  edtts_randn callers (0 = start noise of generate_mel / sample_ddpm; the host-side long-form sampler
 *                         uses 0x51 start noise, 0x52 q_sample noise of the teacher refinement, 0x53 per-chunk coarse noise)
 *     0x10000 + step      ancestral noise of step `step` inside edtts_sample_ddpm
 *     0x20000 + step      q_sample noise of the known frames at step `step` inside edtts_sample_inpaint
 * edtts_randn rejects stream_id >= 0x10000. */
int edtts_randn(float* out, size_t n, uint64_t seed, uint32_t stream_id, uint64_t elem_offset, float scale, void* stream);

/* ---- multistep x0-solver sampler  (schedule.py:440-527, DPMSolverPP.sample with the updates of :339-438) -------
 * For step i = 0 .. num_steps-1 (t = timesteps_host[i], step_idx = i as in schedule.py:475-479):
 *     out = decoder(x, t, sem_idx | sem_features, step_idx=i)
 *     x0  = clamp(p0*x + p1*out, -3, 3)                       (v-prediction: p0 = sqrt(ab_t), p1 = -sqrt(1-ab_t); x0-pred: 0, 1)
 *     mode 1:  x = c0*x + c1*x0                                                               (first_order_update)
 *     mode 2:  x = c0*x + c1*x0 + cB*(rinv*(x0 - h_new))*0.5                                  (second_order_update)
 *     mode 3:  x = c0*x + c1*x0 + cB*(x0 - h_old)*0.5 + cC*(x0 - 2*h_old + h_new)/6           (third_order_update)
 * fused into the last transformer layer of each step; h_new / h_old are the previous two clamped x0 predictions.
 * coef_host: float[num_steps*8] = {mode, p0, p1, c0, c1, rinv, cB, cC} per step, computed by the host from the schedule
 * tables with the reference's expressions.  hist: scratch [2,B,T,n_mels]; x0_all: NULL or [num_steps,B,T,n_mels] to receive
 * every step's x0 (return_intermediates).  Exactly one of sem_idx / sem_features is non-NULL.  Workspace sized with
 * cond_rows = num_steps (<= n_step_emb).  x_out receives the final x. */
int edtts_sample_multistep(const EdttsDims* dims, const void* packed, void* workspace, int B, int T, int S,
                           const int64_t* sem_idx, const float* sem_features, const float* x_T, int num_steps,
                           const int64_t* timesteps_host, const float* coef_host, float* hist, float* x0_all,
                           float* x_out, void* stream);

/* ---- long-form in-painting sampler  (/root/reference/inference_pipeline.py:97-140 inpaint_student_sample, :145-196
 * inpaint_teacher_refine) -------------------------------------------------------------------------------------------
 * A v-prediction sampler on the sem_features context with a CONSTANT step index (3 for the student, 0 for the teacher).  x [B,T,
 * n_mels] holds the start point on entry (noise, or q_sample(x_coarse, t_start) -- the host draws it) and the result on return.
 * t_all / step_all: device int64[num_steps], the timesteps in visiting order and the (constant) step index of every step.
 * For step i = 0 .. num_steps-1 (t = t_all[i], t_next = t_all[i+1] or 0):
 *     if known_mel:  x[:, :overlap_len] = sqrt_ab[t] * known_mel + sqrt_1mab[t] * noise_i        (q_sample of the previous chunk's tail)
 *     v = decoder(x, t, sem_features, step_idx)
 *     if cfg_scale != 1:  v = v_u + cfg_scale * (v - v_u),  v_u = decoder(x, t, zero_features, step_idx)   (classifier-free guidance)
 *     x0 = clamp(sqrt_ab[t] x - sqrt_1mab[t] v, -3, 3);  eps = sqrt_1mab[t] x + sqrt_ab[t] v
 *     x  = sqrt(ab[t_next]) x0 + sqrt(1 - ab[t_next]) eps
 * and finally x[:, :overlap_len] = known_mel.  The blend, the guidance combine and the update are fused into the last transformer
 * layer of the conditional pass; the context K/V of both passes is built once.
 * coef_host: float[num_steps*4] = {sqrt_ab[t], sqrt_1mab[t], sqrt(ab[t_next]), sqrt(1-ab[t_next])}.  known_mel [B,overlap_len,n_mels]
 * or NULL.  noise_k [num_steps,B,overlap_len,n_mels] (the reference's torch.randn_like draws; parity runs inject them) or NULL ->
 * in-kernel Philox keyed by (seed, step, element).  cfg_scale != 1 needs a second workspace (same size), an all-zero feature
 * tensor [B,S,semantic_dim] and a scratch v_uncond [B,T,n_mels].  Workspaces sized with cond_rows = num_steps. */
int edtts_sample_inpaint(const EdttsDims* dims, const void* packed, void* workspace, void* workspace_uncond, int B, int T, int S,
                         const float* sem_features, const float* zero_features, float* x, int num_steps,
                         const int64_t* t_all, const int64_t* step_all, const float* coef_host, const float* known_mel,
                         int overlap_len, const float* noise_k, uint64_t seed, float cfg_scale, float* v_uncond, void* stream);

/* ---- depthwise-separable Conv1d  (layers/conv.py:25-64, DepthwiseSeparableConv.forward) -----------------
 * Standalone exported layer (named by the north star; the decoder never calls it).
 * x [B,C_in,T] channel-first; dw [C_in,k] depthwise taps (Conv1d(k, stride, padding = k/2, groups = C_in, no bias),
 * conv.py:33-41: T_out = (T + 2*(k/2) - k) / stride + 1); pw [C_out,C_in], pb [C_out]; GroupNorm(groups, eps 1e-5, affine
 * gn_w/gn_b [C_out]) then exact (erf) GELU -> y [B,C_out,T_out].
 * C_in <= 80, C_out <= 160, T_out <= 512 AND 127 * stride + ksize <= 260 (the reference's shape class: k = 3 / 5 at stride 1 or 2;
 * the raw rows of one 128-frame pass must fit the kernel's staging tile) run as ONE kernel whose intermediate never leaves
 * registers, and scratch may be NULL; every other shape (stride 3, a large ksize, more channels or frames) takes a three-kernel
 * path and needs scratch of B*C_out*T_out + 2*B*groups floats -- a NULL scratch is then EDTTS_ERR_ARG.
 * (Within the one-kernel class, the layer the reference constructs -- C_out = 160, groups = 8, stride 1, k <= 5 -- runs a
 * group-pipelined form whose stores overlap its MFMAs; the contract is the same.)
 * edtts_dsconv_scratch_floats answers for a given shape: 0 (one kernel, no scratch) or that count. */
int edtts_dsconv_scratch_floats(int B, int C_in, int C_out, int T, int ksize, int stride, int groups, size_t* out_floats);
int edtts_dsconv_forward(const float* x, const float* dw, const float* pw, const float* pb, const float* gn_w,
                         const float* gn_b, int B, int C_in, int C_out, int T, int ksize, int stride, int groups,
                         float* scratch, float* y, void* stream);

/* ---- mel post-processing  (the step after the sampler in the reference's scripts: generate_sample.py:115-145,
 * inference_pipeline.py:382-396) -----------------------------------------------------------------
 * edtts_mel_to_spec: with mean/std [B,n_mels]: denormalize_mel (utils/audio.py:17-19: mel_n * std + mean) -> exp ->; with both
 * NULL the input already is the linear mel spectrogram ->
 * torchaudio.transforms.InverseMelScale (driver "gelsd": the minimum-norm least-squares solution, i.e. multiplication by the
 * pseudo-inverse `pinv` [n_freqs,n_mels] of the transposed mel filter bank, computed once by the host) -> relu.
 * mel_n [B,T,n_mels] -> spec [B,n_freqs,T] (the reference's / torch's layout, power spectrogram).
 * edtts_griffin_lim: torchaudio.functional.griffinlim(rand_init=True, length=None): magnitude = spec^(1/power); n_iter times
 * {istft -> stft(center, reflect) -> angles = (rebuilt - m*prev) / (|.| + 1e-16), m = momentum/(1+momentum)}; final istft.
 * window [n_fft] (hann, win_length = n_fft); twiddle [n_fft/2][2] = exp(-2 pi i q / n_fft) (host table, fp64-evaluated);
 * angles0 [B,n_freqs,T,2] = the complex torch.rand draw the reference starts from (parity runs inject it) or NULL -> hashed
 * uniforms keyed by (seed, element).  scratch: edtts_griffin_lim_scratch_floats floats.  wave_out [B, hop*(T-1)].
 * Compiled for n_fft = 1024 (CFG.n_fft).  PARITY UNPINNED for these two entry points: the reference calls torchaudio, which
 * cannot be installed offline; the oracle restates torchaudio's published algorithm on torch.stft / torch.istft. */
int edtts_mel_to_spec(const float* mel_n, const float* mean, const float* stdv, const float* pinv, int B, int T, int n_mels,
                      int n_freqs, float* spec, void* stream);
int edtts_griffin_lim_scratch_floats(int B, int T, int n_fft, int hop, size_t* out_floats);
int edtts_griffin_lim(const float* spec, int B, int T, int n_fft, int hop, const float* window, const float* twiddle, int n_iter,
                      float momentum, float power, const float* angles0, uint64_t seed, float* scratch, float* wave_out,
                      void* stream);

/* ---- measurement hook (bench.py roofline leg) -----------------------------------------------------------
 * edtts_profile_enable(n > 0): from now on every transformer-layer kernel launch is bracketed by a pair of
 * hipEvents recorded on the stream it is launched on, up to n launches; n = 0 disables and releases the events.
 * A decoder layer is either one fused launch (kind 0) or two: the attention half (kind 0) and the FFN + tail half
 * (kind 1).  edtts_profile_collect synchronises on the recorded events, returns the summed device time (ms) and
 * the launch count per kind (arrays of 2), and resets the counter.  Not graph-capturable while on. */
int edtts_profile_enable(int max_records);
int edtts_profile_collect(double* ms_by_kind, int* launches_by_kind);

/* ---- sub-batches on several streams -------------------------------------------------------------------------
 * The sampler loops (edtts_generate, edtts_sample_multistep, edtts_sample_ddpm) cut a batch whose layer launches are at least two
 * full rounds of one wave per SIMD (B * ceil(T/32) >= 2 * 4 * #CUs: B >= 128 at T = 512 on an MI355X) into sub-batches that walk
 * the same launch sequence on several streams -- `stream` and library-owned non-blocking ones, forked from and joined back into
 * `stream` with events inside the call (graph-capturable; the caller sees ordinary stream semantics) -- so that one sub-batch's
 * waves fill the SIMDs another's finishing launch leaves idle.  How many: as many as still leave each sub-batch two rounds of
 * waves, at least two, at most n (B = 256: two at T = 512, four at T = 1024).  Results do not depend on the cut (bitwise).
 * n = 1 switches the cut off (per-kernel profiling wants launches that do not share the device); the default is 4 (environment
 * EDTTS_SUBSTREAMS at load time), the maximum 8.  Returns the previous value; values outside [1, 8] only query.
 * edtts_substreams_for: the number of sub-batches a call of that shape makes under the current setting. */
int edtts_set_substreams(int n);
int edtts_substreams_for(const EdttsDims* dims, int B, int T);

/* ---- small grids: the cooperative layer kernel ---------------------------------------------------------------
 * A decoder forward whose 32-frame tiles number at most half of the device's SIMDs (B * ceil(T/32) <= 2 * #CUs: B <= 32 at
 * T = 512, and the reference's B = 1 calls) runs its transformer layers with 2 or 4 waves per frame tile (csrc/edtts_coop.h:
 * attention split by heads, GEMMs by output tiles, exchanged through LDS) instead of one; the instance is chosen from the tile
 * count.  Results are bitwise those of the one-wave kernels.  Compiled for the fp32 decoders (160, 4, 80) and (256, 8, 80; without the
 * 32-frame x 2 form, whose tile state does not fit a CU twice).
 * mode: -1 automatic (default; environment EDTTS_COOP at load time), 0 off, 14 / 24 / 22 force (16-frame tiles x 4 waves,
 * 32-frame tiles x 4 waves, 32-frame tiles x 2 waves) -- for tests and measurements.  Returns the previous mode; other values
 * only query. */
int edtts_set_coop(int mode);

#ifdef __cplusplus
}
#endif
#endif /* EDTTS_H_ */
