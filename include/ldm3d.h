/* ldm3d.h - C ABI of libldm3d.so: the MI355X-native (gfx950) 3D latent-diffusion denoising path.
 *
 * This is the drop-in boundary for the hot path of sanazkaviani/3d-latent-diffusion-model.  The reference has
 * no native code: its plug point is the JSON "_target_" class instantiated by define_instance
 * (3d_ldm/utils.py:243-246) and then used through the nn.Module surface; each entry point below cites the
 * reference call it serves.  The host-side mirror (the .py files of 3d-latent-diffusion-model_amd/, loaded with ctypes)
 * exposes these as the same nn.Module / scheduler / inferer objects.  See INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative ldm_status; ldm_last_error() gives the message
 *     (thread local).  Nothing aborts, nothing throws across the ABI.
 *   - all tensor arguments are DEVICE pointers owned by the caller; image/latent tensors are fp32 NCDHW
 *     (the reference's layout); the library converts to its internal NDHWC bf16 layout itself.
 *   - the caller owns the workspace (size from *_workspace_bytes); the library owns weights and plans.
 *     No allocation, no synchronisation on the forward/step path: every launch goes to `stream`
 *     (a hipStream_t passed as void*), so calls may be captured into a hipGraph.
 *   - handles are not thread safe; one process per GPU (torchrun model, 3d_ldm/train_diffusion.py:43-51).
 */
#ifndef LDM3D_H
#define LDM3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDM_MAX_LEVELS 8
#define LDM_ABI_VERSION 1

typedef enum {
    LDM_OK = 0,
    LDM_ERR_BAD_ARG = -1,
    LDM_ERR_UNSUPPORTED = -2,     /* shape / config outside what the kernels implement */
    LDM_ERR_HIP = -3,
    LDM_ERR_NOT_LOADED = -4,      /* forward called before every parameter was uploaded */
    LDM_ERR_WORKSPACE = -5,       /* workspace too small */
    LDM_ERR_RCCL = -6
} ldm_status;

/* kwargs of "diffusion_def" (3d_ldm/config/config_train_16g.json:39-48); MONAI defaults norm_num_groups=32,
 * norm_eps=1e-6 are filled by the host mirror. */
typedef struct {
    int spatial_dims;                         /* only 3 */
    int in_channels, out_channels;
    int num_levels;
    int channels[LDM_MAX_LEVELS];
    int attention_levels[LDM_MAX_LEVELS];
    int num_head_channels[LDM_MAX_LEVELS];
    int num_res_blocks[LDM_MAX_LEVELS];
    int norm_num_groups;
    float norm_eps;
} ldm_unet_cfg;

/* kwargs of "autoencoder_def" (3d_ldm/config/config_train_16g.json:7-28). */
typedef struct {
    int spatial_dims;                         /* only 3 */
    int in_channels, out_channels, latent_channels;
    int num_levels;
    int channels[LDM_MAX_LEVELS];
    int num_res_blocks[LDM_MAX_LEVELS];
    int attention_levels[LDM_MAX_LEVELS];
    int norm_num_groups;
    float norm_eps;
    int with_encoder_nonlocal_attn, with_decoder_nonlocal_attn;
} ldm_vae_cfg;

typedef struct ldm_model ldm_model;           /* a UNet or an AutoencoderKL: weights + cached launch plans */

int ldm_version(void);
const char* ldm_last_error(void);

/* ---- construction (replaces define_instance(args, "diffusion_def" / "autoencoder_def"),
 *      3d_ldm/utils.py:243-246, 3d_ldm/inference.py:71,75, 3d_ldm/train_diffusion.py:90,127) ------------ */
int ldm_unet_create(const ldm_unet_cfg* cfg, ldm_model** out);
int ldm_vae_create(const ldm_vae_cfg* cfg, ldm_model** out);
void ldm_model_destroy(ldm_model* m);

/* ---- parameters: MONAI-shaped state_dict (replaces load_state_dict, 3d_ldm/inference.py:73,77,
 *      3d_ldm/train_diffusion.py:92-95,129-136).  Enumerate names/shapes, then upload fp32 HOST arrays. ----- */
int ldm_model_num_params(const ldm_model* m);
const char* ldm_model_param_name(const ldm_model* m, int i);
int ldm_model_param_ndim(const ldm_model* m, int i);
const int64_t* ldm_model_param_shape(const ldm_model* m, int i);
int ldm_model_load_param(ldm_model* m, const char* name, const float* host_data, size_t numel);
int64_t ldm_model_param_numel_total(const ldm_model* m);

/* ---- DiffusionModelUNet.forward(x, timesteps, context=None)
 *      (reached via inferer(...) 3d_ldm/train_diffusion.py:197-205,260-268 and inferer.sample(...)
 *      3d_ldm/train_diffusion.py:326-333, 3d_ldm/inference.py:94-99).
 *      x:[B,Cx,D,H,W], cond:[B,Cc,D,H,W] or NULL (mode="concat": Cx + Cc == in_channels), timesteps:[B] fp32,
 *      out:[B,out_channels,D,H,W]; all fp32 NCDHW device memory. ---------------------------------------------- */
size_t ldm_unet_workspace_bytes(ldm_model* m, int B, int D, int H, int W);
int ldm_unet_forward(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                     const float* timesteps, float* out, int B, int D, int H, int W,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- one training step of the diffusion UNet (3d_ldm/train_diffusion.py:197-223: inferer(...) -> F.mse_loss ->
 *      loss.backward() -> clip_grad_norm_(1.0) -> Adam.step()).
 *      train_forward == forward, but keeps every activation / statistic in `workspace`; train_backward consumes that
 *      workspace and dLoss/d eps_hat (fp32 [B,out_channels,D,H,W]) and overwrites flat_grads (fp32, one element per
 *      parameter element: parameter i starts at ldm_model_param_offset(m, i) and has its MONAI tensor layout).
 *      load_params_device re-packs the fp32 master parameters (HOST array of n DEVICE pointers, library order) into
 *      the bf16 weight arena after an optimizer step.  The flat layout is what the data-parallel all-reduce
 *      (3d_ldm/train_diffusion.py:147-149 DDP) runs over: one ldm_comm_allreduce of the whole buffer. ----------------- */
int64_t ldm_model_param_offset(const ldm_model* m, int i);
int ldm_model_load_params_device(ldm_model* m, const float* const* device_ptrs, int n, void* stream);
int ldm_model_load_params_flat(ldm_model* m, const float* flat_device, void* stream);   /* all parameters in ONE buffer at their ldm_model_param_offset */
size_t ldm_unet_train_workspace_bytes(ldm_model* m, int B, int D, int H, int W);
int ldm_unet_train_forward(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                           const float* timesteps, float* out, int B, int D, int H, int W,
                           void* workspace, size_t workspace_bytes, void* stream);
int ldm_unet_train_backward(ldm_model* m, const float* grad_out, float* flat_grads, int B, int D, int H, int W,
                            void* workspace, size_t workspace_bytes, void* stream);
/* AutoencoderKL.forward(images) -> (reconstruction, z_mu, z_sigma) with its backward (stage-1 trainer,
 * 3d_ldm/train_autoencoder.py:366-451: recons + KL losses -> loss_g.backward()).  eps: [B,L,d,h,w] N(0,1) draws of the
 * sampling step; d_mu / d_sigma: gradients of the KL term w.r.t. z_mu / z_sigma (NULL = 0); flat_grads as above. */
size_t ldm_vae_train_workspace_bytes(ldm_model* m, int B, int D, int H, int W);
int ldm_vae_train_forward(ldm_model* m, const float* x, const float* eps, float* recon, float* z_mu, float* z_sigma,
                          int B, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
int ldm_vae_train_backward(ldm_model* m, const float* d_recon, const float* d_mu, const float* d_sigma, float* flat_grads,
                           int B, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
/* Optimizer tail on flat fp32 buffers (torch.nn.utils.clip_grad_norm_(params, max_norm) 3d_ldm/train_diffusion.py:216
 * and torch.optim.Adam(lr) :155, torch defaults betas (0.9, 0.999), eps 1e-8, no weight decay):
 *   grad_sq_norm: *out (device fp32 scalar) = sum g^2.
 *   adam_step: g' = g * min(1, max_norm / (sqrt(*sq_norm) + 1e-6)) when sq_norm != NULL and max_norm > 0; then
 *              m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p = p (1 - lr wd) - lr * (m / (1-b1^step)) / (sqrt(v / (1-b2^step)) + eps)
 *              (wd = 0: torch.optim.Adam of train_diffusion.py:155; wd > 0: the decoupled AdamW of train_autoencoder.py:274-279).
 *   sq_norm points at TWO device floats {sum g^2, skipped steps}: the NaN-skip of the reference's trainers (train_diffusion.py:210-212
 *   `continue`s in front of backward when the loss is NaN) without a host read: when sum g^2 is not finite (a NaN loss makes every
 *   gradient NaN; the data-parallel mean carries it to every rank) the step leaves params / exp_avg / exp_avg_sq untouched and adds 1
 *   to sq_norm[1]; the bias corrections use step - sq_norm[1].  The caller zeroes sq_norm[1] once; ldm_grad_sq_norm writes out[0] only.
 *   The skip depends on sq_norm alone: max_norm <= 0 switches the clip factor off, not the finite check (pass sq_norm = NULL for a
 *   plain Adam step with neither).
 *   ldm_grad_sq_norm and ldm_op_mse_loss reduce through ONE scratch buffer per process (allocated on the device that is current at
 *   the first call): call them from one stream of one device at a time. */
int ldm_grad_sq_norm(const float* flat_grads, int64_t n, float* out, void* stream);
/* F.mse_loss(noise_pred, noise) of 3d_ldm/train_diffusion.py:207 together with the gradient loss.backward() (:214) hands to the network:
 * loss_out[0] = mean((pred - target)^2), grad_out (optional) = 2 (pred - target) / n; fp32 device buffers of n elements. */
int ldm_op_mse_loss(const float* pred, const float* target, int64_t n, float* loss_out, float* grad_out, void* stream);
int ldm_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* sq_norm, float max_norm, void* stream);
/* Adam(W) over a model's flat fp32 parameter buffer (layout of ldm_model_param_offset) that re-packs the library's bf16 weight
 * arena from the updated values in the same pass (optimizer.step() + the implicit "weights changed", train_diffusion.py:219). */
int ldm_model_adam_step(ldm_model* m, float* params_flat, const float* grads_flat, float* exp_avg, float* exp_avg_sq, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, const float* sq_norm, float max_norm,
                        void* stream);

/* HIP-graph replay of the UNet forward plan (the sampling loop of 3d_ldm/inference.py:88-99 calls the UNet 1000 times per
 * volume with the same shapes): with on != 0, ldm_unet_forward records its launches into a hipGraph the second time it sees
 * a (x, cond, timesteps, out, workspace, stream) pointer set and replays it afterwards.  Results are identical. */
int ldm_model_set_graph_mode(ldm_model* m, int on);

/* ---- precision of the inference AND training plans of both networks.  The reference computes in fp32 (autocast off: 3d_ldm/train_diffusion.py:177,237;
 *      3d_ldm/inference.py:91-99 has no autocast): LDM_PREC_FP32 runs the same plans on fp32 activations / weights with the
 *      fp32 matrix instruction (v_mfma_f32_32x32x2_f32; the convolutions of the inference plans as three bf16 MFMAs per product on
 *      hi / lo splits of the fp32 operands, fp32-class accuracy) and stays within ~5e-5 rel-L2 of the CPU path; LDM_PREC_BF16 (default)
 *      is the bf16-MFMA headline path (~3e-2 at the benchmark depth, its own rounding floor).  After switching to fp32 upload the
 *      parameters again (the unrounded copies are made at upload time). ------------------------------------------------- */
#define LDM_PREC_BF16 0
#define LDM_PREC_FP32 1
int ldm_model_set_precision(ldm_model* m, int precision);
int ldm_model_get_precision(const ldm_model* m);

/* ---- debug taps: every block output of an inference plan as fp32 NCDHW, optionally followed by overwriting it with the
 *      caller's tensor (teacher forcing), for stage-wise parity tests against the oracle.  kind = "unet" | "enc" | "dec". ---- */
int ldm_model_tap_count(ldm_model* m, const char* kind, int B, int D, int H, int W);
int64_t ldm_model_tap_elems(ldm_model* m, const char* kind, int B, int D, int H, int W);
int ldm_model_tap_info(ldm_model* m, const char* kind, int B, int D, int H, int W, int i, char* name, int name_cap, int dims[5],
                       int64_t* offset);
size_t ldm_model_taps_workspace_bytes(ldm_model* m, const char* kind, int B, int D, int H, int W, int force);
int ldm_unet_forward_taps(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                          const float* timesteps, float* out, int B, int D, int H, int W, float* taps_out, const float* taps_in,
                          void* workspace, size_t workspace_bytes, void* stream);
int ldm_vae_encode_taps(ldm_model* m, const float* x, const float* eps, float* z_mu, float* z_sigma, float* z,
                        int B, int D, int H, int W, float* taps_out, const float* taps_in,
                        void* workspace, size_t workspace_bytes, void* stream);
int ldm_vae_decode_taps(ldm_model* m, const float* z, float* out, int B, int d, int h, int w, float* taps_out, const float* taps_in,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- AutoencoderKL.encode / sampling / decode (3d_ldm/train_diffusion.py:104,180,195,249,258,310,324;
 *      3d_ldm/train_autoencoder.py:366,579).  encode: x:[B,Cin,D,H,W] -> z_mu, z_sigma, z = mu + sigma*eps
 *      (each [B,L,D/f,H/f,W/f], any of the three outputs may be NULL; eps NULL means eps = 0).
 *      decode: z:[B,L,d,h,w] -> out:[B,Cout,d*f,h*f,w*f]. ------------------------------------------------------ */
size_t ldm_vae_encode_workspace_bytes(ldm_model* m, int B, int D, int H, int W);
size_t ldm_vae_decode_workspace_bytes(ldm_model* m, int B, int d, int h, int w);
int ldm_vae_encode(ldm_model* m, const float* x, const float* eps, float* z_mu, float* z_sigma, float* z,
                   int B, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
int ldm_vae_decode(ldm_model* m, const float* z, float* out, int B, int d, int h, int w,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- scheduler arithmetic (monai DDPMScheduler / DDIMScheduler as built at 3d_ldm/inference.py:79-84,
 *      3d_ldm/train_diffusion.py:140-145).  The host mirror keeps the fp32 beta / alpha-bar tables and passes
 *      the per-step scalars; these launches are the element-wise part, n = number of elements.
 *      ddpm:  x0 = (x - sqrt_b*eps)*inv_sqrt_a ; clip ; prev = c0*x0 + c1*x (+ sigma*noise)
 *      ddim:  x0 likewise           ; prev = c0*x0 + dir*eps (+ sigma*noise)
 *      noise / x0_out may be NULL. ---------------------------------------------------------------------------- */
int ldm_ddpm_step(const float* eps, const float* x, const float* noise, float* prev, float* x0_out, int64_t n,
                  float inv_sqrt_a, float sqrt_b, float c0, float c1, float sigma, int clip, void* stream);
int ldm_ddim_step(const float* eps, const float* x, const float* noise, float* prev, float* x0_out, int64_t n,
                  float inv_sqrt_a, float sqrt_b, float c0, float dir, float sigma, int clip, void* stream);
/* add_noise: out = sqrt_a[b]*x0 + sqrt_b[b]*eps; sqrt_a, sqrt_b: [B] device fp32 (3d_ldm/train_diffusion.py:197-205) */
int ldm_add_noise(const float* x0, const float* eps, const float* sqrt_a, const float* sqrt_b, float* out,
                  int B, int64_t per_sample, void* stream);
int ldm_scale(const float* x, float* y, int64_t n, float s, void* stream);

/* ---- device-resident sampler: the scheduler step with its noise drawn inside the kernel (Philox4x32-10 keyed by a seed) and its
 *      coefficients / current timestep read from device memory, so that the loop body of 3d_ldm/inference.py:94-99 (UNet forward +
 *      scheduler.step) is one fixed launch sequence: ldm_unet_denoise_step replays it as ONE HIP graph in graph mode.
 *      coef_host: [n_steps][6] = {1/sqrt(abar_t), sqrt(1 - abar_t), c0, c1 | dir, sigma, t} per step in sampling order. ---------- */
typedef struct ldm_sampler ldm_sampler;
int ldm_sampler_create(const float* coef_host, int n_steps, int kind /* 0 DDPM, 1 DDIM */, int clip, uint64_t seed, ldm_sampler** out);
void ldm_sampler_destroy(ldm_sampler* sp);
int ldm_sampler_reset(ldm_sampler* sp, float* tbuf, int B, void* stream);
int ldm_sampler_step(ldm_sampler* sp, const float* eps, float* x, float* x0_out, int64_t n, float* tbuf, int B, void* stream);
int ldm_sampler_noise(const ldm_sampler* sp, int step, float* out, int64_t n, void* stream);
int ldm_unet_denoise_step(ldm_model* m, ldm_sampler* sp, float* x, int x_channels, const float* cond, int cond_channels,
                          float* tbuf, float* eps_scratch, int B, int D, int H, int W,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- operator level: the kernels the plans are made of, on the library's internal layout (NDHWC bf16 device
 *      tensors, C % 32 == 0).  They replace torch.nn.functional.conv3d / group_norm+silu / softmax-attention as
 *      MONAI's blocks call them and exist for per-kernel parity tests and micro-benchmarks.
 *      conv3d: out[m][co] = sum_tap sum_ci w[tap][co][ci] * cat(xa,xb)[voxel(m,tap)][ci]  (+ optional fused 1x1
 *      conv over cat(x1a,x1b) with w1) + bias + bias2 + temb[n][co] + residual[m][co].  w: [k^3][cout_pad][ca+cb]
 *      bf16, cout_pad % 64 == 0.  stride 2 with pad 0 means F.pad(x,(0,1)*3) then stride-2 conv (AEKLDownsample).
 *      ups = 1 folds a nearest x2 upsample of the input into the loader.  Exactly one of out_bf16 ([M][rup32(cout)])
 *      / out_f32 (NCDHW [N][cout][DHW]) is written.  wgn in {0 auto,1,2,4} picks the tile (BN = 64*wgn), splitk 0 =
 *      auto; scratch holds the fp32 split-K slabs (splitk * M * cout_pad * 4 bytes). ------------------------------ */
int ldm_op_conv3d(const void* xa, int ca, const void* xb, int cb, const void* w, const float* bias,
                  const void* x1a, int c1a, const void* x1b, int c1b, const void* w1, const float* bias2,
                  const float* temb, int temb_stride, const void* residual, void* out_bf16, float* out_f32,
                  int N, int Din, int Hin, int Win, int ksize, int stride, int pad, int ups,
                  int cout, int cout_pad, int wgn, int splitk, void* scratch, size_t scratch_bytes, void* stream);
/* ---- training building blocks (backward of train_diffusion.py:207-219) ----------------
 *      data gradient of a conv = ldm_op_conv3d on dY with flipped + transposed weights (ldm_op_weight_flip_transpose),
 *      pad' = k-1-pad; for a stride-2 forward conv pass ups = 2 (zero-insertion upsample: odd tap positions read 0). */
/*      the same 3^3 stride-1 pad-1 convolution for 64 OUTPUT channels over a large grid (the AutoencoderKL's full-resolution ResBlocks:
 *      3d_ldm/train_autoencoder.py:139-148 construction; Encoder / Decoder blocks at num_channels[0] = 64), conv3_block_kernel: one
 *      (4, th, 16) block of output voxels per workgroup, its input halo copied into LDS once per 32 input channels.  x [N][D][H][W][cin],
 *      w packed [27][64][cin], out [N*D*H*W][64]; stats (optional) [N * rows][64][2], rows = ldm_op_conv3d_block_stats_rows; th = 8 | 4. */
int ldm_op_conv3d_block_stats_rows(int D, int H, int W, int th);
/*      the same for 128 OUTPUT channels (conv3_block128_kernel: eight waves per workgroup, double-buffered halo chunks; the AutoencoderKL's
 *      half-resolution ResBlocks at num_channels[1] = 128): w packed [27][128][cin], out / residual [N*D*H*W][128],
 *      stats [N * ldm_op_conv3d_block_stats_rows(D, H, W, 8)][128][2]. */
int ldm_op_conv3d_block128(const void* x, int cin, const void* w, const float* bias, const float* temb, int temb_stride, const void* residual,
                           void* out, float* stats, int N, int D, int H, int W, void* stream);
/* tests only, experiments builds (make EXTRA=-DLDM_EXPERIMENTS): number of workgroups of conv3_block_kernel's tile loop (0 = default: two per
 * CU; a multiple of 8); returns the previous value.  The product library has no tile-loop form: returns -1. */
int ldm_debug_conv_block_slots(int slots);
int ldm_op_conv3d_block(const void* x, int cin, const void* w, const float* bias, const float* temb, int temb_stride, const void* residual,
                        void* out, float* stats, int N, int D, int H, int W, int th, void* stream);
int ldm_op_weight_flip_transpose(const void* w, void* wt, int ksize, int cout, int cout_pad, int cin, void* stream);
/*      weight gradient: dw[tap][co][ci] (fp32) = sum_m dy[m][co] * x[src(m, tap)][ci]; deterministic (no atomics).
 *      ksplit > 1 splits the voxel range over workgroups: dw = ksplit partial matrices [ksplit][k^3][cout][cin] to sum. */
int ldm_op_conv3d_wgrad(const void* dy, int cdy, const void* x, int cx, float* dw, int cout, int cin,
                        int N, int Din, int Hin, int Win, int ksize, int stride, int pad, int ups, int ksplit, void* stream);
/*      backward of y = act(GroupNorm(cat(xa, xb))): dxa, dxb (bf16 NDHWC, plus acc_a / acc_b when given), dgamma, dbeta (fp32). */
size_t ldm_op_group_norm_bwd_scratch_bytes(int N, int C, int DHW, int groups);
int ldm_op_group_norm_bwd(const void* dy, const void* xa, int ca, const void* xb, int cb, const float* gamma, const float* beta,
                          int groups, float eps, int silu, const void* acc_a, const void* acc_b, void* dxa, void* dxb,
                          float* dgamma, float* dbeta, int N, int DHW, void* scratch, size_t scratch_bytes, void* stream);
/* GroupNorm(groups, eps, affine) over cat(xa, xb), optional fused SiLU -> out [N*DHW][ca+cb] bf16. */
/* ---- data path (3d_ldm/utils.py:94-107): ScaleIntensityRangePercentiles per volume on the device (exact order statistics by radix
 *      select, numpy "linear" interpolation); x / out: B volumes of n fp32 values ------------------------------------------------ */
size_t ldm_op_scale_intensity_percentiles_scratch_bytes(int B);
int ldm_op_scale_intensity_percentiles(const float* x, float* out, int B, int64_t n, float lower, float upper, float b_min, float b_max,
                                       void* scratch, size_t scratch_bytes, void* stream);
/* ---- PatchDiscriminator building blocks (stage-1 GAN tail, 3d_ldm/train_autoencoder.py:150-158,407-424,454-494): 4^3 strided convs as
 *      im2col + the 1x1 GEMM kernels (forward, data gradient through col2im, weight gradient through ldm_op_conv3d_wgrad with ksize 1);
 *      InstanceNorm + LeakyReLU(0.2) = ldm_op_group_norm(_bwd) with groups = C and activation code 2 (0 none, 1 SiLU, 2 LeakyReLU 0.2). */
int ldm_op_im2col(const void* x, void* col, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream);
int ldm_op_col2im(const void* dcol, void* dx, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream);
int ldm_op_leaky_relu(const void* x, void* y, int64_t n, float slope, void* stream);
int ldm_op_leaky_relu_bwd(const void* x, const void* dy, void* dx, int64_t n, float slope, void* stream);
int ldm_op_pack_ncdhw(const float* x, void* out_bf16_ndhwc, int N, int C, int Cs, int64_t DHW, void* stream);
int ldm_op_unpack_ndhwc(const void* act_bf16_ndhwc, float* out, int N, int C, int Cs, int64_t DHW, void* stream);
/* The same PatchDiscriminator building blocks on fp32 NDHWC tensors: the reference trains the discriminator in fp32 when AMP is off
 * (3d_ldm/train_autoencoder.py:150-158 construction, :454-494 discriminator step); `--precision fp32` selects them
 * (ldm3d/discriminator.py).  Exact fp32 MFMA (csrc/f32_path.h, f32_train.h).  K / stored channels are multiples of 16, weight
 * matrices have cout_pad % 64 == 0 rows (rows >= cout zero), bias cout_pad entries.
 *   gemm_f32:        out[M][couts] = x[M][K] w[cout_pad][K]^T + bias          (couts % 4 == 0, cout <= couts <= cout_pad)
 *   gemm_wgrad_f32:  dw[ksplit][cout][K] = partial sums over row ranges of dy[M][cdy]^T x[M][K]
 *   group_norm_f32 / _bwd_f32: y = act(GroupNorm(x)) with act 0 none, 1 SiLU, 2 LeakyReLU(0.2) (InstanceNorm: groups = C), and its
 *                    backward (dx, dgamma, dbeta summed over the batch); scratch from ldm_op_group_norm_f32_scratch_bytes. */
int ldm_op_pack_ncdhw_f32(const float* x, float* out_f32_ndhwc, int N, int C, int Cs, int64_t DHW, void* stream);
int ldm_op_unpack_ndhwc_f32(const float* act_f32_ndhwc, float* out, int N, int C, int Cs, int64_t DHW, void* stream);
int ldm_op_im2col_f32(const float* x, float* col, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream);
int ldm_op_col2im_f32(const float* dcol, float* dx, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream);
int ldm_op_leaky_relu_f32(const float* x, float* y, int64_t n, float slope, void* stream);
int ldm_op_leaky_relu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, float slope, void* stream);
int ldm_op_gemm_f32(const float* x, int K, const float* w, const float* bias, float* out, int64_t M, int cout, int cout_pad, int couts, void* stream);
int ldm_op_gemm_wgrad_f32(const float* dy, int cdy, const float* x, int K, float* dw, int cout, int64_t M, int ksplit, void* stream);
/* The 1x1x1 convolutions of the fp32 inference plans (SABlock's q|k|v and output projections, the ResBlock's nin_shortcut; SURVEY.md
 * section 8a row a2.3) as one launch on fp32 operands: out[M][couts] = (xa | xb)[M][ca + cb] w[cout_pad][ca + cb]^T + bias (+ residual),
 * every product as three bf16 MFMAs on hi / lo splits made in registers (csrc/gemm_light_x3.h; ~1e-5 relative).  ca, cb % 32 == 0
 * (cb = 0: one source), cout_pad, couts % 32 == 0.  stats (optional): [ceil(M / rows)][couts][2] per-tile (sum, sum of squares), rows = 64
 * when big else 32; big = -1: the planner's choice. */
int ldm_op_linear_f32x3(const float* xa, int ca, const float* xb, int cb, const float* w, const float* bias, const float* residual, float* out,
                        float* stats, int64_t M, int cout_pad, int couts, int big, void* stream);
size_t ldm_op_group_norm_f32_scratch_bytes(int N, int C, int DHW, int groups);
int ldm_op_group_norm_f32(const float* x, int C, const float* gamma, const float* beta, int groups, float eps, int act, float* out,
                          int N, int DHW, void* scratch, size_t scratch_bytes, void* stream);
int ldm_op_group_norm_bwd_f32(const float* dy, const float* x, int C, const float* gamma, const float* beta, int groups, float eps, int act,
                              float* dx, float* dgamma, float* dbeta, int N, int DHW, void* scratch, size_t scratch_bytes, void* stream);
/* The split-K conv -> GroupNorm pair as the inference plans launch it at the low-resolution levels (conv + ONE finalize-and-GroupNorm
 * launch, csrc/fin_gn.h; MONAI ResBlock conv1 -> norm2 -> SiLU behind 3d_ldm/train_diffusion.py:197-205): gn_out = GroupNorm(+SiLU) of
 * bf16(conv + bias + temb[n] + residual), conv_out (optional) = that bf16 tensor itself.  splitk >= 2.  Every workgroup of the second launch
 * owns one (sample, group) over all rows (the conv writes its fp32 slabs one plane per group), so no statistics cross workgroups.
 * LDM_ERR_UNSUPPORTED for shapes the plans keep on two launches (channels per group not a power of two in 4 ... 64, > 4096 items per group). */
size_t ldm_op_conv3d_fin_gn_scratch_bytes(int N, int D, int H, int W, int cout_pad, int splitk);
int ldm_op_conv3d_fin_gn(const void* x, int cin, const void* w, const float* bias, const float* temb, int temb_stride, const void* residual,
                         const float* gamma, const float* beta, int groups, float eps, int silu, void* conv_out, void* gn_out,
                         int N, int D, int H, int W, int cout, int cout_pad, int wgn, int splitk, void* scratch, size_t scratch_bytes,
                         void* stream);
/* producer -> GroupNorm pair as the inference plans launch it (conv epilogue / write-through split-K finalize leave the statistics
 * slabs, one-launch GroupNorm(+SiLU) with write-through stores folds them): the per-kernel gate of exactly those kernel variants */
size_t ldm_op_conv3d_gn_scratch_bytes(int N, int D, int H, int W, int cout_pad, int splitk);
int ldm_op_conv3d_gn(const void* x, int cin, const void* w, const float* bias, const float* gamma, const float* beta, int groups, float eps,
                     int silu, void* conv_out, void* gn_out, int N, int D, int H, int W, int cout, int cout_pad, int wgn, int splitk,
                     void* scratch, size_t scratch_bytes, void* stream);
size_t ldm_op_group_norm_scratch_bytes(int N, int C, int DHW);
int ldm_op_group_norm(const void* xa, int ca, const void* xb, int cb, const float* gamma, const float* beta,
                      int groups, float eps, int silu, void* out, int N, int DHW, void* scratch, size_t scratch_bytes,
                      void* stream);
/* softmax(q k^T / 8) v with head_dim 64: qkv [B*N][3C] bf16 (q|k|v, channel = head*64 + d) -> out [B*N][C] bf16. */
int ldm_op_attention(const void* qkv, void* out, int B, int N, int C, void* stream);
/* the same for head_dim = 32 | 64 | 128 | 256 (num_head_channels 32: 3d_ldm/config/config_train_stable.json:45-46; the AutoencoderKL
 * attention blocks are single-head, head_dim = C: config_train_32g.json:21-25); lse (optional) receives the log-sum-exp rows */
int ldm_op_attention_hd(const void* qkv, void* out, float* lse, int B, int N, int C, int head_dim, void* stream);
int ldm_op_attention_bwd_hd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_scratch, void* dqkv,
                            int B, int N, int C, int head_dim, void* stream);
/*      training form (also writes lse [B][C/64][N] fp32) and backward: dqkv [B*N][3C] bf16 from d_o [B*N][C] bf16;
 *      delta_scratch: B * (C/64) * N floats. */
int ldm_op_attention_train(const void* qkv, void* out, float* lse, int B, int N, int C, void* stream);
int ldm_op_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_scratch, void* dqkv,
                         int B, int N, int C, void* stream);

/* ---- measurement support (bench.py): HIP events around every launch of ONE conv tile configuration
 *      (wgm x wgn waves, K depth bk), recorded on the launch stream.  stop() fills
 *      out = {instrumented launches, their total ms, their algorithmic FLOPs, all conv launches, all conv FLOPs}.
 *      plan_conv_cfgs lists the {wgm, wgn, bk | halo << 8, splitk} the planner chose per conv of a plan ("unet"|"enc"|"dec");
 *      halo: 0 = conv_igemm_kernel, 1 / 2 = conv3_halo_kernel (126 x 128 / 254 x 64 tiles), 3 = conv3_block_kernel, 4 = conv3_block128_kernel. -- */
int ldm_profile_start(int wgm, int wgn, int bk, int max_launches);
int ldm_profile_detail(double* flops, double* ms, int max);   /* per instrumented launch; call before ldm_profile_stop; returns their number */
int ldm_profile_stop(double out[5]);
/* Per-op timeline (HIP events around every op of every launch plan that runs while it is on) appended as CSV rows to `path`; NULL or
 * "" = off.  Replaces the torch.profiler window of 3d_ldm/train_autoencoder.py:312-329 (--profile).  Initially: $LDM_PLAN_TRACE. */
int ldm_set_plan_trace(const char* path);
/* diagnostic builds (-DLDM_KSTAMPS) only: in-kernel 100 MHz stamps of the instrumented kernels, [entries][8] in launch order; the
 * product library returns 0 entries */
int ldm_debug_kstamps(unsigned long long* out, int max_entries, int reset);
int ldm_model_plan_conv_cfgs(ldm_model* m, const char* kind, int B, int D, int H, int W, int* cfgs, int max_convs);
/* launches of a cached inference plan ("unet"|"enc"|"dec"; builds it if needed) */
int ldm_model_plan_launches(ldm_model* m, const char* kind, int B, int D, int H, int W);

/* ---- data-parallel collectives (replaces init_process_group("nccl") + DDP all-reduce,
 *      3d_ldm/utils.py:55-63, 3d_ldm/train_diffusion.py:121-123,147-149,281-283): RCCL over xGMI.
 *      unique_id is the 128-byte ncclUniqueId produced by ldm_comm_unique_id on rank 0 and distributed
 *      by the launcher's store.  dtype: 0 = fp32, 1 = bf16; op: 0 = sum, 1 = avg. ------------------------------ */
typedef struct ldm_comm ldm_comm;
int ldm_comm_unique_id(char id[128]);
int ldm_comm_init(int rank, int world, const char id[128], ldm_comm** out);
/* Same communicator object over a caller-supplied transport instead of RCCL: fn must leave the reduction (op) over the `world`
 * ranks in buf, ordered on `stream`; non-zero return = failure.  What the library does around the transport (bucket ranges, issue
 * points, streams, op = avg, the join in front of the optimizer) is identical, which is what lets tests/test_gpu_comm.py play the
 * second rank of a 2-rank job on one GPU.  No reference counterpart (test seam). */
typedef int (*ldm_allreduce_fn)(void* user, void* buf, int64_t count, int dtype, int op, void* stream);
int ldm_comm_init_custom(int rank, int world, ldm_allreduce_fn fn, void* user, ldm_comm** out);
int ldm_comm_allreduce(ldm_comm* c, void* buf, int64_t count, int dtype, int op, void* stream);
int ldm_comm_broadcast(ldm_comm* c, void* buf, int64_t count, int dtype, int root, void* stream);
int ldm_comm_barrier(ldm_comm* c, void* stream);
void ldm_comm_destroy(ldm_comm* c);
int ldm_comm_rank(const ldm_comm* c);
int ldm_comm_world(const ldm_comm* c);
/* Bucketed gradient all-reduce overlapped with backward (replaces DistributedDataParallel's bucket hooks,
 * 3d_ldm/train_diffusion.py:147-149): while a communicator is attached, ldm_unet_train_backward / ldm_vae_train_backward average the
 * flat gradient buffer over the ranks, bucket by bucket (LDM_GRAD_BUCKET_MB, default 48), each collective queued on the
 * communicator's stream as soon as backward has finished that tail range of the buffer; `stream` waits for the last one.
 * ldm_model_grad_sync_trace returns the bucket timeline of the last backward call (n buckets; entry n = end of the call). */
int ldm_model_set_grad_sync(ldm_model* m, ldm_comm* comm);
int ldm_model_grad_sync_trace(ldm_model* m, double* issue_ms, double* done_ms, int64_t* elems, int max);
/* Wire dtype of the bucketed exchange: 0 = fp32 (default, what DDP reduces: 3d_ldm/train_diffusion.py:147-149), 1 = bf16 (each
 * bucket cast into a staging slice on the communicator's stream, all-reduced (avg) as bf16, cast back: half the bytes on xGMI;
 * SURVEY.md section 8a row a7 "382 MB if bf16 grads").  ldm_model_grad_sync_pending: buckets issued that no join has covered yet
 * (0 = nothing of this communicator is in flight ahead of the launch stream; host bookkeeping, no synchronisation). */
int ldm_model_set_grad_wire(ldm_model* m, int dtype);
int ldm_model_grad_sync_pending(const ldm_model* m);
/* Evidence for bench records / tests: {all-reduce calls, all-reduce bytes in the wire dtype, broadcast calls, broadcast bytes} as
 * handed to the transport; whether that transport is RCCL; ncclGetVersion() of the loaded librccl (e.g. 22606; negative = status). */
int ldm_comm_stats(const ldm_comm* c, int64_t out[4]);
int ldm_comm_is_rccl(const ldm_comm* c);
int ldm_comm_rccl_version(void);
/* The exchange schedule of the training plan for this shape, built on the host (no GPU needed): events in launch order with
 * kind 0 = an op leaves final values in flat_grads[lo, lo + n), 1 = bucket [lo, lo + n) handed to the communicator, 2 = join,
 * 3 = unclassified write; op = index of the launch-plan op.  Returns the event count (only `max` are written). */
int ldm_model_grad_schedule(ldm_model* m, int B, int D, int H, int W, int* kind, int64_t* lo, int64_t* n, int* op, int max);

#ifdef __cplusplus
}
#endif
#endif /* LDM3D_H */
