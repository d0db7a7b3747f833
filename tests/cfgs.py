"""Model definitions used across tests (shapes from 3d_ldm/config/config_train_16g.json, BASELINE.json)."""
# benchmark UNet: diffusion_def of config_train_16g.json:39-48 with in = out = 4 latent channels (BASELINE.json)
UNET_FULL = dict(spatial_dims=3, in_channels=4, out_channels=4, channels=[256, 256, 512],
                 attention_levels=[False, True, True], num_head_channels=[0, 64, 64], num_res_blocks=2)
# same topology, small enough for the CPU oracle to run in a second
UNET_TINY = dict(spatial_dims=3, in_channels=4, out_channels=4, channels=[64, 64, 128],
                 attention_levels=[False, True, True], num_head_channels=[0, 64, 64], num_res_blocks=2,
                 norm_num_groups=32)
# concat-conditioned variant (train_diffusion.py:197-205 mode="concat"): in = 2 x latent
UNET_TINY_COND = dict(UNET_TINY, in_channels=8)
# odd level structure: per-level res blocks, attention at level 0, 2 levels
UNET_TINY_ALT = dict(spatial_dims=3, in_channels=3, out_channels=2, channels=[64, 128],
                     attention_levels=[True, False], num_head_channels=64, num_res_blocks=[1, 2],
                     norm_num_groups=16)
# autoencoder_def of config_train_16g.json:7-28 with 1 image channel and 4 latent channels (BASELINE.json config 2)
VAE_FULL = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, channels=[64, 128, 256],
                num_res_blocks=2, norm_num_groups=32, norm_eps=1e-6, attention_levels=[False, False, False],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)
VAE_TINY = dict(spatial_dims=3, in_channels=2, out_channels=2, latent_channels=8, channels=[32, 64, 64],
                num_res_blocks=[1, 2, 1], norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False, False],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)
SCHED = dict(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
# AutoencoderKL with attention: the structure of autoencoder_def in config_train_32g.json:7-28 / config_train_multigpu.json:21-27
# (attention at the last level + non-local attention in encoder and decoder, single head d = C), reduced channels
VAE_TINY_ATTN = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, channels=[32, 64, 64],
                     num_res_blocks=[1, 2, 2], norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False, True],
                     with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True)
# d = 128 (two waves per workgroup in the fp32 attention backward)
VAE_MID_ATTN = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, channels=[32, 64, 128],
                    num_res_blocks=[1, 1, 1], norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False, True],
                    with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=False)
# the full-width one: channels [64, 128, 256], single-head attention at 256 channels (d = 256)
VAE_FULL_ATTN = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=16, channels=[64, 128, 256],
                     num_res_blocks=[1, 2, 2], norm_num_groups=32, norm_eps=1e-6, attention_levels=[False, False, True],
                     with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True)
# num_head_channels 32 (diffusion_def of config_train_stable.json:39-48), reduced channels
UNET_TINY_HEAD32 = dict(spatial_dims=3, in_channels=4, out_channels=4, channels=[64, 64, 128],
                        attention_levels=[False, False, True], num_head_channels=[0, 0, 32], num_res_blocks=2, norm_num_groups=32)
# 96 channels in 32 groups: THREE channels per group (the odd-group paths of the GroupNorm folds), 3 heads of 32
UNET_TINY_ODD = dict(spatial_dims=3, in_channels=4, out_channels=4, channels=[96, 96],
                     attention_levels=[False, True], num_head_channels=[0, 32], num_res_blocks=1, norm_num_groups=32)

# kwargs of autoencoder_def / diffusion_def of the five shipped reference configs (3d_ldm/config/*.json, "@" / "$@" references
# resolved) with the patch sizes of their autoencoder_train / diffusion_train sections: the shapes the reference's own entry scripts run
REF_CONFIGS = {
    "config_train_16g": dict(
        ae=dict(spatial_dims=3, in_channels=2, out_channels=2, latent_channels=8, channels=[64, 128, 256], num_res_blocks=2,
                norm_num_groups=32, norm_eps=1e-06, attention_levels=[False, False, False],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False),
        unet=dict(spatial_dims=3, in_channels=8, out_channels=8, channels=[256, 256, 512], attention_levels=[False, True, True],
                  num_head_channels=[0, 64, 64], num_res_blocks=2),
        ae_patch=(64, 64, 64), unet_patch=(144, 176, 112)),
    "config_train_32g": dict(
        ae=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=16, channels=[64, 128, 256], num_res_blocks=2,
                norm_num_groups=32, norm_eps=1e-06, attention_levels=[False, False, True],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False),
        unet=dict(spatial_dims=3, in_channels=32, out_channels=16, channels=[256, 512, 1024], attention_levels=[False, True, True],
                  num_head_channels=[0, 64, 64], num_res_blocks=2),
        ae_patch=(64, 64, 64), unet_patch=(80, 80, 80)),
    "config_train_multigpu": dict(
        ae=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=16, channels=[64, 128, 256], num_res_blocks=2,
                norm_num_groups=32, norm_eps=1e-06, attention_levels=[False, True, True],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=True),
        unet=dict(spatial_dims=3, in_channels=32, out_channels=16, channels=[256, 512, 1024], attention_levels=[False, True, True],
                  num_head_channels=[0, 64, 64], num_res_blocks=2),
        ae_patch=(64, 64, 64), unet_patch=(64, 64, 64)),
    "config_train_stable": dict(
        ae=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, channels=[32, 64, 128], num_res_blocks=1,
                norm_num_groups=16, norm_eps=1e-06, attention_levels=[False, False, True],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False),
        unet=dict(spatial_dims=3, in_channels=16, out_channels=8, channels=[128, 256, 512], attention_levels=[False, False, True],
                  num_head_channels=[0, 0, 32], num_res_blocks=1),
        ae_patch=(48, 48, 48), unet_patch=(48, 48, 48)),
    "config_optimized": dict(
        ae=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=16, channels=[128, 256, 256], num_res_blocks=2,
                norm_num_groups=32, norm_eps=1e-06, attention_levels=[False, True, True],
                with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=True),
        unet=None, ae_patch=(64, 64, 64), unet_patch=None),
}
