"""MI355X-native (gfx950) 3D latent-diffusion denoising path: host-side mirror of the reference's interface.

Everything numerical runs in ``libldm3d.so`` (hand-written HIP, C ABI in include/ldm3d.h); this package only
adapts it to the objects the reference's entry scripts use: ``define_instance`` (3d_ldm/utils.py:243-246),
``DiffusionModelUNet`` / ``AutoencoderKL`` nn.Modules, ``DDPMScheduler`` / ``DDIMScheduler`` and
``LatentDiffusionInferer`` (3d_ldm/train_diffusion.py:11-12,140-152; 3d_ldm/inference.py:23-24,79-85).
There is no CPU fallback: without the built library or without a GPU the compute calls raise.
"""
__version__ = "0.1.0"
