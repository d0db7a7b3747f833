"""CPU oracle for the 3D latent-diffusion denoising path.  TEST INFRASTRUCTURE ONLY.

This package is a plain ``torch.nn.functional`` (fp32, NCDHW) restatement of the
arithmetic the reference delegates to MONAI >= 1.4 (not vendored under
/root/reference, not installed, not installable: no network):

    monai.networks.nets.DiffusionModelUNet      -> oracle.unet
    monai.networks.nets.AutoencoderKL           -> oracle.autoencoder
    monai.networks.schedulers.DDPM/DDIMScheduler-> oracle.schedulers
    monai.inferers.LatentDiffusionInferer       -> oracle.inferer

and of the reference's own glue around them (3d_ldm/utils.py:243-262,
3d_ldm/train_diffusion.py:172-223, 3d_ldm/inference.py:79-99).

PARITY UNPINNED: the reference ships no golden vectors, fixtures or asserting
tests for this path (3d_ldm/test_losses.py:11-86 prints only) and MONAI cannot
be imported here, so the oracle is pinned by closed-form known-answer tests only
(tests/test_oracle_known_answers.py) - see DESIGN.md "Oracle".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (3d-latent-diffusion-model_amd/, networks/) never
does; it has no CPU fallback and fails loudly without the HIP library.

``emulate_bf16=False`` is the pure fp32 reference: the reference's own
arithmetic (autocast off, 3d_ldm/train_diffusion.py:177) and what the library's
fp32 precision mode (ldm_model_set_precision) is gated against at 1e-3 rel-L2.
``emulate_bf16=True`` reproduces the rounding points of the bf16 HIP path (bf16
weights, bf16 activation storage, fp32 accumulation); a bf16 network of this
depth is chaotic under rounding, so that emulation only measures the bf16 noise
floor (~3e-2 at the benchmark shape) that the bf16 path is gated against, and
the bf16 path is pinned tightly per block by the teacher-forced tap tests
(tests/test_gpu_taps.py).
"""
