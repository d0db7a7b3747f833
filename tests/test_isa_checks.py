"""Build-time ISA checks of the hand-managed hazards (CPU: hipcc cross-compiles the device code, no GPU needed).

csrc/common.h store16<true> issues `global_store_dwordx4 ... sc1` through inline asm, which hides it from hipcc's hazard
recognizer: a > 64-bit VMEM store reads its data VGPRs over several cycles and a VALU write to them in the next wait states
corrupted lanes 12-15 of each 16-lane group in round 1.  The asm therefore carries its own `s_nop 1`; this
test fails if a recompile ever separates the two, and if the hot kernels start spilling registers to scratch."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-latent-diffusion-model_amd", "csrc")


@pytest.fixture(scope="module")
def isa():
    asm, res = os.path.join(CSRC, "ldm3d.s"), os.path.join(CSRC, "resource_usage.txt")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    if not (os.path.exists(asm) and os.path.exists(res)) or os.path.getmtime(asm) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", CSRC, "asm"], check=True, capture_output=True, timeout=900)
    return open(asm).read().splitlines(), open(res).read()


def test_write_through_stores_keep_their_wait_states(isa):
    lines, _ = isa
    idx = [i for i, l in enumerate(lines) if "global_store_dwordx4" in l and " sc1" in l and " sc0" not in l]
    assert len(idx) >= 4                                   # gn_fused_apply_kernel<true>, splitk_finalize_kernel<true>
    for i in idx:
        nxt = next(l.strip() for l in lines[i + 1:] if l.strip() and not l.strip().startswith((";", "//", ".")))
        assert re.match(r"s_nop\s+[1-9]", nxt), (lines[i].strip(), nxt)


def test_hot_kernels_do_not_spill(isa):
    _, res = isa
    blocks = re.split(r"remark: [^\n]*Function Name: ", res)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        hot = ("conv3_halo_kernelILi6ELi0E" in name or "conv_igemm_kernel" in name or "gemm_light_kernel" in name or "conv_f32_kernel" in name or
               "attn_fwd_kernelILi1E" in name or "attn_fwd_kernelILi4E" in name or "conv_wgrad_kernelILi0E" in name or "conv_wgrad_w16_kernelILi0E" in name or
               "conv3_block_kernelILi8ELi0ELb0E" in name or "gemm_light_x3_kernel" in name)     # the one-tile block conv (the plans' form); its tile-loop form is an off-by-default option
        if not hot:
            continue
        seen += 1
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
        if re.search(r"conv3_halo_kernelILi6ELi0ELb[01]ELb1ELb[01]EEv", name):     # <NSB, ABL, TALL, PERSIST = true, MASK_INLINE>
            continue                                       # the persistent (tile-loop) instantiations: checked loop by loop below
        assert m and int(m.group(1)) == 0, (name, m and m.group(1))
    assert seen >= 12


def test_persistent_halo_kernels_keep_scratch_out_of_their_k_loops(isa):
    """conv3_halo_kernel<6, 0, *, true> walks several tiles per workgroup (launches with more tiles than CUs: the AutoencoderKL's 96^3 /
    48^3 levels); the tile loop keeps more values live and the compiler parks some of them in scratch BETWEEN K loops (prologue, K-group
    exchange, epilogue).  That is accepted; a scratch access inside a K loop is not: every loop of the kernel that consists of K steps
    (16 MFMAs per step; 96 in the six-step steady block, 48 / 96 in the tails, 16 in the fused-skip loop) must be free of them.  The
    one-tile-per-workgroup instantiations (every conv of the B = 1 UNet step) must not touch scratch at all (test above)."""
    lines, res = isa
    for tall in "01":
        start = next(i for i, l in enumerate(lines) if l.startswith(f"_Z17conv3_halo_kernelILi6ELi0ELb{tall}ELb1ELb0EEv10ConvParams:"))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        k_loops = 0
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
                continue
            seg = body[labels[m.group(1)]:i]
            mfma = sum("v_mfma" in x for x in seg)
            if 0 < mfma <= 96:
                k_loops += 1
                assert not any("scratch_" in x for x in seg), (tall, labels[m.group(1)], i)
        assert k_loops >= 3, (tall, k_loops)
