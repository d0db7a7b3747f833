"""Per-kernel parity (-m gpu): every HIP kernel, called through the C ABI (ldm_op_*), against plain torch CPU fp32
ops evaluated on the same bf16-rounded inputs.  Tolerance: fp32 accumulation-order noise plus one output rounding
to bf16 -> rel-L2 <= 3e-3 against the UN-rounded fp32 result (a bf16 rounding alone is ~1.7e-3 rms... see below),
and <= 2e-4 against the result rounded to bf16 the same way (the gate that matters: same rounding points)."""
import ctypes as C
import itertools

import pytest
import torch
import torch.nn.functional as F

from util import bf16_round, from_ndhwc, pack_conv_weight, pad_vec, rel_l2, rup, to_ndhwc_bf16

pytestmark = pytest.mark.gpu
TOL_SAME_ROUNDING = 3e-4     # identical rounding points; remaining difference = fp32 summation order + rare bf16 ties
TOL_FP32_OUT = 2e-5          # fp32 outputs (no output rounding)


def _conv_case(cuda, lib, *, n=1, cin=(64, 0), cout=64, dims=(8, 8, 8), k=3, stride=1, pad=1, ups=0, wgn=0, splitk=0,
               skip=None, temb=False, residual=False, f32_out=False, seed=0):
    from ldm3d import _lib
    g = torch.Generator().manual_seed(seed)
    ca, cb = cin
    c = ca + cb
    cas, cbs = rup(ca, 32), (rup(cb, 32) if cb else 0)
    assert cb == 0 or ca == cas, "dual source needs an unpadded first source"
    x = torch.randn((n, c, *dims), generator=g)
    w = torch.randn((cout, c, k, k, k), generator=g) / (c * k ** 3) ** 0.5
    b = 0.1 * torch.randn((cout,), generator=g)
    xr, wr = bf16_round(x), bf16_round(w)
    xin = F.interpolate(xr, scale_factor=2.0, mode="nearest") if ups else xr
    if stride == 2 and pad == 0:
        xin = F.pad(xin, (0, 1, 0, 1, 0, 1))
    ref = F.conv3d(xin, wr, b, stride=stride, padding=pad)
    cout_pad = rup(cout, 64)
    args = dict(x1a=None, c1a=0, x1b=None, c1b=0, w1=None, bias2=None)
    keep = []
    if skip is not None:                      # fused 1x1 conv over a (dual-source) tensor at output resolution
        s_a, s_b = skip
        xs = torch.randn((n, s_a + s_b, *ref.shape[2:]), generator=g)
        ws = torch.randn((cout, s_a + s_b, 1, 1, 1), generator=g) / (s_a + s_b) ** 0.5
        bs = 0.1 * torch.randn((cout,), generator=g)
        ref = ref + F.conv3d(bf16_round(xs), bf16_round(ws), bs)
        x1a = to_ndhwc_bf16(xs[:, :s_a]).to(cuda)
        x1b = to_ndhwc_bf16(xs[:, s_a:]).to(cuda) if s_b else None
        w1 = pack_conv_weight(ws, s_a + s_b, cout_pad).to(cuda)
        b2 = pad_vec(bs, cout_pad).to(cuda)
        keep += [x1a, x1b, w1, b2]
        args = dict(x1a=x1a.data_ptr(), c1a=s_a, x1b=None if x1b is None else x1b.data_ptr(), c1b=s_b, w1=w1.data_ptr(),
                    bias2=b2.data_ptr())
    te = None
    if temb:
        tv = torch.randn((n, cout_pad), generator=g)
        ref = ref + tv[:, :cout, None, None, None]
        te = tv.to(cuda)
    res = None
    if residual:
        rv = bf16_round(torch.randn(ref.shape, generator=g))
        ref = ref + rv
        res = to_ndhwc_bf16(rv).to(cuda)
    xa = to_ndhwc_bf16(x[:, :ca]).to(cuda)
    xb = to_ndhwc_bf16(x[:, ca:]).to(cuda) if cb else None
    wp = pack_conv_weight(w, cas + cbs, cout_pad).to(cuda)
    bp = pad_vec(b, cout_pad).to(cuda)
    do, ho, wo = ref.shape[2:]
    m = n * do * ho * wo
    couts = rup(cout, 32)
    out_bf = torch.empty((n, do, ho, wo, couts), dtype=torch.bfloat16, device=cuda)
    out_f = torch.empty((n, cout, do, ho, wo), dtype=torch.float32, device=cuda)
    scratch = torch.empty((max(1, splitk or 64) * m * cout_pad * 4 + 256,), dtype=torch.uint8, device=cuda)
    st = lib.ldm_op_conv3d(xa.data_ptr(), cas, None if xb is None else xb.data_ptr(), cbs, wp.data_ptr(), bp.data_ptr(),
                           args["x1a"], args["c1a"], args["x1b"], args["c1b"], args["w1"], args["bias2"],
                           None if te is None else te.data_ptr(), cout_pad, None if res is None else res.data_ptr(),
                           None if f32_out else out_bf.data_ptr(), out_f.data_ptr() if f32_out else None,
                           n, *dims, k, stride, pad, ups, cout, cout_pad, wgn, splitk, scratch.data_ptr(), scratch.numel(),
                           torch.cuda.current_stream().cuda_stream)
    _lib.check(st)
    torch.cuda.synchronize()
    if f32_out:
        return rel_l2(out_f.cpu(), ref), TOL_FP32_OUT
    got = from_ndhwc(out_bf.cpu(), cout)
    if couts > cout:
        assert float(out_bf[..., cout:].float().abs().max()) == 0.0, "channel padding must be written as zeros"
    return rel_l2(got, bf16_round(ref)), TOL_SAME_ROUNDING


@pytest.mark.parametrize("wgn,splitk", [(1, 1), (2, 1), (4, 1), (2, 3), (1, 9), (4, 4)])
def test_conv3_tiles_and_splitk(cuda, built_lib, wgn, splitk):
    err, tol = _conv_case(cuda, built_lib, cin=(256, 0), cout=256, dims=(6, 6, 6), wgn=wgn, splitk=splitk)
    assert err <= tol, (wgn, splitk, err)


@pytest.mark.parametrize("cin,cout", [((4, 0), 64), ((32, 0), 4), ((96, 0), 32), ((64, 0), 1), ((128, 64), 128), ((64, 96), 64)])
def test_conv3_channel_shapes(cuda, built_lib, cin, cout):
    """Tiny / padded channel counts, BK=32 path, dual-source concat."""
    err, tol = _conv_case(cuda, built_lib, cin=cin, cout=cout, dims=(5, 7, 6), n=2)
    assert err <= tol, (cin, cout, err)


def test_conv3_ragged_edges(cuda, built_lib):
    """M not a multiple of any tile, batch 3, non-cubic volume."""
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=96, dims=(3, 5, 7), n=3)
    assert err <= tol, err


def test_conv3_stride2_pad1(cuda, built_lib):
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=64, dims=(8, 8, 8), stride=2, pad=1)
    assert err <= tol, err


def test_conv3_stride2_asym_pad(cuda, built_lib):
    """AEKLDownsample: F.pad (0,1) per dim then stride-2 conv without padding."""
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=64, dims=(8, 6, 10), stride=2, pad=0)
    assert err <= tol, err


def test_conv3_fused_nearest_upsample(cuda, built_lib):
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=64, dims=(4, 5, 3), ups=1)
    assert err <= tol, err


def test_conv1x1(cuda, built_lib):
    err, tol = _conv_case(cuda, built_lib, cin=(128, 0), cout=384, dims=(6, 6, 6), k=1, pad=0)
    assert err <= tol, err


def test_conv_epilogue_temb_residual(cuda, built_lib):
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=64, dims=(6, 6, 6), n=2, temb=True, residual=True)
    assert err <= tol, err


@pytest.mark.parametrize("splitk", [1, 4])
def test_conv_fused_skip_1x1_dual_source(cuda, built_lib, splitk):
    """ResBlock conv2 with the 1x1 skip over cat(h, skip) appended as extra K steps."""
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=64, dims=(6, 6, 6), skip=(64, 32), splitk=splitk)
    assert err <= tol, err


@pytest.mark.parametrize("cin,cout,dims,n,skip,temb,splitk", [
    (256, 256, (24, 24, 24), 1, (256, 256), False, 1),    # up-block conv2 at 24^3: 1x1 skip over cat(h, skip) = 8 extra K steps, two sources
    (128, 128, (6, 6, 6), 1, (192, 0), True, 1),          # one source, 3 extra steps, ragged last tile, per-sample channel bias
    (64, 128, (5, 3, 7), 3, (64, 64), False, 1),          # several samples, tiles that end inside a sample, one chunk per source
    (64, 256, (4, 4, 4), 2, (64, 0), False, 1),           # a single extra step (no prefetch ahead)
    # split over K (round 5): the splits share the skip's steps -- the 12^3 / 6^3 ResBlocks with a channel change
    (256, 256, (12, 12, 12), 1, (256, 512), True, 9),     # up_blocks.1 conv2: 12 skip steps over 9 splits (2 each, the last three splits none)
    (512, 512, (6, 6, 6), 1, (512, 512), False, 24),      # up_blocks.0 conv2: 16 skip steps over 24 splits (one each, eight splits none)
    (512, 512, (6, 6, 6), 1, (256, 0), True, 12),         # down_blocks.2 conv2: 4 steps, one source, 12 splits
    (128, 128, (5, 3, 7), 2, (64, 192), False, 3),        # two samples, 4 steps over 3 splits (2 / 2 / 0), chunks from both sources inside one split
])
def test_conv3_halo_kernel_with_the_fused_1x1_skip(cuda, built_lib, cin, cout, dims, n, skip, temb, splitk):
    """conv3_halo_kernel with the ResBlock's 1x1 skip_connection fused as a second K loop at the centre tap (wgn = 2 on an eligible
    conv): MONAI's `skip_connection(x) + conv2(h)` in one launch, as the inference plans run the up-block conv2s at 24^3 (unsplit) and
    at 12^3 / 6^3 (split over K: every split takes a share of the skip's steps)."""
    err, tol = _conv_case(cuda, built_lib, cin=(cin, 0), cout=cout, dims=dims, n=n, wgn=2, splitk=splitk, skip=skip, temb=temb, seed=cin + dims[2] + skip[0])
    assert err <= tol, err


@pytest.mark.parametrize("splitk", [1, 3])
def test_conv_f32_ncdhw_output(cuda, built_lib, splitk):
    err, tol = _conv_case(cuda, built_lib, cin=(64, 0), cout=4, dims=(6, 6, 6), n=2, f32_out=True, splitk=splitk)
    assert err <= tol, err


def test_conv_big_level0_shape(cuda, built_lib):
    """The dominant shape of the benchmark UNet: 256 -> 256 @ 24^3."""
    err, tol = _conv_case(cuda, built_lib, cin=(256, 0), cout=256, dims=(24, 24, 24))
    assert err <= tol, err


@pytest.mark.parametrize("cin,cout,dims,n,splitk", [
    (256, 256, (6, 6, 6), 1, 1), (256, 256, (6, 6, 6), 1, 4), (64, 128, (5, 3, 7), 3, 1), (128, 96, (4, 9, 2), 2, 3),
    (64, 128, (1, 1, 5), 1, 1), (512, 128, (6, 6, 6), 1, 9), (64, 256, (12, 12, 12), 1, 1), (128, 128, (2, 2, 130), 1, 2),
    (64, 128, (3, 3, 1), 2, 1)])
def test_conv3_halo_kernel_shapes(cuda, built_lib, cin, cout, dims, n, splitk):
    """conv3_halo_kernel (wgn = 2 on an eligible conv): W-border masks for narrow / wide lines, tiles that end inside a
    sample, several samples, K splits of whole (kd, kh, chunk) macro steps including single-macro-step ranges."""
    err, tol = _conv_case(cuda, built_lib, cin=(cin, 0), cout=cout, dims=dims, n=n, wgn=2, splitk=splitk, seed=cin + dims[2])
    assert err <= tol, err


@pytest.mark.parametrize("f32_out", [False, True])
def test_conv3_halo_kernel_epilogues(cuda, built_lib, f32_out):
    err, tol = _conv_case(cuda, built_lib, cin=(128, 0), cout=128 if not f32_out else 100, dims=(6, 5, 7), n=2, wgn=2,
                          temb=not f32_out, residual=not f32_out, f32_out=f32_out)
    assert err <= tol, err


@pytest.mark.parametrize("c,groups,dual,silu", [(64, 32, 0, 1), (256, 32, 0, 1), (768, 32, 512, 1), (96, 8, 32, 0), (1024, 32, 512, 1)])
def test_group_norm_silu(cuda, built_lib, c, groups, dual, silu):
    from ldm3d import _lib
    g = torch.Generator().manual_seed(c)
    n, dims = 2, (5, 6, 7)
    x = bf16_round(torch.randn((n, c, *dims), generator=g) * 1.5 + 0.3)
    gamma = 1 + 0.1 * torch.randn((c,), generator=g)
    beta = 0.1 * torch.randn((c,), generator=g)
    ref = F.group_norm(x, groups, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    ca = dual if dual else c
    xa = to_ndhwc_bf16(x[:, :ca], ca).to(cuda)
    xb = to_ndhwc_bf16(x[:, ca:], c - ca).to(cuda) if dual else None
    dhw = dims[0] * dims[1] * dims[2]
    out = torch.empty((n, *dims, c), dtype=torch.bfloat16, device=cuda)
    sb = built_lib.ldm_op_group_norm_scratch_bytes(n, c, dhw)
    scratch = torch.empty((sb,), dtype=torch.uint8, device=cuda)
    gd, bd = gamma.to(cuda), beta.to(cuda)
    _lib.check(built_lib.ldm_op_group_norm(xa.data_ptr(), ca, None if xb is None else xb.data_ptr(), c - ca, gd.data_ptr(),
                                           bd.data_ptr(), groups, 1e-6, silu, out.data_ptr(), n, dhw, scratch.data_ptr(),
                                           scratch.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    err = rel_l2(from_ndhwc(out.cpu(), c), bf16_round(ref))
    assert err <= TOL_SAME_ROUNDING, err


@pytest.mark.parametrize("b,n,c", [(1, 216, 512), (2, 64, 64), (1, 1728, 256), (1, 100, 128)])
def test_attention(cuda, built_lib, b, n, c):
    """softmax(q k^T / 8) v, head_dim 64; compared with fp32 softmax attention on the same bf16 q, k, v."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(n)
    qkv = torch.randn((b, n, 3 * c), generator=g)
    qkv[..., :2 * c] *= 1.7                       # peaky-ish scores
    qkv = bf16_round(qkv)
    h = c // 64

    def split(z):
        return z.reshape(b, n, h, 64).permute(0, 2, 1, 3)
    q, k, v = split(qkv[..., :c]), split(qkv[..., c:2 * c]), split(qkv[..., 2 * c:])
    a = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (a @ v).permute(0, 2, 1, 3).reshape(b, n, c)
    dq = qkv.to(torch.bfloat16).to(cuda)
    out = torch.empty((b, n, c), dtype=torch.bfloat16, device=cuda)
    _lib.check(built_lib.ldm_op_attention(dq.data_ptr(), out.data_ptr(), b, n, c, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    err = rel_l2(out.float().cpu(), ref)
    # P is rounded to bf16 before P.V and the output to bf16: ~2^-9 relative each
    assert err <= 8e-3, err


# ------------------------------------------------------------------------------------------------ training blocks
def _dgrad_case(cuda, lib, cin, cout, dims, k, stride, pad, seed=0):
    """dX of y = conv3d(x, w, stride, pad) for a given dY, as one more conv launch on flipped/transposed weights."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((1, cin, *dims), generator=g, requires_grad=True)
    w = bf16_round(torch.randn((cout, cin, k, k, k), generator=g) / (cin * k ** 3) ** 0.5)
    y = F.conv3d(x, w, None, stride=stride, padding=pad)
    dy = bf16_round(torch.randn(y.shape, generator=g))
    (ref,) = torch.autograd.grad(y, x, dy)
    cout_pad, cin_pad = rup(cout, 64), rup(cin, 64)
    wp = pack_conv_weight(w, cin, cout_pad).to(cuda)                       # forward arena layout [taps][cout_pad][cin]
    wt = torch.empty((k ** 3, cin_pad, rup(cout, 32)), dtype=torch.bfloat16, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ldm_op_weight_flip_transpose(wp.data_ptr(), wt.data_ptr(), k, cout, cout_pad, cin, st))
    dyd = to_ndhwc_bf16(dy).to(cuda)                                       # channels padded to round32(cout)
    out = torch.empty((1, *dims, rup(cin, 32)), dtype=torch.bfloat16, device=cuda)
    scratch = torch.empty((64 << 20,), dtype=torch.uint8, device=cuda)
    zero_bias = torch.zeros((cin_pad,), device=cuda)
    _lib.check(lib.ldm_op_conv3d(dyd.data_ptr(), rup(cout, 32), None, 0, wt.data_ptr(), zero_bias.data_ptr(), None, 0, None, 0, None,
                                 None, None, 0, None, out.data_ptr(), None, 1, *y.shape[2:], k, 1, k - 1 - pad,
                                 2 if stride == 2 else 0, cin, cin_pad, 0, 0, scratch.data_ptr(), scratch.numel(), st))
    torch.cuda.synchronize()
    return rel_l2(from_ndhwc(out.cpu(), cin), bf16_round(ref))


@pytest.mark.parametrize("cin,cout,dims,k,stride,pad", [(64, 64, (6, 6, 6), 3, 1, 1), (128, 64, (5, 7, 6), 3, 1, 1),
                                                        (64, 128, (6, 6, 6), 1, 1, 0), (64, 64, (8, 8, 8), 3, 2, 1),
                                                        (32, 96, (12, 8, 4), 3, 2, 1)])
def test_conv_dgrad(cuda, built_lib, cin, cout, dims, k, stride, pad):
    err = _dgrad_case(cuda, built_lib, cin, cout, dims, k, stride, pad)
    assert err <= TOL_SAME_ROUNDING, err


def _wgrad_case(cuda, lib, cin, cout, dims, k, stride, pad, n=1, ups=0, seed=0, ksplit=1):
    from ldm3d import _lib
    g = torch.Generator().manual_seed(seed)
    x = bf16_round(torch.randn((n, cin, *dims), generator=g))
    w = torch.randn((cout, cin, k, k, k), generator=g, requires_grad=True)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv3d(xin, w, None, stride=stride, padding=pad)
    dy = bf16_round(torch.randn(y.shape, generator=g))
    (ref,) = torch.autograd.grad(y, w, dy)                                  # [cout][cin][k][k][k]
    ref = ref.reshape(cout, cin, k ** 3).permute(2, 0, 1).contiguous()       # -> [taps][cout][cin]
    xd, dyd = to_ndhwc_bf16(x).to(cuda), to_ndhwc_bf16(dy).to(cuda)
    dw = torch.full((ksplit, k ** 3, cout, cin), float("nan"), dtype=torch.float32, device=cuda)
    _lib.check(lib.ldm_op_conv3d_wgrad(dyd.data_ptr(), dyd.shape[-1], xd.data_ptr(), xd.shape[-1], dw.data_ptr(), cout, cin,
                                       n, *dims, k, stride, pad, ups, ksplit, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    return rel_l2(dw.sum(0).cpu(), ref)


@pytest.mark.parametrize("cin,cout,dims,k,stride,pad,n,ups", [
    (64, 64, (6, 6, 6), 3, 1, 1, 1, 0), (128, 256, (5, 7, 6), 3, 1, 1, 2, 0), (256, 128, (8, 8, 8), 1, 1, 0, 1, 0),
    (64, 64, (8, 8, 8), 3, 2, 1, 1, 0), (4, 64, (8, 8, 8), 3, 1, 1, 1, 0), (64, 4, (6, 6, 6), 3, 1, 1, 1, 0),
    (64, 64, (4, 4, 4), 3, 1, 1, 1, 1), (96, 160, (6, 6, 6), 3, 1, 1, 1, 0), (32, 160, (5, 7, 6), 3, 1, 1, 2, 0),
    (128, 64, (6, 5, 7), 3, 1, 1, 2, 0), (160, 40, (4, 6, 9), 3, 1, 1, 1, 0),
    (128, 160, (8, 8, 8), 3, 2, 1, 1, 0), (128, 128, (4, 4, 4), 3, 1, 1, 1, 1), (192, 128, (6, 6, 6), 1, 1, 0, 2, 0)])
def test_conv_wgrad(cuda, built_lib, cin, cout, dims, k, stride, pad, n, ups):
    """fp32 output, fp32 accumulation over voxels: only summation order separates it from autograd.  The 3^3 cases with Cin <= 64 run
    the forms with several taps per workgroup (WgradParams::pair: 1 = taps (2 t, 2 t + 1) share a tile, the 27th tap's partner is empty;
    2 = Cout <= 64 too: all three kw of a (kd, kh); 3 = Cout <= 64 < Cin: kw pairs through the shifted dY rows alone).  Both channel counts above
    64 (no several-taps form): the sixteen-wave kernel conv_wgrad_w16_kernel, here also at stride 2, behind a nearest upsample and as a 1x1."""
    err = _wgrad_case(cuda, built_lib, cin, cout, dims, k, stride, pad, n, ups)
    assert err <= 2e-5, err


@pytest.mark.parametrize("cin,cout,dims,n,ksplit", [
    (128, 256, (5, 7, 9), 2, 1), (192, 128, (4, 6, 8), 2, 2), (96, 160, (6, 6, 12), 1, 1), (256, 256, (12, 12, 12), 1, 3),
    (128, 128, (2, 2, 8), 1, 1), (128, 128, (3, 3, 8), 2, 4), (160, 96, (3, 5, 17), 1, 2), (128, 128, (24, 24, 24), 1, 3),
    (128, 128, (1, 1, 8), 1, 1), (128, 128, (4, 4, 64), 1, 1), (128, 128, (2, 2, 32), 3, 2)])
def test_conv_wgrad_wide_channels_w_borders_and_splits(cuda, built_lib, cin, cout, dims, n, ksplit):
    """3^3 stride-1 weight gradients with both channel counts above 64 (the UNet's 256 / 512-channel levels) over the shapes where a W-halo
    form would have to get its border pairs right: W not a divisor of 64, W = 8, W = 64 and 32, batches (the voxel before sample n's first
    is sample n - 1's last), voxel ranges that are no multiple of 64, splits that end up empty, channel counts that leave the last tile part
    empty, volumes smaller than one K step.  The product library runs conv_wgrad_kernel on them; an experiments build with
    LDM_WGRAD_KW3=1 runs conv_wgrad_kw3_kernel (three kw taps per workgroup over one shared X tile, csrc/conv_wgrad_kw3.h; no faster,
    profiles/r05_wgrad_kw3.txt)."""
    err = _wgrad_case(cuda, built_lib, cin, cout, dims, 3, 1, 1, n, 0, ksplit=ksplit)
    assert err <= 2e-5, err


@pytest.mark.parametrize("dims,n,ksplit", [((12, 12, 12), 1, 3), ((6, 6, 6), 1, 4), ((2, 2, 2), 2, 1), ((5, 3, 2), 1, 2), ((24, 24, 24), 1, 5)])
def test_conv_wgrad_voxel_split_and_tiny_volumes(cuda, built_lib, dims, n, ksplit):
    """Voxel range split over workgroups (partial matrices summed by the caller), including splits that end up empty
    and volumes smaller than one 64-voxel K step."""
    err = _wgrad_case(cuda, built_lib, 64, 96, dims, 3, 1, 1, n, 0, ksplit=ksplit)
    assert err <= 2e-5, err


@pytest.mark.parametrize("c,groups,dual,silu,n", [(64, 32, 0, 1, 1), (256, 32, 0, 1, 2), (768, 32, 512, 1, 1), (96, 8, 32, 0, 2)])
def test_group_norm_silu_backward(cuda, built_lib, c, groups, dual, silu, n):
    """dx (+ accumulated residual gradient), dgamma, dbeta of act(GroupNorm(cat(xa, xb))) against torch autograd on the
    same bf16-rounded inputs; dx is rounded to bf16 by the kernel."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(c + n)
    dims = (4, 6, 5)
    x = bf16_round(torch.randn((n, c, *dims), generator=g) * 1.3 + 0.2).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn((c,), generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn((c,), generator=g)).requires_grad_(True)
    y = F.group_norm(x, groups, gamma, beta, 1e-6)
    if silu:
        y = F.silu(y)
    dy = bf16_round(torch.randn(y.shape, generator=g))
    acc = bf16_round(torch.randn(y.shape, generator=g))
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (x, gamma, beta), dy)
    dx_ref = dx_ref + acc
    ca = dual if dual else c
    xd = x.detach()
    xa = to_ndhwc_bf16(xd[:, :ca], ca).to(cuda)
    xb = to_ndhwc_bf16(xd[:, ca:], c - ca).to(cuda) if dual else None
    aa = to_ndhwc_bf16(acc[:, :ca], ca).to(cuda)
    ab_ = to_ndhwc_bf16(acc[:, ca:], c - ca).to(cuda) if dual else None
    dyd = to_ndhwc_bf16(dy, c).to(cuda)
    dhw = dims[0] * dims[1] * dims[2]
    dxa = torch.empty((n, *dims, ca), dtype=torch.bfloat16, device=cuda)
    dxb = torch.empty((n, *dims, c - ca), dtype=torch.bfloat16, device=cuda) if dual else None
    dgam, dbet = torch.empty((c,), device=cuda), torch.empty((c,), device=cuda)
    scratch = torch.empty((built_lib.ldm_op_group_norm_bwd_scratch_bytes(n, c, dhw, groups),), dtype=torch.uint8, device=cuda)
    gd, bd = gamma.detach().to(cuda), beta.detach().to(cuda)
    P = lambda t: None if t is None else t.data_ptr()
    _lib.check(built_lib.ldm_op_group_norm_bwd(dyd.data_ptr(), xa.data_ptr(), ca, P(xb), c - ca, gd.data_ptr(), bd.data_ptr(), groups,
                                               1e-6, silu, aa.data_ptr(), P(ab_), dxa.data_ptr(), P(dxb), dgam.data_ptr(), dbet.data_ptr(),
                                               n, dhw, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = from_ndhwc(dxa.cpu(), ca)
    if dual:
        got = torch.cat([got, from_ndhwc(dxb.cpu(), c - ca)], 1)
    assert rel_l2(got, bf16_round(dx_ref)) <= 5e-4
    assert rel_l2(dgam.cpu(), dg_ref) <= 1e-4 and rel_l2(dbet.cpu(), db_ref) <= 1e-4


@pytest.mark.parametrize("b,n,c", [(1, 216, 128), (2, 64, 64), (1, 1000, 256)])
def test_attention_backward(cuda, built_lib, b, n, c):
    """dq | dk | dv against torch autograd of fp32 softmax attention on the same bf16 q, k, v, dO (P and dS are rounded to
    bf16 inside the kernels, outputs are bf16: ~2^-9 relative each)."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(n + c)
    qkv = torch.randn((b, n, 3 * c), generator=g)
    qkv[..., :2 * c] *= 1.5
    qkv = bf16_round(qkv).requires_grad_(True)
    h = c // 64

    def split(z):
        return z.reshape(b, n, h, 64).permute(0, 2, 1, 3)
    q, k, v = split(qkv[..., :c]), split(qkv[..., c:2 * c]), split(qkv[..., 2 * c:])
    o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3).reshape(b, n, c)
    do = bf16_round(torch.randn(o.shape, generator=g))
    (ref,) = torch.autograd.grad(o, qkv, do)
    st = torch.cuda.current_stream().cuda_stream
    dq = qkv.detach().to(torch.bfloat16).to(cuda)
    out = torch.empty((b, n, c), dtype=torch.bfloat16, device=cuda)
    lse = torch.empty((b, h, n), dtype=torch.float32, device=cuda)
    _lib.check(built_lib.ldm_op_attention_train(dq.data_ptr(), out.data_ptr(), lse.data_ptr(), b, n, c, st))
    dod = do.to(torch.bfloat16).to(cuda)
    delta = torch.empty((b, h, n), dtype=torch.float32, device=cuda)
    dqkv = torch.full((b, n, 3 * c), float("nan"), dtype=torch.bfloat16, device=cuda)
    _lib.check(built_lib.ldm_op_attention_bwd(dq.data_ptr(), out.data_ptr(), dod.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                              dqkv.data_ptr(), b, n, c, st))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), o.detach()) <= 8e-3
    ref_lse = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) * 0.125, dim=-1)
    assert rel_l2(lse.cpu(), ref_lse) <= 1e-5
    got = dqkv.float().cpu()
    for name, sl in (("dq", slice(0, c)), ("dk", slice(c, 2 * c)), ("dv", slice(2 * c, 3 * c))):
        e = rel_l2(got[..., sl], ref[..., sl])
        assert e <= 1.5e-2, (name, e)


def test_conv3_halo_vs_general_kernel(cuda, built_lib, monkeypatch):
    """Both kernels for the same conv (LDM_CONV_HALO toggles the planner's choice): each within tolerance of torch and of
    each other (they differ only in fp32 summation order over K)."""
    kw = dict(cin=(128, 0), cout=128, dims=(7, 6, 9), n=2, wgn=2, temb=True, residual=True, seed=5)
    e_halo, tol = _conv_case(cuda, built_lib, **kw)
    monkeypatch.setenv("LDM_CONV_HALO", "0")
    e_gen, _ = _conv_case(cuda, built_lib, **kw)
    assert e_halo <= tol and e_gen <= tol, (e_halo, e_gen)


# ---------------------------------------------------------------------------------------------- light GEMM (1x1 convolutions)
@pytest.mark.parametrize("cin,cout,dims,n,residual", [
    ((256, 0), 256, (12, 12, 12), 1, True),      # out_proj shape of the 12^3 level (32 x 32 wave tiles)
    ((512, 0), 1536, (6, 6, 6), 1, False),       # q|k|v of the 6^3 level
    ((128, 128), 96, (5, 7, 3), 2, True),        # two concatenated sources, ragged M, channel padding (cout 96 -> 128 weight rows)
    ((256, 0), 512, (16, 16, 16), 1, False),     # 64 x 64 wave tiles (>= 256 tiles)
    ((128, 0), 64, (4, 4, 4), 1, False),         # a single K chunk
])
def test_gemm_light_kernel(cuda, built_lib, monkeypatch, cin, cout, dims, n, residual):
    """1x1 convolutions with Cin % 128 == 0 run on gemm_light_kernel (fragments straight from global memory); the same
    case on conv_igemm_kernel (LDM_GEMM_LIGHT=0) must also hold: both within the same-rounding tolerance of torch."""
    kw = dict(cin=cin, cout=cout, dims=dims, n=n, k=1, pad=0, residual=residual, seed=11)
    e_light, tol = _conv_case(cuda, built_lib, **kw)
    monkeypatch.setenv("LDM_GEMM_LIGHT", "0")
    e_gen, _ = _conv_case(cuda, built_lib, **kw)
    assert e_light <= tol and e_gen <= tol, (e_light, e_gen)


@pytest.mark.parametrize("cin,cout,dims,n,splitk", [(64, 64, (9, 7, 10), 1, 1), (128, 64, (6, 6, 6), 2, 3), (64, 32, (13, 5, 8), 1, 1)])
def test_conv3_tall_halo_tile(cuda, built_lib, cin, cout, dims, n, splitk):
    """conv3_halo_kernel<6, 0, true>: the 254-voxel x 64-cout tile that carries the Cout = 64 layers of the AutoencoderKL
    (forced here with wgn = 1; ragged last tiles, several samples, split K, channel padding, all epilogue inputs)."""
    err, tol = _conv_case(cuda, built_lib, cin=(cin, 0), cout=cout, dims=dims, n=n, wgn=1, splitk=splitk, temb=True, residual=True, seed=13)
    assert err <= tol, err


@pytest.mark.parametrize("b,n,c,d", [(1, 216, 256, 32), (2, 100, 64, 32), (1, 512, 128, 128), (1, 1000, 256, 256), (2, 216, 256, 128),
                                     (1, 130, 64, 64)])
def test_attention_any_head_dim_forward_backward(cuda, built_lib, b, n, c, d):
    """head_dim 32 (num_head_channels of config_train_stable.json:45-46) and the single-head AutoencoderKL blocks (d = C = 64 / 128 /
    256, config_train_32g.json:21-25): forward, log-sum-exp rows and dq | dk | dv against fp32 softmax attention + torch autograd
    on the same bf16 q, k, v, dO."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(n + c + d)
    qkv = torch.randn((b, n, 3 * c), generator=g)
    qkv[..., :2 * c] *= (64.0 / d) ** 0.25 * 1.5          # keep the score spread comparable across head dims
    qkv = bf16_round(qkv).requires_grad_(True)
    h, scale = c // d, d ** -0.5

    def split(z):
        return z.reshape(b, n, h, d).permute(0, 2, 1, 3)
    q, k, v = split(qkv[..., :c]), split(qkv[..., c:2 * c]), split(qkv[..., 2 * c:])
    o = (torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v).permute(0, 2, 1, 3).reshape(b, n, c)
    do = bf16_round(torch.randn(o.shape, generator=g))
    (ref,) = torch.autograd.grad(o, qkv, do)
    st = torch.cuda.current_stream().cuda_stream
    dq = qkv.detach().to(torch.bfloat16).to(cuda)
    out = torch.full((b, n, c), float("nan"), dtype=torch.bfloat16, device=cuda)
    lse = torch.empty((b, h, n), dtype=torch.float32, device=cuda)
    _lib.check(built_lib.ldm_op_attention_hd(dq.data_ptr(), out.data_ptr(), lse.data_ptr(), b, n, c, d, st))
    dod = do.to(torch.bfloat16).to(cuda)
    delta = torch.empty((b, h, n), dtype=torch.float32, device=cuda)
    dqkv = torch.full((b, n, 3 * c), float("nan"), dtype=torch.bfloat16, device=cuda)
    _lib.check(built_lib.ldm_op_attention_bwd_hd(dq.data_ptr(), out.data_ptr(), dod.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                                 dqkv.data_ptr(), b, n, c, d, st))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), o.detach()) <= 8e-3
    ref_lse = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) * scale, dim=-1)
    assert rel_l2(lse.cpu(), ref_lse) <= 1e-5
    got = dqkv.float().cpu()
    for name, sl in (("dq", slice(0, c)), ("dk", slice(c, 2 * c)), ("dv", slice(2 * c, 3 * c))):
        e = rel_l2(got[..., sl], ref[..., sl])
        assert e <= 1.5e-2, (name, e)
    with pytest.raises(_lib.LdmError):
        _lib.check(built_lib.ldm_op_attention_hd(dq.data_ptr(), out.data_ptr(), None, b, n, c, 48, st))


# the kernel variants the INFERENCE PLANS launch for a conv -> GroupNorm pair: statistics slabs from the conv epilogue or from
# splitk_finalize_kernel<true> (inline-asm write-through stores), folded by gn_fused_apply_kernel<true> (same stores).  The shapes are
# the benchmark UNet's: 24^3 / 12^3 / 6^3 voxels, every tile family (halo 126x128, tall 254x64, general 128x128 / 64x256 / 256x64).
@pytest.mark.parametrize("cin,cout,dims,n,wgn,splitk,groups,silu", [
    (256, 256, (12, 12, 12), 1, 2, 1, 32, True),        # halo tile, statistics from the conv epilogue
    (256, 256, (12, 12, 12), 1, 2, 6, 32, True),        # halo tile, split-K: statistics from the write-through finalize
    (512, 512, (6, 6, 6), 1, 2, 12, 32, True),          # the 6^3 level: 216 rows, ragged 32-row blocks
    (128, 64, (16, 16, 16), 1, 1, 1, 32, False),        # tall halo tile (Cout = 64)
    (64, 256, (8, 8, 8), 2, 4, 1, 32, True),            # general kernel 64 x 256 tile, batch 2
    (96, 128, (8, 8, 8), 2, 2, 3, 16, True),            # general kernel (K step 32), split-K, batch 2 (DHW % 32 == 0)
    (256, 256, (24, 24, 24), 1, 2, 1, 32, True),        # the headline 24^3 conv + norm
])
def test_conv_then_group_norm_as_the_plans_launch_them(cuda, built_lib, cin, cout, dims, n, wgn, splitk, groups, silu):
    from ldm3d import _lib
    g = torch.Generator().manual_seed(cin + cout + dims[0] + splitk)
    x = bf16_round(torch.randn((n, cin, *dims), generator=g))
    w = bf16_round(torch.randn((cout, cin, 3, 3, 3), generator=g) / (27 * cin) ** 0.5)
    b = 0.1 * torch.randn((cout,), generator=g)
    gamma = 1.0 + 0.1 * torch.randn((cout,), generator=g)
    beta = 0.1 * torch.randn((cout,), generator=g)
    y = bf16_round(F.conv3d(x, w, b, padding=1))                    # the conv output is stored bf16; the statistics are of the stored values
    ref = F.group_norm(y, groups, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    cout_pad = rup(cout, 64)
    xa = to_ndhwc_bf16(x).to(cuda)
    wp = pack_conv_weight(w, cin, cout_pad).to(cuda)
    bp = pad_vec(b, cout_pad).to(cuda)
    conv_out = torch.full((n, *dims, cout), float("nan"), dtype=torch.bfloat16, device=cuda)
    gn_out = torch.full((n, *dims, cout), float("nan"), dtype=torch.bfloat16, device=cuda)
    nb = built_lib.ldm_op_conv3d_gn_scratch_bytes(n, *dims, cout_pad, splitk)
    scratch = torch.empty((nb,), dtype=torch.uint8, device=cuda)
    gd, bd = gamma.to(cuda), beta.to(cuda)
    _lib.check(built_lib.ldm_op_conv3d_gn(xa.data_ptr(), cin, wp.data_ptr(), bp.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                          groups, 1e-6, int(silu), conv_out.data_ptr(), gn_out.data_ptr(), n, *dims, cout, cout_pad, wgn, splitk,
                                          scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    e_conv = rel_l2(from_ndhwc(conv_out.cpu(), cout), y)
    # the GroupNorm is checked on the KERNEL's conv output (identical input on both sides): only its own arithmetic + one rounding
    yk = from_ndhwc(conv_out.cpu(), cout)
    ref_k = F.group_norm(yk, groups, gamma, beta, 1e-6)
    if silu:
        ref_k = F.silu(ref_k)
    e_gn = rel_l2(from_ndhwc(gn_out.cpu(), cout), bf16_round(ref_k))
    e_pair = rel_l2(from_ndhwc(gn_out.cpu(), cout), bf16_round(ref))
    print(f"conv {cin}->{cout} {dims} n={n} wgn={wgn} splitk={splitk}: conv {e_conv:.2e}, GroupNorm on the kernel's input {e_gn:.2e}, pair {e_pair:.2e}")
    assert e_conv <= TOL_SAME_ROUNDING, e_conv
    assert e_gn <= TOL_SAME_ROUNDING, e_gn
    assert e_pair <= TOL_SAME_ROUNDING, e_pair


# conv (split over K, planar slabs) -> ONE finalize-and-GroupNorm launch (csrc/fin_gn.h: every workgroup owns a whole (sample, group), no
# statistics cross workgroups), as the inference plans launch it at the 12^3 / 6^3 levels.  Checked: against torch on identical
# bf16-rounded inputs (same rounding points as the two-launch path), against the two-launch path itself (the un-normalised tensor bit
# for bit; the normalised one to the last bf16 ulp: the fold order of the statistics differs), and under back-to-back replays with
# CHANGING inputs (results follow the inputs).
@pytest.mark.parametrize("cin,cout,dims,n,wgn,splitk,groups,silu,temb,res,keep", [
    (256, 256, (12, 12, 12), 1, 2, 6, 32, True, True, False, False),    # ResBlock conv1 -> norm2 at 12^3 (halo tile): nobody reads the raw tensor
    (256, 256, (12, 12, 12), 1, 2, 8, 32, False, False, True, True),    # conv2 (+ residual) -> attention norm: the raw tensor is the residual stream
    (512, 512, (6, 6, 6), 1, 2, 12, 32, True, True, False, False),      # the 6^3 level: 216 rows (ragged last 32-row block), 16 channels per group
    (512, 512, (6, 6, 6), 1, 2, 27, 32, False, False, True, True),
    (96, 128, (8, 8, 8), 2, 2, 3, 16, True, True, False, True),         # general kernel (K step 32), batch 2, 8 channels per group
])
def test_split_k_conv_then_fused_finalize_group_norm(cuda, built_lib, cin, cout, dims, n, wgn, splitk, groups, silu, temb, res, keep):
    from ldm3d import _lib
    g = torch.Generator().manual_seed(cin + cout + dims[0] + splitk)
    cout_pad = rup(cout, 64)
    w = bf16_round(torch.randn((cout, cin, 3, 3, 3), generator=g) / (27 * cin) ** 0.5)
    b = 0.1 * torch.randn((cout,), generator=g)
    gamma = 1.0 + 0.1 * torch.randn((cout,), generator=g)
    beta = 0.1 * torch.randn((cout,), generator=g)
    tvec = 0.3 * torch.randn((n, cout_pad), generator=g) if temb else None
    wp = pack_conv_weight(w, cin, cout_pad).to(cuda)
    bp = pad_vec(b, cout_pad).to(cuda)
    gd, bd = gamma.to(cuda), beta.to(cuda)
    td = tvec.to(cuda) if temb else None
    st = torch.cuda.current_stream().cuda_stream
    nb = max(built_lib.ldm_op_conv3d_fin_gn_scratch_bytes(n, *dims, cout_pad, splitk), built_lib.ldm_op_conv3d_gn_scratch_bytes(n, *dims, cout_pad, splitk))
    scratch = torch.empty((nb,), dtype=torch.uint8, device=cuda)

    def fused(xa, ra, raw_out, gn_out):
        _lib.check(built_lib.ldm_op_conv3d_fin_gn(xa.data_ptr(), cin, wp.data_ptr(), bp.data_ptr(), _lib.ptr(td), cout_pad if temb else 0, _lib.ptr(ra),
                                                  gd.data_ptr(), bd.data_ptr(), groups, 1e-6, int(silu), _lib.ptr(raw_out), gn_out.data_ptr(),
                                                  n, *dims, cout, cout_pad, wgn, splitk, scratch.data_ptr(), scratch.numel(), st))
    worst = 0.0
    for rep in range(3):
        x = bf16_round(torch.randn((n, cin, *dims), generator=g))
        r = bf16_round(torch.randn((n, cout, *dims), generator=g)) if res else None
        y = F.conv3d(x, w, b, padding=1)
        if temb:
            y = y + tvec[:, :cout].reshape(n, cout, 1, 1, 1)
        if res:
            y = y + r
        y = bf16_round(y)
        ref = F.group_norm(y, groups, gamma, beta, 1e-6)
        if silu:
            ref = F.silu(ref)
        xa = to_ndhwc_bf16(x).to(cuda)
        ra = to_ndhwc_bf16(r).to(cuda) if res else None
        raw = torch.full((n, *dims, cout), float("nan"), dtype=torch.bfloat16, device=cuda) if keep else None
        gn_out = torch.full((n, *dims, cout), float("nan"), dtype=torch.bfloat16, device=cuda)
        fused(xa, ra, raw, gn_out)
        e_pair = rel_l2(from_ndhwc(gn_out.cpu(), cout), bf16_round(ref))
        worst = max(worst, e_pair)
        assert e_pair <= TOL_SAME_ROUNDING, e_pair
        if keep:
            assert rel_l2(from_ndhwc(raw.cpu(), cout), y) <= TOL_SAME_ROUNDING
            yk = from_ndhwc(raw.cpu(), cout)                           # GroupNorm of the kernel's own un-normalised tensor
            ref_k = F.group_norm(yk, groups, gamma, beta, 1e-6)
            if silu:
                ref_k = F.silu(ref_k)
            assert rel_l2(from_ndhwc(gn_out.cpu(), cout), bf16_round(ref_k)) <= TOL_SAME_ROUNDING
        if not temb and not res:                                       # the two-launch path takes bias only: compare where the forms coincide
            c2 = torch.empty((n, *dims, cout), dtype=torch.bfloat16, device=cuda)
            g2 = torch.empty((n, *dims, cout), dtype=torch.bfloat16, device=cuda)
            _lib.check(built_lib.ldm_op_conv3d_gn(xa.data_ptr(), cin, wp.data_ptr(), bp.data_ptr(), gd.data_ptr(), bd.data_ptr(), groups, 1e-6, int(silu),
                                                  c2.data_ptr(), g2.data_ptr(), n, *dims, cout, cout_pad, wgn, splitk, scratch.data_ptr(), scratch.numel(), st))
            torch.cuda.synchronize()
            assert torch.equal(c2, raw)
            d = (g2.float() - gn_out.float()).abs()
            assert float(d.max()) <= 2.0 ** -7 * float(g2.float().abs().max()) and float((d > 0).float().mean()) < 1e-2
    # replays with changing inputs: results must follow the inputs (a stale exchange area or a counter left non-zero would show)
    xs = [to_ndhwc_bf16(bf16_round(torch.randn((n, cin, *dims), generator=g))).to(cuda) for _ in range(2)]
    outs = [torch.empty((n, *dims, cout), dtype=torch.bfloat16, device=cuda) for _ in range(2)]
    first = []
    ra = to_ndhwc_bf16(bf16_round(torch.randn((n, cout, *dims), generator=g))).to(cuda) if res else None
    for k in range(2):
        fused(xs[k], ra, None, outs[k])
        first.append(outs[k].clone())
    for it in range(40):
        k = it & 1
        fused(xs[k], ra, None, outs[k])
        if it % 10 == 9:
            assert torch.equal(outs[k], first[k])
    fused(xs[0], ra, None, outs[0])
    assert torch.equal(outs[0], first[0]) and not torch.equal(first[0], first[1])
    print(f"fused finalize + GroupNorm {cin}->{cout} {dims} n={n} splitk={splitk} groups={groups}: pair {worst:.2e}")


@pytest.mark.parametrize("cin,cout,dims,n,th,temb,residual,slots", [
    (64, 64, (8, 16, 32), 1, 8, False, False, 0),      # whole blocks
    (64, 64, (8, 16, 32), 1, 4, False, True, 0),
    (128, 64, (5, 11, 20), 2, 8, True, True, 0),       # ragged in every dimension, two samples, four channel chunks
    (32, 40, (3, 3, 3), 1, 4, True, False, 0),         # smaller than one block; real couts < 64 (padding written as zeros)
    (64, 64, (4, 9, 17), 1, 8, False, False, 0),       # one voxel past a block edge in h and w
    (64, 64, (9, 20, 40), 2, 8, True, True, 8),        # the tile loop: 54 tiles on 8 workgroups (7 / 6 tiles each), ragged, two samples
    (32, 64, (8, 16, 32), 1, 8, False, False, 16),     # the tile loop with a tile count that is no multiple of the grid... 8 tiles on 16: falls back to one tile per workgroup
    (96, 64, (12, 8, 16), 1, 8, False, True, 8),       # three channel chunks per tile, 3 tiles on 8 workgroups -> one-tile form
    (96, 64, (12, 24, 48), 1, 8, False, True, 8),      # three chunks, 27 tiles on 8 workgroups
])
def test_conv3_block_kernel(cuda, built_lib, cin, cout, dims, n, th, temb, residual, slots):
    """conv3_block_kernel (64 output channels, one halo block in LDS per workgroup: the AutoencoderKL's full-resolution ResBlock convs)
    against F.conv3d on the same bf16-rounded operands; its per-block GroupNorm partials are the sums of the stored values."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(cin + dims[2] + th)
    x = torch.randn((n, cin, *dims), generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / (cin * 27) ** 0.5
    b = 0.1 * torch.randn((cout,), generator=g)
    ref = F.conv3d(bf16_round(x), bf16_round(w), b, padding=1)
    te = None
    if temb:
        tv = torch.randn((n, 64), generator=g)
        tv[:, cout:] = 0.0                       # like the bias and the weight rows: padding channels carry zeros
        ref = ref + tv[:, :cout, None, None, None]
        te = tv.to(cuda)
    res = None
    if residual:
        rv = bf16_round(torch.randn(ref.shape, generator=g))
        ref = ref + rv
        res = to_ndhwc_bf16(rv, 64).to(cuda)
    xa = to_ndhwc_bf16(x).to(cuda)
    wp = pack_conv_weight(w, cin, 64).to(cuda)
    bp = pad_vec(b, 64).to(cuda)
    rows = built_lib.ldm_op_conv3d_block_stats_rows(*dims, th)
    out = torch.full((n, *dims, 64), float("nan"), dtype=torch.bfloat16, device=cuda)
    stats = torch.full((n * rows, 64, 2), float("nan"), device=cuda)
    prev = built_lib.ldm_debug_conv_block_slots(slots)
    if slots and prev < 0:
        pytest.skip("the tile-loop form of conv3_block_kernel exists in experiments builds only (make EXTRA=-DLDM_EXPERIMENTS)")
    try:
        _lib.check(built_lib.ldm_op_conv3d_block(xa.data_ptr(), cin, wp.data_ptr(), bp.data_ptr(), None if te is None else te.data_ptr(), 64,
                                                 None if res is None else res.data_ptr(), out.data_ptr(), stats.data_ptr(), n, *dims, th,
                                                 torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        built_lib.ldm_debug_conv_block_slots(max(prev, 0))
    assert torch.isfinite(out.float()).all()
    err = rel_l2(from_ndhwc(out.cpu(), cout), bf16_round(ref))
    assert err <= TOL_SAME_ROUNDING, err
    if cout < 64:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "channel padding must be written as zeros"
    o = out.double().view(n, -1, 64)
    tot = stats.double().view(n, rows, 64, 2).sum(1)
    assert torch.allclose(tot[..., 0], o.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(tot[..., 1], (o * o).sum(1), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("cin,cout,dims,n,temb,residual", [
    (128, 128, (8, 16, 32), 1, False, False),       # whole blocks, four chunks through both buffers
    (64, 128, (5, 11, 20), 2, True, True),          # ragged in every dimension, two samples, two chunks
    (32, 100, (3, 3, 3), 1, True, False),           # one chunk (no second buffer), smaller than a block, real couts < 128
    (96, 128, (4, 9, 17), 1, False, True),          # three chunks (the buffers end on the one they started with)
    (256, 128, (12, 16, 16), 1, False, False),      # eight chunks (the decoder's 256 -> 128)
])
def test_conv3_block128_kernel(cuda, built_lib, cin, cout, dims, n, temb, residual):
    """conv3_block128_kernel (128 output channels: eight waves per workgroup, double-buffered halo chunks copied piece by piece under the K
    loop of the chunk before; the AutoencoderKL's half-resolution ResBlock convs) against F.conv3d on the same bf16-rounded operands; its
    per-block GroupNorm partials are the sums of the stored values."""
    from ldm3d import _lib
    g = torch.Generator().manual_seed(cin + dims[2] + 128)
    x = torch.randn((n, cin, *dims), generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / (cin * 27) ** 0.5
    b = 0.1 * torch.randn((cout,), generator=g)
    ref = F.conv3d(bf16_round(x), bf16_round(w), b, padding=1)
    te = None
    if temb:
        tv = torch.randn((n, 128), generator=g)
        tv[:, cout:] = 0.0
        ref = ref + tv[:, :cout, None, None, None]
        te = tv.to(cuda)
    res = None
    if residual:
        rv = bf16_round(torch.randn(ref.shape, generator=g))
        ref = ref + rv
        res = to_ndhwc_bf16(rv, 128).to(cuda)
    xa = to_ndhwc_bf16(x).to(cuda)
    wp = pack_conv_weight(w, cin, 128).to(cuda)
    bp = pad_vec(b, 128).to(cuda)
    rows = built_lib.ldm_op_conv3d_block_stats_rows(*dims, 8)
    out = torch.full((n, *dims, 128), float("nan"), dtype=torch.bfloat16, device=cuda)
    stats = torch.full((n * rows, 128, 2), float("nan"), device=cuda)
    _lib.check(built_lib.ldm_op_conv3d_block128(xa.data_ptr(), cin, wp.data_ptr(), bp.data_ptr(), None if te is None else te.data_ptr(), 128,
                                                None if res is None else res.data_ptr(), out.data_ptr(), stats.data_ptr(), n, *dims,
                                                torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    err = rel_l2(from_ndhwc(out.cpu(), cout), bf16_round(ref))
    assert err <= TOL_SAME_ROUNDING, err
    if cout < 128:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "channel padding must be written as zeros"
    o = out.double().view(n, -1, 128)
    tot = stats.double().view(n, rows, 128, 2).sum(1)
    assert torch.allclose(tot[..., 0], o.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(tot[..., 1], (o * o).sum(1), rtol=1e-4, atol=1e-2)
