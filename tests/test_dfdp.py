"""Depth-from-DP network against the fixture generated from the reference's
YRStereonet_3D by oracle/gen_golden_dfdp.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def inputs(fx):
    g = torch.Generator().manual_seed(int(fx["input_seed"]))
    xl = torch.rand(1, 3, 128, 128, generator=g)
    xr = torch.roll(xl, 3, dims=-1) * 0.9 + 0.05 * torch.rand(1, 3, 128, 128, generator=g)
    assert xl.double().sum().item() == pytest.approx(float(fx["xl_sum"]), rel=1e-12)
    assert xr.double().sum().item() == pytest.approx(float(fx["xr_sum"]), rel=1e-12)
    cx, cy = torch.rand(2, 3, 4, 24, generator=g), torch.rand(2, 3, 4, 24, generator=g)
    gt = 0.5 + 4.5 * torch.rand(1, 1, 128, 128, generator=g)
    gt[0, 0, :4, :4] = 0.0
    assert gt.double().sum().item() == pytest.approx(float(fx["gt_depth_sum"]), rel=1e-12)
    inputs.gt = gt
    return xl, xr, cx, cy


def build(fx):
    from sdirt_amd.dfdp import DfDPNet
    torch.manual_seed(int(fx["seed"]))
    return DfDPNet().eval()


def test_dfdp_net_draws_the_reference_weights_and_matches_it_on_cpu():
    fx = load_golden("f10_dfdp_net")
    net = build(fx)
    sd = net.state_dict()
    keys = [k[4:] for k in fx.files if k.startswith("sum/")]
    assert sorted(keys) == sorted(k for k, v in sd.items() if v.dtype.is_floating_point)
    for k in keys:                                     # same names, same seeded values
        assert sd[k].double().sum().item() == pytest.approx(float(fx["sum/" + k]), rel=1e-10, abs=1e-10), k
        assert sd[k].double().abs().sum().item() == pytest.approx(float(fx["abs/" + k]), rel=1e-10), k
    xl, xr, cx, cy = inputs(fx)
    from sdirt_amd.dfdp import dp_cost_volume
    np.testing.assert_array_equal(dp_cost_volume(cx, cy, 20).numpy(), fx["cv"])
    with torch.no_grad():
        fl = net.feature(xl)
        disp = net(xl, xr)
    np.testing.assert_allclose(fl[0, :, ::8, ::8].numpy(), fx["feature_l_head"], rtol=1e-4, atol=1e-5)
    assert disp.shape == fx["disp"].shape == (1, 1, 128, 128)
    np.testing.assert_allclose(disp.numpy(), fx["disp"], rtol=1e-4, atol=1e-4)


def test_basenet_depth_mode_matches_the_reference():
    """dfdp/basenet.py:23-49: log-depth transform in place, masked SmoothL1, outputs back in metres."""
    from sdirt_amd.dfdp import Basenet
    fx = load_golden("f10_dfdp_net")
    xl, xr, _, _ = inputs(fx)
    torch.manual_seed(int(fx["seed"]))
    base = Basenet(train_mode="dfdp").eval()
    gt = inputs.gt.clone()
    with torch.no_grad():
        losses, outputs = base.dfdp({"stack_rgb_img": torch.cat((xl, xr), 1), "AiF_img": xl, "gt_depth": gt},
                                    train=True)
    assert losses["total"].item() == pytest.approx(float(fx["loss_total"]), rel=1e-4)
    np.testing.assert_allclose(outputs["pred_depth_est"].numpy(), fx["pred_depth_est"], rtol=2e-4, atol=1e-5)
    assert (outputs["gt_depth"] - inputs.gt).abs().max().item() <= float(fx["gt_depth_roundtrip_err"]) + 1e-6
    assert outputs["gt_depth"] is gt                                   # transformed in place, as the reference
    with pytest.raises(ValueError):
        Basenet(train_mode="other")


def test_basenet_deblur_mode_matches_the_reference():
    """Mydeblur (dddnet.py:32-100, 180-305) under Basenet(train_mode='deblur'): same seeded weights
    (per-tensor checksums), same three losses, same refined depth and deblurred image."""
    from sdirt_amd.dfdp import Basenet
    fx = load_golden("f10_dfdp_net")
    xl, xr, _, _ = inputs(fx)
    torch.manual_seed(int(fx["seed"]))
    base = Basenet(train_mode="deblur").eval()
    sd = base.deblur_net.state_dict()
    keys = [k[5:] for k in fx.files if k.startswith("dsum/")]
    assert sorted(keys) == sorted(sd)
    for k in keys:
        assert sd[k].double().sum().item() == pytest.approx(float(fx["dsum/" + k]), rel=1e-10, abs=1e-10), k
        assert sd[k].double().abs().sum().item() == pytest.approx(float(fx["dabs/" + k]), rel=1e-10, abs=1e-12), k
    with torch.no_grad():
        base.deblur_net.cam_attention.gamma.fill_(0.3)
        losses, outputs = base.dfdp({"stack_rgb_img": torch.cat((xl, xr), 1), "AiF_img": xl,
                                     "gt_depth": inputs.gt.clone()}, train=True)
    got = [losses["depth_est"].item(), losses["depth_fix"].item(), losses["aif"].item()]
    np.testing.assert_allclose(got, fx["deblur_losses"], rtol=2e-4)
    assert losses["total"].item() == pytest.approx(float(fx["deblur_loss_total"]), rel=2e-4)
    np.testing.assert_allclose(outputs["pred_aif"][0, :, ::16, ::16].numpy(), fx["pred_aif_head"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(outputs["pred_depth_fix"][0, :, ::16, ::16].numpy(), fx["pred_depth_fix_head"],
                               rtol=1e-3, atol=2e-5)


@pytest.mark.gpu
def test_dfdp_net_and_cost_volume_kernel_on_the_gpu():
    fx = load_golden("f10_dfdp_net")
    dev = "cuda:0"
    net = build(fx).to(dev)
    xl, xr, cx, cy = inputs(fx)
    from sdirt_amd.dfdp import dp_cost_volume
    cv = dp_cost_volume(cx.to(dev), cy.to(dev), 20)                    # HIP kernel, fp32
    np.testing.assert_array_equal(cv.cpu().numpy(), fx["cv"])
    cvh = dp_cost_volume(cx.to(dev).half(), cy.to(dev).half(), 20)     # HIP kernel, fp16
    np.testing.assert_array_equal(cvh.float().cpu().numpy(), torch.from_numpy(fx["cv"]).half().float().numpy())
    g = torch.Generator().manual_seed(5)
    from sdirt_amd.dfdp import _cost_volume_reference
    for shape, d in (((1, 5, 7, 33), 20), ((2, 2, 3, 9), 6), ((1, 1, 2, 130), 12)):   # ragged widths
        a, b = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
        assert torch.equal(dp_cost_volume(a.to(dev), b.to(dev), d).cpu(), dp_cost_volume(a, b, d))
        # gradients: HIP adjoint kernel vs autograd through the reference's slice assignments
        wgt = torch.rand(shape[0], 2 * shape[1], d, shape[2], shape[3], generator=g)
        a1, b1 = a.clone().requires_grad_(), b.clone().requires_grad_()
        (_cost_volume_reference(a1, b1, d) * wgt).sum().backward()
        a2, b2 = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
        (dp_cost_volume(a2, b2, d) * wgt.to(dev)).sum().backward()
        assert torch.allclose(a2.grad.cpu(), a1.grad, rtol=1e-6, atol=1e-6)
        assert torch.allclose(b2.grad.cpu(), b1.grad, rtol=1e-6, atol=1e-6)
    with torch.no_grad():
        disp = net(xl.to(dev), xr.to(dev))
    assert np.abs(disp.cpu().numpy() - fx["disp"]).max() < 2e-3        # MIOpen fp32 convs vs CPU
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):   # as Basenet.forward runs it
        disp16 = net(xl.to(dev), xr.to(dev))
    assert np.abs(disp16.float().cpu().numpy() - fx["disp"]).max() < 5e-2
    # the deblurring branch on the GPU (fp32 MIOpen convolutions vs the reference's CPU result)
    from sdirt_amd.dfdp import Basenet
    torch.manual_seed(int(fx["seed"]))
    base = Basenet(train_mode="deblur").eval().to(dev)
    with torch.no_grad():
        base.deblur_net.cam_attention.gamma.fill_(0.3)
        losses, outputs = base.dfdp({"stack_rgb_img": torch.cat((xl, xr), 1).to(dev), "AiF_img": xl.to(dev),
                                     "gt_depth": inputs.gt.clone().to(dev)}, train=True)
    assert losses["total"].item() == pytest.approx(float(fx["deblur_loss_total"]), rel=2e-3)
    assert np.abs(outputs["pred_aif"][0, :, ::16, ::16].cpu().numpy() - fx["pred_aif_head"]).max() < 2e-3


@pytest.mark.gpu
def test_inference_layout_one_feature_pass_pixel_major_tensors():
    """DfDPNet in eval mode on the GPU: both views through the feature network in one pass, channels_last / channels_last_3d
    tensors, the cost volume written pixel-major by sdirt_dp_cost_volume_nhwc -- element for element the volume of the
    planar kernel (16-byte and element-wise forms, fp16 and fp32, ragged widths), the network's output that of the
    reference's two calls within fp32 convolution noise, parameters (names, values) untouched."""
    from sdirt_amd.dfdp import DfDPNet, dp_cost_volume
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    for shape, d in (((1, 32, 16, 24), 20), ((2, 8, 5, 33), 20), ((2, 5, 7, 33), 6), ((1, 4, 3, 9), 6), ((1, 3, 2, 130), 12)):
        for dt in (torch.float32, torch.float16):
            a, b = torch.rand(*shape, generator=g).to(dev, dt), torch.rand(*shape, generator=g).to(dev, dt)
            want = dp_cost_volume(a, b, d)
            both = torch.cat((a, b)).contiguous(memory_format=torch.channels_last)       # slices of one pixel-major tensor
            n = shape[0]
            got = dp_cost_volume(both[:n], both[n:], d)
            assert got.is_contiguous(memory_format=torch.channels_last_3d) and not got.is_contiguous()
            assert torch.equal(got, want), (shape, dt)
            # the adjoint of the pixel-major form is the planar kernel's
            wgt = torch.rand(want.shape, generator=g).to(dev, dt)
            a1, b1 = a.clone().requires_grad_(), b.clone().requires_grad_()
            (dp_cost_volume(a1, b1, d) * wgt).sum().backward()
            a2 = a.contiguous(memory_format=torch.channels_last).requires_grad_()
            b2 = b.contiguous(memory_format=torch.channels_last).requires_grad_()
            (dp_cost_volume(a2, b2, d) * wgt).sum().backward()
            assert torch.equal(a1.grad, a2.grad) and torch.equal(b1.grad, b2.grad)
    fx = load_golden("f10_dfdp_net")
    net = build(fx).to(dev)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    xl, xr, _, _ = inputs(fx)
    xl, xr = xl.to(dev), xr.to(dev)
    with torch.no_grad():
        net.inference_layout = False
        two = net(xl, xr)
        net.inference_layout = True
        one = net(xl, xr)
        assert net._laid_out and net.feature.start[0].conv.weight.is_contiguous(memory_format=torch.channels_last)
        assert (one - two).abs().max().item() < 1e-3 and np.abs(one.cpu().numpy() - fx["disp"]).max() < 2e-3
        with torch.autocast("cuda", dtype=torch.float16):
            one16 = net(xl, xr)
        assert np.abs(one16.float().cpu().numpy() - fx["disp"]).max() < 5e-2
    after = net.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    net.train()                                                  # training: the reference's two calls (batch statistics per view)
    out = net(torch.cat((xl, xl.flip(-1))), torch.cat((xr, xr.flip(-1))))     # (two samples: a 1 x 1 context map needs them)
    assert out.requires_grad and torch.isfinite(out).all()
    # ... on planar weights again (the layout follows the path), the same Parameter objects, the same values
    assert not net._laid_out and net.feature.start[0].conv.weight.is_contiguous()
    assert all(torch.equal(net.state_dict()[k], before[k]) for k in before if "running" not in k and "num_batches" not in k)


@pytest.mark.gpu
def test_inference_fusions_equal_the_torch_ops_they_stand_for():
    """Eval mode under torch.no_grad() on the GPU: batch norm + ReLU in one in-place pass (sdirt_bn_relu), Conv2x's
    upsampling between channels_last_3d volumes (sdirt_upsample_trilinear_ndhwc), Disp as one kernel
    (sdirt_disparity_regression), cached fp16 weights under autocast -- each against the torch ops of dddnet.py:539-568,
    585-589 on the same tensors, then the whole network with the switch on and off."""
    import torch.nn as nn
    import torch.nn.functional as F
    import sdirt_amd.dfdp as D
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(3)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    # --- batch norm + ReLU: planar and pixel-major, 16-byte and element-wise forms, with and without ReLU
    for shape in ((2, 32, 6, 16, 24), (1, 64, 10, 32, 48), (2, 32, 17, 24), (1, 5, 7, 9), (2, 3, 3, 5, 7)):
        C = shape[1]
        bn = (nn.BatchNorm3d if len(shape) == 5 else nn.BatchNorm2d)(C).to(dev).eval()
        with torch.no_grad():
            bn.running_mean.copy_(rnd(C)); bn.running_var.copy_(rnd(C).abs() + 0.3)
            bn.weight.copy_(rnd(C)); bn.bias.copy_(rnd(C))
        for dt, tol in ((torch.float32, 2e-6), (torch.float16, 1e-3)):
            for fmt in (torch.contiguous_format, torch.channels_last_3d if len(shape) == 5 else torch.channels_last):
                for relu in (True, False):
                    x = (3 * rnd(*shape)).to(dt).contiguous(memory_format=fmt)
                    with torch.no_grad():
                        want = bn(x.float())
                        want = F.relu(want) if relu else want
                        got = D._bn_relu_(x.clone(memory_format=torch.preserve_format), bn, relu)
                    assert got.dtype == dt and got.stride() == x.stride()
                    err = (got.float() - want).abs().max().item()
                    assert err <= tol * max(1.0, want.abs().max().item()), (shape, dt, fmt, relu, err)
    # a statistics update is seen (the tables are cached per module)
    with torch.no_grad():
        x = rnd(2, 32, 17, 24)
        a = D._bn_relu_(x.clone(), bn2 := nn.BatchNorm2d(32).to(dev).eval(), True)
        bn2.running_mean.add_(1.0)
        b = D._bn_relu_(x.clone(), bn2, True)
        assert not torch.equal(a, b) and torch.allclose(b, F.relu(bn2(x)), atol=1e-6)
    # --- Conv2x's upsampling on channels_last_3d volumes
    up = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
    conv2x = D.Conv2x(64, 64).to(dev).eval()
    for shape in ((1, 64, 5, 32, 48), (2, 8, 3, 5, 7), (1, 5, 2, 3, 4), (1, 4, 1, 6, 5)):
        for dt, tol in ((torch.float32, 2e-6), (torch.float16, 1e-3)):
            x = rnd(*shape).to(dt).contiguous(memory_format=torch.channels_last_3d)
            with torch.no_grad():
                got = conv2x._up2(x)
                want = up(x.float())
            assert got.shape == want.shape and got.dtype == dt and got.is_contiguous(memory_format=torch.channels_last_3d)
            assert (got.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), (shape, dt)
    # --- Disp: trilinear x (2, 4, 4) + softmin + expectation over the shifts
    def disp_ref(x, maxdisp):
        x = F.interpolate(x.float(), [maxdisp, x.shape[3] * 4, x.shape[4] * 4], mode="trilinear", align_corners=False)
        p = F.softmin(torch.squeeze(x, 1), dim=1)
        shifts = torch.arange(-maxdisp // 2, maxdisp // 2, device=x.device).view(1, -1, 1, 1)
        return torch.sum(p * shifts, 1, keepdim=True)
    for shape, md in (((1, 1, 20, 32, 48), 20), ((1, 1, 10, 32, 48), 20), ((2, 1, 10, 8, 12), 20), ((1, 1, 3, 5, 7), 5), ((1, 1, 1, 2, 3), 4), ((1, 1, 32, 4, 4), 64)):
        for dt in (torch.float32, torch.float16):
            x = (4 * rnd(*shape)).to(dt)
            with torch.no_grad():
                got = D.Disp(md).to(dev)(x)
                want = disp_ref(x, md)
            assert got.shape == want.shape and got.dtype == torch.float32
            assert (got - want).abs().max().item() <= 2e-5 * md, (shape, md, dt, (got - want).abs().max().item())
    # --- the whole network, switch on / off (fp32, then as Basenet.forward runs it: fp16 autocast)
    fx = load_golden("f10_dfdp_net")
    net = build(fx).to(dev)
    xl, xr, _, _ = inputs(fx)
    xl, xr = xl.to(dev), xr.to(dev)
    try:
        with torch.no_grad():
            D.inference_fusions = False
            off = net(xl, xr)
            with torch.autocast("cuda", dtype=torch.float16):
                off16 = net(xl, xr)
            D.inference_fusions = True
            on = net(xl, xr)
            with torch.autocast("cuda", dtype=torch.float16):
                on16 = net(xl, xr)
                assert "_sdirt_w16" in net.matching.start[0].conv.__dict__               # cached fp16 weights in use
                on16b = net(xl, xr)
    finally:
        D.inference_fusions = True
    assert (on - off).abs().max().item() < 2e-4 and np.abs(on.cpu().numpy() - fx["disp"]).max() < 2e-3
    assert (on16 - on16b).abs().max().item() < 2e-3          # (MIOpen's picks need not repeat bit for bit)
    assert (on16 - off16).abs().max().item() < 2e-2 and np.abs(on16.float().cpu().numpy() - fx["disp"]).max() < 5e-2
    # a weight update invalidates the cached copy
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        net.matching.start[0].conv.weight.mul_(1.5)
        assert not torch.equal(net(xl, xr), on16)
    # with autograd on (fine-tuning with frozen statistics) the differentiable torch ops run
    out = net(xl, xr)
    assert out.requires_grad
    # ... and a training step goes through the reference's op sequence whatever layout the weights were left in (the
    # inference calls above re-laid them): two samples, fp16 autocast, loss scaling as 2_dfdp_net.py's loop
    net.train()
    opt = torch.optim.AdamW(net.parameters(), 1e-4)
    scaler = torch.amp.GradScaler("cuda")
    x2, y2 = torch.cat((xl, xl.flip(-1))), torch.cat((xr, xr.flip(-1)))
    with torch.autocast("cuda", dtype=torch.float16):
        loss = (net(x2, y2).float() + 0.5).abs().mean()
    scaler.scale(loss).backward()
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert len(grads) > 40 and all(torch.isfinite(g).all() for g in grads) and any(float(g.abs().max()) > 0 for g in grads)
    scaler.step(opt)
    scaler.update()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(128, 128), (256, 384), (192, 320)])
def test_inference_path_reproduces_the_hourglass_output_not_only_the_disparity(size):
    """With seeded weights the softmin is nearly uniform (the fixture's disparities are -0.5 +- 3e-4), so a disparity map
    says little about the tensors behind it.  This compares what Disp RECEIVES -- the hourglass's [B, 1, 20, H/4, W/4] volume --
    between the reference's op sequence on the CPU (the path the F10 fixture pins) and the GPU inference path (one feature
    pass, pixel-major tensors, fused batch norm / upsampling, cached fp16 weights), relative to the volume's own scale."""
    import sdirt_amd.dfdp as D
    fx = load_golden("f10_dfdp_net")
    net = build(fx)
    g = torch.Generator().manual_seed(7)
    B = 2 if size == (192, 320) else 1             # (192 x 320: a batch of two, quarter-size maps the 32 x 32 windows do not divide)
    xl = torch.rand(B, 3, *size, generator=g)
    xr = torch.roll(xl, 2, dims=-1) * 0.9 + 0.05 * torch.rand(B, 3, *size, generator=g)
    seen = {}
    hook = net.disp.register_forward_pre_hook(lambda m, args: seen.__setitem__("x", args[0].detach().float().cpu().contiguous()))
    try:
        with torch.no_grad():
            net(xl, xr)
            want = seen.pop("x")
            net = net.to("cuda:0")
            net(xl.cuda(), xr.cuda())
            got32 = seen.pop("x")
            with torch.autocast("cuda", dtype=torch.float16):
                net(xl.cuda(), xr.cuda())
            got16 = seen.pop("x")
            D.inference_fusions = False
            net.inference_layout = False
            with torch.autocast("cuda", dtype=torch.float16):
                net(xl.cuda(), xr.cuda())
            ref16 = seen.pop("x")
    finally:
        hook.remove()
        D.inference_fusions = True
    scale = want.abs().max().item()
    assert want.shape == (B, 1, 20, size[0] // 4, size[1] // 4) and want.std().item() > 0.02 * scale      # a tensor with structure
    e32 = (got32 - want).abs().max().item() / scale
    e16 = (got16 - want).abs().max().item() / scale
    r16 = (ref16 - want).abs().max().item() / scale
    print(f"hourglass output {tuple(want.shape)}: scale {scale:.3e}; GPU inference path vs CPU reference ops: fp32 {e32:.2e}, "
          f"fp16 autocast {e16:.2e} (the reference's op sequence under the same autocast: {r16:.2e}) of the scale")
    assert e32 < 2e-3
    assert e16 < max(3e-2, 2.0 * r16)


@pytest.mark.gpu
def test_cost_volume_other_cuda_dtypes_take_the_reference_formulation():
    """bf16 autocast and float64 gradcheck of the depth network on the GPU: dtypes the
    kernel is not built for go through the reference's own slice-assignment formulation."""
    from sdirt_amd.dfdp import dp_cost_volume, _cost_volume_reference
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 6, 12, generator=g).cuda()
    y = torch.randn(1, 4, 6, 12, generator=g).cuda()
    want = dp_cost_volume(x, y, 8)
    for dt in (torch.bfloat16, torch.float64):
        got = dp_cost_volume(x.to(dt), y.to(dt), 8)
        assert got.dtype == dt and torch.equal(got, _cost_volume_reference(x.to(dt), y.to(dt), 8))
        assert torch.allclose(got.float(), want, atol=2e-2)
    xd = x.double().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a: dp_cost_volume(a, y.double(), 8), (xd,), nondet_tol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("k,shape", [(32, (1, 128, 128, 192)), (8, (2, 128, 128, 192)), (4, (1, 3, 12, 20)), (1, (1, 2, 5, 7))])
def test_window_average_equals_avg_pool2d(dtype, k, shape):
    """WindowAverage (sdirt_avg_pool_windows) against nn.AvgPool2d((k, k), stride=(k, k)) -- the two context branches of the
    feature extractor (dddnet.py:376-385) -- forward within one rounding of the result type (fp32 accumulation in another
    order), the adjoint exactly; shapes the kernel does not take (a side the window does not divide) go to torch."""
    from sdirt_amd.dfdp import WindowAverage
    g = torch.Generator(device="cuda").manual_seed(k)
    x = torch.randn(shape, device="cuda", generator=g).to(dtype)
    ours, ref = WindowAverage((k, k), stride=(k, k)), torch.nn.AvgPool2d((k, k), stride=(k, k))
    a, b = ours(x), ref(x)
    assert a.shape == b.shape and a.dtype == b.dtype
    tol = 2e-3 if dtype == torch.float16 else 2e-6
    assert (a.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item())
    x1 = x.float().clone().requires_grad_(True)
    x2 = x.float().clone().requires_grad_(True)
    w = torch.randn(b.shape, device="cuda", generator=g)
    (ours(x1) * w).sum().backward()
    (ref(x2) * w).sum().backward()
    assert torch.allclose(x1.grad, x2.grad, rtol=0, atol=1e-7)
    if shape[1] % 8 == 0:                                    # the same map pixel-major (DfDPNet's inference layout)
        c = ours(x.contiguous(memory_format=torch.channels_last))
        assert c.shape == b.shape and c.is_contiguous()
        assert (c.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item())
    odd = torch.randn(1, 2, 10, 13, device="cuda").to(dtype)
    assert torch.equal(WindowAverage((4, 4), stride=(4, 4))(odd), torch.nn.AvgPool2d((4, 4), stride=(4, 4))(odd))
