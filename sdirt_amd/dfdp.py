"""Depth-from-dual-pixel network (consumer of the simulated DP pairs).

`DfDPNet` is the reference's YRStereonet_3D (dfdp/dddnet/dddnet.py:103-152, 358-568) with the
same parameter names, so its checkpoints load unchanged:

  feature   2-D CNN to 1/4 resolution, 32 channels (3 stride-1/2 convs, 2 dilated convs, two
            average-pool context branches, fusion)                        dddnet.py:358-407
  cost      signed-shift DP cost volume [B, 64, 20, H/4, W/4]              dddnet.py:136-148
            -> ONE HIP kernel (sdirt_dp_cost_volume) instead of zero-fill + 40 sliced copies
  matching  3-D conv hourglass -> [B, 1, 20, H/4, W/4]                     dddnet.py:409-446
  disp      trilinear x(2, 4, 4) upsample, softmin over the 20 shifts, expectation over
            shifts -10..9                                                  dddnet.py:543-568

The convolutions are stock torch.nn (MIOpen on ROCm) -- dense conv work the vendor library
covers.  Unlike the reference (dddnet.py:564 hard-codes torch.cuda.current_device()), the
module also runs on the CPU, where the cost volume falls back to the reference's slice copies
(used by the CPU tests of the host logic only).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .basics import dptr, stream_ptr


def _cost_volume_reference(x, y, d_max):
    """The signed-shift cost volume in stock torch ops (zero-fill, then 2 * d_max slice copies, as
    dddnet.py:136-148 builds it).  DfDPNet is a stock-torch module (MIOpen / CPU convolutions) and
    runs wherever torch runs; this is what CPU tensors take.  CUDA tensors always take the HIP
    kernel -- there is no silent fallback from it."""
    B, C, H, W = x.shape
    cost = torch.zeros(B, C * 2, d_max, H, W).type_as(x)
    for i in range(d_max):
        gap = i - d_max // 2
        keep = slice(None, gap) if gap < 0 else slice(gap, None)
        cost[:, :C, i, :, keep] = x[:, :, :, keep]
        cost[:, C:, i, :, keep] = y[:, :, :, -gap:] if gap < 0 else (y[:, :, :, :-gap] if gap > 0 else y)
    return cost


class _CostVolume(torch.autograd.Function):
    """sdirt_dp_cost_volume / sdirt_dp_cost_volume_backward (one kernel each way)."""

    @staticmethod
    def forward(ctx, x, y, d_max):
        B, C, H, W = x.shape
        y = y.to(x.dtype)
        ctx.shape, ctx.d_max = (B, C, H, W), d_max
        half = 1 if x.dtype == torch.float16 else 0
        cl = torch.channels_last
        if not x.is_contiguous() and x.is_contiguous(memory_format=cl) and y.is_contiguous(memory_format=cl):
            # pixel-major feature maps (the inference layout of DfDPNet) -> a channels_last_3d volume: what the
            # hourglass's convolutions compute in, nothing transposed in between
            cost = torch.empty((B, 2 * C, d_max, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last_3d)
            _lib.check(_lib.lib().sdirt_dp_cost_volume_nhwc(dptr(x), dptr(y), B, C, d_max, H, W, half,
                                                            dptr(cost), stream_ptr(x.device)))
            return cost
        x, y = x.contiguous(), y.contiguous()
        cost = torch.empty((B, 2 * C, d_max, H, W), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().sdirt_dp_cost_volume(dptr(x), dptr(y), B, C, d_max, H, W, half,
                                                   dptr(cost), stream_ptr(x.device)))
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        B, C, H, W = ctx.shape
        g = grad_cost.contiguous()
        gx = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device)
        gy = torch.empty_like(gx)
        _lib.check(_lib.lib().sdirt_dp_cost_volume_backward(dptr(g), B, C, ctx.d_max, H, W,
                                                            1 if g.dtype == torch.float16 else 0,
                                                            dptr(gx), dptr(gy), stream_ptr(g.device)))
        return gx, gy, None


def dp_cost_volume(x, y, d_max=20):
    """dddnet.py:155-178 / 136-148: x, y [B,C,H,W] -> [B,2C,d_max,H,W] (differentiable)."""
    if x.is_cuda and x.dtype in (torch.float16, torch.float32):
        return _CostVolume.apply(x, y, d_max)
    # CPU tensors, and the dtypes the kernel is not built for (bf16 autocast, float64 gradcheck):
    # the reference's own formulation in stock torch ops -- a documented path chosen by dtype,
    # never a fallback from a failed launch (the fp16 / fp32 kernel raises on any error)
    return _cost_volume_reference(x, y, d_max)


class _WindowAverage(torch.autograd.Function):
    """sdirt_avg_pool_windows; the adjoint spreads grad / k^2 over each window."""

    @staticmethod
    def forward(ctx, x, k):
        B, C, H, W = x.shape
        ctx.k = k
        vl = 8 if x.dtype == torch.float16 else 4
        if (not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last) and C % vl == 0 and C // vl <= 256
                and x.data_ptr() % 16 == 0):
            # a pixel-major map (the inference layout of DfDPNet): averaged as it lies (no re-laying copy of the map in front)
            out = torch.empty((B, C, H // k, W // k), dtype=x.dtype, device=x.device)
            _lib.check(_lib.lib().sdirt_avg_pool_windows_nhwc(dptr(x), B, H, W, C, k, 1 if x.dtype == torch.float16 else 0,
                                                              dptr(out), stream_ptr(x.device)))
            return out
        x = x.contiguous()
        out = torch.empty((B, C, H // k, W // k), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().sdirt_avg_pool_windows(dptr(x), B * C, H, W, k, 1 if x.dtype == torch.float16 else 0,
                                                     dptr(out), stream_ptr(x.device)))
        return out

    @staticmethod
    def backward(ctx, grad):
        k = ctx.k
        return (grad / (k * k)).repeat_interleave(k, dim=-2).repeat_interleave(k, dim=-1), None


class WindowAverage(nn.AvgPool2d):
    """nn.AvgPool2d((k, k), stride=(k, k)) (dddnet.py:376-385: the context branches pool over 32 x 32 and 8 x 8 windows).
    CUDA fp16 / fp32 maps whose sides the window divides take sdirt_avg_pool_windows (torch's kernel gives one thread a
    whole window: 240 us per call at 512 x 768 against ~10); anything else is nn.AvgPool2d itself.  No parameters: the
    reference's checkpoints load unchanged."""

    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        square = (self.kernel_size, self.stride) in ((k, k), ((k, k), (k, k)), ((k, k), k), (k, (k, k)))
        if (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.float32) and square and self.padding in (0, (0, 0))
                and x.shape[-2] % k == 0 and x.shape[-1] % k == 0 and x.shape[-1] <= 12288):
            return _WindowAverage.apply(x, k)
        return super().forward(x)


#: eval mode, CUDA tensors, no autograd (torch.no_grad()): BasicConv's batch norm + ReLU as ONE in-place pass
#: (sdirt_bn_relu), its convolution with a cached fp16 copy of the weights under fp16 autocast (autocast re-casts every
#: weight on every entry of an autocast region: 22 casts per frame), Conv2x's upsampling between channels_last_3d volumes
#: (sdirt_upsample_trilinear_ndhwc) and Disp as one kernel (sdirt_disparity_regression).  Same arithmetic in the same
#: precision as the torch ops they stand for; False: the torch ops.
inference_fusions = True


def _hooked(m):
    """Forward hooks on a submodule (feature taps, quantisation observers ...) see the module called: no fusion past them."""
    return bool(m._forward_hooks or m._forward_pre_hooks)


def _layout(x):
    """(outer, inner) of a [B, C, *spatial] tensor as sdirt_bn_relu counts them, or None for strides it does not take."""
    spatial = 1
    for n in x.shape[2:]:
        spatial *= n
    if x.is_contiguous():
        return x.shape[0], spatial
    fmt = {4: torch.channels_last, 5: torch.channels_last_3d}.get(x.dim())
    vl = 8 if x.dtype == torch.float16 else 4
    if fmt is not None and x.is_contiguous(memory_format=fmt) and (x.shape[1] <= 256 or (x.shape[1] % vl == 0 and x.shape[1] <= 256 * vl)):
        return x.shape[0] * spatial, 1
    return None


def _bn_tables(bn, device):
    """(mean, invstd, gamma, beta) fp32 on `device`, cached on the module until one of its tensors is written."""
    src = (bn.running_mean, bn.running_var, bn.weight, bn.bias)
    key = (str(device), float(bn.eps)) + tuple((t._version, t.data_ptr()) if t is not None else None for t in src)
    hit = bn.__dict__.get("_sdirt_tables")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            mean = bn.running_mean.detach().to(device, torch.float32).contiguous()
            invstd = torch.rsqrt(bn.running_var.detach().to(device, torch.float32) + bn.eps).contiguous()
            gamma = bn.weight.detach().to(device, torch.float32).contiguous() if bn.weight is not None else torch.ones_like(mean)
            beta = bn.bias.detach().to(device, torch.float32).contiguous() if bn.bias is not None else torch.zeros_like(mean)
        hit = bn.__dict__["_sdirt_tables"] = (key, (mean, invstd, gamma, beta))
    return hit[1]


def _bn_relu_(x, bn, relu):
    """dddnet.py:539-543 on a convolution's output, eval mode: one in-place pass; shapes / dtypes / strides the kernel does
    not take go through the torch ops (chosen here, before anything is launched)."""
    if bn is None:
        return F.relu(x, inplace=True) if relu else x
    lay = _layout(x) if x.dtype in (torch.float16, torch.float32) and bn.track_running_stats and bn.running_mean is not None else None
    if lay is None:
        x = bn(x)
        return F.relu(x, inplace=True) if relu else x
    mean, invstd, gamma, beta = _bn_tables(bn, x.device)
    _lib.check(_lib.lib().sdirt_bn_relu(dptr(x), lay[0], x.shape[1], lay[1], dptr(mean), dptr(invstd), dptr(gamma), dptr(beta),
                                        1 if relu else 0, 1 if x.dtype == torch.float16 else 0, stream_ptr(x.device)))
    return x


def _conv_cached(conv, x):
    """conv(x); under fp16 autocast with a cached fp16 copy of the (fp32) weight -- the cast autocast would make, made once
    per weight update instead of once per autocast region."""
    w = conv.weight
    if not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16 and w.dtype == torch.float32
            and conv.bias is None and getattr(conv, "padding_mode", "zeros") == "zeros"):
        return conv(x)
    key = (w._version, w.data_ptr(), w.stride())
    hit = conv.__dict__.get("_sdirt_w16")
    if hit is None or hit[0] != key:
        hit = conv.__dict__["_sdirt_w16"] = (key, w.detach().to(torch.float16))           # (keeps the memory format)
    w16 = hit[1]
    if isinstance(conv, nn.ConvTranspose3d):
        return F.conv_transpose3d(x, w16, None, conv.stride, conv.padding, conv.output_padding, conv.groups, conv.dilation)
    if isinstance(conv, nn.ConvTranspose2d):
        return F.conv_transpose2d(x, w16, None, conv.stride, conv.padding, conv.output_padding, conv.groups, conv.dilation)
    if isinstance(conv, nn.Conv3d):
        return F.conv3d(x, w16, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if isinstance(conv, nn.Conv2d):
        return F.conv2d(x, w16, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return conv(x)


class BasicConv(nn.Module):
    """conv (2-D / 3-D, optionally transposed, no bias) [+ batch norm] [+ ReLU]; dddnet.py:513-541."""

    def __init__(self, cin, cout, deconv=False, is_3d=False, bn=True, relu=True, **kw):
        super().__init__()
        conv = {(False, False): nn.Conv2d, (False, True): nn.ConvTranspose2d,
                (True, False): nn.Conv3d, (True, True): nn.ConvTranspose3d}[(is_3d, deconv)]
        self.conv = conv(cin, cout, bias=False, **kw)
        self.bn = (nn.BatchNorm3d if is_3d else nn.BatchNorm2d)(cout)
        self.use_bn, self.relu = bn, relu

    def forward(self, x):
        if (inference_fusions and x.is_cuda and not self.training and not torch.is_grad_enabled()
                and not _hooked(self.conv) and not _hooked(self.bn)):
            return _bn_relu_(_conv_cached(self.conv, x), self.bn if self.use_bn else None, self.relu)
        x = self.conv(x)
        if self.use_bn:
            x = self.bn(x)
        return F.relu(x, inplace=True) if self.relu else x


def _convbn(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, dilation=1, bias=False),
                         nn.BatchNorm2d(cout))


class Feature(nn.Module):
    def __init__(self):
        super().__init__()
        self.start = nn.Sequential(BasicConv(3, 32, kernel_size=3, padding=1),
                                   BasicConv(32, 64, kernel_size=3, stride=1, padding=1),
                                   BasicConv(64, 64, kernel_size=3, stride=2, padding=1))
        self.layer1 = nn.Sequential(BasicConv(64, 128, kernel_size=3, stride=1, padding=4, dilation=4),
                                    BasicConv(128, 128, kernel_size=3, stride=1, padding=8, dilation=8),
                                    BasicConv(128, 128, kernel_size=3, stride=2, padding=1))
        self.branch1 = nn.Sequential(WindowAverage((32, 32), stride=(32, 32)), _convbn(128, 32),
                                     nn.ReLU(inplace=True))
        self.branch3 = nn.Sequential(WindowAverage((8, 8), stride=(8, 8)), _convbn(128, 32),
                                     nn.ReLU(inplace=True))
        self.end = nn.Sequential(BasicConv(192, 96, kernel_size=3, stride=1, padding=1),
                                 BasicConv(96, 32, kernel_size=1, bn=False, relu=False, padding=0))

    def forward(self, x):
        x = self.layer1(self.start(x))
        size = x.shape[2:]
        ctx = [F.interpolate(b(x), size, mode="bilinear", align_corners=True)
               for b in (self.branch1, self.branch3)]
        return self.end(torch.cat(ctx + [x], 1))


class Conv2x(nn.Module):
    """Upsample x2 (trilinear), conv, concatenate the skip tensor, conv; dddnet.py:570-602."""

    def __init__(self, cin, cout, is_3d=True):
        super().__init__()
        self.conv1 = BasicConv(cin, cout, False, is_3d, kernel_size=3, stride=1, padding=1)
        self.conv2 = BasicConv(cout * 2, cout, False, is_3d, kernel_size=3, stride=1, padding=1)
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)

    def _up2(self, x):
        if (inference_fusions and x.is_cuda and x.dim() == 5 and not torch.is_grad_enabled() and not x.is_contiguous()
                and x.is_contiguous(memory_format=torch.channels_last_3d) and x.dtype in (torch.float16, torch.float32)):
            B, C, D, H, W = x.shape
            out = torch.empty((B, C, 2 * D, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last_3d)
            _lib.check(_lib.lib().sdirt_upsample_trilinear_ndhwc(dptr(x), B, C, D, H, W, 2 * D, 2 * H, 2 * W,
                                                                 1 if x.dtype == torch.float16 else 0, dptr(out), stream_ptr(x.device)))
            return out
        return self.up2(x)

    def forward(self, x, rem):
        x = self.conv1(self._up2(x))
        assert x.size() == rem.size()
        return self.conv2(torch.cat((x, rem), 1))


class Matching(nn.Module):
    def __init__(self):
        super().__init__()
        c3 = dict(is_3d=True, kernel_size=3, padding=1)
        self.start = nn.Sequential(BasicConv(64, 32, **c3), BasicConv(32, 48, stride=2, **c3),
                                   BasicConv(48, 64, **c3))
        self.conv1a = nn.Sequential(BasicConv(64, 64, stride=2, **c3), BasicConv(64, 64, **c3))
        self.deconv1a = Conv2x(64, 64, is_3d=True)
        self.end = nn.Sequential(
            BasicConv(64, 64, is_3d=True, kernel_size=4, padding=1, stride=2, deconv=True),
            BasicConv(64, 1, is_3d=True, kernel_size=3, padding=1, stride=1, bn=False, relu=False))

    def forward(self, x):
        x = self.start(x)
        return self.end(self.deconv1a(self.conv1a(x), x))


class Disp(nn.Module):
    def __init__(self, maxdisp=20):
        super().__init__()
        self.maxdisp = maxdisp

    def forward(self, x):
        if (inference_fusions and x.is_cuda and not torch.is_grad_enabled() and x.shape[1] == 1 and x.shape[2] <= 32
                and self.maxdisp <= 64 and x.dtype in (torch.float16, torch.float32)):
            x = x.contiguous()
            B, _, D0, H0, W0 = x.shape
            out = torch.empty((B, 1, 4 * H0, 4 * W0), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().sdirt_disparity_regression(dptr(x), B, D0, H0, W0, self.maxdisp, 4 * H0, 4 * W0,
                                                             1 if x.dtype == torch.float16 else 0, dptr(out), stream_ptr(x.device)))
            return out
        x = F.interpolate(x, [self.maxdisp, x.shape[3] * 4, x.shape[4] * 4], mode="trilinear",
                          align_corners=False)
        p = F.softmin(torch.squeeze(x, 1), dim=1)
        shifts = torch.arange(-self.maxdisp // 2, self.maxdisp // 2, device=x.device).view(1, -1, 1, 1)
        return torch.sum(p * shifts, 1, keepdim=True)


class DfDPNet(nn.Module):
    """YRStereonet_3D: (left, right) DP views [B,3,H,W] (H, W multiples of 128) -> signed
    disparity [B,1,H,W] in pixels of the quarter-resolution shift axis, range (-10, 9)."""

    def __init__(self, maxdisp=20):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature = Feature()
        self.matching = Matching()
        self.disp = Disp(maxdisp)
        for m in self.modules():                     # dddnet.py:115-120
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    #: inference on the GPU (eval mode): BOTH views through the feature network in ONE pass (a batch of two: batch norm
    #: uses its running statistics in eval mode, so the halves are what two calls give), the 2-D network in channels_last,
    #: the cost volume and the 3-D hourglass in channels_last_3d -- the layouts MIOpen's MI355X convolution kernels
    #: compute in (no transposes around them).  512 x 768, fp16 autocast, after the find pass: 2.50 -> 1.94 ms, 1.45 with
    #: `inference_fusions` (tools/dfdp_ab.py, profiles/r06/dfdp_ab.txt).  Training keeps the reference's two calls (batch
    #: statistics per view).
    inference_layout = True

    def _lay_out(self, pixel_major):
        """The weights' memory format follows the path that is about to run: channels_last / channels_last_3d for the inference
        path, planar for the reference's op sequence (training above all: MIOpen's training kernels were seen to fault on
        pixel-major weights with small maps under fp16 autocast).  Strides only -- values, names, dtypes, the Parameter
        objects an optimiser holds stay; a switch re-lays ~5 MB, once per change of mode."""
        if getattr(self, "_laid_out", False) != pixel_major:
            self.feature.to(memory_format=torch.channels_last if pixel_major else torch.contiguous_format)
            self.matching.to(memory_format=torch.channels_last_3d if pixel_major else torch.contiguous_format)
            self._laid_out = pixel_major

    def forward(self, xl, yr):
        if self.training or not self.inference_layout or not xl.is_cuda or xl.shape != yr.shape or xl.dtype != yr.dtype:
            self._lay_out(False)
            cost = dp_cost_volume(self.feature(xl), self.feature(yr), self.maxdisp)          # dddnet.py:123-148
            return self.disp(self.matching(cost))
        self._lay_out(True)
        B = xl.shape[0]
        f = self.feature(torch.cat((xl, yr)).contiguous(memory_format=torch.channels_last))
        return self.disp(self.matching(dp_cost_volume(f[:B], f[B:], self.maxdisp)))


class Basenet(nn.Module):
    """dfdp/basenet.py:9-105, depth-estimation mode: the DP pair rendered by PSFNet.render
    ([B,6,H,W], left views then right views) -> log-depth estimate, SmoothL1 loss on the pixels
    with valid ground truth; train_mode='deblur' adds the Mydeblur branch (refined depth + deblurred
    image, basenet.py:29-31, 65-69)."""

    def __init__(self, train_mode="dfdp"):
        super().__init__()
        if train_mode not in ("dfdp", "deblur"):
            raise ValueError("train_mode must be 'dfdp' or 'deblur'")
        self.train_mode = train_mode
        self.dfdp_net = DfDPNet()
        self.deblur_net = Mydeblur()        # always constructed, as in the reference (checkpoint keys)

    def forward(self, input_dict):
        with torch.autocast("cuda", dtype=torch.float16, enabled=input_dict["stack_rgb_img"].is_cuda):
            return self.dfdp(input_dict, train=True)

    def linear(self, depth):
        """basenet.py:88-92: metres -> log-metres IN PLACE where depth is valid; remembers the mask."""
        self.mask = (depth > 1e-9).detach()
        depth[self.mask] = torch.log(depth[self.mask])
        return depth

    def inverse_linear(self, depth, mask=None):
        if mask is None:
            depth[self.mask] = torch.exp(depth[self.mask])
            return depth
        return torch.exp(depth)

    def compute_loss(self, results, gts):
        l1 = nn.SmoothL1Loss(reduction="mean")
        est = l1(results["pred_depth_est"][self.mask], gts["gt_depth"][self.mask])
        losses = {"depth_est": est, "total": est}
        if self.train_mode == "deblur":                                      # basenet.py:65-69
            losses["depth_fix"] = l1(results["pred_depth_fix"][self.mask], gts["gt_depth"][self.mask])
            losses["aif"] = l1(results["pred_aif"], gts["gt_aif"])
            losses["total"] = est * 2 + losses["depth_fix"] + losses["aif"]
        return losses

    def dfdp(self, input_dict, train=False):
        stack, gt_aif = input_dict["stack_rgb_img"], input_dict["AiF_img"]
        left, right = stack[:, 0:3], stack[:, 3:]
        gt_depth = self.linear(input_dict["gt_depth"])
        depth_est = self.dfdp_net(left, right)
        results = {"pred_depth_est": depth_est}
        if self.train_mode == "deblur":
            results["pred_depth_fix"], results["pred_aif"] = self.deblur_net(left, right, depth_est)
        losses = None
        if train:
            losses = self.compute_loss(results, {"gt_depth": gt_depth, "gt_aif": gt_aif})
        outputs = {"gt_depth": self.inverse_linear(gt_depth), "gt_aif": gt_aif, "gt_l": None, "gt_r": None,
                   "rt_render_l": left, "rt_render_r": right,
                   "pred_depth_est": self.inverse_linear(depth_est.to(torch.float32))}
        if self.train_mode == "deblur":
            outputs["pred_depth_fix"] = self.inverse_linear(results["pred_depth_fix"].to(torch.float32))
            outputs["pred_aif"] = results["pred_aif"]
        return losses, outputs

    def inference(self, input_dict):
        gt_depth, gt_aif = self.linear(input_dict["depth"]), input_dict["AiF_img"]
        stack = input_dict["stack_rgb_img"]
        left, right = stack[:, 0:3], stack[:, 3:]
        depth_est = self.dfdp_net(left, right)
        out = {"gt_depth": self.inverse_linear(gt_depth), "gt_aif": gt_aif, "gt_l": left, "gt_r": right,
               "rt_render_l": None, "rt_render_r": None,
               "pred_depth_est": self.inverse_linear(depth_est.to(torch.float32), mask=False)}
        if self.train_mode == "deblur":
            fix, aif = self.deblur_net(left, right, depth_est)
            out["pred_depth_fix"], out["pred_aif"] = self.inverse_linear(fix.to(torch.float32), mask=False), aif
        return out


# ---- optional deblurring branch (dddnet.py:15-100, 180-305) --------------------------------------
def _init_deblur(m):
    """dddnet.py:15-29 (`weight_init`)."""
    name = m.__class__.__name__
    if name.find("Conv") != -1 and hasattr(m, "kernel_size"):
        n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        m.weight.data.normal_(0, 0.5 * (2.0 / n) ** 0.5)
        if m.bias is not None:
            m.bias.data.zero_()
    elif name.find("BatchNorm") != -1:
        m.weight.data.fill_(1)
        m.bias.data.zero_()
    elif name.find("Linear") != -1:
        m.weight.data.normal_(0, 0.01)
        m.bias.data = torch.ones(m.bias.data.size())


def _res_pair(c):
    return nn.Sequential(nn.Conv2d(c, c, kernel_size=3, padding=1), nn.ReLU(),
                         nn.Conv2d(c, c, kernel_size=3, padding=1))


class Encoder(nn.Module):
    """Three resolution levels (32, 64, 128 channels), two residual conv pairs each."""

    def __init__(self, in_channel=3, out_channel=128):
        super().__init__()
        self.layer1 = nn.Conv2d(in_channel, 32, kernel_size=3, padding=1)
        self.layer2, self.layer3 = _res_pair(32), _res_pair(32)
        self.layer5 = nn.Conv2d(32, 64, kernel_size=3, stride=2, padding=1)
        self.layer6, self.layer7 = _res_pair(64), _res_pair(64)
        self.layer9 = nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1)
        self.layer10 = _res_pair(128)
        self.layer11 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=3, padding=1), nn.ReLU(),
                                     nn.Conv2d(128, out_channel, kernel_size=3, padding=1))

    def forward(self, x):
        x = self.layer1(x)
        for blk in (self.layer2, self.layer3):
            x = blk(x) + x
        x = self.layer5(x)
        for blk in (self.layer6, self.layer7):
            x = blk(x) + x
        x = self.layer9(x)
        for blk in (self.layer10, self.layer11):
            x = blk(x) + x
        return x


class Decoder(nn.Module):
    def __init__(self, in_channel=128, out_channel=3):
        super().__init__()
        self.layer13 = nn.Sequential(nn.Conv2d(in_channel, 128, kernel_size=3, padding=1), nn.ReLU(),
                                     nn.Conv2d(128, 128, kernel_size=3, padding=1))
        self.layer14 = _res_pair(128)
        self.layer16 = nn.ConvTranspose2d(128, 64, kernel_size=4, stride=2, padding=1)
        self.layer17, self.layer18 = _res_pair(64), _res_pair(64)
        self.layer20 = nn.ConvTranspose2d(64, 32, kernel_size=4, stride=2, padding=1)
        self.layer21, self.layer22 = _res_pair(32), _res_pair(32)
        self.layer24 = nn.Conv2d(32, out_channel, kernel_size=3, padding=1)

    def forward(self, x):
        for blk in (self.layer13, self.layer14):
            x = blk(x) + x
        x = self.layer16(x)
        for blk in (self.layer17, self.layer18):
            x = blk(x) + x
        x = self.layer20(x)
        for blk in (self.layer21, self.layer22):
            x = blk(x) + x
        return self.layer24(x)


class CAM_Module(nn.Module):
    """Channel attention: softmax(max(E) - E) V with E = X X^T over flattened pixels, gated by gamma."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        B, C, H, W = x.shape
        q = x.view(B, C, -1)
        energy = torch.bmm(q, q.permute(0, 2, 1))
        att = torch.softmax(torch.max(energy, -1, keepdim=True)[0].expand_as(energy) - energy, dim=-1)
        return self.gamma * torch.bmm(att, q).view(B, C, H, W) + x


class ConvBlock(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=1, bias=True, activation="prelu", norm=None):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride, padding, bias=bias)
        self.norm = norm
        if norm == "batch":
            self.bn = nn.BatchNorm2d(cout)
        elif norm == "instance":
            self.bn = nn.InstanceNorm2d(cout)
        self.activation = activation
        acts = {"relu": lambda: nn.ReLU(True), "prelu": nn.PReLU, "lrelu": lambda: nn.LeakyReLU(0.2, True),
                "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}
        if activation is not None:
            self.act = acts[activation]()

    def forward(self, x):
        x = self.conv(x)
        if self.norm is not None:
            x = self.bn(x)
        return self.act(x) if self.activation is not None else x


class Mydeblur(nn.Module):
    """dddnet.py:32-100: three-level patch hierarchy (quarters -> halves -> full frame) of
    encoder/decoder pairs on cat(left, right, disparity), a channel-attention branch on the view
    difference; returns (refined disparity [B,1,H,W], deblurred image [B,3,H,W])."""

    def __init__(self):
        super().__init__()
        self.feat = 128
        self.encoder1 = Encoder(7, self.feat).apply(_init_deblur)
        self.encoder2 = Encoder(7, self.feat).apply(_init_deblur)
        self.encoder3 = Encoder(7, self.feat).apply(_init_deblur)
        self.decoder3 = Decoder(self.feat, 7).apply(_init_deblur)
        self.decoder2 = Decoder(self.feat, 7).apply(_init_deblur)
        self.decoder1 = Decoder(self.feat, 3).apply(_init_deblur)
        self.decoderd = Decoder(self.feat, 1).apply(_init_deblur)
        self.cam_attention = CAM_Module(self.feat)
        self.down = ConvBlock(4, self.feat, 8, 4, 2, activation="sigmoid", norm=None)
        self.conv = ConvBlock(self.feat, 1, 3, 1, 1, activation="sigmoid", norm=None)

    def forward(self, image_left, image_right, est_blurdisp):
        H, W = image_left.shape[2:]
        lv1 = torch.cat((image_left, image_right, est_blurdisp), 1)
        top, bot = lv1[:, :, :H // 2], lv1[:, :, H // 2:]
        quads = [top[..., :W // 2], top[..., W // 2:], bot[..., :W // 2], bot[..., W // 2:]]
        f3 = [self.encoder3(q) for q in quads]
        f3_top, f3_bot = torch.cat(f3[:2], 3), torch.cat(f3[2:], 3)
        f3_all = torch.cat((f3_top, f3_bot), 2)
        r3_top, r3_bot = self.decoder3(f3_top), self.decoder3(f3_bot)
        f2_all = torch.cat((self.encoder2(top + r3_top), self.encoder2(bot + r3_bot)), 2) + f3_all
        f1 = self.encoder1(lv1 + self.decoder2(f2_all)) + f2_all
        att = self.cam_attention(self.down(torch.cat((image_left - image_right, est_blurdisp), 1)))
        return self.decoderd(f1 + att), self.decoder1(f1 + att)
