"""PSFNet (deeplens/psfnet.py): the lens + the network that represents its PSF field.

The lens half sets up the state the ray-traced PSF path needs and generates
training / test batches straight from the HIP kernels (already on the GPU).  The
network half -- init / load, the fitting loop, `pred` with the mirror trick for
the right sub-pixel, and `render` (depth map -> per-pixel L/R kernels -> the
per-pixel convolution kernel of render_psf.py) -- keeps the reference's
signatures; the MLP layers are stock torch.nn GEMMs.

Not rebuilt (comparison baselines outside the path, psfnet.py:381-527, 768-868):
pred_DPDNet / pred_Modeling / pred_Learn2reduce (need deeplens/related_psf),
ThinLens, the matplotlib visualisers, and the map-network variant
(get_training_psf_map / calc_psf_map, which the reference itself cannot run: wrong
kwarg at psfnet.py:307).
"""
import logging
import os

import numpy as np
import torch

from .optics import Lensgroup
from .psfnet_arch import MLP, MLPConv, initialize_weights
from .render_psf import local_psf_render_fast, psfnet_render

DMIN = 200      # [mm]  psfnet.py:15
DMAX = 20000    # [mm]  psfnet.py:16

_SIDE_PRIORITY = 0      # (tools/fit_budget_ab.py varies it)

_PSFNET_SENSOR_Z = {"rf35mm": 80.447, "rf50mm": 62.25}          # psfnet.py:42-48


class PSFNet(Lensgroup):
    """psfnet.py:18-57: Lensgroup + kernel size + depth range + the PSF network."""

    def __init__(self, filename, model_name="mlp", kernel_size=11, sensor_res=(512, 512),
                 device="cuda", **lens_kwargs):
        super().__init__(filename=filename, sensor_res=sensor_res, device=device, **lens_kwargs)
        self.in_features = 4
        #: render(): fuse pred's flip / normalise into the convolution kernel (GPU only)
        self.fused_render = True
        #: render(): evaluate the MLP with the single-kernel MFMA path instead of torch.nn layers
        self.fused_mlp = True
        self.kernel_size = kernel_size
        self.model_name = model_name
        self.init_net()
        self.spp = 4096
        self.patch_size = 64
        self.psf_grid = [sensor_res[0] // self.patch_size, sensor_res[1] // self.patch_size]
        self.d_max = -DMAX
        self.d_min = -DMIN
        for key, z in _PSFNET_SENSOR_Z.items():
            if filename.find(key) != -1:
                self.d_sensor = z
                break
        else:
            raise ValueError("filename is not correct")          # psfnet.py:46-48 prints and exits
        self.foc_d_arr = np.array([-999.9, -1000, -1000.1], dtype=np.float32) + self.d_sensor
        self.foc_z_arr = (self.foc_d_arr - self.d_min) / (self.d_max - self.d_min)
        self.foc_d = np.array([-1000.0], dtype=np.float32) + self.d_sensor

    def depth2z(self, depth):
        z = (depth - self.d_min) / (self.d_max - self.d_min)
        return torch.clamp(z, min=0, max=1)

    def z2depth(self, z):
        return z * (self.d_max - self.d_min) + self.d_min

    def _warp_z(self, z_gauss, foc_z):
        # psfnet.py:190-193 / 229-232: z[z_gauss > 0] = (1 - foc_z) * z_gauss / 3 + foc_z, z[z_gauss < 0] = foc_z * z_gauss / 3 + foc_z,
        # 0 where z_gauss == 0 -- the same elementwise arithmetic without the four boolean-mask gathers / scatters
        zero = torch.zeros_like(z_gauss)
        return torch.where(z_gauss > 0, (1 - foc_z) * z_gauss / 3 + foc_z, torch.where(z_gauss < 0, foc_z * z_gauss / 3 + foc_z, zero))

    def get_training_data(self, bs=256, spp=4096, _defer=False):
        """psfnet.py:170-202: (inp [bs,3] in [-1,1]^2 x [0,1], psf [bs,ks,ks] on the GPU).
        _defer: the second item is a PendingPSF (kernel enqueued, Newton trip check in .wait())."""
        foc_z = np.random.choice(self.foc_z_arr)
        x = (torch.rand(bs) - 0.5) * 2
        y = (torch.rand(bs) - 0.5) * 2
        z = self._warp_z(torch.clamp(torch.randn(bs), min=-3, max=3), foc_z)
        inp = torch.stack((x, y, z), dim=-1)
        points = torch.stack((x, y, self.z2depth(z)), dim=-1)
        if _defer:
            return inp, self.psf_diff(points=points, ks=self.kernel_size, spp=spp, _defer=True)
        return inp, self.psf(points=points, ks=self.kernel_size, spp=spp)

    def get_test_data(self, bs=1024, spp=65536):
        """psfnet.py:204-241: 32x32 grid paired ELEMENT-WISE with bs depths."""
        foc_z = self.foc_z_arr[1]
        g = 32
        x, y = torch.meshgrid(torch.linspace(-1 + 1 / (2 * g), 1 - 1 / (2 * g), g),
                              torch.linspace(1 - 1 / (2 * g), -1 + 1 / (2 * g), g), indexing="xy")
        x, y = x.reshape(-1), y.reshape(-1)
        z = self._warp_z(torch.linspace(-3, 3, bs), foc_z)
        inp = torch.stack((x, y, z), dim=-1)
        points = torch.stack((x, y, self.z2depth(z)), dim=-1)
        return inp, self.psf(points=points, ks=self.kernel_size, spp=spp)

    # ------------------------------------------------------------------ network
    def init_net(self):
        """psfnet.py:63-90.  Input (x, y, z) in [-1,1]^2 x [0,1] -> ks x ks kernel."""
        ks = self.kernel_size
        if self.model_name == "mlp":
            self.psfnet = MLP(in_features=3, out_features=ks ** 2, hidden_features=512,
                              hidden_layers=8)
        elif self.model_name == "mlpconv":
            self.psfnet = MLPConv(in_features=3, ks=ks, channels=1)
        elif self.model_name == "siren":
            raise NotImplementedError
        else:
            # 'mlp+lum' cannot be constructed in the reference either (psfnet_arch.py:63)
            raise Exception("Unsupported PSF network architecture.")
        self.psfnet.apply(initialize_weights)
        self.psfnet.to(self.device)

    def load_net(self, net_path):
        """psfnet.py:92-99: copy every tensor of the checkpoint whose shape matches."""
        own = self.psfnet.state_dict()
        ckpt = torch.load(net_path, map_location=self.device)
        own.update({k: v for k, v in ckpt.items() if k in own and own[k].shape == v.shape})
        self.psfnet.load_state_dict(own)
        if hasattr(self.psfnet, "invalidate_packed"):
            self.psfnet.invalidate_packed()

    def train_psfnet(self, iters=10000, bs=128, lr=1e-4, spp=2048, evaluate_every=1000,
                     result_dir="./results/temp", pipelined=None, figures=True):
        """psfnet.py:101-168: fit the network to PSFs ray-traced on the fly.  AdamW, cosine
        schedule over iters//3, MSE on max-normalised kernels, fp16 autocast + loss scaling on
        the GPU.  Every `evaluate_every` steps: checkpoint + L1/L2 of sum-normalised kernels
        on the 1024-point test set (logged) and the figure of five target / prediction pairs of the
        current batch (the prediction re-evaluated with the weights after the step).
        Returns the list of per-step training losses.

        pipelined (default: on for CUDA devices): the loop is launch-bound at the reference's
        batch size (64 rows through 11 small GEMMs, forward and backward), so the WHOLE step --
        forward, backward, the loss scaler's unscale / inf check, the fused AdamW update, the
        scaler's update -- is captured once in a hipGraph and replayed (the learning rate of the
        step is a device scalar the host fills from the schedule before each replay), and batch
        i+1 is ray-traced on a second stream while step i runs.  Same draws from the RNGs in the same order (prefetching stops at an
        evaluation step, whose test-set draws come first, and resumes after it), same arithmetic
        per step; `pipelined=False` is the plain loop.  figures=False: no matplotlib, no iterN.png (the
        checkpoints and the logged test errors stay); with figures on, matplotlib is imported before the first step,
        so a missing one fails at the start and not after evaluate_every steps."""
        psfnet = self.psfnet
        plt = None
        if figures:
            from .plots import _pyplot
            plt = _pyplot()
        psfnet.train()
        on_gpu = torch.device(self.device).type == "cuda"
        pipelined = on_gpu if pipelined is None else (pipelined and on_gpu)
        l2, l1 = torch.nn.MSELoss(reduction="mean"), torch.nn.L1Loss(reduction="mean")
        if pipelined:
            # the step runs inside a graph: the learning rate is a device scalar (a Python float would be frozen into the
            # capture), and the schedule is evaluated in Python floats on a stand-in optimiser -- the very numbers the plain
            # loop's scheduler hands its optimiser -- and filled into that scalar before each replay
            lr_dev = torch.tensor(float(lr), dtype=torch.float32, device=self.device)
            optim = torch.optim.AdamW(psfnet.parameters(), lr_dev, fused=True, capturable=True)
            shadow = torch.optim.SGD([torch.zeros(1)], lr)
            sche = torch.optim.lr_scheduler.CosineAnnealingLR(shadow, T_max=int(iters) // 3, eta_min=0)
            shadow.step()                                     # (nothing to update; tells the scheduler its order of calls is fine)
        else:
            optim = torch.optim.AdamW(psfnet.parameters(), lr)
            sche = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=int(iters) // 3, eta_min=0)
        scaler = torch.amp.GradScaler("cuda", enabled=on_gpu)
        amp = lambda: torch.autocast("cuda", dtype=torch.float16, enabled=on_gpu)  # noqa: E731

        def evaluate(i, batch_inp, batch_psf):
            with torch.no_grad(), amp():
                psfnet.eval()
                # psfnet.py:131-146: five (target, prediction) pairs of the current batch
                if plt is not None:
                    shown = psfnet(batch_inp[:5]).float().cpu()
                    fig, axs = plt.subplots(5, 2)
                    for j in range(min(5, shown.shape[0])):
                        axs[j, 0].imshow(batch_psf[j].float().cpu().numpy())
                        axs[j, 1].imshow(shown[j].numpy())
                    fig.suptitle(f"GT/Pred PSFs at iter {i + 1}")
                    fig.savefig(os.path.join(result_dir, f"iter{i + 1}.png"), dpi=300)
                    plt.close(fig)
                torch.save(psfnet.state_dict(),
                           os.path.join(result_dir, f"iter{i + 1}_PSFNet_{self.model_name}.pkl"))
                inp, psf = self.get_test_data()
                inp, psf = inp.to(self.device), psf.to(self.device)
                pred = psfnet(inp)
                psf = psf / psf.sum((-1, -2), keepdim=True)
                pred = pred / pred.sum((-1, -2), keepdim=True)
                logging.info(f"{i}, {l1(pred, psf).item()}, {l2(pred, psf).item()}")
                psfnet.train()

        losses = []
        if not pipelined:
            for i in range(iters + 1):
                with amp():
                    inp, psf = self.get_training_data(bs=bs, spp=spp)
                    inp, psf = inp.to(self.device), psf.to(self.device)
                    loss = l2(psfnet(inp), psf)
                    optim.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(optim)
                scaler.update()
                sche.step()
                losses.append(loss.detach())
                if (i + 1) % evaluate_every == 0:
                    evaluate(i, inp, psf)
        else:
            main = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device, priority=_SIDE_PRIORITY)

            produced = [0]

            def produce():
                # enqueue only: the Newton trip check of this batch (newton.py) runs when the batch
                # is consumed, two iterations later, by which time its kernel has long finished
                produced[0] += 1
                with torch.cuda.stream(side):
                    inp, pend = self.get_training_data(bs=bs, spp=spp, _defer=True)
                    inp = inp.pin_memory().to(self.device, non_blocking=True)
                return inp, pend

            def consume(item):
                inp, pend = item
                with torch.cuda.stream(side):
                    psf = pend.wait()                          # re-launches (rare) go to `side` too
                    ready = torch.cuda.Event()
                    ready.record(side)
                return inp, psf, ready

            side.wait_stream(main)
            first = consume(produce())
            static_inp = torch.empty_like(first[0])
            static_psf = torch.empty_like(first[1])
            main.wait_event(first[2])
            static_inp.copy_(first[0]); static_psf.copy_(first[1])

            def whole_step():
                with amp():
                    loss = l2(psfnet(static_inp), static_psf)
                scaler.scale(loss).backward()
                scaler.step(optim)                            # fused AdamW: unscale, inf check and update on the device
                scaler.update()
                return loss

            # warm-up off the capture stream (lazy state: the optimiser's moments, the scaler's scale) -- on a COPY of the
            # run's state: whatever three extra steps on the first batch did is put back before the capture
            saved = {k: v.detach().clone() for k, v in psfnet.state_dict().items()}
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(3):
                    optim.zero_grad(set_to_none=True)
                    whole_step()
                with torch.no_grad():
                    for k, v in psfnet.state_dict().items():
                        v.copy_(saved[k])
                    for st in optim.state.values():
                        for t in st.values():
                            if torch.is_tensor(t):
                                t.zero_()
                    scaler._scale.fill_(scaler._init_scale)
                    scaler._growth_tracker.zero_()
            main.wait_stream(side)
            optim.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = whole_step()
            # the capture itself ran no kernel, but its Python side advanced the scaler's bookkeeping exactly as one
            # step + update does; nothing to undo there
            queue = [first]
            evals_done = [0]

            def refill():
                # up to two batches ahead -- but never past an evaluation that has not run yet:
                # evaluate(i) draws the test set from the same CPU generators BEFORE batch i+1 in
                # the plain loop (and in the reference), so batch b waits for b // evaluate_every
                # evaluations; the queue drains at those boundaries and fills up again after them
                while (len(queue) < 3 and produced[0] < iters + 1
                       and produced[0] // evaluate_every <= evals_done[0]):
                    queue.append(produce())
            refill()
            for i in range(iters + 1):
                item = queue.pop(0)
                inp, psf, ready = item if len(item) == 3 else consume(item)
                main.wait_event(ready)
                static_inp.copy_(inp); static_psf.copy_(psf)
                inp.record_stream(main); psf.record_stream(main)
                lr_dev.fill_(shadow.param_groups[0]["lr"])     # this step's rate, as the plain loop's optimiser reads it
                graph.replay()                                 # grads are rewritten, not accumulated
                refill()
                sche.step()
                losses.append(static_loss.detach().clone())
                if (i + 1) % evaluate_every == 0:
                    evaluate(i, static_inp, static_psf)
                    evals_done[0] += 1
                    refill()
        torch.save(psfnet.state_dict(), os.path.join(result_dir, f"PSFNet_{self.model_name}.pkl"))
        return [float(v) for v in losses]

    def pred(self, inp):
        """psfnet.py:317-336: network kernels for the left sub-pixel at (x, y, z) and, by the
        sensor's mirror symmetry, for the right one as fliplr of the kernel at (-x, y, z);
        each (L, R) pair is normalised by its joint sum.  inp [..., 3] -> [..., 2, ks, ks].
        As in the reference, `inp[..., 0]` is negated in place."""
        if (inp.is_cuda and self.fused_mlp and not torch.is_grad_enabled()
                and getattr(self.psfnet, "fused_supported", lambda: False)()):
            both = self.psfnet.forward_fused(inp, mirror=True)    # one kernel for L and R
            inp[..., 0] = inp[..., 0] * (-1)
        else:
            mirrored = inp.clone()
            mirrored[..., 0] = mirrored[..., 0] * (-1)
            both = self.psfnet(torch.stack((inp, mirrored)))      # one GEMM chain for L and R
            inp[..., 0] = mirrored[..., 0]
        psf = torch.stack((both[0], torch.flip(both[1], dims=[-1])), dim=-3)
        psf = psf / (psf.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1) + 1e-9)
        assert psf.shape[-1] == self.kernel_size
        return psf

    def pred_coc(self, inp, is_z=True):
        """psfnet.py:338-379: thin-lens baseline -- Gaussian of the circle of confusion clipped
        to its radius, halved left/right of the kernel centre according to the side of focus.
        inp [B,W,H,3] (z or depth last) -> [B,W,H,2,ks,ks]."""
        ks, dev = self.kernel_size, inp.device
        ax = torch.linspace(-ks / 2 + 1 / 2, ks / 2 - 1 / 2, ks, device=dev)
        gx, gy = torch.meshgrid(ax, ax, indexing="xy")
        z = inp[..., -1]
        foc_dist = torch.tensor(self.foc_d).to(dev)
        ps = self.sensor_size[0] / self.sensor_res[0]
        depth = z * (self.d_max - self.d_min) + self.d_min if is_z else z
        coc = torch.abs(depth - foc_dist) * self.foclen ** 2 / (-depth * self.fnum * (-foc_dist - self.foclen))
        radius = (torch.clamp(coc / ps, min=0.1) / 2).unsqueeze(-1).unsqueeze(-1)
        rr = gx ** 2 + gy ** 2
        thin = torch.exp(-rr / (2 * radius ** 2)) * (rr < radius ** 2)
        keep_right = torch.ones(ks, ks, device=dev)
        keep_right[..., 0:ks // 2] = 0                            # the reference's `l_mask`
        keep_left = torch.ones(ks, ks, device=dev)
        keep_left[..., ks // 2 + 1:] = 0                          # the reference's `r_mask`
        near = (depth > foc_dist).unsqueeze(-1).unsqueeze(-1)
        far = (depth < foc_dist).unsqueeze(-1).unsqueeze(-1)
        one = torch.ones_like(thin)
        psf_l = thin * torch.where(near, keep_right, one) * torch.where(far, keep_left, one)
        psf_r = thin * torch.where(near, keep_left, one) * torch.where(far, keep_right, one)
        psf = torch.stack((psf_l, psf_r), dim=-3)
        return psf / (psf.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1) + 1e-6)

    # ------------------------------------------------------------------ tone curves
    _TONE_LO = (0.89129432, 0.27217316, -0.00246187)              # psfnet.py:591
    _TONE_HI = (5.94018909e-01, 1.20060450e+01, -5.24983855e-03)  # psfnet.py:592

    def fit_degamma(self, x):
        """psfnet.py:589-598: 8-bit code value -> luminance, blend of two rational fits."""
        (a1, b1, c1), (a2, b2, c2) = self._TONE_LO, self._TONE_HI
        lo = 1 / (1 / (a1 * x + b1) + c1)
        hi = 1 / (1 / (a2 * x + b2) + c2)
        t = torch.clamp(x / 100, max=1)
        return hi * t + lo * (1 - t)

    def degamma(self, img_gamma):
        if img_gamma.is_cuda and img_gamma.dtype == torch.float32 and not img_gamma.requires_grad:
            return self._tone(img_gamma, 0)
        return self.fit_degamma(img_gamma * 255.)

    @staticmethod
    def _tone(t, mode):
        """degamma (mode 0) / clip(gamma(.), 0, 1) (mode 1) in one kernel, same fp32 arithmetic."""
        from . import _lib
        from .basics import dptr, stream_ptr
        t = t.contiguous()
        out = torch.empty_like(t)
        _lib.check(_lib.lib().sdirt_tone_curve(dptr(t), t.numel(), mode, dptr(out), stream_ptr(t.device)))
        return out

    def fit_gamma(self, l):
        """psfnet.py:605-615: inverse of fit_degamma."""
        (a1, b1, c1), (a2, b2, c2) = self._TONE_LO, self._TONE_HI
        x1 = (1 / (1 / (l + 1e-9) - c1) - b1) / a1
        x2 = (1 / (1 / (l + 1e-9) - c2) - b2) / a2
        t = torch.clamp((x1 + x2) / 2 / 100, max=1)
        return x2 * t + x1 * (1 - t)

    def gamma(self, img_degamma):
        return self.fit_gamma(img_degamma) / 255.

    def noise(self, render, shape):
        """psfnet.py:627-640: Gaussian noise with a left/right ramp (training augmentation).  Same
        draws from numpy's and torch's generators, in the same order; the ramp is built on the
        device instead of being expanded to [N,C,H,W] on the host and copied over."""
        N, C, H, W = shape
        noise_map = torch.randn_like(render) * (0.05 * np.random.rand())
        lo, hi = np.random.rand() / 2, np.random.rand() / 2 + 0.5
        ramp = torch.linspace(lo, hi, W, device=render.device)
        weight = torch.cat([ramp.expand(N, C, H, W), torch.flip(ramp, [-1]).expand(N, C, H, W)], dim=1)
        render += noise_map * weight
        return render

    @torch.no_grad()
    def render(self, img, depth, foc_dist, train=False):
        """psfnet.py:642-714, batched branch: all-in-focus img [N,C,H,W] + depth [N,1,H,W] (mm,
        negative) + foc_dist [N] -> dual-pixel capture [N,2C,H,W] (left views, then right).
        Per pixel: network L/R kernels at (x, y, z(depth)), convolution in linear light."""
        if img.dim() != 4:
            # psfnet.py:659-675 feeds a 4-vector (x, y, z, foc_z) to the 3-input network and
            # fails there; only the batched branch is usable in the reference
            raise ValueError("render expects img [N,C,H,W], depth [N,1,H,W], foc_dist [N]")
        depth = depth + self.d_sensor
        N, C, H, W = img.shape
        z = self.depth2z(depth).squeeze(1)
        key = (N, H, W, str(img.device))
        if getattr(self, "_render_grid", (None,))[0] != key:
            # the reference rebuilds this grid on the host at every call (psfnet.py:682-688)
            x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="xy")
            self._render_grid = (key, x.unsqueeze(0).repeat(N, 1, 1).to(img.device),
                                 y.unsqueeze(0).repeat(N, 1, 1).to(img.device))
        _, x, y = self._render_grid
        o = torch.stack((x, y, z), -1).float()
        if img.is_cuda and self.fused_render:
            # one GEMM chain over [(x,y,z); (-x,y,z)], then flip + normalise + convolve in one
            # HIP kernel straight from the raw fp16 outputs (sdirt_psfnet_render)
            if self.fused_mlp and getattr(self.psfnet, "fused_supported", lambda: False)():
                # the whole network in one kernel, activations in LDS (sdirt_psfnet_mlp)
                raw = self.psfnet.forward_fused(o, mirror=True)
            else:
                mirrored = o.clone()
                mirrored[..., 0] = -mirrored[..., 0]
                raw = self.psfnet(torch.stack((o, mirrored)))
            render_lr = psfnet_render(self.degamma(img), raw[0], raw[1], self.kernel_size)
        else:
            psf = self.pred(o)
            render_lr = local_psf_render_fast(self.degamma(img), psf, self.kernel_size)
        render = torch.cat(render_lr, dim=1)
        if not train and render.is_cuda and render.dtype == torch.float32:
            return self._tone(render, 1)                      # gamma + clip in one pass
        render = self.gamma(render)
        if train:
            render = self.noise(render, img.shape)
        return torch.clip(render, 0.0, 1.0)

    # ------------------------------------------------------------------ checks the reference ships
    def time_compare_psf(self, verbose=True):
        """psfnet.py:570-586, the one timing harness the reference ships: 512*768/16 = 24576 random
        points (x, y in [0, 1), z2depth of a uniform z), 2 * GEO_SPP = 4096 spp, kernel size of the
        model, the PSFs copied to the host INSIDE the timed span; then one network prediction for a
        128 x 192 field, copied to the host likewise.  Wall-clock seconds, as the reference prints
        them -> (ray_tracing_seconds, network_seconds).  The reference's `.to('cpu')` is Lensgroup.to_host here:
        the same copy into page-locked memory (a pageable destination was a third of the ray-tracing span)."""
        import time
        from .basics import GEO_SPP
        inp = torch.rand(512 * 768 // 16, 3)
        inp[:, 2] = self.z2depth(inp[:, 2])
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        start_time = time.time()
        psfl = self.to_host(self.psf(points=inp, ks=self.kernel_size, center=True, spp=GEO_SPP * 2))
        t_trace = time.time() - start_time
        if verbose:
            print(f"ray_tracing time cost: {t_trace}s")
        inp = torch.rand(1, 512 // 4, 768 // 4, 3).to(self.device)
        start_time = time.time()
        with torch.no_grad():
            psf2 = self.to_host(self.pred(inp).detach())
        t_net = time.time() - start_time
        if verbose:
            print(f"network time cost: {t_net}s")
        assert psfl.shape == (24576, self.kernel_size, self.kernel_size) and psf2.shape[-3:] == (2, self.kernel_size, self.kernel_size)
        return t_trace, t_net

    def compare_psf(self, spp=None, save_dir=".", figures=True):
        """psfnet.py:529-568: ray-traced vs predicted (L, R) kernels at three field points (0, 0.4, 0.8 of the
        diagonal) and two depths, each pair written as `rt_<depth>_v0x.png` / `pred_<depth>_v0x.png` into
        `save_dir` (the reference: the working directory) -> {depth: (traced [3,2,ks,ks], predicted [3,2,ks,ks])}."""
        from .basics import GEO_SPP
        spp = spp or GEO_SPP * 100
        x = torch.tensor([0, 0.4, 0.8])
        out = {}
        for d_ori in (-500.0, -20000.0):
            depth = d_ori + self.d_sensor
            pts = torch.stack((x, x, torch.full_like(x, depth)), dim=-1)
            psfl = self.psf(points=pts, ks=self.kernel_size, center=True, spp=spp)
            pts[..., 0] = pts[..., 0] * (-1)
            psfr = torch.flip(self.psf(points=pts, ks=self.kernel_size, center=True, spp=spp),
                              dims=[-1])
            z = self.depth2z(torch.tensor(depth))
            inp = torch.stack((x, x, torch.full_like(x, z)), dim=-1).to(self.device)
            traced, pred = torch.stack((psfl, psfr), dim=1), self.pred(inp).detach()
            out[int(d_ori)] = (traced, pred)
            if figures:
                for j, tag in enumerate(("v00", "v04", "v08")):
                    self.vis_psf_map(traced[j].cpu(), filename=os.path.join(save_dir, f"rt_{int(d_ori)}_{tag}.png"))
                    self.vis_psf_map(pred[j].cpu(), filename=os.path.join(save_dir, f"pred_{int(d_ori)}_{tag}.png"))
        return out

    def vis_psf_map(self, psf, filename=None, normal=True):
        """psfnet.py:728-762: [N,k,k] kernels side by side in grey (each over its own maximum when `normal`), or a
        [N,C,k,k] field as a C x N (N x N when square) mosaic on a 0 ... 0.1 scale; saved at 300 dpi."""
        from .plots import _pyplot
        plt = _pyplot()
        psf = psf.detach().cpu().float()
        if psf.dim() == 4:
            N, C = psf.shape[:2]
            rows = N if N == C else C
            fig, axs = plt.subplots(rows, N, squeeze=False)
            for i in range(rows):
                for j in range(N):
                    axs[i, j].imshow((psf[i, j] if N == C else psf[j, i]).numpy(), vmin=0.0, vmax=0.1)
        else:
            fig, axs = plt.subplots(1, psf.shape[0], squeeze=False)
            for i in range(psf.shape[0]):
                k = psf[i] / psf[i].max() if normal else psf[i]
                axs[0, i].imshow(k.numpy(), vmin=0.0, vmax=1, cmap="gray")
                axs[0, i].axis("off")
        if filename is not None:
            fig.savefig(filename, dpi=300)
        plt.close(fig)
