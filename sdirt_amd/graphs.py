"""A fixed-shape call chain captured once in a hipGraph and replayed (torch.cuda.CUDAGraph on ROCm).

The headline path needs none of this -- a PSF-volume step is ONE library call and one kernel (volume.VolumeStepper).  The
chains around it are many small launches: BASELINE config 5's frame (PSFNet.render -> DfDPNet forward) is ~85 of them in
under 5 ms, and between two dependent launches the GPU idles for the few microseconds the command processor needs to hand
over.  A graph removes most of that: the same kernels with the same arguments, enqueued as one unit.  (The fitting loop's
forward + backward is captured the same way, psfnet.py: train_psfnet(pipelined=True).)

    frame = GraphedCall(lambda: net(*simulate(img, depth)), warmup=3)    # tensors the lambda reads are the STATIC inputs:
    img.copy_(next_img); depth.copy_(next_depth)                         # refresh them in place,
    out = frame()                                                        # replay; `out` is the same tensor(s) every time

Everything the chain launches must be launched on torch's current stream (this package's kernels are: basics.stream_ptr),
must not synchronise with the host, and must have made its one-off choices before the capture (MIOpen's find pass, cached
fp16 weights, packed MLP weights, trip tables): that is what the warm-up calls are for.  A graph replays POINTERS: whatever
the chain reads besides the static inputs (weights, cached copies, tables) must stay where it was -- a module that is trained
or re-laid (DfDPNet switches its weights' memory format between its inference and its training path) needs a new capture.
"""
import torch


class GraphedCall:
    def __init__(self, fn, warmup=3, device=None):
        self.fn = fn
        dev = torch.device(device if device is not None else torch.cuda.current_device())
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                         # warm-up off the stream that will capture
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
