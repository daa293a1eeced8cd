import sys, time, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
lens = bench.build_lens(dev, "rf50mm", 62.25)
pts = bench.volume_points(1).to(dev)
outs = [tuple(torch.empty((pts.shape[0], 65, 65), device=dev) for _ in range(2)) for _ in range(2)]
prev = None; ts = []
torch.cuda.synchronize()
for i in range(40):
    t0 = time.perf_counter()
    p = lens.psf_lr(pts, ks=65, spp=4096, out=outs[i % 2], defer=True)
    t1 = time.perf_counter()
    if prev is not None: prev.wait()
    prev = p
    ts.append((t1 - t0, time.perf_counter() - t1))
prev.wait(); torch.cuda.synchronize()
print("enqueue ms:", " ".join(f"{a*1e3:.2f}" for a, _ in ts))
print("wait ms:   ", " ".join(f"{b*1e3:.2f}" for _, b in ts))
