import sys, time, torch
sys.path.insert(0, ".")
import bench
lens = bench.build_lens(torch.device("cuda:0"), "rf50mm", 62.25)
pts = bench.volume_points(1).to("cuda:0")
out = tuple(torch.empty((pts.shape[0], 65, 65), device="cuda:0") for _ in range(2))
for pol in ("reference", "adaptive", "max", "reference", "adaptive"):
    lens.trip_policy = pol
    for _ in range(2): lens.psf_lr(pts, ks=65, spp=4096, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): lens.psf_lr(pts, ks=65, spp=4096, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"{pol:10s} {dt * 1e3:.2f} ms/step  {pts.shape[0] * 4096 / dt / 1e9:.2f} Grays/s")
