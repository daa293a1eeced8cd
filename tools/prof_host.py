import cProfile, pstats, sys, torch, time
sys.path.insert(0, ".")
from sdirt_amd import Lensgroup
lens = Lensgroup("sdirt_amd/data/rf50mm.json", sensor_res=(512,768), device="cuda:0")
pts = torch.tensor([[0.3,0.2,-800.0],[-0.7,0.6,-5000.0]])
for _ in range(20): lens.psf_lr(pts, ks=17, spp=2048)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): lens.psf_lr(pts, ks=17, spp=2048)
torch.cuda.synchronize(); print("psf_lr 2 points:", (time.perf_counter()-t)/200*1e3, "ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): lens.psf_lr(pts, ks=17, spp=2048)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
