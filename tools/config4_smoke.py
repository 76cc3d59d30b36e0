import os, sys, time, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import bench
from trainers import FirstStepTrainer
torch.manual_seed(0)
tr = FirstStepTrainer(enc_filters=(256,256,256,256,256), dec_filters=(256,256,256,256,512), dict_size=1024, device="cuda")
img, noise = bench.synthetic_batch(2, 512, 5, torch.device("cuda"))
for i in range(3):
    t0=time.perf_counter(); out = tr.training_step({"image": img}, noise=noise); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    sc = tr.scalars(out)
    print("step %d: %.0f ms total %.4g recon %.4f commit %.4f codes used %d mem %.1f GB" % (i, dt*1e3, sc["total"], sc["recon"], sc["commit"], int(torch.unique(out["ids_1"]).numel()), torch.cuda.max_memory_allocated()/1e9), flush=True)
