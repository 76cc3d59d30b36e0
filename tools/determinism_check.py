import os, sys, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import bench
from trainers import FirstStepTrainer
def run():
    torch.manual_seed(0)
    tr = FirstStepTrainer(device="cuda")
    pool=[bench.synthetic_batch(8, 128, 100+s, torch.device("cuda")) for s in range(2)]
    for i in range(3):
        img, noise = pool[i%2]
        out = tr.training_step({"image": img}, noise=noise)
    torch.cuda.synchronize()
    return [p.detach().clone() for p in list(tr.encoder.parameters())+list(tr.decoder.parameters())], float(out["total"].detach()), tr.encoder.vq.embed.clone()
a=run(); b=run()
same=sum(int(torch.equal(x,y)) for x,y in zip(a[0],b[0])); print("params bit-identical: %d / %d; total %r vs %r; codebook equal %s" % (same, len(a[0]), a[1], b[1], torch.equal(a[2],b[2])))
mx=max(float((x-y).abs().max()) for x,y in zip(a[0],b[0])); print("max abs param diff", mx)
