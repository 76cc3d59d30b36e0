"""2-rank data-parallel rehearsal on ONE GPU (gloo backend, both ranks on cuda:0): the real HIP path with the
gradient reducer, synchronised StyledDenorm statistics and the VQ EMA all-reduce, compared with a single process
on the concatenated batch.  (RCCL needs one GPU per rank; the driver exercises that at N = 2/4/8.)"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
root = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "medical-image-editing_amd"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
forced = os.environ.get("VQW_DP_FORCE", "0") == "1"      # one rank, every collective issued all the same (hipops.ops)
if world > 1 or forced:
    dist.init_process_group(os.environ.get("VQW_TEST_BACKEND", "gloo"), rank=rank, world_size=world)
from trainers import FirstStepTrainer, FlipViews, LossWeights
from networks import UNetEncoder, UNetDecoder
from oracle.vqwnet_ref import synthetic_slices
torch.manual_seed(5)
K = 10
enc = UNetEncoder(1, [16, 32, 32, 64, 64], K, 0.99, 'torch', False, 1, True)
dec = UNetDecoder(16, 1, [32, 32, 64, 64, 128], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False)
B, S = 4, 64
with torch.no_grad():
    enc.vq.embed.mul_(0.7); enc.vq.cluster_size.fill_(B * S * S / K)
    enc.vq.embed_avg.copy_(enc.vq.embed.t() * enc.vq.cluster_size[None, :])
tr = FirstStepTrainer(dict_size=K, momentum=0.99, views=FlipViews(border=2), encoder=enc, decoder=dec, device="cuda:0",
                      data_parallel=world > 1 or forced, loss_weight=LossWeights(cross=float(os.environ.get("VQW_TEST_CROSS_W", "1"))))
img, noise = synthetic_slices(B, S, 11)
lo, hi = (rank * B // world, (rank + 1) * B // world)
o = tr.training_step({"image": img[lo:hi].cuda()}, noise=noise[lo:hi].cuda())
torch.cuda.synchronize()
from hipops import ops as _ops
res = {"total": float(o["total"].detach()), "ids_1": o["ids_1"].cpu(),
       "collectives": (_ops.collective_calls, tr.reducer.launches if tr.reducer is not None else 0),
       "backend": dist.get_backend() if dist.is_initialized() else "",
       "vq": {k: v.cpu() for k, v in enc.vq.state_dict().items()},
       "bn": {k: v.cpu() for k, v in dec.state_dict().items() if "running_" in k},
       "g": {k: p.grad.cpu() for k, p in list(enc.named_parameters())[:8] + list(dec.named_parameters())[:40:4]},
       "p": {k: p.detach().cpu() for k, p in list(dec.named_parameters())[:6]}}
torch.save(res, out + ".%d" % rank)
if world > 1 or forced:
    dist.barrier(); dist.destroy_process_group()
'''


def _run(world, tmp_path, tag, port, cross_w=1.0, extra_env=None):
    script = tmp_path / "dpw.py"
    script.write_text(WORKER)
    out = str(tmp_path / tag)
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(r),
                   VQW_TEST_CROSS_W=str(cross_w))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        o = p.communicate(timeout=500)[0].decode()
        assert p.returncode == 0, o[-3000:]
    return [torch.load(out + ".%d" % r) for r in range(world)]


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_rccl_world_size_one_equals_plain_step(tmp_path, overlap):
    """The data-parallel step over RCCL itself (backend "nccl"), as far as one GPU allows: a process group of ONE rank with
    every collective forced on (VQW_DP_FORCE=1) - the SyncBN statistics of both views on their two streams, the VQ EMA
    statistics, the gradient buckets flattened behind the weight-gradient lanes and all-reduced asynchronously on
    ProcessGroupNCCL's stream, finish() and the optimiser step.  With one rank every all-reduce is the identity and the mean
    divides by 1, so losses, ids, VQ buffers, BN running statistics, averaged gradients and updated parameters must equal the
    non-distributed step BIT FOR BIT; a missing stream dependency shows up as a difference."""
    # (the overlapped schedule needs a parameter's gradient final when it is announced, so its weight-gradient slabs are folded
    # call by call; the plain step folds them in one batched launch at the end of the pass - another fixed summation order.
    # Its reference therefore runs with the batched fold off: VQW_FOLD_DEFER=0)
    plain = _run(1, tmp_path, "plain1", 29631, extra_env={"VQW_FOLD_DEFER": "0"} if overlap == "1" else None)[0]
    # overlap "0": the gradient buckets are exchanged in finish() (default); "1": launched from inside the backward pass
    rccl = _run(1, tmp_path, "rccl1", 29632, extra_env={"VQW_DP_FORCE": "1", "VQW_TEST_BACKEND": "nccl", "VQW_DP_OVERLAP": overlap})[0]
    assert rccl["backend"] == "nccl" and plain["backend"] == ""
    assert plain["collectives"] == (0, 0)
    n_small, n_buckets = rccl["collectives"]
    assert n_small >= 2 + 16 and n_buckets >= 1, rccl["collectives"]      # 2 VQ + SyncBN forward / backward of both views
    assert rccl["total"] == plain["total"]
    assert torch.equal(rccl["ids_1"], plain["ids_1"])
    for grp in ("vq", "bn", "g", "p"):
        for k in plain[grp]:
            assert torch.equal(rccl[grp][k], plain[grp][k]), "%s %s differs between the RCCL step and the plain step" % (grp, k)


def test_two_rank_dp_matches_single_process(tmp_path):
    single = _run(1, tmp_path, "single", 29621)[0]
    dp = _run(2, tmp_path, "dp", 29622)
    # replicas stay identical
    for k in dp[0]["p"]:
        assert torch.equal(dp[0]["p"][k], dp[1]["p"][k]), "ranks diverged on " + k
    for k in dp[0]["vq"]:
        assert torch.allclose(dp[0]["vq"][k], dp[1]["vq"][k], rtol=0, atol=0), "VQ buffers differ across ranks: " + k
    # global-batch semantics: ids, VQ EMA buffers, BN running statistics equal the single-process run
    assert torch.equal(torch.cat([dp[0]["ids_1"], dp[1]["ids_1"]]), single["ids_1"])
    for k in single["vq"]:
        assert torch.allclose(dp[0]["vq"][k], single["vq"][k], rtol=2e-5, atol=1e-6), k
    for k in single["bn"]:
        assert torch.allclose(dp[0]["bn"][k], single["bn"][k], rtol=1e-4, atol=1e-6), k
    # averaged gradients ~ single-process gradients (the cross loss averages per-rank means, so not bit-equal)
    errs = []
    gmax = max(float(g.norm()) for g in single["g"].values())
    for k, g in single["g"].items():
        if float(g.norm()) < 1e-4 * gmax:       # analytically-zero gradients (bias in front of a norm): noise only
            continue
        errs.append(float((dp[0]["g"][k] - g).norm() / g.norm()))
    assert max(errs) < 0.15 and sorted(errs)[len(errs) // 2] < 0.05, errs
    mean_total = 0.5 * (dp[0]["total"] + dp[1]["total"])
    assert abs(mean_total - single["total"]) <= 0.05 * abs(single["total"])


def test_two_rank_dp_gradients_equal_single_process_without_cross_loss(tmp_path):
    """The tight form of the gradient check above.  The cross-view loss is a mean over the (sample, code) pairs PRESENT on a
    rank, so the mean of the per-rank losses is a different function from the global-batch loss (DDP semantics, as in the
    reference) and a few per cent apart in gradient.  With its weight at zero every remaining term (commit, reconstruction;
    SyncBN and VQ statistics are global) is an equal-size mean, and the averaged data-parallel gradients ARE the
    single-process gradients: fp32 rounding and the occasional ReLU / max-pool flip are all that is left - a mis-scaled
    bucket or a lost gradient would be orders of magnitude above this bound."""
    single = _run(1, tmp_path, "single0", 29623, cross_w=0.0)[0]
    dp = _run(2, tmp_path, "dp0", 29624, cross_w=0.0)
    assert torch.equal(torch.cat([dp[0]["ids_1"], dp[1]["ids_1"]]), single["ids_1"])
    errs = []
    gmax = max(float(g.norm()) for g in single["g"].values())
    for k, g in single["g"].items():
        if float(g.norm()) < 1e-4 * gmax:
            continue
        assert torch.equal(dp[0]["g"][k], dp[1]["g"][k]), "ranks hold different averaged gradients for " + k
        errs.append(float((dp[0]["g"][k] - g).norm() / g.norm()))
    assert max(errs) < 2e-2 and sorted(errs)[len(errs) // 2] < 2e-3, errs
    mean_total = 0.5 * (dp[0]["total"] + dp[1]["total"])
    assert abs(mean_total - single["total"]) <= 1e-4 * abs(single["total"])


WORKER2 = r'''
import os, sys, torch, torch.distributed as dist
root = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "medical-image-editing_amd"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
from trainers import SecondStepTrainer, GanLossWeights
from networks import UNetEncoder, UNetDecoder, NLayerDiscriminator
from oracle.vqwnet_ref import synthetic_slices
torch.manual_seed(6)
K = 8
enc = UNetEncoder(1, [16, 16, 32, 32, 32], K, 0.99, 'torch', False, 1, True)
dec = UNetDecoder(16, 1, [16, 32, 32, 32, 64], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False)
dis = NLayerDiscriminator(1, 1, n_filters=16, n_layers=3)
B, S = 4, 64
tr = SecondStepTrainer(enc, dec, dis, loss_weight=GanLossWeights(1.0, 0.2, 0.7), lr=1e-3, device="cuda:0", data_parallel=world > 1)
img, _ = synthetic_slices(B, S, 12)
lo, hi = (rank * B // world, (rank + 1) * B // world)
o = tr.training_step(img[lo:hi].cuda())
torch.cuda.synchronize()
res = {"gen_total": float(o["gen_total"].detach()), "dis_total": float(o["dis_total"].detach()), "recon": float(o["recon"].detach()),
       "ids": o["ids"].cpu(),
       "bn": {k: v.cpu() for k, v in list(dec.state_dict().items()) + [("dis." + k, v) for k, v in dis.state_dict().items()] if "running_" in k},
       "pdec": {k: p.detach().cpu() for k, p in list(dec.named_parameters())[:40:5]},
       "pdis": {k: p.detach().cpu() for k, p in dis.named_parameters()}}
torch.save(res, out + ".%d" % rank)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def test_second_step_two_rank_dp(tmp_path):
    """SecondStepTrainer(data_parallel=True) on two ranks (gloo, one GPU) against one process on the whole batch:
    identical replicas, BatchNorm running statistics of the global batch, parameters after the generator and the
    discriminator update within rounding of the single-process step (losses are per-rank means, so averaged)."""
    global WORKER
    keep, WORKER = WORKER, WORKER2
    try:
        single = _run(1, tmp_path, "s2_single", 29631)[0]
        dp = _run(2, tmp_path, "s2_dp", 29632)
    finally:
        WORKER = keep
    for grp in ("pdec", "pdis"):
        for k in dp[0][grp]:
            assert torch.equal(dp[0][grp][k], dp[1][grp][k]), "ranks diverged on " + k
    assert torch.equal(torch.cat([dp[0]["ids"], dp[1]["ids"]]), single["ids"])
    for k in single["bn"]:
        assert torch.allclose(dp[0]["bn"][k], single["bn"][k], rtol=2e-4, atol=2e-6), k
    for key in ("gen_total", "dis_total", "recon"):
        mean = 0.5 * (dp[0][key] + dp[1][key])
        assert abs(mean - single[key]) <= 2e-4 * max(1.0, abs(single[key])), (key, mean, single[key])
    worst = 0.0
    for grp in ("pdec", "pdis"):
        for k, v in single[grp].items():
            d = float((dp[0][grp][k] - v).abs().max())
            worst = max(worst, d)
            assert d <= 2.5e-3, (k, d)            # one Adam step of lr 1e-3: a sign flip of a ~0 gradient moves 2e-3
    assert worst > 0.0


VQ_REF_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "medical-image-editing_amd"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from networks.vq import VQ
g = np.load(os.path.join(root, "tests", "golden", "vq_dist.npz"))
for tag in ("k10", "k64"):
    e0 = torch.from_numpy(g[tag + "/embed0"].copy())
    K, D = e0.shape
    vq = VQ(emb_dim=D, dict_size=K, momentum=float(g[tag + "/momentum"]), eps=1e-5, knn_backend="torch")
    vq.dist_mode = "reference"
    vq.embed.copy_(e0); vq.embed_avg.copy_(e0.t()); vq.cluster_size.zero_()
    vq.to("cuda:0").train()
    for call in (1, 2):
        x = torch.from_numpy(g["%s/r%d/x%d" % (tag, rank, call)]).to("cuda:0")
        q, commit, ids = vq(x)
        agree = np.mean(ids.transpose(1, 2).cpu().numpy() == g["%s/r%d/ids%d" % (tag, rank, call)])
        assert agree > 0.999, (tag, rank, call, agree)
        ref_c = float(g["%s/r%d/commit%d" % (tag, rank, call)])
        assert abs(float(commit) - ref_c) <= 5e-5 * abs(ref_c), (tag, rank, call, float(commit), ref_c)
        for b in ("embed", "cluster_size", "embed_avg"):
            ref = torch.from_numpy(g["%s/r%d/%s_after%d" % (tag, rank, b, call)])
            err = float((getattr(vq, b).cpu() - ref).norm() / ref.norm())
            assert err < (1e-5 if agree == 1.0 else 5e-3), (tag, rank, call, b, err)
dist.barrier(); dist.destroy_process_group()
'''


def test_vq_reference_dist_mode_matches_reference_fixture(tmp_path):
    """`dist_mode="reference"` (rank-mean embed_sum, local counts: vq_module.py:187-193) on two gloo ranks sharing the
    GPU, against tests/golden/vq_dist.npz - buffers the REFERENCE's VQModule held after two calls under WORLD_SIZE=2."""
    script = tmp_path / "vqw.py"
    script.write_text(VQ_REF_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29651", WORLD_SIZE="2", RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o = p.communicate(timeout=300)[0].decode()
        assert p.returncode == 0, o[-3000:]


DDP_WORKER = r'''
import copy, os, sys, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "medical-image-editing_amd"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from torch.nn.parallel import DistributedDataParallel as DDP
from networks import UNetEncoder, UNetDecoder
from trainers.data_parallel import GradientAllReducer
from hipops import ops
from oracle.vqwnet_ref import synthetic_slices
torch.manual_seed(5)
K = 10
enc0 = UNetEncoder(1, [16, 32, 32, 64, 64], K, 0.99, "torch", False, 1, True)
dec0 = UNetDecoder(16, 1, [32, 32, 64, 64, 128], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False)
img, _ = synthetic_slices(4, 64, 11)
x = img[rank * 2:(rank + 1) * 2].cuda()

def loss_of(enc, dec):
    q, commit, ids = enc(x)
    rec = dec(q)
    return ops.mse_loss(rec, x) + commit

# (A) torch DistributedDataParallel: its reducer hears of a gradient through the parameter's AccumulateGrad hook
encA, decA = copy.deepcopy(enc0).cuda(), copy.deepcopy(dec0).cuda()
encA.eval(); decA.train()          # VQ in eval mode: no EMA update in this comparison; SyncBN statistics in the decoder
dA_enc, dA_dec = DDP(encA, device_ids=[0]), DDP(decA, device_ids=[0])
loss_of(dA_enc, dA_dec).backward()
torch.cuda.synchronize()
gA = {("enc." + k): p.grad.clone() for k, p in encA.named_parameters()}
gA.update({("dec." + k): p.grad.clone() for k, p in decA.named_parameters()})
assert all(g is not None for g in gA.values())

# (B) this package's reducer: out-of-band weight gradients on side streams, bucketed all-reduce
encB, decB = copy.deepcopy(enc0).cuda(), copy.deepcopy(dec0).cuda()
encB.eval(); decB.train()
params = [p for p in list(decB.parameters())[::-1] + list(encB.parameters())[::-1] if p.requires_grad]
red = GradientAllReducer(params, bucket_bytes=1 << 20)
red.prepare()
ops.reset_pending(params)
loss_of(encB, decB).backward()
red.finish()
torch.cuda.synchronize()
gB = {("enc." + k): p.grad for k, p in encB.named_parameters()}
gB.update({("dec." + k): p.grad for k, p in decB.named_parameters()})
gmax = max(float(g.norm()) for g in gA.values())
worst = 0.0
for k in gA:
    na = float(gA[k].norm())
    if na < 1e-6 * gmax:
        continue
    e = float((gA[k] - gB[k]).norm()) / na
    worst = max(worst, e)
    assert e < 2e-4, (k, e)
# and DDP really averaged over the ranks: the gradient is not the local one
encC, decC = copy.deepcopy(enc0).cuda(), copy.deepcopy(dec0).cuda()
encC.eval(); decC.train()
loss_of(encC, decC).backward()
torch.cuda.synchronize()
k = "dec.conv1x1.weight"
assert float((decC.conv1x1.weight.grad - gA[k]).norm()) > 1e-3 * float(gA[k].norm())
print("rank", rank, "worst DDP-vs-reducer gradient difference %.2e" % worst)
dist.barrier(); dist.destroy_process_group()
'''


def test_torch_ddp_wrapper_equals_gradient_all_reducer(tmp_path):
    """The modules under torch.nn.parallel.DistributedDataParallel (what the reference launches with, run_vqwnet.py:112-121;
    gloo, two ranks on one GPU): inside a DDP forward the conv weight gradients take the autograd route, so DDP's reducer
    sees every parameter; the averaged gradients equal the ones this package's own reducer produces."""
    script = tmp_path / "ddpw.py"
    script.write_text(DDP_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29661", WORLD_SIZE="2", RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o = p.communicate(timeout=500)[0].decode()
        assert p.returncode == 0, o[-3000:]


KMEANS_WORKER = r'''
import os, sys, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "medical-image-editing_amd"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from networks import UNetEncoder
from hipops import ops
torch.manual_seed(1)
K = 8
enc = UNetEncoder(1, [16, 16, 32, 32, 32], K, 0.99, "torch", False, world, False).cuda().eval()
imgs = torch.randn(4, 1, 32, 32, generator=torch.Generator().manual_seed(7))
mine = imgs[rank * 2:(rank + 1) * 2].cuda()
with torch.no_grad():
    feat_all = enc.feature_extraction(imgs.cuda())
    enc(mine, rank=rank)
expect, _ = ops.kmeans_codebook(feat_all.permute(0, 2, 3, 1).reshape(-1, 16), K, seed=0)
assert torch.allclose(enc.vq.embed, expect, rtol=1e-5, atol=1e-6), float((enc.vq.embed - expect).abs().max())
both = [torch.zeros_like(enc.vq.embed) for _ in range(world)]
dist.all_gather(both, enc.vq.embed)
assert torch.equal(both[0], both[1])
dist.barrier(); dist.destroy_process_group()
'''


def test_kmeans_initialisation_two_ranks(tmp_path):
    """initialize_embed under two ranks (unet_encoder.py:67-68, 84-88): features all-gathered, rank 0 clusters, the codebook
    is broadcast - both replicas hold the k-means of the GLOBAL batch's features."""
    script = tmp_path / "kmw.py"
    script.write_text(KMEANS_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", WORLD_SIZE="2", RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o = p.communicate(timeout=300)[0].decode()
        assert p.returncode == 0, o[-3000:]
