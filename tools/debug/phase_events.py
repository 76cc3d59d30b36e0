#!/usr/bin/env python3
"""Where the two views' forward passes sit in an UNTRACED step: HIP events around encoder / decoder forward calls on the stream
they run on, times relative to the step's first event (a kernel trace slows the host down and shifts the schedule).

    python tools/debug/phase_events.py [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from trainers import build_first_step_trainer  # noqa: E402
from utils import load_json  # noqa: E402

dev = torch.device("cuda:0")
cfg = load_json(os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"))
torch.manual_seed(0)
tr = build_first_step_trainer(cfg, device=dev, data_parallel=False)
B, S = int(cfg.dataset.batch_size), int(cfg.dataset.image_size)
pool = [bench.synthetic_batch(B, S, 1234 + s, dev) for s in range(4)]
marks = []


def wrap(mod, name):
    fwd = mod.forward

    def f(*a, **k):
        s = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
        out = fwd(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
        marks.append((name, s.cuda_stream, e0, e1))
        return out
    mod.forward = f


wrap(tr.decoder, "decoder.forward")
wrap(tr.encoder, "encoder.forward")
fe = tr.encoder.feature_extraction


def fe_wrapped(*a, **k):
    s = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
    out = fe(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
    marks.append(("encoder.feature_extraction", s.cuda_stream, e0, e1))
    return out


tr.encoder.feature_extraction = fe_wrapped
fl = tr.forward_losses


def fl_wrapped(*a, **k):
    out = fl(*a, **k)

    def hook(name):
        def h(g):
            s = torch.cuda.current_stream()
            e = torch.cuda.Event(enable_timing=True); e.record(s)
            marks.append((name, s.cuda_stream, e, e))
            return None
        return h
    for key in ("recon_1", "recon_2", "embed_1", "embed_2"):
        if out[key].requires_grad:
            out[key].register_hook(hook("grad reaches " + key))
    return out


tr.forward_losses = fl_wrapped
opt_step = tr.enc_optim.step


def opt_wrapped(*a, **k):
    s = torch.cuda.current_stream()
    e = torch.cuda.Event(enable_timing=True); e.record(s)
    marks.append(("optimiser begins", s.cuda_stream, e, e))
    return opt_step(*a, **k)


tr.enc_optim.step = opt_wrapped
chain = torch.cuda.Stream(device=dev, priority=-1)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
starts, ends = [], []
for i in range(steps):
    img, noise = pool[i % 4]
    with torch.cuda.stream(chain):
        e = torch.cuda.Event(enable_timing=True); e.record(chain); starts.append((e, len(marks)))
        tr.training_step({"image": img}, noise=noise)
        e = torch.cuda.Event(enable_timing=True); e.record(chain); ends.append(e)
torch.cuda.synchronize()
for k in range(steps - 2, steps):
    e0, m0 = starts[k]
    m1 = starts[k + 1][1] if k + 1 < steps else len(marks)
    print("step %d: %.2f ms" % (k, e0.elapsed_time(ends[k])))
    ids = {}
    for name, sid, a, b in marks[m0:m1]:
        ids.setdefault(sid, len(ids) + 1)
        print("   stream %d  %-28s %7.2f .. %7.2f ms" % (ids[sid], name, e0.elapsed_time(a), e0.elapsed_time(b)))
