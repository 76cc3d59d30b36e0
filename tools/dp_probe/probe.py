#!/usr/bin/env python3
"""Single-GPU probe of what collectives cost NEXT TO the one-workgroup-per-CU conv kernels.

Every torch.distributed collective of the data-parallel step (SyncBN statistics, VQ statistics, gradient buckets) is
replaced by a stand-in on a dedicated "communicator" stream with the same stream ordering ProcessGroupNCCL uses
(communicator stream waits for the producer, consumer waits for the communicator stream).  The stand-in launches a
few workgroups that need `--lds` bytes of LDS and spin ~`--us` microseconds, i.e. what an RCCL kernel needs from the
GPU, and doubles the tensor (a world of two identical replicas).  Compares step time against the plain step.

    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dp_probe/lds_probe.so tools/dp_probe/lds_probe.hip
    python tools/dp_probe/probe.py --lds 65536 --us 30
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def install_fake_world(lds, us, blocks):
    probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_probe.so"))
    probe.lds_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    comm = torch.cuda.Stream()
    sink = torch.zeros(4, device="cuda")
    spin = int(us * 100)            # clock64 ticks at 100 MHz
    count = [0]

    class Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def all_reduce(t, op=None, group=None, async_op=False):
        count[0] += 1
        cur = torch.cuda.current_stream()
        comm.wait_event(cur.record_event())
        with torch.cuda.stream(comm):
            rc = probe.lds_probe_launch(sink.data_ptr(), blocks, lds, spin, comm.cuda_stream)
            assert rc == 0
            t.mul_(2)
            ev = comm.record_event()
        t.record_stream(comm)
        if async_op:
            return Work(ev)
        cur.wait_event(ev)
        return None

    dist.is_initialized = lambda: True
    dist.is_available = lambda: True
    dist.get_world_size = lambda group=None: 2
    dist.get_rank = lambda group=None: 0
    dist.all_reduce = all_reduce
    return count


def run(dp, args):
    from trainers import FirstStepTrainer
    import bench
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    tr = FirstStepTrainer(device=dev, data_parallel=dp)
    pool = [bench.synthetic_batch(args.batch, args.size, 1234 + s, dev) for s in range(4)]
    chain = torch.cuda.Stream(device=dev, priority=-1)

    def step(i):
        img, noise = pool[i % len(pool)]
        with torch.cuda.stream(chain):
            return tr.training_step({"image": img}, noise=noise)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lds", type=int, default=65536)
    ap.add_argument("--us", type=float, default=30.0)
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    base = run(False, args)
    count = install_fake_world(args.lds, args.us, args.blocks)
    dp = run(True, args)
    n = count[0] / (args.steps + 3)
    print("plain step %.1f ms | with %d stand-in collectives per step (%d B LDS, %d workgroups, ~%.0f us each): %.1f ms  (%+.1f%%)"
          % (base, n, args.lds, args.blocks, args.us, dp, 100 * (dp / base - 1)))


if __name__ == "__main__":
    main()
