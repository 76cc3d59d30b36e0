#!/usr/bin/env python3
"""Direct (virtual up-sampling in the loader) vs collapsed forms of the 3x3-over-upsampled convolutions."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
from hipops import _lib
L = _lib.load()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N = 32
for cin, cout, H in ((64, 32, 256), (128, 64, 128), (256, 128, 64), (512, 256, 32)):
    h = H // 2
    x = torch.randn(N, h, h, cin, device="cuda"); w = torch.randn(cout, 3, 3, cin, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda"); y = torch.empty(N, H, H, cout, device="cuda"); dy = torch.randn_like(y)
    wt = torch.empty(cin * 9 * cout, device="cuda"); gfull = torch.empty(N, H, H, cin, device="cuda"); g0 = torch.empty_like(x)
    ws = torch.empty(L.vqw_conv3x3_up2_ws_bytes(cin, cout), dtype=torch.uint8, device="cuda")
    _lib.check(L.vqw_conv3x3_up2_prepare(p(w), p(ws), ws.numel(), cin, cout, st()))
    _lib.check(L.vqw_pack_dgrad_weights(p(w), p(wt), cout, cin, 3, st()))
    t_df = timeit(lambda: _lib.check(L.vqw_conv2d_fwd(p(x), cin, 1, None, 0, p(w), p(b), p(y), N, H, H, cout, 3, 1, 0, st())))
    t_cf = timeit(lambda: _lib.check(L.vqw_conv3x3_up2_fwd(p(x), p(ws), p(b), p(y), N, h, h, cin, cout, 0, st())))
    def dgrad_direct():
        _lib.check(L.vqw_conv2d_fwd(p(dy), cout, 0, None, 0, p(wt), None, p(gfull), N, H, H, cin, 3, 1, 0, st()))
        _lib.check(L.vqw_input_grad_gather(p(gfull), cin, 0, cin, 1, p(g0), 0, N, H, H, st()))
    t_dd = timeit(dgrad_direct)
    t_cd = timeit(lambda: _lib.check(L.vqw_conv3x3_up2_dgrad(p(dy), p(ws), p(g0), N, h, h, cin, cout, st())))
    t_pr = timeit(lambda: _lib.check(L.vqw_conv3x3_up2_prepare(p(w), p(ws), ws.numel(), cin, cout, st())))
    print("%3d->%3d @%3d  fwd direct %.3f ms  collapsed %.3f ms | dgrad direct(+gather) %.3f ms  collapsed %.3f ms | prepare %.3f ms"
          % (cin, cout, H, t_df, t_cf, t_dd, t_cd, t_pr), flush=True)
