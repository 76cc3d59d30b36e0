"""Per-tap / per-block error of the weight gradient the library computes for a plain 3x3 layer vs an fp64 reference."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
from hipops import _lib
L = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, H, W, Cin, Cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (2, 32, 32, 32, 32))]
torch.manual_seed(0)
x = torch.randn(N, H, W, Cin, device="cuda")
dy = torch.randn(N, H, W, Cout, device="cuda")
dw = torch.empty(Cout, 3, 3, Cin, device="cuda")
ws = torch.empty(L.vqw_conv2d_wgrad_ws_bytes(Cin, 0, N, H, W, Cout, 3), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(L.vqw_conv2d_wgrad(p(x), Cin, 0, None, 0, p(dy), p(dw), None, p(ws), ws.numel(), N, H, W, Cout, 3, 1, 0, st))
torch.cuda.synchronize()
xr = x.double().cpu().permute(0, 3, 1, 2)
dr = dy.double().cpu().permute(0, 3, 1, 2)
ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dr, padding=1).permute(0, 2, 3, 1)      # OHWI
got = dw.double().cpu()
print("overall rel", float((got - ref).norm() / ref.norm()))
for ky in range(3):
    print("tap row %d:" % ky, ["%.2e" % float((got[:, ky, kx] - ref[:, ky, kx]).norm() / ref[:, ky, kx].norm()) for kx in range(3)])
for cb in range(0, Cout, 16):
    print("co %3d:" % cb, ["%.1e" % float((got[cb:cb + 16, :, :, ib:ib + 16] - ref[cb:cb + 16, :, :, ib:ib + 16]).norm() / ref[cb:cb + 16, :, :, ib:ib + 16].norm()) for ib in range(0, Cin, 16)])
