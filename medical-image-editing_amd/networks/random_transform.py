"""Two-view augmentation on the device (reference: networks/random_transform.py:10-112).

Same class name, constructor (`config.modules` + one sub-config per module), `forward(x) -> (x, clear_x)`,
`forward_transform(ids)` and `reverse_transform(ids)` as the reference, which assembles them from kornia 0.5.1.  kornia
is not part of this build: the random parameters are drawn on the host from a seeded generator, the per-sample 3x3
matrices are kept like kornia's `return_transform=True` ones (source pixel -> destination pixel), and the pixel work
runs in the HIP kernels of csrc/augment.hip.  The arithmetic is the one oracle/augment_ref.py states (parity against
kornia unpinned: it is not installed offline).  A grey image expanded to RGB and converted back (reference :77, :91-92)
is the identity up to one rounding, so the image stays single-channel here.
"""
import torch
import torch.nn as nn

from hipops import ops


def _cfg(config, name, key, default=None):
    sub = config.get(name) if isinstance(config, dict) else getattr(config, name, None)
    if sub is None:
        return default
    if isinstance(sub, dict):
        return sub.get(key, default)
    return getattr(sub, key, default)


class RandomTransform(nn.Module):
    GEOMETRIC = ("RandomHorizontalFlip", "RandomAffine")
    PHOTOMETRIC = ("ColorJitter", "RandomGaussianBlur", "RandomPosterize", "RandomGaussianNoise")

    def __init__(self, config, seed=0):
        super().__init__()
        modules = config["modules"] if isinstance(config, dict) else config.modules
        self.config = config
        unknown = [m for m in modules if m not in self.GEOMETRIC + self.PHOTOMETRIC]
        if unknown:
            raise ValueError("RandomTransform: unknown augmentation modules %s" % unknown)
        self.geometrics = [m for m in modules if m in self.GEOMETRIC]
        self.photometrics = [m for m in modules if m in self.PHOTOMETRIC]
        self.generator = torch.Generator().manual_seed(int(seed))
        self._transforms, self._on_device = [], []

    # -- host-side parameter sampling (float64, one row per sample)
    def _uniform(self, n, lo, hi):
        return lo + (hi - lo) * torch.rand(n, generator=self.generator, dtype=torch.float64)

    def _bernoulli(self, n, p):
        return torch.rand(n, generator=self.generator, dtype=torch.float64) < float(p)

    def _geometric_matrix(self, name, B, H, W):
        m = torch.eye(3, dtype=torch.float64).repeat(B, 1, 1)
        if name == "RandomHorizontalFlip":
            on = self._bernoulli(B, _cfg(self.config, name, "p", 0.5))
            m[on, 0, 0] = -1.0
            m[on, 0, 2] = W - 1.0
            return m
        deg = _cfg(self.config, name, "degrees", 0.0) or 0.0
        tr = _cfg(self.config, name, "translate", None) or (0.0, 0.0)
        sh = _cfg(self.config, name, "shear", None) or 0.0
        deg = (-deg, deg) if not isinstance(deg, (tuple, list)) else tuple(deg)
        sh = (-sh, sh) if not isinstance(sh, (tuple, list)) else tuple(sh)[:2]
        on = self._bernoulli(B, _cfg(self.config, name, "p", 0.5))
        ang = self._uniform(B, deg[0], deg[1])
        tx = self._uniform(B, -tr[0] * W, tr[0] * W)
        ty = self._uniform(B, -tr[1] * H, tr[1] * H)
        sx = self._uniform(B, sh[0], sh[1])
        # back @ rot @ shear @ to_centre, written out (vectorised over the batch)
        cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
        a = torch.deg2rad(ang)
        c, s_, t = torch.cos(a), torch.sin(a), -torch.tan(torch.deg2rad(sx))
        r00, r01 = c, c * t + s_           # rot @ shear, first row
        r10, r11 = -s_, -s_ * t + c        # second row
        aff = torch.eye(3, dtype=torch.float64).repeat(B, 1, 1)
        aff[:, 0, 0], aff[:, 0, 1], aff[:, 0, 2] = r00, r01, cx + tx - (r00 * cx + r01 * cy)
        aff[:, 1, 0], aff[:, 1, 1], aff[:, 1, 2] = r10, r11, cy + ty - (r10 * cx + r11 * cy)
        m[on] = aff[on]
        return m

    def _photometric_op(self, name, B):
        """-> ("photometric", params (B,4) float32, needs_noise) or ("blur", taps, apply (B,) uint8)"""
        on = self._bernoulli(B, _cfg(self.config, name, "p", 0.5))
        p = torch.zeros(B, 4, dtype=torch.float64)
        p[:, 1] = 1.0
        p[:, 2] = 8.0
        if name == "ColorJitter":
            br = float(_cfg(self.config, name, "brightness", 0.0) or 0.0)
            ct = float(_cfg(self.config, name, "contrast", 0.0) or 0.0)
            badd = self._uniform(B, max(0.0, 1.0 - br), 1.0 + br) - 1.0
            cmul = self._uniform(B, max(0.0, 1.0 - ct), 1.0 + ct)
            p[on, 0] = badd[on]
            p[on, 1] = cmul[on]
            return ("photometric", p.float(), False)
        if name == "RandomPosterize":
            lo = int(_cfg(self.config, name, "bits", 3))
            bits = torch.randint(lo, 9, (B,), generator=self.generator).double()
            p[on, 2] = bits[on]
            return ("photometric", p.float(), False)
        if name == "RandomGaussianNoise":
            p[on, 3] = float(_cfg(self.config, name, "std", 1.0))
            return ("photometric", p.float(), True)
        k = int(_cfg(self.config, name, "kernel", 3))
        sigma = float(_cfg(self.config, name, "sigma", 1.0))
        r = torch.arange(k, dtype=torch.float64) - (k - 1) / 2.0
        g = torch.exp(-(r * r) / (2.0 * sigma * sigma))
        return ("blur", (g / g.sum()).float(), on.to(torch.uint8))

    # -- reference entry points
    def forward(self, x):
        """x: (B, 1, H, W) in [0, 1].  Returns (augmented, clear) like random_transform.py:76-94."""
        B, _, H, W = x.shape
        self._transforms, self._on_device = [], []
        for name in self.geometrics:
            m = self._geometric_matrix(name, B, H, W)
            self._transforms.append(m)
            # both directions go to the device once: (destination->source of the forward warp, of the inverse warp)
            pair = torch.stack([torch.linalg.inv(m), m]).float().to(x.device, non_blocking=True)
            self._on_device.append((pair[0], pair[1]))
            x = ops.warp_image(x, pair[0])
        clear_x = x.detach().clone()
        for name in self.photometrics:
            op = self._photometric_op(name, B)
            if op[0] == "blur":
                x = ops.gauss_blur(x, op[1].to(x.device), op[2].to(x.device))
            else:
                noise = torch.randn(x.shape, device=x.device, dtype=torch.float32) if op[2] else None
                x = ops.photometric(x, op[1].to(x.device), noise)
        return x, clear_x

    def forward_transform(self, x):
        """Warp an id map (B, H, W) into this view's frame (nearest, 0 = out of frame); int32 result."""
        for minv, _ in self._on_device:
            x = ops.warp_labels(x, minv)
        return x

    def reverse_transform(self, x):
        """Warp an id map of this view back to the un-augmented frame (inverse matrices, reverse order)."""
        for _, mfwd in reversed(self._on_device):
            x = ops.warp_labels(x, mfwd)      # destination->source of the inverse warp = the forward matrix
        return x
