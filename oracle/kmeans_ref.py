"""CPU oracle of the codebook initialisation `UNetEncoder.initialize_embed` (reference `src/networks/unet_encoder.py:66-91`,
wired at `src/trainers/base.py:201`).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path (hipops.ops.kmeans_codebook on the VQ kernels) never does.

The reference delegates to the third-party `kmeans_pytorch.kmeans` (`requirements.txt:52`: kmeans-pytorch==0.3.0), which is
not installed and cannot be fetched: **parity unpinned** against the reference's own output.  This file restates that
package's published algorithm (Lloyd iterations on squared Euclidean distances, `torch.argmin` assignment = ties to the lowest
centre index, per-centre mean, stop when (sum_k ||c_k - c_k'||_2)^2 < tol, tol = 1e-4) around the reference's call site
(`kmeans(X=features (P, D), num_clusters=dict_size, distance='euclidean')`, centres copied into `vq.embed`), with the two
deviations the build defines and documents:

  * start: `dict_size` distinct rows drawn by `torch.randperm(P, generator=seed)[:dict_size]` (the package draws them with
    `np.random.choice(P, K, replace=False)` from numpy's global state - not reproducible across processes / ranks);
  * an empty cluster keeps its centre (the package's `selected.mean(dim=0)` of zero rows is NaN and poisons the codebook).

Arithmetic: distances and means in float64 over the float32 inputs, centres rounded to float32 once per iteration (the HIP
path searches in fp32 with the VQ score formula and accumulates the means in double): assignments must agree wherever the
top-1 / top-2 distance gap is clear of fp32 rounding - `min_gap` of the returned trace says how clear the fixture is.
"""
import numpy as np
import torch


def initial_rows(P, K, seed):
    """Row indices of the starting centres: the same draw as hipops.ops.kmeans_codebook."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.randperm(P, generator=g)[:K].numpy()


def kmeans(features, dict_size, seed=0, tol=1e-4, max_iter=100):
    """-> (centres (K, D) float32, assignment (P,) int64 of the LAST search, trace): trace[i] = dict(inertia = sum of squared
    distances to the assigned centre, shift = sum_k ||delta_k||, empty = clusters without members, min_gap = smallest
    top-1 / top-2 squared-distance gap of the iteration's search)."""
    x = np.asarray(features.detach().cpu() if torch.is_tensor(features) else features, dtype=np.float32)
    P, D = x.shape
    K = int(dict_size)
    if P < K:
        raise RuntimeError("k-means needs at least dict_size = %d feature rows, got %d" % (K, P))
    x64 = x.astype(np.float64)
    xn = (x64 * x64).sum(1)
    centres = x[initial_rows(P, K, seed)].copy()
    trace, ids = [], None
    for _ in range(int(max_iter)):
        c64 = centres.astype(np.float64)
        d2 = xn[:, None] - 2.0 * (x64 @ c64.T) + (c64 * c64).sum(1)[None, :]        # (P, K)
        ids = d2.argmin(1)                                                          # ties -> lowest index
        part = np.partition(d2, 1, axis=1) if K > 1 else np.concatenate([d2, d2 + np.inf], 1)
        best = d2[np.arange(P), ids]
        new = c64.copy()
        counts = np.bincount(ids, minlength=K)
        sums = np.zeros((K, D))
        np.add.at(sums, ids, x64)
        has = counts > 0
        new[has] = sums[has] / counts[has, None]
        new32 = new.astype(np.float32)
        shift = float(np.sqrt(((new32.astype(np.float64) - c64) ** 2).sum(1)).sum())
        trace.append(dict(inertia=float(np.maximum(best, 0.0).sum()), shift=shift, empty=int((~has).sum()),
                          min_gap=float((part[:, 1] - part[:, 0]).min())))
        centres = new32
        if shift ** 2 < tol:
            break
    return centres, ids.astype(np.int64), trace


def blobs(P, D, K, seed, spread=4.0, noise=0.35):
    """Synthetic feature rows: K Gaussian blobs (centres ~ spread * N(0, 1), members + noise * N(0, 1)), shuffled."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    c = torch.randn(K, D, generator=g) * spread
    lab = torch.randint(0, K, (P,), generator=g)
    x = c[lab] + noise * torch.randn(P, D, generator=g)
    return x.float(), lab, c.float()
