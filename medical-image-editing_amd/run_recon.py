"""Mask-guided reconstruction from an edited label map (reference: src/run_recon.py:169-228, the arithmetic of
`inner`): ids -> codebook lookup -> mask * rescale -> decoder (eval).  File polling / NIfTI / PNG I/O of the
reference's interactive loop is out of scope; this is the device-side part as one call."""
import torch

from hipops import ops


@torch.no_grad()
def reconstruct(encoder, decoder, label_map):
    """label_map: (B,H,W) integer map, 0 = masked-out (run_recon.py:179-186) -> recon (B,1,H,W) in (-1,1)."""
    encoder.eval()
    decoder.eval()
    mask, ids0, scale = ops.mask_scale(label_map)                       # mask, max(map,1)-1, numel/sum(mask)
    embed = ops.vq_lookup(ids0, encoder.vq.embed, mask=mask, scale=scale)  # lookup * mask * scale in one kernel
    return decoder(embed)


# ----------------------------------------------------------------------------------------------------
# file side of the reference's loop (run_recon.py:83-112, 150-228): checkpoint -> models, edited NIfTI label map ->
# reconstruction -> NIfTI / PNG.  nibabel is replaced by utils/nifti.py; matplotlib is used when importable.
# ----------------------------------------------------------------------------------------------------
LUNG_WINDOW = {'width': 1500, 'center': -550, 'scale': 2.0}            # run_recon.py:74-78


def save_as_nifti(data, path):
    """run_recon.py:83-87: (H, W) tensor -> transposed, both axes reversed, identity affine."""
    import numpy as np
    from utils import nifti
    a = data.detach().float().cpu().numpy() if torch.is_tensor(data) else np.asarray(data, dtype=np.float32)
    nifti.save(np.ascontiguousarray(a.transpose(1, 0)[::-1, ::-1]), path)


def load_from_nifti(path):
    """run_recon.py:90-95"""
    from utils import nifti
    data, _ = nifti.load(path)
    if data.ndim == 3:
        data = data[:, :, 0]
    return data.transpose(1, 0)[::-1, ::-1].copy()


def denormalize(image, width, center, scale):
    """utils/__init__.py:43-51"""
    vmax, vmin = center + width // 2, center - width // 2
    return (image / scale + 0.5) * (vmax - vmin) + vmin


def normalize(image, width=1500, center=-550, scale=2.0):
    """utils/__init__.py:17-28 (out of place)"""
    import numpy as np
    vmax, vmin = center + width // 2, center - width // 2
    return ((np.clip(image, vmin, vmax) - vmin) / (vmax - vmin) - 0.5) * scale


def load_model(config, device="cuda"):
    """run_recon.py:114-153: build the two networks from a config object and restore them from `resume_checkpoint`."""
    from networks import UNetEncoder, UNetDecoder
    from utils.checkpoint import init_from_ckpt
    encoder = UNetEncoder(in_channels=config.in_channels, filters=config.enc_filters, dict_size=config.dict_size,
                          momentum=config.momentum, knn_backend=config.knn_backend, use_styled_up_block=False, num_gpus=4,
                          init_embed=False)
    decoder = UNetDecoder(in_channels=config.enc_filters[0], out_channels=config.in_channels, filters=config.dec_filters,
                          use_dropblock=config.use_dropblock, block_size=config.block_size, start_value=config.start_value,
                          stop_value=config.stop_value, nr_steps=config.nr_steps, dropped_skip_layers=config.dropped_skip_layers,
                          use_styled_up_block=True, use_pixel_shuffle=config.use_pixel_shuffle)
    init_from_ckpt(config.resume_checkpoint, encoder, 'encoder', 'encoder.')
    init_from_ckpt(config.resume_checkpoint, decoder, 'decoder', 'decoder.')
    return encoder.to(device).eval(), decoder.to(device).eval()


def reconstruct_file(encoder, decoder, edited_path, out_nifti=None, out_png=None, window=None, flipud=False, device="cuda"):
    """One pass of `inner` (run_recon.py:169-228): edited label map file -> reconstruction (numpy, (H, W)).
    window = (width, center, scale) re-windows to the lung window as denorm_norm does (:156-167); flipud as CRCConfig."""
    import numpy as np
    m = load_from_nifti(edited_path).astype(np.int32)
    if flipud:
        m = np.flipud(m).copy()
    label = torch.from_numpy(m).long().unsqueeze(0).to(device)
    recon = reconstruct(encoder, decoder, label)[0, 0].float().cpu().numpy()
    if window is not None:
        recon = normalize(denormalize(recon, *window), **LUNG_WINDOW)
    if flipud:
        recon = np.flipud(recon).copy()
    if out_nifti:
        save_as_nifti(recon, out_nifti)
    if out_png:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.axis('off')
        plt.imshow(recon, cmap='gray', vmin=-1, vmax=1)          # utils/__init__.py:162-166
        plt.savefig(out_png, bbox_inches='tight', dpi=300)
        plt.clf()
    return recon
