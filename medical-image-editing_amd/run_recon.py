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
