"""Checkpoint wire format of the reference (PyTorch-Lightning `.ckpt` = torch.save'd dict whose 'state_dict' holds the
LightningModule's parameters under the attribute prefixes `encoder.`, `decoder.`, `dis.`): trainers/base.py:85-114,
run_recon.py:98-112.  The modules of this build keep the reference's state_dict keys, so these are plain filters."""
import torch


def _state_dict(path):
    return torch.load(path, map_location='cpu')['state_dict']


def load_first_stage_from_ckpt(path, encoder, decoder=None, load_only_enc=False):
    """base.py:85-102: encoder strictly, decoder with strict=False."""
    sd = _state_dict(path)
    enc = {k[len('encoder.'):]: v for k, v in sd.items() if k.startswith('encoder')}
    dec = {k[len('decoder.'):]: v for k, v in sd.items() if k.startswith('decoder')}
    encoder.load_state_dict(enc, strict=True)
    if not load_only_enc and decoder is not None:
        decoder.load_state_dict(dec, strict=False)
    return encoder, decoder


def load_discriminator_from_ckpt(path, dis):
    """base.py:104-113"""
    sd = _state_dict(path)
    dis.load_state_dict({k[len('dis.'):]: v for k, v in sd.items() if k.startswith('dis')}, strict=True)
    return dis


def init_from_ckpt(path, model, key_name, delete_string='model.'):
    """run_recon.py:98-112: keep the keys that start with `key_name`, strip `delete_string` from those that carry it."""
    sd = _state_dict(path)
    new = {}
    for k, v in sd.items():
        if k.startswith(delete_string):
            new[k[len(delete_string):]] = v
        elif k.startswith(key_name):
            new[k] = v
    model.load_state_dict(new, strict=True)
    return model


def save_lightning_style_ckpt(path, encoder=None, decoder=None, dis=None, extra=None):
    """Write a checkpoint the reference's loaders accept (plain contiguous tensors, attribute prefixes)."""
    sd = {}
    for pre, m in (("encoder.", encoder), ("decoder.", decoder), ("dis.", dis)):
        if m is not None:
            for k, v in m.state_dict().items():
                sd[pre + k] = v.detach().cpu().contiguous().clone()
    d = {"state_dict": sd}
    d.update(extra or {})
    torch.save(d, path)
