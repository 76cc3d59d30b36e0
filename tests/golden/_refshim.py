"""Loader for the upstream reference's hot-path files (generator-side only).

Used ONLY by tests/golden/make_golden.py in the build container, where
/root/reference exists.  Nothing under tests/ that runs on the GPU box imports
this module.  It registers empty stand-in *packages* (`networks`, `networks.vq`,
`functions`) whose __path__ points at the reference directories, so that the
reference's relative imports resolve to its own files without executing its
package __init__ files (those pull kornia / lightning / nibabel, absent
offline).  Two third-party names the hot-path files import but never call on
this path are provided as inert stubs:

  * kmeans_pytorch.kmeans           (unet_encoder.py:4, only used by initialize_embed)
  * utils.get_world_size/is_distributed  (vq_module.py:20-21; semantics of
    utils/__init__.py:109-114: keyed off env WORLD_SIZE)
"""
import importlib.util
import os
import sys
import types

REF_SRC = "/root/reference/src"


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_SRC, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference sources not present (this only runs in the build container)")
    km = types.ModuleType("kmeans_pytorch")

    def _kmeans(*a, **k):
        raise RuntimeError("kmeans_pytorch is not available offline")
    km.kmeans = _kmeans
    sys.modules["kmeans_pytorch"] = km

    ut = types.ModuleType("utils")
    ut.get_world_size = lambda: int(os.environ.get("WORLD_SIZE", 1))
    ut.is_distributed = lambda: ut.get_world_size() > 1
    sys.modules["utils"] = ut

    nets = _pkg("networks", os.path.join(REF_SRC, "networks"))
    vq = _pkg("networks.vq", os.path.join(REF_SRC, "networks", "vq"))
    fn = _pkg("functions", os.path.join(REF_SRC, "functions"))

    _load("networks.vq.grad_approximation", "networks/vq/grad_approximation.py")
    vqm = _load("networks.vq.vq_module", "networks/vq/vq_module.py")
    vq.VQ = vqm.VQModule
    vq.VQModule = vqm.VQModule
    _load("networks.initialize", "networks/initialize.py")
    _load("networks.dropblock", "networks/dropblock.py")
    nets.blocks = _load("networks.blocks", "networks/blocks.py")
    nets.aspp = _load("networks.aspp", "networks/aspp.py")
    nets.unet_encoder = _load("networks.unet_encoder", "networks/unet_encoder.py")
    nets.unet_decoder = _load("networks.unet_decoder", "networks/unet_decoder.py")
    nets.vqwnet = _load("networks.vqwnet", "networks/vqwnet.py")
    fn.embed_loss = _load("functions.embed_loss", "functions/embed_loss.py")
    fn.onehot = _load("functions.onehot", "functions/onehot.py")
    fn.seg_loss = _load("functions.seg_loss", "functions/seg_loss.py")
    # second training step: PatchGAN discriminator + hinge loss
    _load("networks.actnorm", "networks/actnorm.py")
    nets.discriminator = _load("networks.discriminator", "networks/discriminator.py")
    fn.gan_loss = _load("functions.gan_loss", "functions/gan_loss.py")
    return types.SimpleNamespace(
        NLayerDiscriminator=nets.discriminator.NLayerDiscriminator, hinge_d_loss=fn.gan_loss.hinge_d_loss,
        blocks=nets.blocks, aspp=nets.aspp, vq_module=vqm,
        dropblock=sys.modules["networks.dropblock"],
        UNetEncoder=nets.unet_encoder.UNetEncoder,
        UNetDecoder=nets.unet_decoder.UNetDecoder,
        VQWNet=nets.vqwnet.VQWNet,
        EmbeddingLoss=fn.embed_loss.EmbeddingLoss,
        OneHotEncoder=fn.onehot.OneHotEncoder,
        SoftDiceLoss=fn.seg_loss.SoftDiceLoss, FocalLoss=fn.seg_loss.FocalLoss,
    )


def load_reference_utils():
    """utils/__init__.py of the reference (CT windows, norm / denorm, load_json).  Its imports of nibabel, matplotlib,
    Lightning-based `.logger` and `.init_seed` are never touched by those functions and are given as inert stubs; the
    module is loaded under the name `refutils` so that the `utils` stub above stays in place."""
    import importlib.machinery
    for name in ("nibabel",):
        if importlib.util.find_spec(name) is None and name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, mpl.pyplot
    for sub, names in (("logger", ("ModelSaver", "Logger")), ("init_seed", ("InitSeedAndSaveConfig",))):
        m = types.ModuleType("refutils." + sub)
        for n in names:
            setattr(m, n, type(n, (), {}))
        sys.modules["refutils." + sub] = m
    path = os.path.join(REF_SRC, "utils", "__init__.py")
    spec = importlib.util.spec_from_file_location("refutils", path, submodule_search_locations=[])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refutils"] = mod
    spec.loader.exec_module(mod)
    return mod
