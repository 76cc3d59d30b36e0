"""TEST INFRASTRUCTURE ONLY — CPU restatement of the second training step's discriminator path.

Follows the reference: networks/discriminator.py:18-87 (NLayerDiscriminator: 4x4 convs, padding 1, stride 2 except the
last two, BatchNorm2d + LeakyReLU(0.2)), functions/gan_loss.py:6-10 (hinge_d_loss), trainers/single_window_trainer.py:
434-488 (_train_second_step_nl_dis).  Functional torch-CPU code over a state dict with the reference's key names
(`main.0.weight`, ...).  Parity PINNED: tests/golden/gan.npz holds outputs / gradients / updated buffers produced by the
reference's own NLayerDiscriminator and hinge_d_loss (tests/golden/make_golden.py::gen_gan).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import torch
import torch.nn.functional as F


def discriminator_layout(n_layers=3):
    """[(kind, index in `main`, stride)] in the order of discriminator.py:56-81."""
    lay = [("conv", 0, 2), ("lrelu", 1, 0)]
    i = 2
    for _ in range(1, n_layers):
        lay += [("conv", i, 2), ("bn", i + 1, 0), ("lrelu", i + 2, 0)]
        i += 3
    lay += [("conv", i, 1), ("bn", i + 1, 0), ("lrelu", i + 2, 0), ("conv", i + 3, 1)]
    return lay


def discriminator_forward(P, x, training=True, n_layers=3, momentum=0.1, eps=1e-5):
    """P: state dict (running stats are updated in place when training, like nn.BatchNorm2d)."""
    for kind, idx, stride in discriminator_layout(n_layers):
        pre = "main.%d." % idx
        if kind == "conv":
            x = F.conv2d(x, P[pre + "weight"], P.get(pre + "bias"), stride=stride, padding=1)
        elif kind == "bn":
            x = F.batch_norm(x, P[pre + "running_mean"], P[pre + "running_var"], P[pre + "weight"], P[pre + "bias"],
                             training, momentum, eps)
            if training and pre + "num_batches_tracked" in P:
                P[pre + "num_batches_tracked"] += 1
        else:
            x = F.leaky_relu(x, 0.2)
    return x


def hinge_d_loss(logits_real, logits_fake):
    """gan_loss.py:6-10"""
    loss_real = torch.mean(F.relu(1. - logits_real))
    loss_fake = torch.mean(F.relu(1. + logits_fake))
    return 0.5 * (loss_real + loss_fake)


def generator_loss(logits_fake):
    """single_window_trainer.py:463"""
    return -torch.mean(logits_fake)
