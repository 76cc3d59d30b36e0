"""GAN losses of the second training step (reference: functions/gan_loss.py:6-17)."""
from hipops import ops


def hinge_d_loss(logits_real, logits_fake):
    """0.5 * (mean(relu(1 - real)) + mean(relu(1 + fake)))"""
    return ops.weighted_sum([ops.hinge_real(logits_real), ops.hinge_fake(logits_fake)], [0.5, 0.5])


def generator_loss(logits_fake):
    """-mean(dis(recon)) (single_window_trainer.py:463)"""
    return ops.neg_mean(logits_fake)
