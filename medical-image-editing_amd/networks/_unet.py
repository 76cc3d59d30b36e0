"""Shared construction / traversal of one U-Net half (four ResBlock levels, a bottleneck, four up blocks).

The reference spells every level out as its own attribute (`down_conv1_1` ... `up_conv2_1`); checkpoints, parameter
order, the order in which the initialiser consumes the RNG stream and the DP bucket order all hang on those names and
on the order they are created in.  Here the names are generated; the creation order is the reference's
(down 1..n, bottleneck, up n..1).
"""
from .blocks import ResBlock, DoubleConv


def add_half(net, stage, in_channels, filters, make_up):
    """Register down_conv{stage}_k (k = 1..n), double_conv{stage}, up_conv{stage}_k (k = n..1) on `net`.
    make_up(k) builds the up block of level k (1-based, level 1 = full resolution)."""
    n = len(filters) - 1
    widths = [in_channels] + list(filters)
    for k in range(1, n + 1):
        net.add_module("down_conv%d_%d" % (stage, k), ResBlock(widths[k - 1], widths[k]))
    net.add_module("double_conv%d" % stage, DoubleConv(filters[n - 1], filters[n]))
    for k in range(n, 0, -1):
        net.add_module("up_conv%d_%d" % (stage, k), make_up(k))
    return n


def level_modules(net, stage, n):
    downs = [getattr(net, "down_conv%d_%d" % (stage, k)) for k in range(1, n + 1)]
    ups = [getattr(net, "up_conv%d_%d" % (stage, k)) for k in range(n, 0, -1)]
    return downs, getattr(net, "double_conv%d" % stage), ups


def run_half(net, stage, n, x):
    """down path (collecting skips) -> bottleneck -> up path (consuming the skips deepest first)."""
    downs, mid, ups = level_modules(net, stage, n)
    skips = []
    for d in downs:
        x, s = d(x)
        skips.append(s)
    x = mid(x)
    for u in ups:
        x = u(x, skips.pop())
    return x
