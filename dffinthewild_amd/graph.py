"""Layer table of the depth-from-focus network: the host-side mirror of the weight contract.

Every conv of ``DFF_net`` (reference ``Depth_Estimation_Test/Depth_Estimation_Network.py:15-57,
131-330``) is one :class:`ConvSpec` row; the state-dict keys the reference's ``nn.Sequential``
nesting produces (384 for ``Network``) are derived from the rows, so that a
checkpoint written by the reference loads into :class:`dffinthewild_amd.Network` unchanged.
The HIP engine carries the same table in C++ (``class Table`` in ``csrc/dffw_engine.cpp``); ``tests/test_boundary.py``
checks the two against each other through ``dffw_param_info``.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple


@dataclass(frozen=True)
class ConvSpec:
    conv_key: str                 # state-dict prefix of the conv ("….weight" is appended)
    cin: int
    cout: int
    kernel: Tuple[int, int, int]  # (slice, row, col)
    transposed: bool = False
    bn_key: Optional[str] = None  # state-dict prefix of the BatchNorm3d that follows, if any
    bias: bool = False
    live: bool = True             # False: parameter exists in checkpoints but forward never reads it

    def weight_shape(self):
        a, b = (self.cin, self.cout) if self.transposed else (self.cout, self.cin)
        return (a, b) + tuple(self.kernel)


def _cb(prefix, cin, cout, k, **kw):
    """conv + BatchNorm pair as produced by the reference's convbn_3d (DEN.py:286-289)."""
    return ConvSpec(prefix + ".0", cin, cout, k, bn_key=prefix + ".1", **kw)


def _deconv(prefix, cin, cout):
    """ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1) + BatchNorm (DEN.py:41-48, 196-202, 258-262)."""
    return ConvSpec(prefix + ".0", cin, cout, (3, 3, 3), transposed=True, bn_key=prefix + ".1")


def _srd(prefix, c):
    """SRD block: two per-slice 3x3 convs then the cross-slice attention pair (DEN.py:295-330)."""
    return [
        _cb(prefix + ".Focus_Measure.conv.0", c, c, (1, 3, 3)),
        _cb(prefix + ".Focus_Measure.conv.2", c, c, (1, 3, 3)),
        ConvSpec(prefix + ".N_ch_attention.0", c, c, (3, 1, 1)),
        ConvSpec(prefix + ".N_ch_attention.2", c, c, (1, 1, 1)),
    ]


def _efd(prefix, cin, cout):
    """EFD block: strided conv branch + max-pool/conv branch (DEN.py:306-315)."""
    return [
        _cb(prefix + ".stride_conv", cin, cout, (3, 3, 3)),
        _cb(prefix + ".max_pooling.1", cin, cout, (3, 3, 3)),
    ]


def _hourglass(prefix, p):
    """Refinement hourglass (DEN.py:240-264); pre_conv is a dead parameter (never used in forward)."""
    k3 = (3, 3, 3)
    return [
        _cb(prefix + ".conv0.0", 2 * p, p, k3),
        _cb(prefix + ".conv1.0", p, 2 * p, k3),
        _cb(prefix + ".pre_conv.0", 2 * p, 2 * p, (1, 1, 1), live=False),
        _cb(prefix + ".conv2", 2 * p, 2 * p, k3),
        _cb(prefix + ".conv3.0", 2 * p, 2 * p, k3),
        _cb(prefix + ".conv4.0", 2 * p, 2 * p, k3),
        _deconv(prefix + ".conv5", 2 * p, 2 * p),
        _deconv(prefix + ".conv6", 2 * p, p),
    ]


def dff_net_convs(prefix: str = "DFF_net") -> List[ConvSpec]:
    """All conv rows of DFF_net in the reference's parameter registration order."""
    P = prefix
    k3 = (3, 3, 3)
    rows: List[ConvSpec] = []
    rows.append(_cb(P + ".FM_measure.Focus_extraction.0", 3, 8, (1, 9, 9)))
    rows += _srd(P + ".FM_measure.Focus_extraction.2", 8)
    rows += _efd(P + ".FM_conv1.0", 8, 16)
    rows += _srd(P + ".FM_conv1.1", 16)
    rows += _efd(P + ".FM_conv2.0", 16, 32)
    rows += _srd(P + ".FM_conv2.1", 32)
    S = P + ".SPP_module"
    for scale, c0 in (("8", 32), ("16", 64), ("32", 64)):
        rows.append(_cb(f"{S}.dres{scale}_0.0", 32, c0, k3))
        rows.append(_cb(f"{S}.dres{scale}_0.2", c0, c0, k3))
        rows.append(_cb(f"{S}.dres{scale}_1.0", c0, c0, k3))
        rows.append(_cb(f"{S}.dres{scale}_1.2", c0, c0, k3))
    rows.append(ConvSpec(S + ".conv1", 32, 64, k3))
    rows.append(_cb(S + ".conv2.0", 64, 64, k3))
    rows.append(ConvSpec(S + ".conv3", 64, 128, k3))
    rows.append(_cb(S + ".conv4.0", 128, 128, k3))
    rows.append(_deconv(S + ".conv8", 128, 64))
    rows.append(_deconv(S + ".conv9", 64, 32))
    rows.append(_cb(S + ".combine1.0", 128, 64, k3))
    rows.append(_cb(S + ".combine2.0", 192, 128, k3))
    rows.append(_cb(S + ".redir1", 32, 32, (1, 1, 1)))
    rows.append(_cb(S + ".redir2", 64, 64, (1, 1, 1)))
    rows.append(_cb(S + ".redir3", 128, 128, (1, 1, 1), live=False))
    rows.append(_cb(P + ".confidence.0", 32, 32, k3))
    rows.append(ConvSpec(P + ".confidence.2", 32, 1, k3))
    rows.append(_cb(P + ".dres0.0", 32, 64, k3))
    rows.append(_cb(P + ".dres0.2", 64, 64, k3))
    rows.append(_deconv(P + ".deconv_1", 64, 32))
    rows += _hourglass(P + ".dres2", 32)
    rows.append(_deconv(P + ".deconv_2", 32, 16))
    rows += _hourglass(P + ".dres3", 16)
    rows.append(_deconv(P + ".deconv_3", 16, 8))
    rows += _hourglass(P + ".dres4", 8)
    rows.append(ConvSpec(P + ".classif1.0", 32, 1, (1, 1, 1)))
    rows.append(ConvSpec(P + ".classif2.0", 16, 1, (1, 1, 1)))
    rows.append(ConvSpec(P + ".classif3.0", 8, 1, (1, 1, 1)))
    return rows


def _of_block(prefix, cin, cout):
    """resnet_block_2d_OF (End_to_End.py:135-145): convbn 1x3x3 (stride s) -> relu -> convbn 1x3x3, plus a
    bias-free 1x1x1 (stride s) shortcut called ``feature``.  Registration order: conv, then feature."""
    return [
        _cb(prefix + ".conv.0", cin, cout, (1, 3, 3)),
        _cb(prefix + ".conv.2", cout, cout, (1, 3, 3)),
        ConvSpec(prefix + ".feature", cin, cout, (1, 1, 1)),
    ]


def _alpha_head(prefix, cin, c):
    """conv1/conv2/conv3 of FlowNetwork (End_to_End.py:37-69): three convbn 1x3x3 + relu, then a biased
    1x3x3 conv to the 3 warp parameters (the AdaptiveAvgPool3d that follows has no parameters)."""
    return [
        _cb(prefix + ".0", cin, c, (1, 3, 3)),
        _cb(prefix + ".2", c, c, (1, 3, 3)),
        _cb(prefix + ".4", c, c, (1, 3, 3)),
        ConvSpec(prefix + ".6", c, 3, (1, 3, 3), bias=True),
    ]


FLOW_PLANES = 8     # FlowNetwork(inplanes=8), End_to_End.py:11
FLOW_SLICES = 10    # AdaptiveAvgPool3d((10,1,1)), End_to_End.py:46,57,68: the alignment net is built for 10 slices


def flow_net_convs(prefix: str = "optical_flow_aggregation") -> List[ConvSpec]:
    """All conv rows of FlowNetwork (End_to_End.py:18-69) in registration order (138 state-dict entries)."""
    P, C = prefix, FLOW_PLANES
    rows: List[ConvSpec] = []
    rows += _of_block(P + ".OF_feature.0", 3, C) + _of_block(P + ".OF_feature.1", C, C)
    rows += _of_block(P + ".OF_feature1.0", C, 2 * C) + _of_block(P + ".OF_feature1.1", 2 * C, 2 * C)
    rows += _of_block(P + ".OF_feature2.0", 2 * C, 4 * C) + _of_block(P + ".OF_feature2.1", 4 * C, 4 * C)
    rows += _alpha_head(P + ".conv1", 8 * C + 2, 8 * C)
    rows += _alpha_head(P + ".conv2", 4 * C + 2, 4 * C)
    rows += _alpha_head(P + ".conv3", 2 * C + 2, 2 * C)
    return rows


def e2e_convs() -> List[ConvSpec]:
    """End_to_End.Network (End_to_End.py:9-16): DFF_net registered first, then optical_flow_aggregation
    (522 state-dict entries)."""
    return dff_net_convs("DFF_net") + flow_net_convs("optical_flow_aggregation")


# role tags used by synth.py to pick a distribution per entry
ROLE_CONV, ROLE_CONVT, ROLE_BIAS = "conv", "convT", "bias"
ROLE_BN_W, ROLE_BN_B, ROLE_BN_MEAN, ROLE_BN_VAR, ROLE_BN_NBT = "bn_w", "bn_b", "bn_mean", "bn_var", "bn_nbt"


def param_entries(convs: List[ConvSpec]):
    """Yield ``(key, shape, role, is_buffer)`` for every state-dict entry, in registration order
    (conv weight [, bias], then the BatchNorm's weight, bias, running_mean, running_var,
    num_batches_tracked — the order ``nn.Module.state_dict`` emits)."""
    for c in convs:
        yield (c.conv_key + ".weight", c.weight_shape(), ROLE_CONVT if c.transposed else ROLE_CONV, False)
        if c.bias:
            yield (c.conv_key + ".bias", (c.cout,), ROLE_BIAS, False)
        if c.bn_key:
            yield (c.bn_key + ".weight", (c.cout,), ROLE_BN_W, False)
            yield (c.bn_key + ".bias", (c.cout,), ROLE_BN_B, False)
            yield (c.bn_key + ".running_mean", (c.cout,), ROLE_BN_MEAN, True)
            yield (c.bn_key + ".running_var", (c.cout,), ROLE_BN_VAR, True)
            yield (c.bn_key + ".num_batches_tracked", (), ROLE_BN_NBT, True)


def check_stack_shape(FS_shape, focus_shape=None):
    """Shape contract of ``Network.forward``: FS is (B,3,N,H,W) with H and W
    multiples of 32 (two stride-2 stages then an 8x8 pool, DEN.py:149,309); focus_dists must
    broadcast against (B,N,H,W) (DEN.py:90)."""
    if len(FS_shape) != 5:
        raise ValueError(f"FS must be 5-D (B,3,N,H,W), got {tuple(FS_shape)}")
    B, C, N, H, W = (int(v) for v in FS_shape)
    if C != 3:
        raise ValueError(f"FS must have 3 colour channels on dim 1 (B,3,N,H,W), got {C}")
    if N < 1 or B < 1:
        raise ValueError("FS needs at least one sample and one focal slice")
    if H % 32 or W % 32 or H < 32 or W < 32:
        raise ValueError(f"H and W must be positive multiples of 32 (pad with -1 like the reference loaders), got {H}x{W}")
    if focus_shape is not None:
        fs = tuple(int(v) for v in focus_shape)
        if len(fs) != 4:
            raise ValueError(f"focus_dists must be 4-D (B,N,H,W) or (B,N,1,1), got {fs}")
        for got, want, name in zip(fs, (B, N, H, W), "BNHW"):
            if got != want and got != 1:
                raise ValueError(f"focus_dists dim {name}={got} does not broadcast to {want}")
    return B, N, H, W


def check_e2e_shape(FS_shape, focus_shape, fov_shape):
    """Shape contract of ``End_to_End.Network.forward(FS, focus_dists, FOVs)`` (End_to_End.py:13-16,
    Test_dataloader.py:54-70): FS (B,3,10,H,W); focus_dists broadcastable to (B,10,H,W); FOVs one relative
    field of view per slice, (B,1,10,1,1) (any shape with B*10 elements in (b,n) order is accepted)."""
    B, N, H, W = check_stack_shape(FS_shape, focus_shape)
    if N != FLOW_SLICES:
        raise ValueError(f"the alignment network is built for {FLOW_SLICES} focal slices "
                         f"(AdaptiveAvgPool3d((10,1,1)), End_to_End.py:46), got {N}")
    n = 1
    for v in fov_shape:
        n *= int(v)
    if n != B * N:
        raise ValueError(f"FOVs must hold one value per (sample, slice) = {B}x{N}, got shape {tuple(fov_shape)}")
    return B, N, H, W
