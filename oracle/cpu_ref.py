"""ORACLE — test infrastructure, not product code.

CPU restatement (PyTorch-CPU, fp32, functional) of the reference's depth-from-focus forward pass,
``DFF_net.forward`` in ``/root/reference/Depth_Estimation_Test/Depth_Estimation_Network.py:74-127``, and of
the End_to_End variant (``/root/reference/End_to_End/End_to_End.py``: FlowNetwork alignment + DFF_net).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``dffinthewild_amd/`` does.

Where the arithmetic lives: the reference owns no operator code — every multiply happens inside
PyTorch (README pins torch==1.6.0; this image has 2.10.0).  The operator semantics relied on are
all PyTorch defaults and are restated below where they are used: zero padding, ``ceil_mode=False``
pooling, ``align_corners=False`` bilinear resize (``F.upsample`` default), softplus beta=1 /
threshold=20, eval-mode BatchNorm with eps=1e-5.

Pinning: the reference ships no tests or golden vectors for this path, so
parity is pinned by running the reference itself in the build container on synthetic weights
(``oracle/make_goldens.py``) and committing inputs-by-recipe + outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against them to <=1e-5 relative L2.

The graph is written here as plain functions over a flat ``{state-dict key: tensor}`` mapping so
it shares no module structure with the product (``dffinthewild_amd``) or with the reference.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm3d default, relied on by DEN.py:286-289


class _W:
    """Read-only view of a state dict with a key prefix."""

    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def sub(self, name):
        return _W(self.sd, self.prefix + name + ".")


def _bn(x, w, key):
    """Eval-mode BatchNorm3d: (x-mean)/sqrt(var+eps)*gamma+beta (DEN.py:289)."""
    return F.batch_norm(x, w(key + ".running_mean"), w(key + ".running_var"),
                        w(key + ".weight"), w(key + ".bias"), False, 0.0, BN_EPS)


def _conv(x, w, key, stride=1, pad=0, dil=1):
    return F.conv3d(x, w(key + ".weight"), None, stride, pad, dil)


def _conv_bn(x, w, key, stride=1, pad=1, dil=1):
    """convbn_3d: bias-free Conv3d followed by BatchNorm3d (DEN.py:286-289)."""
    return _bn(_conv(x, w, key + ".0", stride, pad, dil), w, key + ".1")


def _up_bn(x, w, key):
    """ConvTranspose3d k3 s(1,2,2) p1 output_padding(0,1,1) + BatchNorm3d (DEN.py:41-42 etc.)."""
    y = F.conv_transpose3d(x, w(key + ".0.weight"), None, (1, 2, 2), 1, (0, 1, 1))
    return _bn(y, w, key + ".1")


def _slice_res_block(x, w):
    """resnet_block_2d (DEN.py:295-304): relu(x + BN(conv1x3x3(relu(BN(conv1x3x3(x))))))."""
    y = F.relu(_conv_bn(x, w, "conv.0", 1, (0, 1, 1)))
    y = _conv_bn(y, w, "conv.2", 1, (0, 1, 1))
    return F.relu(x + y)


def _srd(x, w):
    """SRD (DEN.py:317-330): per-slice residual block, then the cross-slice attention pair
    conv3x1x1 -> relu -> conv1x1x1 -> relu (no BN, no bias) added back to the features."""
    feat = _slice_res_block(x, w.sub("Focus_Measure"))
    a = F.relu(_conv(feat, w, "N_ch_attention.0", 1, (1, 0, 0)))
    a = F.relu(_conv(a, w, "N_ch_attention.2", 1, 0))
    return feat + a


def _efd(x, w):
    """EFD (DEN.py:306-315): relu(BN(conv3^3 s(1,2,2)(x)) + BN(conv3^3(maxpool(1,2,2)(x))))."""
    a = _conv_bn(x, w, "stride_conv", (1, 2, 2), 1)
    b = _conv_bn(F.max_pool3d(x, (1, 2, 2), (1, 2, 2)), w, "max_pooling.1", 1, 1)
    return F.relu(a + b)


def _stem(FS, w):
    """FM_module (DEN.py:131-143): dilated 1x9x9 conv 3->8 + BN + ReLU, then SRD(8)."""
    x = F.relu(_conv_bn(FS, w, "Focus_extraction.0", 1, (0, 8, 8), (1, 2, 2)))
    return _srd(x, w.sub("Focus_extraction.2"))


def _two_conv(x, w, key, last_relu):
    y = F.relu(_conv_bn(x, w, key + ".0"))
    y = _conv_bn(y, w, key + ".2")
    return F.relu(y) if last_relu else y


def _pyramid(x, w):
    """hourglassup.forward (DEN.py:212-238): three average-pooled scales, per-scale residual
    stacks, strided fusion downwards and transposed-conv fusion back up to 1/8 resolution."""
    s8 = F.avg_pool3d(x, (1, 2, 2), (1, 2, 2))
    s16 = F.avg_pool3d(x, (1, 4, 4), (1, 4, 4))
    s32 = F.avg_pool3d(x, (1, 8, 8), (1, 8, 8))

    def scale(t, tag):
        r = _two_conv(t, w, f"dres{tag}_0", True)
        return _two_conv(r, w, f"dres{tag}_1", False) + r

    s8, s16, s32 = scale(s8, "8"), scale(s16, "16"), scale(s32, "32")
    d1 = _conv(s8, w, "conv1", (1, 2, 2), 1)                      # no BN (DEN.py:184)
    d1 = F.relu(_conv_bn(torch.cat((d1, s16), 1), w, "combine1.0"))
    c2 = F.relu(_conv_bn(d1, w, "conv2.0"))
    d2 = _conv(c2, w, "conv3", (1, 2, 2), 1)                      # no BN (DEN.py:189)
    d2 = F.relu(_conv_bn(torch.cat((d2, s32), 1), w, "combine2.0"))
    c4 = F.relu(_conv_bn(d2, w, "conv4.0"))
    u8 = F.relu(_up_bn(c4, w, "conv8") + _conv_bn(c2, w, "redir2", 1, 0))
    u9 = F.relu(_up_bn(u8, w, "conv9") + _conv_bn(s8, w, "redir1", 1, 0))
    return u9


def _hourglass(x, w, presqu, postsqu):
    """hourglass.forward (DEN.py:265-284).  Returns (out, pre_1) where pre_1 is conv0's output."""
    pre1 = F.relu(_conv_bn(x, w, "conv0.0"))
    out = F.relu(_conv_bn(pre1, w, "conv1.0", (1, 2, 2), 1))
    pre = _conv_bn(out, w, "conv2")
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
    out = F.relu(_conv_bn(pre, w, "conv3.0", (1, 2, 2), 1))
    out = F.relu(_conv_bn(out, w, "conv4.0"))
    skip = presqu if presqu is not None else pre
    out = F.relu(_up_bn(out, w, "conv5") + skip)
    out = _up_bn(out, w, "conv6")
    return out, pre1


def _regress(score, focus_dists, H, W):
    """The inline regression block, four times in DEN.py:86-90,110-126: bilinear resize of the
    per-slice scores (slices act as channels; F.upsample default = align_corners=False),
    p = softplus(s)+1e-6, p /= sum_N p, depth = sum_N focus_dists*p."""
    if score.shape[-2:] != (H, W):
        score = F.interpolate(score, size=[H, W], mode="bilinear", align_corners=False)
    p = F.softplus(score) + 1e-6
    p = p / p.sum(dim=1, keepdim=True)
    return torch.sum(focus_dists * p, dim=1)


def dff_forward(sd, FS, focus_dists, prefix="DFF_net.", taps=None):
    """Restatement of DFF_net.forward (DEN.py:74-127).

    ``sd``: mapping state-dict key -> float32 CPU tensor.  ``FS``: (B,3,N,H,W).  ``focus_dists``:
    anything broadcastable against (B,N,H,W).  Returns ``(mid_out, pred1, pred2, pred3)``, each
    (B,H,W).  If ``taps`` is a dict it receives the intermediate volumes named as in SURVEY.md
    section 8c (V1, V2, V3, FS_volume, conf, cost1, cost2, cost3).
    """
    w = _W(sd, prefix)
    H, W = FS.shape[-2:]
    v1 = _stem(FS, w.sub("FM_measure"))                                   # DEN.py:77
    v2 = _srd(_efd(v1, w.sub("FM_conv1.0")), w.sub("FM_conv1.1"))         # DEN.py:78
    v3 = _srd(_efd(v2, w.sub("FM_conv2.0")), w.sub("FM_conv2.1"))         # DEN.py:80
    vol = _pyramid(v3, w.sub("SPP_module"))                               # DEN.py:82
    conf = F.relu(_conv_bn(vol, w, "confidence.0"))
    conf = _conv(conf, w, "confidence.2", 1, 1).squeeze(1)                # DEN.py:83-84
    mid_out = _regress(conf, focus_dists, H, W)                           # DEN.py:86-90

    x = F.relu(_conv_bn(vol, w, "dres0.0"))
    x = F.relu(_conv_bn(x, w, "dres0.2"))
    x = _up_bn(x, w, "deconv_1")                                          # DEN.py:92-94
    out, pre = _hourglass(torch.cat((x, v3), 1), w.sub("dres2"), None, None)
    s1 = x + out
    cost1 = _conv(s1, w, "classif1.0").squeeze(1)                         # DEN.py:95-97

    x2 = _up_bn(s1, w, "deconv_2")
    out, pre = _hourglass(torch.cat((x2, v2), 1), w.sub("dres3"), pre, out)
    s2 = x2 + out
    cost2 = _conv(s2, w, "classif2.0").squeeze(1)                         # DEN.py:99-103

    x3 = _up_bn(s2, w, "deconv_3")
    out, _ = _hourglass(torch.cat((x3, v1), 1), w.sub("dres4"), pre, out)
    s3 = x3 + out
    cost3 = _conv(s3, w, "classif3.0").squeeze(1)                         # DEN.py:105-108

    pred1 = _regress(cost1, focus_dists, H, W)
    pred2 = _regress(cost2, focus_dists, H, W)
    pred3 = _regress(cost3, focus_dists, H, W)                            # DEN.py:110-126
    if taps is not None:
        taps.update(V1=v1, V2=v2, V3=v3, FS_volume=vol, conf=conf,
                    cost1=cost1, cost2=cost2, cost3=cost3)
    return mid_out, pred1, pred2, pred3


def fov_warp(x, alpha, FOVs):
    """Restatement of FlowNetwork.FOV_warp (End_to_End/End_to_End.py:106-134) with per-sample semantics
    (the reference only ever runs batch 1, TRS.py:23; its batch>1 broadcast of alpha is a bug):
    flow_x = (W//2)*(FOV+alpha0-1)*linspace(-1,1,W) + alpha1, flow_y likewise with H and alpha2; sample x at
    (x-flow_x, y-flow_y, n) with trilinear grid_sample, zeros padding, align_corners=True.
    x (B,C,N,H,W); alpha (B,3,N,1,1); FOVs (B,1,N,1,1).  Returns (warped, flow (B,2,N,H,W))."""
    B, C, N, H, W = x.shape
    outs, flows = [], []
    for b in range(B):
        a = alpha[b].reshape(3, N, 1, 1)
        f = a[0] + FOVs[b].reshape(N, 1, 1)
        lx = torch.linspace(-1, 1, steps=W).reshape(1, 1, W)
        ly = torch.linspace(-1, 1, steps=H).reshape(1, H, 1)
        fx = (W // 2) * (f - 1) * lx + a[1]
        fy = (H // 2) * (f - 1) * ly + a[2]
        fx, fy = fx.expand(N, H, W), fy.expand(N, H, W)
        xs = torch.arange(W, dtype=torch.float32).reshape(1, 1, W).expand(N, H, W)
        ys = torch.arange(H, dtype=torch.float32).reshape(1, H, 1).expand(N, H, W)
        zs = torch.arange(N, dtype=torch.float32).reshape(N, 1, 1).expand(N, H, W)
        gx = 2.0 * (xs - fx) / max(W - 1, 1) - 1.0
        gy = 2.0 * (ys - fy) / max(H - 1, 1) - 1.0
        gz = 2.0 * zs / max(N - 1, 1) - 1.0
        grid = torch.stack((gx, gy, gz), dim=-1).unsqueeze(0)
        outs.append(F.grid_sample(x[b:b + 1], grid, align_corners=True))
        flows.append(torch.stack((fx, fy), 0).unsqueeze(0))
    return torch.cat(outs, 0), torch.cat(flows, 0)


def _of_block(x, w, stride):
    """resnet_block_2d_OF (End_to_End.py:135-145): relu(feature(x) + BN(conv(relu(BN(conv_s(x)))))) where
    ``feature`` is a bias-free 1x1x1 conv with the same (1,s,s) stride."""
    y = F.relu(_conv_bn(x, w, "conv.0", (1, stride, stride), (0, 1, 1)))
    y = _conv_bn(y, w, "conv.2", 1, (0, 1, 1))
    return F.relu(_conv(x, w, "feature", (1, stride, stride), 0) + y)


def _alpha_head(vol, w):
    """conv1/2/3 of FlowNetwork (End_to_End.py:37-69): 3x(convbn 1x3x3 + relu), biased 1x3x3 conv to 3
    channels, AdaptiveAvgPool3d((10,1,1)) — for the 10-slice stacks the network is built for this is the
    plain mean over H and W of every slice."""
    y = vol
    for i in (0, 2, 4):
        y = F.relu(_conv_bn(y, w, str(i), 1, (0, 1, 1)))
    y = F.conv3d(y, w("6.weight"), w("6.bias"), 1, (0, 1, 1))
    return F.adaptive_avg_pool3d(y, (10, 1, 1))


def flow_forward(sd, FS, FOVs, prefix="optical_flow_aggregation.", taps=None):
    """Restatement of FlowNetwork.forward (End_to_End.py:71-105), per-sample semantics for batch>1.

    Three feature levels (full, 1/2, 1/4 resolution); coarse to fine, each level is warped with the
    warp parameters found so far, paired with its last slice (the reference slice) and the flow field,
    and an alpha head regresses a per-slice update (scale term damped by 0.001).  Returns the warped
    focal stack (B,3,N,H,W) and the accumulated alpha (B,3,N,1,1)."""
    w = _W(sd, prefix)
    B, _, N, H, W = FS.shape
    FOVs = FOVs.reshape(B, 1, N, 1, 1)
    fe1 = _of_block(_of_block(FS, w.sub("OF_feature.0"), 1), w.sub("OF_feature.1"), 1)
    fe2 = _of_block(_of_block(fe1, w.sub("OF_feature1.0"), 2), w.sub("OF_feature1.1"), 1)
    fe3 = _of_block(_of_block(fe2, w.sub("OF_feature2.0"), 2), w.sub("OF_feature2.1"), 1)
    alpha = torch.zeros(B, 3, N, 1, 1)
    for tag, fe, head in (("3", fe3, "conv1"), ("2", fe2, "conv2"), ("1", fe1, "conv3")):
        warped, flow = fov_warp(fe, alpha, FOVs)
        ref = warped[:, :, -1:].expand_as(warped)                    # End_to_End.py:82,90,98
        step = _alpha_head(torch.cat((ref, warped, flow), 1), w.sub(head)).clone()
        if taps is not None:
            taps["head" + tag] = step.clone()                        # before the 0.001 damping
        step[:, 0] = 0.001 * step[:, 0]                              # End_to_End.py:86,94,102
        alpha = alpha + step
        if taps is not None:
            taps["alpha" + tag] = alpha.clone()
    warped_FS, _ = fov_warp(FS, alpha, FOVs)
    return warped_FS, alpha


def e2e_forward(sd, FS, focus_dists, FOVs, taps=None):
    """Restatement of End_to_End.Network.forward (End_to_End.py:13-16): align the stack, then DFF_net on
    the aligned stack.  Returns (mid_out, pred1, pred2, pred3, aligned FS) like End_to_End.py:259."""
    warped, _ = flow_forward(sd, FS, FOVs, taps=taps)
    return dff_forward(sd, warped, focus_dists, "DFF_net.", taps) + (warped,)


def to_torch_state(sd_numpy):
    """numpy state dict (dffinthewild_amd.synth) -> CPU torch tensors."""
    return {k: torch.from_numpy(v) for k, v in sd_numpy.items()}


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2, the parity metric of BASELINE.json (b = reference)."""
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b).clamp_min(1e-30))


def rmse(a, b):
    """sqrt(mean((a-b)^2)): mask_rmse of the reference's metrics.py:102-103 with an all-true mask."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float(torch.sqrt(torch.mean((a - b) ** 2)))
