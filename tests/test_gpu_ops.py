"""GPU: each kernel family of libdffw.so, called through the C ABI, against the same PyTorch
operator the reference invokes (computed on the CPU in fp32).  Tolerances are relative L2:
split-bf16 5e-5 (expected ~1e-5), fp16 3e-3, bf16 2e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 5e-5, "fp16": 3e-3, "bf16": 2e-2}


def rel(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def eng(lib_built):
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from dffinthewild_amd import engine
    return engine


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def bn_params(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 + torch.rand(c, generator=g), torch.rand(c, generator=g) - 0.5,
            torch.rand(c, generator=g) - 0.5, 0.5 + torch.rand(c, generator=g))


def ref_bn(y, bn):
    return F.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)


# (name, Cin, Cout, kernel, stride, pad, dilation, N, H, W)  — every conv family of DFF_net
CONV_CASES = [
    ("stem_1x9x9_dil2", 3, 8, (1, 9, 9), (1, 1, 1), (0, 8, 8), (1, 2, 2), 3, 32, 40),
    ("slice_1x3x3", 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), 4, 16, 24),
    ("attn_3x1x1", 16, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), 5, 8, 16),
    ("point_1x1x1", 32, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 3, 8, 8),
    ("c3_16_8", 16, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 4, 16, 16),
    ("c3_32_16", 32, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 3, 8, 24),
    ("c3_64_32", 64, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 2, 8, 8),
    ("c3_192_128", 192, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 2, 8, 8),
    ("c3_s2_8_16", 8, 16, (3, 3, 3), (1, 2, 2), (1, 1, 1), (1, 1, 1), 3, 16, 32),
    ("c3_s2_64_128", 64, 128, (3, 3, 3), (1, 2, 2), (1, 1, 1), (1, 1, 1), 2, 16, 16),
    ("c3_one_slice", 8, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1, 8, 8),
    ("score_c3_32_1", 32, 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 3, 8, 8),
    ("score_1x1x1_8_1", 8, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 3, 16, 16),
    # output channel counts that do not fill the kernel's power-of-two tile count (48 -> 4 tiles, 96 -> 8): the padding tiles must not store
    # (the straight-line epilogue has no per-tile channel guard, so these launches must stay on the generic one)
    ("c3_32_48", 32, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 3, 8, 24),
    ("c3_32_96", 32, 96, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 2, 8, 16),
]


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: c[0])
def test_conv_families(eng, case, prec):
    name, cin, cout, k, s, p, d, N, H, W = case
    B = 2
    x = rnd(B, cin, N, H, W, seed=1)
    w = rnd(cout, cin, *k, seed=2, scale=(2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5 * 1.7)
    bn = bn_params(cout, 3) if cout > 1 else None
    ref = F.conv3d(x, w, None, s, p, d)
    if bn:
        ref = ref_bn(ref, bn)
        ref = F.relu(ref)
    got = eng.op_conv3d(x.cuda(), w, stride=s, pad=p, dilation=d, bn=bn, relu=1 if bn else 0, precision=prec)
    if cout == 1:
        ref = ref.squeeze(1)
    assert got.shape == ref.shape
    assert rel(got, ref) <= TOL[prec], (name, prec, rel(got, ref))


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("cin,cout,B,N,H,W,relu,wgs", [(16, 32, 2, 10, 128, 128, 1, 0), (16, 32, 4, 1, 128, 64, 0, 16), (16, 32, 2, 2, 128, 128, 1, 40),
                                                       (16, 16, 4, 10, 128, 128, 1, 0), (16, 16, 4, 3, 256, 64, 0, 24), (16, 16, 8, 1, 64, 128, 1, 8),
                                                       (32, 32, 2, 10, 128, 128, 1, 0), (32, 32, 4, 1, 128, 64, 0, 16), (32, 32, 2, 3, 128, 128, 0, 40),
                                                       (32, 64, 2, 10, 128, 128, 1, 0), (32, 64, 8, 2, 64, 128, 0, 24)])
def test_conv_roll_strided_16_channels(eng, cin, cout, B, N, H, W, relu, wgs, prec, monkeypatch):
    """conv_roll_s2 (dffw_conv_roll.hip): 3x3x3 stride (1,2,2) as a rolling window over whole pixel records.  16 input channels
    (`FM_conv2.0.stride_conv`, `dres3.conv1`: 16 -> 32; `dres4.conv3`: 16 -> 16; DEN.py:306-315, 252-256) and 32 with the
    contraction split between wave pairs (`dres3.conv3`: 32 -> 32; `dres2.conv1`, SPP `conv1`: 32 -> 64 as two launches over the output
    channel halves).  Slice counts 1, 2, 3, 10, non-square maps, one column per workgroup and long column streams, with and without
    the ReLU epilogue, three arithmetics; against F.conv3d and against conv_tile (DFFW_NO_ROLL_S2) on the same input."""
    x = rnd(B, cin, N, H, W, seed=51)
    w = rnd(cout, cin, 3, 3, 3, seed=52, scale=(2.0 / (cin * 27)) ** 0.5 * 1.7)
    bn = bn_params(cout, 53)
    ref = ref_bn(F.conv3d(x, w, None, (1, 2, 2), 1), bn)
    if relu:
        ref = F.relu(ref)
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    got = eng.op_conv3d(x.cuda(), w, stride=(1, 2, 2), pad=1, bn=bn, relu=relu, precision=prec)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_s2<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL[prec], rel(got, ref)
    monkeypatch.setenv("DFFW_NO_ROLL_S2", "1")
    alt = eng.op_conv3d(x.cuda(), w, stride=(1, 2, 2), pad=1, bn=bn, relu=relu, precision=prec)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]
    assert rel(got, alt) <= TOL[prec] * 0.2


SMALL_CASES = [
    # name, Cin, Cout, kernel, stride, pad, transposed, B, N, H, W, residual
    ("c3_64_64_7x7", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, 1, 5, 7, 7, True),
    ("c3_192_128_3x3", 192, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, 1, 5, 3, 3, False),
    ("c3_32_64_7x7", 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, 2, 3, 7, 5, False),
    ("p1_64_64_14x14", 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, 1, 5, 14, 14, False),
    ("p1_32_32_28x28", 32, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, 1, 5, 28, 28, False),
    ("t3_128_64_7to14", 128, 64, (3, 3, 3), (1, 2, 2), (1, 1, 1), True, 1, 5, 7, 7, True),
    ("s2_64_128_6to3", 64, 128, (3, 3, 3), (1, 2, 2), (1, 1, 1), False, 1, 5, 6, 6, False),
    ("score_32_1_4x4", 32, 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, 1, 2, 4, 4, False),
    ("c3_one_point", 16, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, 1, 1, 1, 1, False),
]


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("case", SMALL_CASES, ids=lambda c: c[0])
def test_conv_small_grids(eng, case, prec, monkeypatch):
    """conv_small (dffw_kernels.hip): grids too small for an LDS tile -- the 1/32-resolution pyramid scale of a 224 x 224 stack is
    5 x 7 x 7 (DEN.py:212-238 at BASELINE config 2) -- as one workgroup per (16 grid points, 16 output channels) whose waves split
    the contraction depth and sum through LDS.  3x3x3 at stride 1 and (1,2,2), the four sub-pixel phases of the transposed conv,
    1x1x1 (depth 1-2 chunks: fewer waves), a one-channel score output, a single grid point; residual + ReLU epilogue; against the
    PyTorch operator and against conv_igemm (DFFW_NO_SMALL) on the same input."""
    name, cin, cout, k, st, pd, transposed, B, N, H, W, residual = case
    x = rnd(B, cin, N, H, W, seed=41)
    wshape = (cin, cout, *k) if transposed else (cout, cin, *k)
    w = rnd(*wshape, seed=42, scale=(2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5 * 1.7)
    bn = bn_params(cout, 43) if cout > 1 else None
    if transposed:
        ref = F.conv_transpose3d(x, w, None, st, pd, (0, 1, 1))
    else:
        ref = F.conv3d(x, w, None, st, pd)
    if bn:
        ref = ref_bn(ref, bn)
    res = rnd(*ref.shape, seed=44) if residual else None
    if residual:
        ref = ref + res
    if bn:
        ref = F.relu(ref)
    kw = dict(stride=st, pad=pd, transposed=transposed, bn=bn, residual=res.cuda() if residual else None, relu=1 if bn else 0, precision=prec)
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_small<"), eng.last_conv_kernel()
    if cout == 1:
        ref = ref.squeeze(1)
    assert got.shape == ref.shape
    assert rel(got, ref) <= TOL[prec], (name, rel(got, ref))
    monkeypatch.setenv("DFFW_NO_SMALL", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_igemm<"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]
    assert rel(got, alt) <= TOL[prec] * 0.2


TEAM_CASES = [
    # name, Cin, Cout, stride, B, N, H, W, residual, teams
    ("dres8_32_32", 32, 32, 1, 1, 10, 32, 32, False, 2),        # two 16-channel stages, one per team
    ("dres8_32_32_res_n5", 32, 32, 1, 1, 5, 28, 28, True, 2),   # BASELINE config 2's grid: edge tiles in y and x
    ("dres16_64_64", 64, 64, 1, 1, 10, 16, 16, True, 4),        # four stages, four teams
    ("dres32_64_64_b2_n3", 64, 64, 1, 2, 3, 8, 8, False, 4),    # an 8 x 8 grid: half of every operand tile outside it
    ("dres0_32_64", 32, 64, 1, 1, 10, 32, 32, False, 2),        # 64 outputs: four output-channel workgroups per tile
    ("conv1_s2_32_64", 32, 64, 2, 1, 10, 32, 32, False, 4),     # stride (1,2,2): four 8-channel stages
    ("conv_s2_16_32_b2_n3", 16, 32, 2, 2, 3, 32, 64, False, 2),
    ("conv_s2_16_16_res", 16, 16, 2, 1, 5, 56, 56, True, 2),
]


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("case", TEAM_CASES, ids=lambda c: c[0])
def test_conv_tile_teams(eng, case, prec, monkeypatch):
    """conv_tile's team configurations (round 6; dffw_conv_tile.hip, template argument KT): the few-tile layers of a batch-1 forward -- the 1/8 ... 1/32-resolution
    pyramid and hourglass layers (DEN.py:212-238, 265-284) -- split their contraction depth over KT teams of four waves INSIDE the workgroup and add the teams'
    accumulators up in LDS, instead of a split-K launch that leaves fp32 partials in memory for a splitk_finish launch.  3x3x3 at stride 1 and (1,2,2), 32 ... 128
    channels, whole and edge tiles, slice counts 3 / 5 / 10, residual + ReLU; against F.conv3d, against the split-K launch pair (DFFW_NO_TEAMS: only the order of
    the fp32 sums differs) and bitwise against a second run."""
    name, cin, cout, stride, B, N, H, W, residual, kt = case
    x = rnd(B, cin, N, H, W, seed=61)
    w = rnd(cout, cin, 3, 3, 3, seed=62, scale=(2.0 / (cin * 27)) ** 0.5 * 1.7)
    bn = bn_params(cout, 63)
    ref = ref_bn(F.conv3d(x, w, None, (1, stride, stride), 1), bn)
    res = rnd(*ref.shape, seed=64) if residual else None
    ref = F.relu(ref + res) if residual else F.relu(ref)
    kw = dict(stride=(1, stride, stride), pad=1, bn=bn, residual=res.cuda() if residual else None, relu=1, precision=prec)
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "100000")   # (the rolling kernels take some of these shapes at this size: not under test here)
    monkeypatch.setenv("DFFW_TEAM_MAX_WGS", "100000")     # (the engine keeps split-K for launches of more than 320 team workgroups)
    got = eng.op_conv3d(x.cuda(), w, **kw)
    kn = eng.last_conv_kernel()
    assert kn.startswith("dffw::conv_tile<") and kn.endswith(", %d>" % kt), kn   # (last template argument: teams per workgroup)
    assert rel(got, ref) <= TOL[prec], (name, rel(got, ref))
    again = eng.op_conv3d(x.cuda(), w, **kw)
    assert torch.equal(got, again)
    monkeypatch.setenv("DFFW_NO_TEAMS", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    kn = eng.last_conv_kernel()
    assert kn.startswith("dffw::conv_tile<") and kn.endswith(", true, false, 1>"), kn   # the split-K instantiation
    assert rel(alt, ref) <= TOL[prec]
    assert rel(got, alt) <= TOL[prec] * 0.1, rel(got, alt)


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("cin,cout,B,N,H,W,residual,kt", [(128, 64, 1, 10, 8, 8, True, 4), (64, 32, 1, 10, 16, 16, True, 2), (128, 64, 1, 5, 14, 14, False, 4),
                                                          (64, 64, 2, 3, 12, 20, False, 2)])
def test_conv_tile_teams_transposed(eng, cin, cout, B, N, H, W, residual, kt, prec, monkeypatch):
    """The transposed 3x3x3 convs of the few-tile regime (SPP `conv8` 128 -> 64 on the 8 x 8 grid, `conv9` 64 -> 32 on 16 x 16 at batch 1, DEN.py:228-238): their four
    sub-pixel passes are workgroups of their own (grid.z) and each workgroup's 32-channel stages one team each (conv_tile's KT, round 6) instead of a walk over
    the stages.  Against F.conv_transpose3d, against the walk (DFFW_NO_TEAMS), and bitwise against a second run."""
    x = rnd(B, cin, N, H, W, seed=71)
    w = rnd(cin, cout, 3, 3, 3, seed=72, scale=(2.0 / (cin * 27 / 4)) ** 0.5 * 1.7)
    bn = bn_params(cout, 73)
    ref = ref_bn(F.conv_transpose3d(x, w, None, (1, 2, 2), 1, (0, 1, 1)), bn)
    res = rnd(*ref.shape, seed=74) if residual else None
    ref = F.relu(ref + res) if residual else F.relu(ref)
    kw = dict(stride=(1, 2, 2), pad=1, transposed=True, bn=bn, residual=res.cuda() if residual else None, relu=1, precision=prec)
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "100000")
    monkeypatch.setenv("DFFW_ROLLT_MIN_UNITS", "100000")
    got = eng.op_conv3d(x.cuda(), w, **kw)
    kn = eng.last_conv_kernel()
    assert kn.startswith("dffw::conv_tile<0, 2," if prec == "bf16x3" else "dffw::conv_tile<") and kn.endswith(", %d>" % kt), kn
    assert rel(got, ref) <= TOL[prec], rel(got, ref)
    assert torch.equal(got, eng.op_conv3d(x.cuda(), w, **kw))
    monkeypatch.setenv("DFFW_NO_TEAMS", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().endswith(", 1>"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]
    assert rel(got, alt) <= TOL[prec] * 0.1, rel(got, alt)


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("cout,N,H,W,zsplit,residual,wgs", [(8, 10, 128, 128, 1, True, 0), (16, 5, 128, 128, 1, False, 24), (8, 1, 64, 256, 1, False, 8),
                                                             (16, 10, 64, 256, 2, True, 40), (16, 7, 128, 128, 3, False, 0), (8, 2, 128, 128, 2, True, 16),
                                                             (8, 10, 128, 128, 1, False, 0), (8, 7, 64, 256, 3, False, 24), (8, 2, 128, 128, 2, False, 16),
                                                             (8, 3, 128, 128, 3, False, 8)])
def test_conv_roll_rolling_window(eng, cout, N, H, W, zsplit, residual, wgs, prec, monkeypatch):
    """conv_roll (dffw_conv_roll.hip): the 16-channel 3x3x3 stride-1 layers of the full-resolution hourglass
    (DEN.py:240-284, dres4.conv0 / conv2) as a rolling window along the slices; every slice count incl. 1 and 2,
    a slice range split over 1-3 workgroups, both output widths, with and without the residual + ReLU epilogue,
    one column per workgroup (wgs=0: the grid covers them) and long streams of columns per workgroup (wgs=8..40).
    The same case with DFFW_NO_ROLL=1 must take conv_tile and agree."""
    B, cin = 2, 16
    x = rnd(B, cin, N, H, W, seed=21)
    w = rnd(cout, cin, 3, 3, 3, seed=22, scale=(2.0 / (cin * 27)) ** 0.5 * 1.7)
    bn = bn_params(cout, 23)
    res = rnd(B, cout, N, H, W, seed=24) if residual else None
    ref = ref_bn(F.conv3d(x, w, None, 1, 1), bn)
    ref = F.relu(ref + res) if residual else F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_ZSPLIT", str(zsplit))
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    got = eng.op_conv3d(x.cuda(), w, pad=1, bn=bn, residual=res.cuda() if residual else None, relu=1, precision=prec)
    # the pair-form layers without a residual run the software-pipelined step in split-bf16 storage (conv_rollx, round 4)
    pipelined = cout == 8 and not residual and prec == "bf16x3"
    assert eng.last_conv_kernel().startswith("dffw::conv_rollx_pair<" if pipelined else "dffw::conv_roll<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL[prec], rel(got, ref)
    if pipelined:   # ... and the serial step must give the same values (three accumulators instead of two: fp32 re-association only)
        monkeypatch.setenv("DFFW_NO_ROLLX", "1")
        old = eng.op_conv3d(x.cuda(), w, pad=1, bn=bn, residual=None, relu=1, precision=prec)
        assert eng.last_conv_kernel().startswith("dffw::conv_roll<"), eng.last_conv_kernel()
        assert rel(got, old) <= 2e-6, rel(got, old)
    monkeypatch.setenv("DFFW_NO_ROLL", "1")
    alt = eng.op_conv3d(x.cuda(), w, pad=1, bn=bn, residual=res.cuda() if residual else None, relu=1, precision=prec)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]


@pytest.mark.parametrize("N,H,W,zsplit,wgs,relu", [(10, 128, 128, 1, 0, 1), (1, 64, 256, 1, 8, 1), (7, 64, 256, 3, 24, 0), (2, 128, 128, 2, 16, 1), (3, 128, 128, 3, 8, 1),
                                                    (5, 256, 64, 1, 0, 1)])
def test_conv_rollx_k2(eng, N, H, W, zsplit, wgs, relu, monkeypatch):
    """conv_rollx_k2 (dffw_conv_rollx.hip): 3x3x3 stride 1, 32 -> 16 channels (`dres3.conv0`, DEN.py:240-284) as a software-pipelined rolling
    window with the contraction split over the two 16-channel input halves (8 waves, partial tiles exchanged through LDS): every slice count
    incl. 1 and 2, split slice ranges, one column per workgroup and long streams, with and without ReLU; against F.conv3d and against
    conv_tile on the same input (DFFW_NO_ROLLX)."""
    B, cin, cout = 2, 32, 16
    x = rnd(B, cin, N, H, W, seed=41)
    w = rnd(cout, cin, 3, 3, 3, seed=42, scale=(2.0 / (cin * 27)) ** 0.5 * 1.7)
    bn = bn_params(cout, 43)
    ref = ref_bn(F.conv3d(x, w, None, 1, 1), bn)
    if relu:
        ref = F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_ZSPLIT", str(zsplit))
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    got = eng.op_conv3d(x.cuda(), w, pad=1, bn=bn, relu=relu, precision="bf16x3")
    assert eng.last_conv_kernel().startswith("dffw::conv_rollx_k2<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL["bf16x3"], rel(got, ref)
    monkeypatch.setenv("DFFW_NO_ROLLX", "1")
    alt = eng.op_conv3d(x.cuda(), w, pad=1, bn=bn, relu=relu, precision="bf16x3")
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(got, alt) <= 2e-5, rel(got, alt)


@pytest.mark.parametrize("B,N,H,W,wgs,relu,residual", [(3, 10, 64, 64, 0, 1, False), (2, 1, 32, 64, 8, 1, True), (3, 7, 64, 32, 24, 0, True), (2, 2, 40, 48, 16, 1, False),
                                                        (5, 3, 8, 16, 0, 0, False), (1, 5, 128, 96, 0, 1, True)])
def test_conv_slice32(eng, B, N, H, W, wgs, relu, residual, monkeypatch):
    """conv_slice32 (dffw_conv_slice.hip): per-slice 1x3x3, 32 -> 32 channels (`FM_conv2.1.Focus_Measure.conv.{0,2}`, DEN.py:295-304; the 32 -> 32
    convs of End_to_End's alignment heads, E2E.py:33-61) as a streaming kernel with the whole filter resident in every wave: every slice count
    incl. 1 and 2, one column per workgroup and long streams, a single 8 x 16 column per sample, image borders in every position, ReLU and
    residual on and off; against F.conv3d and against conv_tile on the same input (DFFW_NO_SLICE32)."""
    cin = cout = 32
    x = rnd(B, cin, N, H, W, seed=81)
    w = rnd(cout, cin, 1, 3, 3, seed=82, scale=(2.0 / (cin * 9)) ** 0.5 * 1.7)
    bn = bn_params(cout, 83)
    res = rnd(B, cout, N, H, W, seed=84) if residual else None
    ref = ref_bn(F.conv3d(x, w, None, 1, (0, 1, 1)), bn)
    if residual:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "1")
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    kw = dict(pad=(0, 1, 1), bn=bn, relu=relu, residual=res.cuda() if residual else None, precision="bf16x3")
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_slice32<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL["bf16x3"], rel(got, ref)
    monkeypatch.setenv("DFFW_NO_SLICE32", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(got, alt) <= 2e-5, rel(got, alt)


@pytest.mark.parametrize("B,N,H,W,wgs,relu", [(3, 10, 64, 64, 0, 1), (2, 1, 32, 64, 8, 1), (3, 7, 64, 32, 24, 0), (2, 2, 40, 48, 16, 1), (5, 3, 8, 16, 0, 0), (1, 5, 120, 160, 0, 1)])
def test_conv_slice64(eng, B, N, H, W, wgs, relu, monkeypatch):
    """conv_slice64 (dffw_conv_slice.hip): per-slice 1x3x3, 64 -> 64 channels (`optical_flow_aggregation.conv1.{2,4}.0`, the level-3 alignment head of
    End_to_End, E2E.py:33-46) with one 16-channel output tile's filter resident per wave, eight waves per 8 x 16 column: every slice count incl. 1 and 2,
    one column per workgroup and long streams, a single column per sample, image borders in every position, ReLU on and off; against F.conv3d and
    against conv_tile on the same input (DFFW_NO_SLICE32)."""
    cin = cout = 64
    x = rnd(B, cin, N, H, W, seed=91)
    w = rnd(cout, cin, 1, 3, 3, seed=92, scale=(2.0 / (cin * 9)) ** 0.5 * 1.7)
    bn = bn_params(cout, 93)
    ref = ref_bn(F.conv3d(x, w, None, 1, (0, 1, 1)), bn)
    if relu:
        ref = F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "1")
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    kw = dict(pad=(0, 1, 1), bn=bn, relu=relu, precision="bf16x3")
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_slice64<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL["bf16x3"], rel(got, ref)
    again = eng.op_conv3d(x.cuda(), w, **kw)
    assert torch.equal(got, again)
    monkeypatch.setenv("DFFW_NO_SLICE32", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(got, alt) <= 2e-5, rel(got, alt)


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (64, 64), (32, 64)])
@pytest.mark.parametrize("N,H,W,zsplit,wgs,relu,residual", [(10, 64, 64, 1, 0, 1, False), (1, 32, 64, 1, 8, 1, True), (7, 64, 32, 3, 24, 0, True), (2, 40, 24, 2, 16, 1, False),
                                                            (3, 64, 64, 3, 8, 0, False), (5, 8, 8, 1, 0, 1, True), (3, 60, 80, 1, 0, 1, True), (4, 28, 36, 2, 8, 1, False)])
def test_conv_rollk(eng, cin, cout, N, H, W, zsplit, wgs, relu, residual, monkeypatch):
    """conv_rollk (dffw_conv_rollk.hip): 3x3x3 stride 1, 32 / 64 -> 32 / 64 channels (`dres2.conv0`, `dres0.*`, `dres2.conv2`, `dres3.conv2/4`, the
    pyramid's `dres8_*`, `confidence.0`; DEN.py:33-40,155-192,243-258) as a rolling window over 8 x 8 columns with the contraction split over
    the workgroup's waves (filter resident in registers, partial sums exchanged through LDS per half-step; 64 outputs = two launches): every
    slice count incl. 1 and 2, split slice ranges, one column per workgroup and long streams, a single 8 x 8 column per sample, partial columns at
    the bottom / right edge (60 x 80: the 1/8-resolution volume of a 480 x 640 stack; 28 x 36), ReLU and residual on and off; against F.conv3d and
    against conv_tile on the same input (DFFW_NO_ROLLK)."""
    B = 3
    x = rnd(B, cin, N, H, W, seed=61)
    w = rnd(cout, cin, 3, 3, 3, seed=62, scale=(2.0 / (cin * 27)) ** 0.5 * 1.7)
    bn = bn_params(cout, 63)
    res = rnd(B, cout, N, H, W, seed=64) if residual else None
    ref = ref_bn(F.conv3d(x, w, None, 1, 1), bn)
    if residual:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_ZSPLIT", str(zsplit))
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "1")
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    kw = dict(pad=1, bn=bn, relu=relu, residual=res.cuda() if residual else None, precision="bf16x3")
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_rollk<%d," % (cin // 8)), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL["bf16x3"], rel(got, ref)
    again = eng.op_conv3d(x.cuda(), w, **kw)
    assert torch.equal(got, again)                      # the partial sums are added in a fixed order
    if cout == 64:                                      # the two 32-channel output halves as two launches instead of grid.y = 2 of one: same arithmetic
        monkeypatch.setenv("DFFW_ROLLK_MERGE_BELOW", "1")
        two = eng.op_conv3d(x.cuda(), w, **kw)
        assert eng.last_conv_kernel().startswith("dffw::conv_rollk<"), eng.last_conv_kernel()
        assert torch.equal(got, two)
        monkeypatch.delenv("DFFW_ROLLK_MERGE_BELOW")
    monkeypatch.setenv("DFFW_NO_ROLLK", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert not eng.last_conv_kernel().startswith("dffw::conv_rollk<"), eng.last_conv_kernel()
    assert rel(got, alt) <= 2e-5, rel(got, alt)


ROLLT_SHAPES = [(10, 32, 32, 0, 1, False), (1, 16, 24, 8, 1, True), (7, 24, 16, 24, 0, True), (2, 30, 40, 16, 1, False), (3, 8, 8, 0, 0, False),
                (5, 12, 20, 8, 1, True), (2, 60, 80, 0, 1, True), (4, 14, 18, 8, 0, True)]


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (64, 64), (32, 64), (32, 16), (16, 16)])
@pytest.mark.parametrize("N,H,W,wgs,relu,residual", ROLLT_SHAPES)
def test_conv_rollt(eng, cin, cout, N, H, W, wgs, relu, residual, monkeypatch):
    """conv_rollt (dffw_conv_rollt.hip): ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1), 32 / 64 -> 32 / 64 channels (`deconv_1`, `dres2.conv5`, `dres3.conv5`, the
    pyramid's `conv9`; DEN.py:41-42, 194-200, 260-264) as a rolling window over 8 x 8 columns of the input grid with the resident filter split over the
    waves by output phase (64 inputs: phase (1,1) also over K between a wave pair that exchanges one partial per operand tile; 32 -> 16, `deconv_2` / `dres3.conv6`:
    the wide form, eight waves = two roles x four pixel sub-blocks of an 8 x 16 column, one pass per step, ring of five; 16 -> 16, `dres4.conv5`: the same with the upper K
    octets on zero weights): every slice count
    incl. 1 and 2, one column per workgroup and long streams, a single column per sample, partial columns at the bottom / right edge (30 x 40 and 60 x 80:
    the 1/16- and 1/8-resolution volumes of a 480 x 640 stack; 12 x 20, 14 x 18), ReLU and residual on and off; against F.conv_transpose3d, bitwise
    repeatable, and against conv_tile on the same input (DFFW_NO_ROLLT)."""
    B = 3
    x = rnd(B, cin, N, H, W, seed=81)
    w = rnd(cin, cout, 3, 3, 3, seed=82, scale=(2.0 / (cin * 27)) ** 0.5 * 3)
    bn = bn_params(cout, 83)
    res = rnd(B, cout, N, 2 * H, 2 * W, seed=84) if residual else None
    ref = ref_bn(F.conv_transpose3d(x, w, None, (1, 2, 2), 1, (0, 1, 1)), bn)
    if residual:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "1")
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    kw = dict(transposed=True, stride=(1, 2, 2), pad=1, bn=bn, relu=relu, residual=res.cuda() if residual else None, precision="bf16x3")
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel() == "dffw::conv_rollt<%d, %d, %s>" % (cin, 1 if residual else 0, "true" if cout == 16 else "false"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL["bf16x3"], rel(got, ref)
    again = eng.op_conv3d(x.cuda(), w, **kw)
    assert torch.equal(got, again)                      # the K-split pair adds its two partials in a fixed order
    monkeypatch.setenv("DFFW_NO_ROLLT", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith(("dffw::conv_tile<", "dffw::conv_roll_t32<")), eng.last_conv_kernel()
    assert rel(got, alt) <= 2e-5, rel(got, alt)


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (32, 16), (16, 16)])
@pytest.mark.parametrize("N,H,W,wgs,relu", [(10, 32, 32, 0, 0), (1, 16, 24, 8, 0), (2, 30, 40, 16, 1), (3, 8, 8, 0, 0), (5, 12, 20, 8, 0)])
@pytest.mark.parametrize("pre,cls", [(True, True), (True, False), (False, True)])
def test_conv_rollt_second_output_and_classifier(eng, cin, cout, N, H, W, wgs, relu, pre, cls, monkeypatch):
    """conv_rollt<.., 2>: the hourglass's last layer (`dres2.conv6`, DEN.py:96-97, 264: `out = conv6(...)`, `out_in = x + out`, `cost = classif(out_in)`): the
    value before the skip add as a second output and the 1x1x1 32 -> 1 classifier folded into the epilogue -- its dot spans the two 16-channel output tiles
    of a pixel, i.e. two waves, each adding its partial to the zeroed score volume (two addends: order-independent, so bitwise repeatable; 16 outputs,
    `dres3.conv6`: one)."""
    B = 2
    x = rnd(B, cin, N, H, W, seed=91)
    w = rnd(cin, cout, 3, 3, 3, seed=92, scale=(2.0 / (cin * 27)) ** 0.5 * 3)
    bn = bn_params(cout, 93)
    res = rnd(B, cout, N, 2 * H, 2 * W, seed=94)
    cw = rnd(1, cout, 1, 1, 1, seed=95)
    ref_pre = ref_bn(F.conv_transpose3d(x, w, None, (1, 2, 2), 1, (0, 1, 1)), bn)
    ref = ref_pre + res
    if relu:
        ref = F.relu(ref)
    ref_cls = F.conv3d(ref, cw).squeeze(1)
    monkeypatch.setenv("DFFW_ROLL_MIN_UNITS", "1")
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    kw = dict(transposed=True, stride=(1, 2, 2), pad=1, bn=bn, relu=relu, residual=res.cuda(), precision="bf16x3", want_pre=pre, cls_weight=cw if cls else None)
    y, yp, sc = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel() == "dffw::conv_rollt<%d, 2, %s>" % (cin, "true" if cout == 16 else "false"), eng.last_conv_kernel()
    assert rel(y, ref) <= TOL["bf16x3"], rel(y, ref)
    if pre:
        assert rel(yp, ref_pre) <= TOL["bf16x3"], rel(yp, ref_pre)
    if cls:
        assert rel(sc, ref_cls) <= TOL["bf16x3"], rel(sc, ref_cls)
    y2, yp2, sc2 = eng.op_conv3d(x.cuda(), w, **kw)
    assert torch.equal(y, y2) and (not pre or torch.equal(yp, yp2)) and (not cls or torch.equal(sc, sc2))
    monkeypatch.setenv("DFFW_NO_ROLLT", "1")
    ya, ypa, sca = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith(("dffw::conv_tile<", "dffw::conv_roll_t32<")), eng.last_conv_kernel()
    assert rel(y, ya) <= 2e-5 and (not pre or rel(yp, ypa) <= 2e-5) and (not cls or rel(sc, sca) <= 2e-5)


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("N,H,W,zsplit,residual,wgs", [(10, 64, 256, 1, True, 0), (5, 128, 128, 1, False, 16), (1, 64, 256, 1, True, 8),
                                                        (7, 64, 256, 3, False, 24), (2, 128, 128, 2, True, 0)])
def test_conv_roll_transposed(eng, N, H, W, zsplit, residual, wgs, prec, monkeypatch):
    """conv_roll_t: ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1), 16 -> 8 channels (`deconv_3`, `dres4.conv6`, DEN.py:41-48) as a
    rolling window over the input slices with the two x phases of an input column in the two halves of the result tile;
    + BN + residual + ReLU, every slice count incl. 1 and 2, split slice ranges, short and long column streams.  The same
    case with DFFW_NO_ROLL=1 takes conv_tile's 4-pass form and must agree with the reference operator as well."""
    B, cin, cout = 2, 16, 8
    x = rnd(B, cin, N, H, W, seed=31)
    w = rnd(cin, cout, 3, 3, 3, seed=32, scale=(2.0 / (cin * 27)) ** 0.5 * 3)
    bn = bn_params(cout, 33)
    res = rnd(B, cout, N, 2 * H, 2 * W, seed=34) if residual else None
    ref = ref_bn(F.conv_transpose3d(x, w, None, (1, 2, 2), 1, (0, 1, 1)), bn)
    ref = F.relu(ref + res) if residual else F.relu(ref)
    monkeypatch.setenv("DFFW_ROLL_ZSPLIT", str(zsplit))
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    kw = dict(transposed=True, stride=(1, 2, 2), pad=1, bn=bn, residual=res.cuda() if residual else None, relu=1, precision=prec)
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_t<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL[prec], rel(got, ref)
    monkeypatch.setenv("DFFW_NO_ROLL", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]


def test_no_lean_roll_switch_reaches_every_rolling_kernel(eng, monkeypatch):
    """DFFW_NO_LEAN_ROLL must select the generic-epilogue instantiation of conv_roll_t32 and conv_roll_s2 as well (the switch travels in
    ConvArgs::dbg and those launch paths masked it off, so the lean-vs-generic parity cases compared the lean kernel with itself)."""
    B, N = 2, 3
    x = rnd(B, 32, N, 64, 256, seed=71)
    w = rnd(32, 16, 3, 3, 3, seed=72, scale=0.1)
    bn = bn_params(16, 73)
    kw = dict(transposed=True, stride=(1, 2, 2), pad=1, bn=bn, relu=1, precision="bf16x3")
    monkeypatch.setenv("DFFW_NO_ROLLT", "1")             # (round 6: 32 -> 16 channels run on conv_rollt's wide form; conv_roll_t32 keeps the 16-input layers and is the fallback)
    a = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_t32<") and eng.last_conv_kernel().endswith("true>"), eng.last_conv_kernel()
    x2 = rnd(B, 16, N, 128, 256, seed=74)
    w2 = rnd(16, 16, 3, 3, 3, seed=75, scale=0.1)
    kw2 = dict(stride=(1, 2, 2), pad=1, bn=bn, relu=1, precision="bf16x3")
    a2 = eng.op_conv3d(x2.cuda(), w2, **kw2)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_s2<") and eng.last_conv_kernel().endswith("true>"), eng.last_conv_kernel()
    monkeypatch.setenv("DFFW_NO_LEAN_ROLL", "1")
    b = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_t32<") and eng.last_conv_kernel().endswith("false>"), eng.last_conv_kernel()
    b2 = eng.op_conv3d(x2.cuda(), w2, **kw2)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_s2<") and eng.last_conv_kernel().endswith("false>"), eng.last_conv_kernel()
    assert torch.equal(a, b) and torch.equal(a2, b2)     # same arithmetic, same order


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("cin", [32, 16])
@pytest.mark.parametrize("N,H,W,residual,wgs", [(10, 64, 256, True, 0), (1, 128, 128, False, 8), (2, 64, 256, True, 16), (5, 128, 128, False, 24)])
def test_conv_roll_transposed_32(eng, N, H, W, residual, wgs, cin, prec, monkeypatch):
    """conv_roll_t32: ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1), 32 -> 16 channels (`deconv_2`, `dres3.conv6`, DEN.py:41-48) as two
    rolling sweeps, one per output row phase (each with only that phase's taps resident); + BN + residual + ReLU.  cin = 16
    (`dres4.conv5`) runs on the same kernel with its upper channel octets reading zeros."""
    B, cout = 2, 16
    x = rnd(B, cin, N, H, W, seed=51)
    w = rnd(cin, cout, 3, 3, 3, seed=52, scale=(2.0 / (cin * 27)) ** 0.5 * 3)
    bn = bn_params(cout, 53)
    res = rnd(B, cout, N, 2 * H, 2 * W, seed=54) if residual else None
    ref = ref_bn(F.conv_transpose3d(x, w, None, (1, 2, 2), 1, (0, 1, 1)), bn)
    ref = F.relu(ref + res) if residual else F.relu(ref)
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    monkeypatch.setenv("DFFW_NO_ROLLT", "1")             # (round 6: in split-bf16 the 32 -> 16 layers run on conv_rollt's wide form; this is its fallback and the 16-input kernel)
    kw = dict(transposed=True, stride=(1, 2, 2), pad=1, bn=bn, residual=res.cuda() if residual else None, relu=1, precision=prec)
    got = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_t32<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL[prec], rel(got, ref)
    monkeypatch.setenv("DFFW_NO_ROLL", "1")
    alt = eng.op_conv3d(x.cuda(), w, **kw)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]


@pytest.mark.parametrize("prec", ["bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("N,H,W,wgs", [(10, 128, 256, 0), (1, 128, 256, 8), (2, 256, 128, 16), (5, 64, 512, 24)])
def test_conv_roll_strided(eng, N, H, W, wgs, prec, monkeypatch):
    """conv_roll_efd<..., false>: 3x3x3 stride (1,2,2) 8 -> 16 channels (dres4.conv1, DEN.py:252) as a rolling window with
    the stride-2 footprint stored even columns first; vs F.conv3d and vs conv_tile on the same input."""
    B, cin, cout = 2, 8, 16
    x = rnd(B, cin, N, H, W, seed=41)
    w = rnd(cout, cin, 3, 3, 3, seed=42, scale=(2.0 / (cin * 27)) ** 0.5 * 1.7)
    bn = bn_params(cout, 43)
    ref = F.relu(ref_bn(F.conv3d(x, w, None, (1, 2, 2), 1), bn))
    if wgs:
        monkeypatch.setenv("DFFW_ROLL_WGS", str(wgs))
    got = eng.op_conv3d(x.cuda(), w, stride=(1, 2, 2), pad=1, bn=bn, relu=1, precision=prec)
    assert eng.last_conv_kernel().startswith("dffw::conv_roll_efd<"), eng.last_conv_kernel()
    assert rel(got, ref) <= TOL[prec], rel(got, ref)
    monkeypatch.setenv("DFFW_NO_ROLL", "1")
    alt = eng.op_conv3d(x.cuda(), w, stride=(1, 2, 2), pad=1, bn=bn, relu=1, precision=prec)
    assert eng.last_conv_kernel().startswith("dffw::conv_tile<"), eng.last_conv_kernel()
    assert rel(alt, ref) <= TOL[prec]


@pytest.mark.parametrize("prec", ["bf16x3", "fp16"])
@pytest.mark.parametrize("cin,cout,N,H,W", [(16, 8, 3, 8, 8), (64, 32, 2, 8, 16), (128, 64, 2, 4, 4), (32, 32, 1, 8, 8)])
def test_transposed_conv_phases(eng, cin, cout, N, H, W, prec):
    """ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1) as 4 sub-pixel phases (DEN.py:41-42), + BN + residual + ReLU."""
    B = 2
    x = rnd(B, cin, N, H, W, seed=4)
    w = rnd(cin, cout, 3, 3, 3, seed=5, scale=(2.0 / (cin * 27)) ** 0.5 * 3)
    bn = bn_params(cout, 6)
    res = rnd(B, cout, N, 2 * H, 2 * W, seed=7)
    ref = F.relu(ref_bn(F.conv_transpose3d(x, w, None, (1, 2, 2), 1, (0, 1, 1)), bn) + res)
    got = eng.op_conv3d(x.cuda(), w, transposed=True, stride=(1, 2, 2), pad=1, bn=bn, residual=res.cuda(), relu=1, precision=prec)
    assert rel(got, ref) <= TOL[prec], rel(got, ref)


def test_conv_epilogue_variants(eng):
    """relu=2 is relu(acc)+res (the SRD attention add, DEN.py:329); relu=0 with residual is the
    un-activated skip sum (DEN.py:96)."""
    B, cin, cout, N, H, W = 1, 16, 16, 3, 8, 8
    x, w, res = rnd(B, cin, N, H, W, seed=8), rnd(cout, cin, 1, 1, 1, seed=9), rnd(B, cout, N, H, W, seed=10)
    y = F.conv3d(x, w)
    assert rel(eng.op_conv3d(x.cuda(), w, residual=res.cuda(), relu=2), F.relu(y) + res) <= 5e-5
    assert rel(eng.op_conv3d(x.cuda(), w, residual=res.cuda(), relu=0), y + res) <= 5e-5
    assert rel(eng.op_conv3d(x.cuda(), w, residual=res.cuda(), relu=1), F.relu(y + res)) <= 5e-5


@pytest.mark.parametrize("prec", ["bf16x3", "fp16"])
def test_pools(eng, prec):
    x = rnd(2, 32, 3, 16, 32, seed=11)
    assert rel(eng.op_pool(x.cuda(), 2, "max", prec), F.max_pool3d(x, (1, 2, 2), (1, 2, 2))) <= TOL[prec]
    for k in (2, 4, 8):
        assert rel(eng.op_pool(x.cuda(), k, "avg", prec), F.avg_pool3d(x, (1, k, k), (1, k, k))) <= TOL[prec]


def test_softplus_pointwise_sweep(eng):
    """softplus_fast (dffw_kernels.hip: log2 / rcp on the hardware transcendentals, no series, no IEEE divide) point by point instead of through a
    whole-map norm: a two-slice head with focus distances (0, 1) returns p1 / (p0 + p1), p = softplus(v) + 1e-6 (DEN.py:88-90), i.e. the RATIO of two
    softplus values -- dense sweep of v over [-30, 25] against (a) the neighbour v + 0.5 (both values small in the tail: relative accuracy there) and
    (b) the anchor v = 0; covers e^v below fp32 resolution (w == 1 select), the 1e-6 floor and the linear branch above the threshold 20.
    Reference in float64 with torch's own threshold rule."""
    v = torch.arange(-30.0, 25.0, 1.0 / 64).float()
    n = v.numel()
    w = 64
    hgt = (n + w - 1) // w
    pad = hgt * w - n
    vv = torch.cat([v, v[-1:].repeat(pad)]).reshape(1, 1, hgt, w)
    fd = torch.tensor([0.0, 1.0]).reshape(1, 2, 1, 1)

    def sp64(x):
        x = x.double()
        return torch.where(x > 20.0, x, torch.log1p(torch.exp(x))) + 1e-6

    for other in (vv + 0.5, torch.zeros_like(vv)):
        score = torch.cat([vv, other], 1).contiguous()
        got = eng.op_regress(score.cuda(), fd.cuda(), hgt, w).cpu().double().reshape(-1)
        p0, p1 = sp64(score[0, 0].float()).reshape(-1), sp64(score[0, 1].float()).reshape(-1)
        ref = p1 / (p0 + p1)
        err = ((got - ref).abs() / ref).max()
        assert float(err) <= 1.5e-6, float(err)


@pytest.mark.parametrize("N,h,w,scale", [(10, 8, 8, 8), (5, 8, 12, 4), (15, 16, 16, 2), (4, 32, 32, 1), (1, 8, 8, 8)])
@pytest.mark.parametrize("layout", ["dense", "bcast"])
def test_regression_head(eng, N, h, w, scale, layout):
    """bilinear(align_corners=False) + softplus + normalise + expectation (DEN.py:86-90) in fp32."""
    B, H, W = 2, h * scale, w * scale
    score = rnd(B, N, h, w, seed=12, scale=30.0)
    score[0, :, 0, 0] = -40.0            # all-negative pixel: the 1e-6 floor decides
    score[1, 0, 1, 1] = 45.0             # softplus linear branch (threshold 20)
    fd = torch.linspace(0.1, 1.5, N).reshape(1, N, 1, 1).repeat(B, 1, 1, 1)
    fd = fd * torch.tensor([1.0, 2.0]).reshape(B, 1, 1, 1)
    fd_in = fd.expand(B, N, H, W).contiguous() if layout == "dense" else fd
    up = F.interpolate(score, size=[H, W], mode="bilinear", align_corners=False) if scale > 1 else score
    p = F.softplus(up) + 1e-6
    ref = torch.sum(fd_in * (p / p.sum(1, keepdim=True)), 1)
    got = eng.op_regress(score.cuda(), fd_in.cuda(), H, W)
    assert rel(got, ref) <= 2e-6, rel(got, ref)
    assert float((got.cpu() - ref).abs().max()) <= 5e-6
