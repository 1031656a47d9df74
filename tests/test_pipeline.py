"""Input assembly and output post-processing around the forward.
CPU part: the oracle (oracle/pipeline_ref.py) against outputs of matplotlib's own colour map (tests/golden/io_jet.npz)
and the library's colour table against the same golden.  GPU part (-m gpu): the HIP kernels through the C ABI against
the oracle, bit-exact for the byte / index work, 1e-6 for the float64 metric means."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_ref as ref

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "io_jet.npz"))
# outputs of the reference's own loader classes (oracle/make_goldens_pack.py: test_Dataloader.py / Test_dataloader.py compiled in
# place, decode calls replaced by seeded arrays)
PACK_CASES = ["fs6", "hci", "ddff", "smartphone", "middlebury", "real_scenes"]


def pack_case(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"io_pack_{name}.npz"))
    crop = tuple(int(v) for v in g["crop"])
    return g, str(g["layout"]), (None if crop[0] < 0 else crop), bool(int(g["norm64"]))


@pytest.mark.parametrize("name", PACK_CASES)
def test_oracle_pack_stack_matches_reference_loaders(name):
    """oracle pack_stack == the tensor the reference's loader returns, bit for bit, for all six loaders: FS6 and Middlebury
    (float64 normalisation of a uint8 / float64 array, one rounding), HCI / DDFF / Smartphone / Real_Scenes (float32 arrays);
    crops of the Smartphone (84 / 63 per side) and Real_Scenes (1/12 per side) loaders; -1 padding to multiples of 32."""
    g, layout, crop, norm64 = pack_case(name)
    want = g["FS"]
    got = ref.pack_stack(g["raw"], layout, crop, norm64=norm64)
    assert got.shape == want.shape and got.dtype == want.dtype == np.float32
    assert np.array_equal(got, want)
    if name in ("fs6", "middlebury"):                                   # the float32 arithmetic would NOT reproduce these loaders
        assert not np.array_equal(ref.pack_stack(g["raw"], layout, crop), want)


def test_real_scene_inputs_match_reference_loader():
    """focus_dists = 1/d and the relative fields of view exactly as End_to_End/Test_dataloader.py:27-54 returns them (golden from the
    reference's Real_Scenes class), through the oracle and the host-side helper; the crop helper gives the loader's border."""
    from dffinthewild_amd import pipeline
    g, _, crop, _ = pack_case("real_scenes")
    d, f = [float(v) for v in g["focus_distance_m"]], float(g["focal_length"])
    ofd, ofov = ref.real_scene_inputs(d, f)
    assert np.array_equal(ofd, g["focus_dists"]) and np.array_equal(ofov, g["rel_fov"])
    fd, fov = pipeline.real_scene_inputs(d, f, device="cpu")
    assert np.array_equal(fd[0].numpy(), g["focus_dists"]) and np.array_equal(fov[0].numpy(), g["rel_fov"])
    H, W = g["raw"].shape[:2]
    assert pipeline.real_scene_crop(H, W) == crop
    assert tuple(g["before_pad"]) == (crop[2], crop[3], 3, 10)


def test_oracle_colorize_matches_matplotlib_golden():
    got = ref.colorize(GOLD["depth"][0], size=tuple(GOLD["crop"]))
    assert np.array_equal(got, GOLD["rgb_minmax"])                      # test_real_scenes.py:40-52 sequence
    got = ref.colorize(GOLD["depth_fixed"], vrange=tuple(GOLD["fixed_range"]))
    assert np.array_equal(got, GOLD["rgb_fixed"])                       # fixed range, under / over / NaN pixels


def test_library_colour_table_is_matplotlibs(lib_built):
    lib = ctypes.CDLL(lib_built)
    buf = (ctypes.c_uint8 * 768)()
    assert lib.dffw_jet_lut(buf) == 0
    assert np.array_equal(np.frombuffer(buf, np.uint8).reshape(256, 3), GOLD["lut_u8"])


def test_oracle_pack_stack_layouts_agree():
    rng = np.random.RandomState(3)
    nhwc = rng.randint(0, 256, size=(5, 37, 50, 3)).astype(np.uint8)
    a = ref.pack_stack(nhwc, "NHWC")
    assert a.shape == (3, 5, 64, 64) and a.dtype == np.float32
    assert np.array_equal(a, ref.pack_stack(np.transpose(nhwc, (1, 2, 3, 0)), "HWCN"))
    assert np.array_equal(a, ref.pack_stack(np.transpose(nhwc, (1, 2, 0, 3)), "HWNC"))
    assert np.all(a[:, :, 37:, :] == -1) and np.all(a[:, :, :, 50:] == -1)
    assert a[1, 2, 3, 4] == np.float32(nhwc[2, 3, 4, 1]) / np.float32(127.5) - np.float32(1.0)


def test_oracle_fs6_float64_normalisation_all_byte_values():
    """The FS6 / DefocusNet loader (test_Dataloader.py:31-39) normalises in float64 (its accumulator np.zeros((256,256,3,0))
    is float64) and rounds to float32 once in torch.Tensor(); the other loaders divide float32 arrays.  All 256 byte values:
    the oracle's norm64 mode equals the loader's own NumPy lines, and differs from the float32 form for some values."""
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)          # one "image" holding every byte value
    mats_input = np.zeros((16, 16, 3, 0))                                              # test_Dataloader.py:31 (float64)
    mats_input = np.concatenate((mats_input, np.expand_dims(img, axis=-1)), axis=3)    # :34
    mats_input = mats_input / 127.5 - 1.0                                              # :36
    mats_input = np.transpose(mats_input, (2, 3, 0, 1))                                # :39
    want = torch.Tensor(mats_input).numpy()                                            # :42 (the single rounding)
    assert mats_input.dtype == np.float64 and want.dtype == np.float32
    got = ref.pack_stack(img[..., None], "HWCN", norm64=True)
    assert np.array_equal(got[:, :, :16, :16], want)
    f32 = ref.pack_stack(img[..., None], "HWCN")
    ndiff = int((got[0, 0, :16, :16] != f32[0, 0, :16, :16]).sum())
    assert 0 < ndiff < 256, ndiff                                                      # the two arithmetics are NOT interchangeable
    assert np.max(np.abs(got - f32)) <= np.spacing(np.float32(1.0))                    # ... but never more than one ulp apart


def test_oracle_metrics_known_answers():
    gt = np.full((4, 4), 2.0, np.float32)
    est = gt.copy()
    est[0, 0] = 4.0          # one pixel off by 2x: fails every accuracy threshold (1.25, 1.5625, 1.953)
    mask = np.ones((4, 4), bool)
    mask[3, 3] = False
    m = ref.masked_metrics(est, gt, mask, conf=np.ones((4, 4), np.float32))
    assert m[0] == 15 and np.isclose(m[1], 1.0 / 15) and np.isclose(m[3], 4.0 / 15) and np.isclose(m[4], 2.0 / 15)
    assert np.isclose(m[7], 14.0 / 15) and np.isclose(m[8], 14.0 / 15) and np.isclose(m[9], 14.0 / 15) and np.isclose(m[10], m[3])


METRIC_CASES = ["ddff_like", "fs6_like", "ragged", "one_pixel"]


def metrics_golden(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "io_metrics.npz"))
    return [z[f"{name}/{k}"] for k in ("est", "gt", "mask", "conf", "want")]


@pytest.mark.parametrize("name", METRIC_CASES)
def test_oracle_metrics_match_the_reference_functions(name):
    """tests/golden/io_metrics.npz = outputs of the reference's own mask_* functions (metrics.py:90-127), produced by
    oracle/make_goldens_metrics.py; the restatement does the same float32 NumPy arithmetic, so it agrees to rounding of the
    pairwise float32 sums."""
    est, gt, mask, conf, want = metrics_golden(name)
    got = ref.masked_metrics(est, gt, mask, conf)
    assert got[0] == want[0] == mask.sum()
    assert np.allclose(got, want, rtol=1e-6, atol=0), (got, want)


# the one real scene the reference ships (End_to_End/Datasets/balls/focus_distance.txt, focal_length.txt; 10 JPEGs of 1280x720)
BALLS_FOCUS = [7223525.792447, 208.130591, 58.335774, 39.112869, 27.335806, 22.975371, 17.020994, 15.317949, 12.74398, 10.0]
BALLS_FOCAL = 0.2


def test_real_scene_inputs_follow_the_loader():
    """pipeline.real_scene_inputs / real_scene_crop against the loader's own NumPy lines (End_to_End/Test_dataloader.py:20-23,
    37-53,75) executed here on the `balls` scene's values, and against the oracle."""
    from dffinthewild_amd import pipeline
    focus_dists = np.asarray(BALLS_FOCUS)
    relative_Fov = (1 / BALLS_FOCAL - 1 / focus_dists)
    relative_Fov = relative_Fov / np.min(relative_Fov)
    relative_Fov = np.expand_dims(np.expand_dims(np.expand_dims(relative_Fov, axis=0), axis=2), axis=2)
    fd_want = torch.Tensor(1 / np.expand_dims(np.expand_dims(focus_dists, axis=1), axis=2))          # (N,1,1)
    fov_want = torch.Tensor(relative_Fov)                                                             # (1,N,1,1)
    fd, fov = pipeline.real_scene_inputs(BALLS_FOCUS, BALLS_FOCAL, device="cpu")
    assert fd.shape == (1, 10, 1, 1) and fov.shape == (1, 1, 10, 1, 1) and fd.dtype == fov.dtype == torch.float32
    assert torch.equal(fd[0], fd_want) and torch.equal(fov[0], fov_want)                              # DataLoader(batch_size=1) adds dim 0
    ofd, ofov = ref.real_scene_inputs(BALLS_FOCUS, BALLS_FOCAL)
    assert np.array_equal(ofd, fd_want.numpy()) and np.array_equal(ofov, fov_want.numpy())
    assert float(fov.min()) == 1.0 and float(fov[0, 0, 0]) > 1.0
    # 720 x 1280 JPEGs: crop 60 / 106 per side -> 600 x 1068, padded to 608 x 1088 by pack_stack
    assert pipeline.real_scene_crop(720, 1280) == (60, 106, 600, 1068)
    with pytest.raises(ValueError):
        pipeline.real_scene_crop(11, 640)


def test_oracle_unpack_stack_follows_the_driver():
    """oracle unpack_stack == test_real_scenes.py:44-47 executed literally for one stack."""
    rng = np.random.default_rng(3)
    x = (rng.random((1, 3, 4, 10, 12), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[0, :, 0, 0, :4] = np.array([-1.0, 1.0, 0.0, np.float32(1.0) - np.float32(2 ** -24)], np.float32)
    test_warp_FS = np.squeeze(127.5 * (x + 1.0)).astype(np.uint8)
    test_warp_FS = np.transpose(test_warp_FS, (2, 3, 0, 1))[:, :, :, :]
    got = ref.unpack_stack(x, size=(9, 7))
    assert got.shape == (1, 4, 9, 7, 3) and got.dtype == np.uint8
    for i in range(4):
        assert np.array_equal(got[0, i], test_warp_FS[:9, :7, :, i])
    assert got[0, 0, 0, 0].tolist() == [0, 0, 0] and got[0, 0, 0, 1].tolist() == [255, 255, 255] and got[0, 0, 0, 2].tolist() == [127, 127, 127]


# ---- GPU ------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pl(lib_built):
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from dffinthewild_amd import pipeline
    return pipeline


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["u8", "f32"])
@pytest.mark.parametrize("layout,shape,crop", [("NHWC", (5, 37, 50, 3), None), ("HWCN", (64, 96, 3, 10), None), ("HWNC", (70, 45, 4, 3), (3, 5, 60, 33)),
                                               ("NHWC", (1, 32, 32, 3), None), ("HWCN", (130, 100, 3, 10), (10, 8, 108, 84))])
def test_pack_stack_bit_exact(pl, layout, shape, crop, dtype):
    rng = np.random.RandomState(11)
    raw = rng.randint(0, 256, size=shape).astype(np.uint8 if dtype == "u8" else np.float32)
    want = ref.pack_stack(raw, layout, crop)
    got = pl.pack_stack(torch.from_numpy(raw).cuda(), layout, crop)
    assert got.shape == (1,) + want.shape
    assert np.array_equal(got[0].cpu().numpy(), want)
    # batch of two different stacks, second one a non-contiguous view
    big = torch.from_numpy(np.stack([raw, raw[::-1].copy()])).cuda()
    got2 = pl.pack_stack(big, layout, crop)
    assert np.array_equal(got2[0].cpu().numpy(), want) and np.array_equal(got2[1].cpu().numpy(), ref.pack_stack(raw[::-1], layout, crop))


@pytest.mark.gpu
@pytest.mark.parametrize("name", PACK_CASES)
def test_pack_stack_matches_reference_loaders(pl, lib_built, name):
    """dffw_pack_stack == the reference loader's tensor bit for bit (goldens of all six loaders), from the uint8 source and from its
    float32 copy; dffw_forward_raw on the same source == forward of the loader's tensor (the stem's raw loader does the same
    arithmetic while staging)."""
    g, layout, crop, norm64 = pack_case(name)
    want = g["FS"]
    norm = "f64" if norm64 else "f32"
    for raw in (g["raw"], g["raw"].astype(np.float32)):
        got = pl.pack_stack(torch.from_numpy(raw).cuda(), layout, crop, norm=norm)
        assert got.shape == (1,) + want.shape
        assert np.array_equal(got[0].cpu().numpy(), want), (name, raw.dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ddff", "middlebury", "real_scenes"])
def test_forward_raw_matches_forward_of_reference_loader_tensor(pl, lib_built, name):
    from dffinthewild_amd import graph, synth
    from dffinthewild_amd.Depth_Estimation_Network import Network
    g, layout, crop, norm64 = pack_case(name)
    FS = torch.from_numpy(g["FS"]).unsqueeze(0).cuda()
    N = FS.shape[2]
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(list(graph.param_entries(graph.dff_net_convs())), seed=0, profile="smooth").items()}
    model = Network()
    model.load_state_dict(sd)
    model = model.cuda().eval()
    fd = torch.linspace(0.1, 1.5, N).reshape(1, N, 1, 1).cuda()
    with torch.no_grad():
        a = model(FS, fd)
        b = model.forward_raw(torch.from_numpy(g["raw"]).cuda(), fd, layout, crop=crop, norm="f64" if norm64 else "f32")
    for x, y in zip(a, b):
        assert torch.equal(x, y), name


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_pack_stack_fs6_float64_normalisation_bit_exact(pl, dtype):
    """norm="f64" (DFFW_RAW_NORM_F64): every byte value, both source dtypes, against the oracle's float64 path (which the CPU
    test above pins to the loader's own NumPy lines); norm="f32" on the same input keeps the float32 arithmetic."""
    vals = np.arange(256, dtype=np.uint8)
    raw = np.stack([np.roll(vals, k) for k in range(40)]).reshape(40, 256, 1, 1).repeat(3, axis=2).repeat(5, axis=3)   # (H=40, W=256, 3, N=5)
    raw = raw.astype(np.uint8 if dtype == "u8" else np.float32)
    t = torch.from_numpy(raw).cuda()
    got64 = pl.pack_stack(t, "HWCN", norm="f64")[0].cpu().numpy()
    got32 = pl.pack_stack(t, "HWCN")[0].cpu().numpy()
    assert np.array_equal(got64, ref.pack_stack(raw, "HWCN", norm64=True))
    assert np.array_equal(got32, ref.pack_stack(raw, "HWCN"))
    assert (got64 != got32).any()
    with pytest.raises(ValueError):
        pl.pack_stack(t, "HWCN", norm="f16")


@pytest.mark.gpu
def test_pack_stack_rejects_bad_input(pl):
    with pytest.raises(RuntimeError):
        pl.pack_stack(torch.zeros(2, 32, 32, 3, dtype=torch.uint8))
    with pytest.raises(ValueError):
        pl.pack_stack(torch.zeros(2, 32, 32, 4, dtype=torch.uint8).cuda())
    with pytest.raises(ValueError):
        pl.pack_stack(torch.zeros(2, 32, 32, 3, dtype=torch.uint8).cuda(), crop=(0, 0, 40, 8))


@pytest.mark.gpu
def test_colorize_matches_matplotlib_golden_and_oracle(pl):
    d = torch.from_numpy(GOLD["depth"]).cuda()
    rgb, rng = pl.colorize(d, size=tuple(int(v) for v in GOLD["crop"]), return_range=True)
    assert np.array_equal(rgb[0].cpu().numpy(), GOLD["rgb_minmax"])
    assert float(rng[0, 0]) == float(GOLD["depth"].min()) and float(rng[0, 1]) == float(GOLD["depth"].max())
    lo, hi = (float(v) for v in GOLD["fixed_range"])
    got = pl.colorize(torch.from_numpy(GOLD["depth_fixed"]).cuda(), vrange=(lo, hi))
    assert np.array_equal(got.cpu().numpy(), GOLD["rgb_fixed"])
    # a batch with different ranges per map, full BASELINE size
    g = torch.Generator().manual_seed(5)
    big = torch.rand(3, 256, 256, generator=g) * torch.tensor([1.0, 5.0, 0.01]).reshape(3, 1, 1) + 0.1
    out = pl.colorize(big.cuda(), size=(250, 231)).cpu().numpy()
    for b in range(3):
        assert np.array_equal(out[b], ref.colorize(big[b].numpy(), size=(250, 231)))


@pytest.mark.gpu
@pytest.mark.parametrize("with_conf", [False, True])
def test_masked_metrics_match_numpy(pl, with_conf):
    g = torch.Generator().manual_seed(9)
    B, H, W, h, w = 3, 64, 96, 61, 90
    est = torch.rand(B, H, W, generator=g) * 1.4 + 0.1
    gt = torch.rand(B, h, w, generator=g) * 1.4 + 0.1
    mask = torch.rand(B, h, w, generator=g) > 0.3
    mask[2] = False
    mask[2, 5, 7] = True                       # a single valid pixel
    conf = torch.rand(B, h, w, generator=g) if with_conf else None
    got = pl.masked_metrics(est.cuda(), gt.cuda(), mask.cuda(), conf.cuda() if with_conf else None).cpu().numpy()
    again = pl.masked_metrics(est.cuda(), gt.cuda(), mask.cuda(), conf.cuda() if with_conf else None).cpu().numpy()
    assert np.array_equal(got, again, equal_nan=True)          # fixed reduction order
    for b in range(B):
        want = ref.masked_metrics(est[b, :h, :w].numpy(), gt[b].numpy(), mask[b].numpy(), conf[b].numpy() if with_conf else None)
        n = 12 if with_conf else 10
        assert np.allclose(got[b, :n], want[:n], rtol=2e-6, atol=0), (b, got[b], want)
        if not with_conf:
            assert np.isnan(got[b, 10]) and np.isnan(got[b, 11])


@pytest.mark.gpu
@pytest.mark.parametrize("name", METRIC_CASES)
def test_masked_metrics_match_the_reference_functions(pl, name):
    """dffw_metrics against the reference's own mask_* functions (tests/golden/io_metrics.npz).  The prediction arrives padded
    (test.py:124-126 crops it), here by 3 rows / 5 columns of junk.  Per-pixel terms are float32 on both sides; the reference sums
    them pairwise in float32, dffw_metrics in float64: 2e-6."""
    est, gt, mask, conf, want = metrics_golden(name)
    h, w = gt.shape
    padded = np.full((1, h + 3, w + 5), 7.0, np.float32)
    padded[0, :h, :w] = est
    got = pl.masked_metrics(torch.from_numpy(padded).cuda(), torch.from_numpy(gt)[None].cuda(), torch.from_numpy(mask)[None].cuda(),
                            torch.from_numpy(conf)[None].cuda()).cpu().numpy()[0]
    assert got[0] == want[0]
    assert np.allclose(got, want, rtol=2e-6, atol=0), (got, want)


@pytest.mark.gpu
def test_unpack_stack_bit_exact(pl):
    """dffw_unpack_stack == the oracle byte for byte: random stacks in [-1,1] with the exact end points, values just inside the byte
    boundaries, two stacks, crops; and the documented out-of-range behaviour (x86 NumPy: int32 conversion, low byte)."""
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 3, 5, 40, 72, generator=g) * 2 - 1
    k = torch.arange(0, 256, dtype=torch.float32)
    x[0, 0, 0, 0, :64] = (k[:64] / 127.5 - 1.0)                      # byte boundaries from below / above
    x[0, 1, 0, 1, :64] = torch.nextafter(k[64:128] / 127.5 - 1.0, torch.tensor(2.0))
    x[1, 2, 4, 2, :64] = torch.nextafter(k[192:] / 127.5 - 1.0, torch.tensor(-2.0))
    x[1, 0, 0, 3, :2] = torch.tensor([-1.0, 1.0])
    for size in (None, (33, 70), (1, 1)):
        got = pl.unpack_stack(x.cuda(), size=size).cpu().numpy()
        assert np.array_equal(got, ref.unpack_stack(x.numpy(), size=size)), size
    wild = torch.tensor([1.5, -1.25, 3.0, -3.0, 1000.0, float("nan"), float("inf"), -float("inf"), 1e20]).reshape(1, 1, 1, 1, 9).repeat(1, 3, 1, 1, 1)
    got = pl.unpack_stack(wild.cuda()).cpu().numpy()[0, 0, 0, :, 0]
    t = (127.5 * (wild[0, 0, 0, 0].numpy().astype(np.float32) + np.float32(1.0))).astype(np.float32)
    want = [(int(v) & 255) if np.isfinite(v) and abs(v) < 2.0 ** 31 else 0 for v in t]
    assert got.tolist() == want, (got.tolist(), want)
    with pytest.raises(ValueError):
        pl.unpack_stack(x[:, :2].cuda())
    with pytest.raises(ValueError):
        pl.unpack_stack(x.cuda(), size=(41, 72))
    with pytest.raises(RuntimeError):
        pl.unpack_stack(x)                              # CPU tensor: no fallback


@pytest.mark.gpu
def test_raw_stack_to_colour_map_end_to_end(pl):
    """uint8 stack -> pack_stack -> Network -> colorize / metrics: equals the reference tensor contract fed by hand."""
    from dffinthewild_amd import graph, synth
    from dffinthewild_amd.Depth_Estimation_Network import Network
    entries = list(graph.param_entries(graph.dff_net_convs()))
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries, 0, "smooth").items()}
    model = Network()
    model.load_state_dict(sd)
    model = model.cuda().eval()
    rng = np.random.RandomState(2)
    raw = rng.randint(0, 256, size=(5, 50, 70, 3)).astype(np.uint8)          # DDFF-like (N,H,W,3), needs padding to 64x96
    fd = pl.focus_dists(np.linspace(0.1, 1.5, 5), 1)
    with torch.no_grad():
        a = model(pl.pack_stack(torch.from_numpy(raw).cuda(), "NHWC"), fd)[3]
        b = model(torch.from_numpy(ref.pack_stack(raw, "NHWC")).unsqueeze(0).cuda(), fd)[3]
    assert torch.equal(a, b) and a.shape == (1, 64, 96)
    rgb = pl.colorize(a, size=(50, 70), vrange=(0.1, 1.5))
    assert np.array_equal(rgb[0].cpu().numpy(), ref.colorize(a[0].cpu().numpy(), size=(50, 70), vrange=(0.1, 1.5)))


@pytest.mark.gpu
@pytest.mark.parametrize("layout,shape,crop", [("NHWC", (10, 250, 231, 3), None), ("HWCN", (300, 280, 3, 5), (20, 17, 256, 256)), ("HWNC", (50, 70, 4, 3), None)])
@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_forward_raw_is_bit_identical_to_pack_then_forward(pl, layout, shape, crop, dtype):
    """dffw_forward_raw: the stem's loader normalises / pads the raw stack itself (tiled stem kernel for the two large
    shapes, the expand-first fallback for the small one); results must equal pack_stack + forward bit for bit."""
    from dffinthewild_amd import graph, synth
    from dffinthewild_amd.Depth_Estimation_Network import Network
    entries = list(graph.param_entries(graph.dff_net_convs()))
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries, 0, "smooth").items()}
    model = Network()
    model.load_state_dict(sd)
    model = model.cuda().eval()
    rng = np.random.RandomState(4)
    raw = torch.from_numpy(np.stack([rng.randint(0, 256, size=shape) for _ in range(2)]).astype(np.uint8 if dtype == "u8" else np.float32)).cuda()
    N = shape[{"NHWC": 0, "HWCN": 3, "HWNC": 2}[layout]]
    fd = pl.focus_dists(np.linspace(0.1, 1.5, N), 2)
    with torch.no_grad():
        a = model.forward_raw(raw, fd, layout, crop)
        b = model(pl.pack_stack(raw, layout, crop), fd)
        a64 = model.forward_raw(raw, fd, layout, crop, norm="f64")      # the FS6 loader's float64 normalisation in the stem loader
        b64 = model(pl.pack_stack(raw, layout, crop, norm="f64"), fd)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for x, y in zip(a64, b64):
        assert torch.equal(x, y)
    assert not torch.equal(a64[3], a[3])
