#!/usr/bin/env python3
"""Golden vectors for the input assembly: what the REFERENCE's own loaders hand to `model(FS, focus_dists)`.

The six loader classes of `/root/reference/Depth_Estimation_Test/test_Dataloader.py` (FS6_dataset :13-55, HCI_dataset :56-91,
DDFF12dataset_benchmark :93-143, Smartphone :144-236, Middlebury :238-289) and `/root/reference/End_to_End/Test_dataloader.py`
(Real_Scenes :8-75) cannot be imported here: their modules import cv2 / h5py / OpenEXR, which this image lacks, and the data sets
are absent.  Everything between the decode calls is plain NumPy / torch, though.  So this script reads the two files where they
lie, takes the ClassDef nodes, and compiles exactly those, unmodified, into a namespace in which the DECODE dependencies are
stand-in data sources: `cv2.imread(path)` returns a seeded uint8 image, `h5py.File(path)` a dict of seeded arrays, `listdir` /
`open` the file names and the two text files of a scene.  `__init__` and `__getitem__` of every class then run as the reference
wrote them - the arithmetic (`/127.5 - 1.0` in float32 or float64, the transposes, the crops, the -1 padding, the focus-distance
tiling, 1/d, the relative fields of view) is the reference's code executing, not a restatement, and nothing of it is written into
this repo.  (`np.int`, removed from NumPy 1.24+, is aliased to `int` for the duration: the Smartphone loader spells it that way.)

Run here only (TEST INFRASTRUCTURE; /root/reference does not exist on the GPU box):
    python oracle/make_goldens_pack.py   ->   tests/golden/io_pack_*.npz
Each file: raw = the decoded source array in the layout the loader holds it (what dffw_pack_stack / oracle pack_stack take),
layout / crop / norm64 = how pack_stack is to be called on it, FS = the loader's output tensor (3,N,Hp,Wp) float32, plus the focus
distances (and relative FOVs) the loader returns.
"""
import ast
import io
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

REF_DEN = "/root/reference/Depth_Estimation_Test/test_Dataloader.py"
REF_E2E = "/root/reference/End_to_End/Test_dataloader.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def image(h, w, tag, plain=False):
    """(h,w,3) uint8 'decoded image': every byte value occurs and the strides differ per axis (a transposed or mirrored read
    cannot pass).  plain: strictly periodic, for the two large cases (512 x 512 x 10, 504 x 378 x 10), so that their float32
    outputs compress to a fixture of a few hundred KiB."""
    y, x, c = np.meshgrid(np.arange(h), np.arange(w), np.arange(3), indexing="ij")
    return ((3 * y + 7 * x + 59 * c + 101 * tag + (0 if plain else (y * x) // 13)) % 256).astype(np.uint8)


class FakeCv2:
    IMREAD_UNCHANGED = -1

    def __init__(self, table):
        self.table = table
        self.calls = []

    def imread(self, path, *flags):
        self.calls.append(path)
        return self.table[path].copy()


class FakeOs:
    """the two members the loaders touch: os.listdir and os.environ (+ os.path for Real_Scenes)"""
    path = os.path

    def __init__(self, listing):
        self.listing = listing
        self.environ = {}

    def listdir(self, p):
        return list(self.listing[p])


def classes(path, names, ns):
    tree = ast.parse(open(path).read(), path)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), [n.name for n in keep]
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def base_ns(**extra):
    ns = {"np": np, "torch": torch, "F": F, "Dataset": Dataset, "random": __import__("random"),
          "join": os.path.join, "tqdm": lambda it, **kw: it}
    ns.update(extra)
    return ns


def save(name, **arrays):
    p = os.path.join(OUT, f"io_pack_{name}.npz")
    np.savez_compressed(p, **arrays)
    print(f"{name}: FS {arrays['FS'].shape} {arrays['FS'].dtype}, raw {arrays['raw'].shape}, {os.path.getsize(p) / 1024:.0f} KiB")


def fs6():
    # test_Dataloader.py:13-46: five 256x256 images per sample, float64 concatenate -> /127.5 - 1.0 in float64 -> torch.Tensor
    root = "Datasets/fs_6/test/"
    names_all = [f"s{i:02d}All.tif" for i in range(5)]
    table = {root + n: image(256, 256, i) for i, n in enumerate(names_all)}
    cv2 = FakeCv2(table)
    ns = base_ns(cv2=cv2, listdir=lambda p: names_all + ["s00Dpt.exr"], isfile=lambda p: True)
    cls = classes(REF_DEN, ["FS6_dataset"], ns)["FS6_dataset"]
    ds = cls()
    ds.read_dpt = lambda p: np.full((256, 256), 0.7, np.float16)       # OpenEXR decode (depth map: not part of the model input)
    FS, dpt, fd, mask = ds[0]
    raw = np.stack([table[root + n] for n in names_all], axis=3)           # (H,W,3,N): what `mats_input` holds before the division
    save("fs6", raw=raw, layout=np.asarray("HWCN"), crop=np.asarray([-1, -1, -1, -1]), norm64=np.asarray(1),
         FS=FS.numpy(), focus_dists=fd.numpy())


def hci():
    # test_Dataloader.py:56-91: hdf5 stack (10,512,512,3) -> float32 -> (512,512,3,10) -> /127.5 - 1.0 -> (3,10,512,512)
    stack = np.stack([image(512, 512, 10 + i, plain=True) for i in range(10)], axis=0)[None]
    disp = np.linspace(-2.5, 2.5, 512 * 512, dtype=np.float32).reshape(1, 512, 512)
    h5 = {"stack_val": stack, "disp_val": disp, "focus_position_disp": np.linspace(-2.0, 2.0, 10, dtype=np.float32)[None]}
    ns = base_ns(h5py=type("H5", (), {"File": staticmethod(lambda p, m: h5)}))
    ds = classes(REF_DEN, ["HCI_dataset"], ns)["HCI_dataset"]()
    FS, gt, fd, mask = ds[0]
    save("hci", raw=stack[0], layout=np.asarray("NHWC"), crop=np.asarray([-1, -1, -1, -1]), norm64=np.asarray(0),
         FS=FS.numpy(), focus_dists=fd.numpy()[:, ::64, ::64])            # (the tiled map is constant per slice: a sample of it)


def ddff():
    # test_Dataloader.py:93-143: hdf5 stack (S,H,W,3) -> float32 /127.5 - 1.0 -> -1 padding to multiples of 32 -> (3,S,Hp,Wp)
    stack = np.stack([image(75, 110, 20 + i) for i in range(4)], axis=0)[None]     # a ragged size (real stacks: 10 x 383 x 552)
    ns = base_ns(h5py=type("H5", (), {"File": staticmethod(lambda p, m: {"stack_test": stack})}))
    ds = classes(REF_DEN, ["DDFF12dataset_benchmark"], ns)["DDFF12dataset_benchmark"]()
    FS, fd = ds[0]
    save("ddff", raw=stack[0], layout=np.asarray("NHWC"), crop=np.asarray([-1, -1, -1, -1]), norm64=np.asarray(0),
         FS=np.asarray(FS), focus_dists=fd.numpy()[:, ::96, ::96])


def smartphone():
    # test_Dataloader.py:144-236: 504x378 JPEGs, centre crop [84:-84, 63:-63] -> (336,252,10,3) float32 -> /127.5 - 1.0 ->
    # (3,10,336,252) -> F.pad to (352,256) with -1
    root = "Datasets/Real_data_DP/test/"
    idx = np.rint(np.linspace(0, 48, 10, endpoint=True)).astype(int)
    table = {f"{root}scaled_images/scene0/{j}/result_scaled_image_center.jpg": image(504, 378, 30 + k, plain=True) for k, j in enumerate(idx)}
    table[f"{root}merged_depth/scene0/result_merged_depth_center.png"] = image(504, 378, 3)[:, :, 0]
    table[f"{root}merged_conf/scene0/result_merged_conf_center.exr"] = (image(504, 378, 4) / 200.0).astype(np.float32)
    cv2 = FakeCv2(table)
    ns = base_ns(cv2=cv2, os=FakeOs({root + "scaled_images/": ["scene0"]}))
    np.int = int                                                            # removed alias the loader still uses (:150)
    try:
        ds = classes(REF_DEN, ["Smartphone"], ns)["Smartphone"]()
    finally:
        del np.int
    with np.errstate(divide="ignore"):
        FS, gt, fd, mask, conf = ds[0]
    raw = np.stack([table[f"{root}scaled_images/scene0/{j}/result_scaled_image_center.jpg"] for j in idx], axis=2)   # (H,W,N,3)
    save("smartphone", raw=raw, layout=np.asarray("HWNC"), crop=np.asarray([84, 63, 336, 252]), norm64=np.asarray(0),
         FS=FS.numpy(), focus_dists=fd.numpy()[:, ::88, ::64])


def middlebury():
    # test_Dataloader.py:238-289: 15 uint8 images concatenated (H,W,3,15) -> /127.5 - 1.0 on the uint8 array (float64) ->
    # torch.Tensor -> np.pad with -1
    paths = [f"mb/im{i:02d}.png" for i in range(15)]
    table = {p: image(70, 100, 40 + i) for i, p in enumerate(paths)}
    table["mb/disp.exr"] = np.full((70, 100), 30.0, np.float32)
    cv2 = FakeCv2(table)
    listing = " ".join(paths + ["mb/disp.exr"]) + "\n"
    ns = base_ns(cv2=cv2, os=FakeOs({}), open=lambda p, m="r": io.StringIO(listing))
    ds = classes(REF_DEN, ["Middlebury"], ns)["Middlebury"]()
    FS, depth, fd, mask = ds[0]
    raw = np.stack([table[p] for p in paths], axis=3)                      # (H,W,3,N)
    save("middlebury", raw=raw, layout=np.asarray("HWCN"), crop=np.asarray([-1, -1, -1, -1]), norm64=np.asarray(1),
         FS=np.asarray(FS), focus_dists=fd.numpy()[:, ::32, ::32])


def real_scenes():
    # End_to_End/Test_dataloader.py:8-75: ten images, 1/12 border crop, float32 /127.5 - 1.0, (3,10,H,W), -1 padding;
    # focus_dists = 1/d, relative FOV = (1/f - 1/d) / min(...)
    root, scene = "Datasets/", "balls"
    files = [f"{i:02d}.jpg" for i in range(10)]
    H, W = 150, 220                                                          # crop 12 / 18 per side -> 126 x 184 -> padded 128 x 192
    table = {f"{root}{scene}/{f}": image(H, W, 50 + i) for i, f in enumerate(files)}
    cv2 = FakeCv2(table)
    d = [0.10 + 0.17 * i for i in range(10)]
    texts = {f"{root}{scene}/focus_distance.txt": "".join(f"{v!r}\n" for v in d), f"{root}{scene}/focal_length.txt": "0.0262\n"}
    ns = base_ns(cv2=cv2, os=FakeOs({root: [scene], f"{root}{scene}/": files + ["focus_distance.txt", "focal_length.txt"]}),
                 open=lambda p, m="r": io.StringIO(texts[p]))
    ds = classes(REF_E2E, ["Real_Scenes"], ns)["Real_Scenes"]()
    FS, fd, fov, before = ds[0]
    ch, cw = H // 12, W // 12
    raw = np.stack([table[f"{root}{scene}/{f}"] for f in files], axis=3)    # (H,W,3,N)
    save("real_scenes", raw=raw, layout=np.asarray("HWCN"), crop=np.asarray([ch, cw, H - 2 * ch, W - 2 * cw]), norm64=np.asarray(0),
         FS=np.asarray(FS), focus_dists=fd.numpy(), rel_fov=fov.numpy(), focus_distance_m=np.asarray(d), focal_length=np.asarray(0.0262),
         before_pad=np.asarray(before))


if __name__ == "__main__":
    for f in (fs6, hci, ddff, smartphone, middlebury, real_scenes):
        f()
    sys.exit(0)
