"""The C ABI from a plain C host (examples/c_host.c): built with gcc against include/dffw.h, libdffw.so and the HIP runtime only
(no PyTorch in that process), run on the GPU, compared with the reference golden.  CPU part: the example compiles."""
import os
import struct
import subprocess

import numpy as np
import pytest

from dffinthewild_amd import graph, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_host.c")
LIBDIR = os.path.join(ROOT, "dffinthewild_amd")


def build(tmp_path):
    exe = str(tmp_path / "c_host")
    # a plain C compiler: the HIP runtime API header needs its platform macro when hipcc is not the driver
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", SRC, "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           "-L" + LIBDIR, "-ldffw", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_c_host_compiles_against_the_header(lib_built, tmp_path):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_c_host_forward_matches_reference_golden(lib_built, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "den_batch2_bcast.npz"))
    B, N, H, W = (int(g[k]) for k in ("B", "N", "H", "W"))
    entries = list(graph.param_entries(graph.dff_net_convs()))
    sd = synth.state_dict_numpy(entries, seed=int(g["wseed"]), profile=str(g["profile"]))
    with open(tmp_path / "weights.bin", "wb") as f:
        floats = [(k, v) for k, v in sd.items() if v.dtype == np.float32]
        f.write(struct.pack("<i", len(floats)))
        for k, v in floats:
            name = k.encode()
            f.write(struct.pack("<i", len(name)) + name + struct.pack("<q", v.size) + np.ascontiguousarray(v).tobytes())
    FS = synth.focal_stack(B, N, H, W, seed=int(g["iseed"]))
    fd = synth.focus_dists(B, N, 1, 1).reshape(B, N)
    with open(tmp_path / "input.bin", "wb") as f:
        f.write(struct.pack("<4i", B, N, H, W) + FS.tobytes() + np.ascontiguousarray(fd).tobytes())
    exe = build(tmp_path)
    run = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "input.bin"), str(tmp_path / "out.bin")],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(4, B, H, W)
    for i, name in enumerate(("mid_out", "pred1", "pred2", "pred3")):
        ref = g[name].astype(np.float64)
        err = np.linalg.norm(out[i].astype(np.float64).ravel() - ref.ravel()) / np.linalg.norm(ref.ravel())
        assert err <= 1e-3, (name, err)


@pytest.mark.gpu
def test_c_host_e2e_forward_matches_reference_golden(lib_built, tmp_path):
    """The End_to_End variant from the same C host (`c_host ... e2e`): DFFW_NET_E2E engine from the 522-entry state dict,
    dffw_forward_e2e, against the golden the reference's End_to_End.Network produced (BASELINE config 5's call path without torch)."""
    import glob
    from oracle.make_goldens_e2e import net_inputs
    path = [p for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "e2e_net_*.npz"))) if "smooth_64x96" in p][0]
    g = np.load(path)
    sd = synth.state_dict_numpy(list(graph.param_entries(graph.e2e_convs())), seed=int(g["wseed"]), profile=str(g["profile"]))
    FS, fd, fov = net_inputs(int(g["H"]), int(g["W"]), int(g["iseed"]))
    FS, fd, fov = (np.asarray(t, dtype=np.float32) for t in (FS, fd, fov))
    B, _, N, H, W = FS.shape
    with open(tmp_path / "weights.bin", "wb") as f:
        floats = [(k, np.asarray(v)) for k, v in sd.items() if np.asarray(v).dtype == np.float32]
        f.write(struct.pack("<i", len(floats)))
        for k, v in floats:
            name = k.encode()
            f.write(struct.pack("<i", len(name)) + name + struct.pack("<q", v.size) + np.ascontiguousarray(v).tobytes())
    with open(tmp_path / "input.bin", "wb") as f:
        f.write(struct.pack("<4i", B, N, H, W) + np.ascontiguousarray(FS).tobytes() +
                np.ascontiguousarray(fd.reshape(B, N)).tobytes() + np.ascontiguousarray(fov.reshape(B, N)).tobytes())
    exe = build(tmp_path)
    run = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "input.bin"), str(tmp_path / "out.bin"), "e2e"],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    maps, aligned = out[:4 * B * H * W].reshape(4, B, H, W), out[4 * B * H * W:].reshape(B, 3, N, H, W)

    def rel(a, b):
        a, b = a.astype(np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
        return np.linalg.norm(a - b) / np.linalg.norm(b)

    checked = 0
    for i, name in enumerate(("mid_out", "pred1", "pred2", "pred3")):
        if name in g.files:
            assert rel(maps[i], g[name]) <= 1e-3, name
            checked += 1
    assert checked >= 1
    if "aligned" in g.files:
        assert rel(aligned, g["aligned"]) <= 1e-3
