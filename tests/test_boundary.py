"""CPU: the drop-in boundary — weight contract, call contract and the C ABI surface.  No GPU compute."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn as nn

from dffinthewild_amd import graph, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng(lib_built):
    from dffinthewild_amd import engine
    return engine


def test_library_exports_every_declared_symbol(eng):
    header = open(os.path.join(ROOT, "include", "dffw.h")).read()
    declared = set(re.findall(r"\b(dffw_[a-z0-9_]+)\s*\(", header))
    assert declared == set(eng.ABI_SYMBOLS)
    raw = ctypes.CDLL(eng.LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), sym
    assert b"gfx950" in eng.lib.dffw_version()


def test_native_param_table_matches_python_table(eng):
    native = eng.param_table()
    rows = list(graph.param_entries(graph.dff_net_convs()))
    assert len(native) == len(rows) == 384
    for (nk, nshape, flags), (k, shape, role, is_buf) in zip(native, rows):
        assert nk == k and tuple(nshape) == tuple(shape)
        assert bool(flags & 1) == is_buf
        assert bool(flags & 2) == (role == graph.ROLE_BN_NBT)
    dead = [k for k, _, f in native if f & 4]
    assert len(dead) == 24 and all(("pre_conv" in k or "redir3" in k) for k in dead)


def test_native_param_table_e2e(eng):
    """DFFW_NET_E2E: the 522 entries of End_to_End.Network in the reference's registration order."""
    native = eng.param_table(eng.NET_E2E)
    rows = list(graph.param_entries(graph.e2e_convs()))
    assert len(native) == len(rows) == 522
    for (nk, nshape, flags), (k, shape, role, is_buf) in zip(native, rows):
        assert nk == k and tuple(nshape) == tuple(shape)
        assert bool(flags & 1) == is_buf
    from dffinthewild_amd.End_to_End import Network
    m = Network()
    assert list(m.state_dict().keys()) == [k for k, *_ in rows]
    new = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(rows).items()}
    m.load_state_dict({"module." + k: v for k, v in new.items()})
    m.eval()
    with pytest.raises(RuntimeError):                      # CPU tensors: no fallback
        m(torch.zeros(1, 3, 10, 32, 32), torch.zeros(1, 10, 1, 1), torch.ones(1, 1, 10, 1, 1))
    with pytest.raises(ValueError):                        # the heads pool to 10 slices
        m(torch.zeros(1, 3, 5, 32, 32), torch.zeros(1, 5, 1, 1), torch.ones(1, 1, 5, 1, 1))


def test_state_dict_contract(eng):
    from dffinthewild_amd import Network
    m = Network()
    sd = m.state_dict()
    assert len(sd) == 384
    assert sd["DFF_net.FM_measure.Focus_extraction.0.0.weight"].shape == (8, 3, 1, 9, 9)
    assert sd["DFF_net.deconv_1.0.weight"].shape == (64, 32, 3, 3, 3)       # ConvTranspose3d (Cin,Cout,...)
    assert sd["DFF_net.SPP_module.combine2.0.0.weight"].shape == (128, 192, 3, 3, 3)
    assert sd["DFF_net.dres4.conv6.1.num_batches_tracked"].dtype == torch.long
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 4_038_832
    # synthetic state (and a DataParallel-prefixed copy) loads strictly
    entries = list(graph.param_entries(graph.dff_net_convs()))
    new = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries).items()}
    m.load_state_dict(new)
    m.load_state_dict({"module." + k: v for k, v in new.items()})
    assert torch.equal(m.state_dict()["DFF_net.classif3.0.weight"], new["DFF_net.classif3.0.weight"])
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in new.items() if "classif3" not in k})


def test_reference_call_sequence_on_cpu_fails_loudly(eng):
    """test.py:30-32,78,85 sequence works up to the model call; the call itself needs the GPU."""
    from dffinthewild_amd.Depth_Estimation_Network import Network
    model = Network()
    model = model.cpu()
    model = nn.DataParallel(model)
    model.module.load_state_dict(model.module.state_dict())
    model.eval()
    FS = torch.zeros(1, 3, 4, 32, 32)
    fd = torch.zeros(1, 4, 32, 32)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        model.module(FS, fd)


def test_shape_contract_errors(eng):
    from dffinthewild_amd import Network
    m = Network().eval()
    fd = torch.zeros(1, 4, 1, 1)
    with pytest.raises(ValueError, match="multiples of 32"):
        m(torch.zeros(1, 3, 4, 48, 48), fd)
    with pytest.raises(ValueError, match="3 colour channels"):
        m(torch.zeros(1, 4, 3, 32, 32), torch.zeros(1, 3, 1, 1))       # (B,N,3,H,W) is NOT the layout
    with pytest.raises(ValueError, match="broadcast"):
        m(torch.zeros(1, 3, 4, 32, 32), torch.zeros(1, 5, 1, 1))
    with pytest.raises(RuntimeError, match="eval"):
        Network()(torch.zeros(1, 3, 4, 32, 32), fd)
    with pytest.raises(ValueError):
        Network(precision="fp8")


def test_c_abi_argument_errors_without_gpu(eng):
    lib = eng.lib
    assert lib.dffw_param_count(7) < 0 and b"unknown net" in lib.dffw_last_error()
    h = ctypes.c_void_p()
    assert lib.dffw_engine_create(0, 0, None, 0, 0, ctypes.byref(h)) == -1
    assert lib.dffw_workspace_bytes(None, 1, 1, 32, 32) < 0


def test_network_deepcopy_and_pickle(eng, tmp_path):
    """The usual nn.Module idioms (EMA / SWA clone, torch.save(model)): a copy carries the parameters only — never the lock
    or the cache of ctypes engine handles — and has its own empty engine cache."""
    import copy
    import io
    import pickle
    from dffinthewild_amd import Network
    from dffinthewild_amd.End_to_End import Network as E2E
    for cls in (Network, E2E):
        m = cls().eval()
        m._engines[99] = ("fingerprint", object())        # stands for a packed engine: must not travel
        c = copy.deepcopy(m)
        assert c is not m and c._engines == {} and c._engines is not m._engines and c._guard is not m._guard
        assert c._token is not m._token and c._inherited_fp is None and not c.training and c.precision == m.precision
        for (k1, v1), (k2, v2) in zip(m.state_dict().items(), c.state_dict().items()):
            assert k1 == k2 and v1.data_ptr() != v2.data_ptr() and torch.equal(v1, v2)
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        r = torch.load(buf, weights_only=False)
        assert isinstance(r, cls) and r._engines == {} and list(r.state_dict()) == list(m.state_dict())
        r.load_state_dict(m.state_dict())                  # a restored module is fully functional (token, lock recreated)
        assert r._token[0] == 1
        assert pickle.loads(pickle.dumps(m))._engines == {}
    # the engine wrapper itself refuses: a second owner of the dffw_engine* would free it twice
    e = eng.Engine.__new__(eng.Engine)
    for op in (copy.copy, copy.deepcopy, pickle.dumps):
        with pytest.raises(TypeError):
            op(e)


def test_comm_entry_points_fail_cleanly_without_gpu(eng):
    """dffw_comm_* / dffw_allgather (RCCL, bound with dlopen at first use): bad arguments and a missing GPU give error codes
    and messages, never a crash."""
    lib = eng.lib
    assert lib.dffw_allgather(None, None, None, 0, None) == -1 and b"null" in lib.dffw_last_error()
    assert lib.dffw_comm_rank(None) == -1 and lib.dffw_comm_size(None) == -1
    lib.dffw_comm_destroy(None)
    h = ctypes.c_void_p()
    assert lib.dffw_comm_init_rank(0, 2, 5, b"\0" * eng.COMM_ID_BYTES, ctypes.byref(h)) == -1 and not h.value
    if not torch.cuda.is_available():
        rc = lib.dffw_comm_init_rank(0, 1, 0, b"\0" * eng.COMM_ID_BYTES, ctypes.byref(h))
        assert rc < 0 and not h.value and lib.dffw_last_error()


def test_isa_wait_lint_finds_the_round6_hazard_and_nothing_else():
    """tools/isa_wait_lint.py (round 6): no instruction hipcc places between an inline-asm load and the wait that covers it may read the load's destination registers.
    The shipped conv_slice kernels are clean; the development build without the scheduling barrier between wait and tie (-DDFFW_SLICE_HAZARD=2) reproduces the copies that made
    conv_slice64_head wrong by 2e-3 (profiles/r06_wait_tie_hazard.txt).  (hipcc cross-compiles gfx950 without a GPU; two compilations of one source, ~1 min.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    src = os.path.join(root, "dffinthewild_amd", "csrc", "dffw_conv_slice.hip")
    tool = os.path.join(root, "tools", "isa_wait_lint.py")
    clean = subprocess.run([sys.executable, tool, src], capture_output=True, text=True)
    assert clean.returncode == 0, clean.stdout
    bad = subprocess.run([sys.executable, tool, src], capture_output=True, text=True, env=dict(os.environ, LINT_DEFS="-DDFFW_SLICE_HAZARD=2"))
    assert bad.returncode == 1 and "conv_slice64_head" in bad.stdout, bad.stdout
