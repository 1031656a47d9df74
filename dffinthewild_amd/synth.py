"""Deterministic synthetic tensors (weights, BN statistics, focal stacks).

No checkpoint or dataset of the reference is available (``/root/reference/.MISSING_LARGE_BLOBS``),
so tests, goldens and ``bench.py`` all run on synthetic state.  The generator below is a
counter-based PRNG that uses only integer arithmetic and IEEE +,-,* in float64, so the same
``(key, seed)`` gives bit-identical float32 tensors on every host (no libm / SIMD ``log``/``cos``
differences between the build container and the GPU box).  The golden fixtures under
``tests/golden/`` were produced by feeding exactly these tensors to the reference network.

Weight statistics follow the reference's own initialiser (He-normal with
``n = k_d*k_h*k_w*C_out``, ``Depth_Estimation_Network.py:59-73``) but the BatchNorm affine
parameters and running statistics are randomised: the reference's default (gamma=1, beta=0,
mean=0, var=1) would make BN an identity and hide folding bugs.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per 64-bit counter value (vectorised, wrapping uint64)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(key: str, n: int, seed: int = 0, lane: int = 0) -> np.ndarray:
    """``n`` float64 values in [0,1) with 24 random bits each, keyed on ``(key, seed, lane)``."""
    base = (_fnv1a(key) ^ ((seed * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)
            ^ ((lane * 0xA0761D6478BD642F) & 0xFFFFFFFFFFFFFFFF))
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(_splitmix64(np.uint64(base) + ctr * np.uint64(0x632BE59BD9B4E019)))
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def bell(key: str, n: int, seed: int = 0) -> np.ndarray:
    """Zero-mean unit-variance bell-shaped values: sqrt(3)*(u1+u2+u3+u4-2) (Irwin-Hall, exact f64)."""
    s = uniform01(key, n, seed, 0) + uniform01(key, n, seed, 1) \
        + uniform01(key, n, seed, 2) + uniform01(key, n, seed, 3)
    return (s - 2.0) * 1.7320508075688772


def tensor_for(key: str, shape, role: str, seed: int = 0, profile: str = "smooth") -> np.ndarray:
    """Synthetic tensor for one state-dict entry; ``role`` is the tag from ``graph.param_entries``.

    profile ``"smooth"``: the 1-channel score convs (classif1-3, confidence.2) are damped so the
    per-slice scores stay within a few units (soft-argmin in its responsive regime, as for a
    trained net), and the 3-channel warp-parameter convs of the End_to_End alignment heads so that
    the predicted shifts stay at a pixel or two.  profile ``"he"``: every conv uses the plain He scale, which saturates the
    soft-argmin (the hardest case for numerical parity).
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    if role == "bn_nbt":
        return np.zeros(shape, dtype=np.int64)
    if role in ("conv", "convT"):
        kd, kh, kw = shape[2:]
        cout = shape[1] if role == "convT" else shape[0]
        std = (2.0 / (kd * kh * kw * cout)) ** 0.5          # He scale, DEN.py:62-64
        if profile == "smooth" and cout == 1:
            std *= 0.25
        if profile == "smooth" and cout == 3:
            std *= 0.15                                      # alpha heads of FlowNetwork: shifts of a pixel or two
        return (bell(key, n, seed) * std).astype(np.float32).reshape(shape)
    u = uniform01(key, n, seed)
    if role == "bn_w":
        v = 0.6 + 0.8 * u
    elif role in ("bn_b", "bias"):
        v = 0.3 * (u - 0.5)
    elif role == "bn_mean":
        v = 0.4 * (u - 0.5)
    elif role == "bn_var":
        v = 0.5 + 1.0 * u
    else:
        raise ValueError(f"unknown role {role!r} for {key}")
    return v.astype(np.float32).reshape(shape)


def state_dict_numpy(entries, seed: int = 0, profile: str = "smooth"):
    """``entries``: iterable of ``(key, shape, role, is_buffer)`` from ``graph.param_entries``."""
    return {k: tensor_for(k, s, r, seed, profile) for k, s, r, _ in entries}


def focal_stack(B: int, N: int, H: int, W: int, seed: int = 1000) -> np.ndarray:
    """FS ~ U[-1,1) float32 in the reference's loader layout (B,3,N,H,W) (test_Dataloader.py:39)."""
    u = uniform01("focal_stack", B * 3 * N * H * W, seed)
    return (2.0 * u - 1.0).astype(np.float32).reshape(B, 3, N, H, W)


def focus_dists(B: int, N: int, H: int = 1, W: int = 1, lo: float = 0.1, hi: float = 1.5) -> np.ndarray:
    """linspace(lo,hi,N) tiled to (B,N,H,W); H=W=1 gives the broadcast layout of Test_dataloader.py:52-54."""
    if N == 1:
        d = np.array([lo], dtype=np.float64)
    else:
        d = lo + (hi - lo) * np.arange(N, dtype=np.float64) / (N - 1)
    out = np.broadcast_to(d.astype(np.float32).reshape(1, N, 1, 1), (B, N, H, W))
    return np.ascontiguousarray(out)
