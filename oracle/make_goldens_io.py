"""Writes tests/golden/io_jet.npz from matplotlib itself (the library the reference calls: test.py:129
`cm.get_cmap('jet')`, test_real_scenes.py:46): the 256-entry table as uint8 and one depth map pushed through the exact
post-processing sequence of End_to_End/test_real_scenes.py:40-52.  Run in the build container:  python oracle/make_goldens_io.py"""
import os

import matplotlib
import numpy as np

cmap = matplotlib.colormaps["jet"]          # == cm.get_cmap('jet') of the matplotlib the reference pins
lut = (255 * cmap(np.arange(256) / 255.0)[:, :3]).astype(np.uint8)

rng = np.random.RandomState(7)
depth = (rng.rand(1, 96, 160).astype(np.float32) * 1.4 + 0.1)
depth[0, 3, 5] = depth.max()                 # x == 1 exactly after normalisation
shape = (83, 150)
# test_real_scenes.py:40, 46-52
t = (depth - np.min(depth)) / (np.max(depth) - np.min(depth))
color = cmap(t)[..., :3]
color = 255 * np.squeeze(color)
color = color.astype(np.uint8)[:shape[0], :shape[1], :]
# test.py:130-132 (fixed range, values below / above the range and one NaN)
d2 = depth[0].copy()
d2[0, 0] = np.nan
lo, hi = 0.3, 1.2
color_fixed = (255 * cmap((d2 - lo) / (hi - lo))[..., :3]).astype(np.uint8)

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "io_jet.npz")
np.savez_compressed(out, lut_u8=lut, depth=depth, crop=np.asarray(shape), rgb_minmax=color, depth_fixed=d2,
                    fixed_range=np.asarray([lo, hi], np.float32), rgb_fixed=color_fixed,
                    matplotlib_version=np.asarray(matplotlib.__version__))
print("wrote", out, lut.shape, color.shape)
