"""Shared helpers for the tests: seeded inputs and comparisons."""
import numpy as np

from syzygy_amd import abi, scene

RTOL = 1e-4  # north_star: 1e-4 relative fp32
# Absolute floor for "relative" comparisons: 1/16 of one UNORM16 step of the scene colour
# for image-space values; LUT tests pass their own.
ATOL_COLOR = 1.0 / 65535.0 / 16.0


def rel_err(a, b, atol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    denom = np.maximum(np.maximum(np.abs(a), np.abs(b)), atol / RTOL)
    return np.abs(a - b) / denom


def assert_close(a, b, rtol=RTOL, atol=ATOL_COLOR, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert (nan_a == nan_b).all(), f"{what}: NaN patterns differ ({nan_a.sum()} vs {nan_b.sum()})"
    ok = ~nan_a
    diff = np.abs(a[ok].astype(np.float64) - b[ok].astype(np.float64))
    bound = atol + rtol * np.maximum(np.abs(a[ok]), np.abs(b[ok])).astype(np.float64)
    bad = diff > bound
    if bad.any():
        i = np.argmax(diff - bound)
        raise AssertionError(
            f"{what}: {bad.sum()} of {bad.size} values outside rtol={rtol} atol={atol}; worst |d|={diff[i]:.3e} "
            f"a={a[ok][i]!r} b={b[ok][i]!r}")
    return float((diff / np.maximum(bound, 1e-300)).max()) if diff.size else 0.0


class Inputs:
    """Packed parameter blocks for one frame."""

    def __init__(self, width, height, elevation_degrees=70.0, spots=64, camera=None, grid=(6, 4), atmosphere_edit=None):
        self.width, self.height = width, height
        self.synthetic = scene.SyntheticScene(grid=grid)
        self.atmosphere = scene.default_atmosphere(scene.sun_euler_for_elevation(elevation_degrees))
        if atmosphere_edit is not None:
            atmosphere_edit(self.atmosphere)
        self.atm, self.sun, self.moon = scene.atmosphere_baked(self.atmosphere, self.synthetic.bounds)
        self.camera = camera if camera is not None else scene.default_camera()
        self.cam = scene.camera_packed(self.camera, width / height)
        self.dirs = (abi.DirectionalLightPacked * 2)(self.sun, self.moon)
        self.spot_count = spots
        self.spots = scene.spot_ring(spots)
        self.rect = abi.Rect(0, 0, width, height)


def rowtile(height, block_rows, rank, nranks):
    from syzygy_amd import lib

    rows = lib().szg_rowtile_local_rows(height, block_rows, rank, nranks)
    return abi.RowTile(block_rows, rank, nranks, rows)


def global_rows(height, block_rows, rank, nranks):
    """Global row index of every local row of `rank` (cyclic row blocks)."""
    out = []
    nblocks = (height + block_rows - 1) // block_rows
    for b in range(rank, nblocks, nranks):
        out.extend(range(b * block_rows, min(height, (b + 1) * block_rows)))
    return np.array(out, dtype=np.int64)
