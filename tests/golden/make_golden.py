"""Generates tests/golden/oracle_golden.npz from the CPU oracle (oracle/szg_oracle.cpp).

The reference ships no golden vectors for this path and cannot be run here, so
these fixtures pin the ORACLE (and through the GPU parity tests, the kernels) against
accidental change; they are data produced by this repository's own code:

    python -m tests.golden.make_golden        # rewrites oracle_golden.npz

Contents: transmittance LUT 64x16 and the four corners + centre of the 256x64 and 512x128
LUTs, a 64x32 sky-view LUT, and a 64x40 frame (G-buffer planes, depth, lights and composite
results) at three sun elevations (70, 5, -3 degrees) with 6 spot lights; the raster oracle's G-buffer of the
editor's start-up scene (64x40) and one 64x64 sun shadow map of it.
"""
import os

import numpy as np

from oracle import binding as ob
from syzygy_amd import abi, meshes, scene  # noqa: F401
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))


def generate():
    out = {}
    for elevation in (70.0, 5.0, -3.0):
        tag = f"e{int(elevation):+d}"
        inp = util.Inputs(64, 40, elevation_degrees=elevation, spots=6)
        tl = ob.transmittance_lut(inp.atm, 64, 16)
        sl = ob.skyview_lut(inp.atm, inp.cam, tl, 64, 32)
        f = ob.HostFrame(64, 40)
        ob.gbuffer_fill(f, inp.rect, None, inp.cam, inp.synthetic.fill)
        ob.lights(f, inp.rect, None, None, inp.cam, inp.dirs, 2, 1, inp.spots, 6)
        out[f"{tag}_lights_f32"] = f.debug.copy()
        out[f"{tag}_lights_unorm16"] = f.color.copy()
        ob.composite(f, inp.rect, None, None, inp.atm, inp.cam, inp.dirs, 0, tl, sl)
        out[f"{tag}_composite_f32"] = f.debug.copy()
        out[f"{tag}_composite_unorm16"] = f.color.copy()
        out[f"{tag}_skyview_64x32"] = sl
        if elevation == 70.0:
            out["transmittance_64x16"] = tl
            out["gbuffer_position"] = f.position.copy()
            out["gbuffer_normal"] = f.normal.copy()
            out["gbuffer_diffuse"] = f.diffuse.copy()
            out["gbuffer_orm"] = f.orm.copy()
            out["depth"] = f.depth.copy()
            for (w, h) in ((256, 64), (512, 128)):
                big = ob.transmittance_lut(inp.atm, w, h, threads=8)
                pts = [(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (h // 2, w // 2), (h // 3, (2 * w) // 3)]
                out[f"transmittance_{w}x{h}_probes"] = np.stack([big[y, x] for (y, x) in pts])
    # raster oracle (oracle/szg_oracle_raster.cpp): the reference's default scene
    import ctypes as C

    from syzygy_amd import lib

    inp = util.Inputs(64, 40, elevation_degrees=40.0, spots=0)
    ms = meshes.reference_default_scene()
    f = ob.HostFrame(64, 40)
    ob.gbuffer_raster(f, inp.rect, None, inp.cam, ms)
    out["raster_depth"] = f.depth.copy()
    out["raster_position"] = f.position.copy()
    out["raster_normal"] = f.normal.copy()
    out["raster_diffuse"] = f.diffuse.copy()
    out["raster_orm"] = f.orm.copy()
    pv = abi.Mat4()
    lib().szg_mat4_mul(C.byref(inp.sun.projection), C.byref(inp.sun.view), C.byref(pv))
    out["raster_sun_shadow_64"] = ob.shadow_raster(pv, 64, ms)
    return out


if __name__ == "__main__":
    data = generate()
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(data)} arrays")
