"""Random parity sweep of the shadow-map rasteriser on the GPU box: random spot lights (position, direction, field of view,
near / far), depth-bias settings, map sizes and scenes (default scene, triangle soups, hostile geometry), GPU vs oracle
bit for bit.
usage: python tests/sweeps/random_sweep_shadow.py FIRST_SEED LAST_SEED"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as ob
from syzygy_amd import abi, lib, meshes, pipelines as pl, scene
from syzygy_amd.pipelines import _memcpy2d_from
from tests.test_raster import _hostile_scene, _soup

bad = 0
no_directional = pl.TStagedBuffer(abi.DirectionalLightPacked, 1)
no_directional.recordCopyToDevice()  # nothing staged: zero directional lights
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([17, 64, 200, 256]))
    kind = int(rng.integers(0, 3))
    ms = {0: lambda: meshes.reference_default_scene(), 1: lambda: _soup(seed, int(rng.integers(1, 500)), spread=float(rng.uniform(5, 120))),
          2: lambda: _hostile_scene(seed, bool(rng.integers(0, 2)))}[kind]()
    spots = (abi.SpotLightPacked * 2)()
    for k in range(2):
        position = (float(rng.uniform(-40, 40)), float(rng.uniform(-50, -2)), float(rng.uniform(-40, 40)))
        target = (float(rng.uniform(-10, 10)), float(rng.uniform(-10, 0)), float(rng.uniform(-10, 10)))
        forward = tuple(t - p for t, p in zip(target, position))
        p = abi.SpotlightParams()
        lib().szg_spotlight_params_default(abi.f3(1, 1, 1), abi.f3(*position), abi.f3(*[float(v) for v in scene.eulers_from_forward(forward)]), C.byref(p))
        p.verticalFOVDegrees = float(rng.uniform(10.0, 150.0))
        p.horizontalScale = float(rng.uniform(0.5, 2.0))
        p.near_plane, p.far_plane = float(10.0 ** rng.uniform(-2, 0.5)), float(10.0 ** rng.uniform(1.5, 4))
        lib().szg_make_spot(C.byref(p), C.byref(spots[k]))
    bias = (0.0, 0.0) if rng.integers(0, 2) else (float(rng.uniform(-4, 4)), float(rng.uniform(-3, 3)))
    deferred = pl.DeferredShadingPipeline((16, 16), max_spot_lights=2, max_shadow_maps=2, shadow_map_dim=dim)
    deferred.setConfiguration(abi.DeferredConfiguration(bias[0], bias[1]))
    deferred.recordShadowRaster(None, no_directional, spots, ms)
    torch.cuda.synchronize()
    sm = deferred.shadowMaps()
    for slot in range(2):
        pv = abi.Mat4()
        lib().szg_mat4_mul(C.byref(spots[slot].projection), C.byref(spots[slot].view), C.byref(pv))
        want = ob.shadow_raster(pv, dim, ms, bias[0], bias[1], threads=8)
        got = _memcpy2d_from(sm.maps[slot], dim * 4, dim).cpu().numpy().view(np.float32).reshape(dim, dim)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        if not same.all():
            bad += 1
            print("seed", seed, "MISMATCH slot", slot, "kind", kind, "dim", dim, "bias", bias, (~same).sum(), "texels", flush=True)
    deferred.cleanup()
print("done, mismatching seeds:", bad, "processed up to", seed)
