"""Random parity sweep of both rasterisers with HOSTILE INPUTS on the GPU box: the camera block, light projection / view
matrices, model matrices (and their inverse transposes) and vertex records (positions, normals, uvs) are overwritten with
NaN, +-inf, zeros, negative, denormal, huge values and raw random bit patterns; textures of odd shapes (1 x N, N x 1, 3 x 5).
G-buffer raster and shadow raster, GPU vs oracle bit for bit (NaN == NaN).
usage: python tests/sweeps/random_sweep_raster_fuzz.py FIRST_SEED LAST_SEED"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as ob
from syzygy_amd import abi, lib, meshes, pipelines as pl, scene
from syzygy_amd.pipelines import _memcpy2d_from
from tests import util
from tests.test_raster import _planes_equal, _soup

SPECIAL = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, -1.0, 1.0e-45, 1.0e-38, 3.0e38, -3.0e38, 1.0e-20, 1.0e20, 0.5, 2.0, 1.0e6]


def poison_words(rng, words, count):
    for _ in range(count):
        i = int(rng.integers(0, len(words)))
        mode = rng.random()
        if mode < 0.5:
            words[i] = SPECIAL[int(rng.integers(0, len(SPECIAL)))]
        elif mode < 0.7:
            words.view(np.uint32)[i] = int(rng.integers(0, 2 ** 32))
        else:
            words[i] = np.float32(rng.normal(0, 1) * 10.0 ** rng.uniform(-20, 20))


def floats_of(block):
    return np.frombuffer((C.c_char * C.sizeof(block)).from_buffer(block), np.float32)


no_directional = pl.TStagedBuffer(abi.DirectionalLightPacked, 1)
no_directional.recordCopyToDevice()
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    W, H, DIM = int(rng.integers(9, 120)), int(rng.integers(5, 80)), int(rng.choice([17, 64, 128]))
    inp = util.Inputs(W, H, spots=2)
    base = _soup(seed, int(rng.integers(2, 120)), spread=float(rng.uniform(5, 80)))[0]
    v, idx = base.vertices.copy(), base.indices.copy()
    shapes = [(1, int(rng.integers(1, 9))), (int(rng.integers(1, 9)), 1), (3, 5), (64, 64)]
    material = {k: (rng.integers(0, 256, shapes[int(rng.integers(0, 4))] + (4,), dtype=np.uint8), bool(rng.integers(0, 2)))
                for k in ("color", "normal", "orm") if rng.random() < 0.8}
    models = [meshes.transform_matrix((float(rng.normal(0, 5)), float(rng.normal(-5, 3)), float(rng.normal(5, 5))),
                                      tuple(float(a) for a in rng.uniform(-3, 3, 3)), tuple(float(s) for s in rng.uniform(0.2, 4, 3)))
              for _ in range(int(rng.integers(1, 4)))]
    where = rng.random()
    if where < 0.3:
        poison_words(rng, v.view(np.float32).reshape(-1), int(rng.integers(1, 12)))
    elif where < 0.55:
        poison_words(rng, floats_of(models[int(rng.integers(0, len(models)))]), int(rng.integers(1, 4)))
    elif where < 0.8:
        poison_words(rng, floats_of(inp.cam), int(rng.integers(1, 4)))
    mesh = meshes.MeshInstanced(v, idx, [(0, len(idx), material)], models)
    if 0.55 <= where < 0.6:  # and the inverse transposes, which the reference computes on the host (scene.cpp:210)
        poison_words(rng, mesh._mits_np.reshape(-1), 2)
    ms = [mesh] + (meshes.reference_default_scene() if rng.random() < 0.3 else [])
    spots = (abi.SpotLightPacked * 2)(inp.spots[0], inp.spots[1])
    if where >= 0.8:
        poison_words(rng, floats_of(spots[int(rng.integers(0, 2))]), int(rng.integers(1, 4)))
    bias = (0.0, 0.0) if rng.integers(0, 2) else (float(rng.uniform(-4, 4)), float(rng.uniform(-3, 3)))

    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    cameras.push(inp.cam)
    cameras.recordCopyToDevice()
    target = pl.SceneTexture(W, H)
    deferred = pl.DeferredShadingPipeline((W, H), max_spot_lights=2, max_shadow_maps=2, shadow_map_dim=DIM)
    deferred.setConfiguration(abi.DeferredConfiguration(bias[0], bias[1]))
    deferred.recordGBufferRaster(None, inp.rect, target, 0, cameras, ms)
    deferred.recordShadowRaster(None, no_directional, spots, ms)
    torch.cuda.synchronize()
    planes, depth = deferred.download_gbuffer(W, H), target.depth.cpu().numpy()
    want = ob.HostFrame(W, H)
    ob.gbuffer_raster(want, inp.rect, None, inp.cam, ms, threads=8)
    problems = []
    same_depth = (depth.view(np.uint32) == want.depth.view(np.uint32)) | (np.isnan(depth) & np.isnan(want.depth))
    if not same_depth.all():
        problems.append(f"depth {int((~same_depth).sum())}")
    try:
        _planes_equal(planes, want.planes())
    except AssertionError as e:
        problems.append(str(e))
    sm = deferred.shadowMaps()
    for slot in range(2):
        pv = abi.Mat4()
        lib().szg_mat4_mul(C.byref(spots[slot].projection), C.byref(spots[slot].view), C.byref(pv))
        want_map = ob.shadow_raster(pv, DIM, ms, bias[0], bias[1], threads=8)
        got_map = _memcpy2d_from(sm.maps[slot], DIM * 4, DIM).cpu().numpy().view(np.float32).reshape(DIM, DIM)
        same = (got_map.view(np.uint32) == want_map.view(np.uint32)) | (np.isnan(got_map) & np.isnan(want_map))
        if not same.all():
            problems.append(f"shadow map {slot}: {int((~same).sum())} texels")
    if problems:
        bad += 1
        print("seed", seed, "MISMATCH", "; ".join(problems), "W,H", W, H, "poisoned", "vertices" if where < 0.3 else "model" if where < 0.55 else
              "camera" if where < 0.8 else "light", flush=True)
    deferred.cleanup()
print("done, mismatching seeds:", bad, "processed up to", seed)
