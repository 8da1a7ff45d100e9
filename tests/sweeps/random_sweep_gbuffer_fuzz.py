"""Random parity sweep with HOSTILE G-buffers on the GPU box: the analytic fill is overwritten, in patches and single
pixels, with NaN, +-inf, zeros, negative values, denormals, huge and out-of-range values and raw random bit patterns in every
plane (diffuse incl. its alpha flag, specular incl. black = SURVEY Q6, normal, world position incl. underground and
planet-scale values, ORM) and in the depth buffer (geometry flagged as sky and the reverse). lights + atmosphere on the GPU
vs the oracle on the same planes: the fp32 frame bit-identical including the NaN pattern, the RGBA16 image equal.
usage: python tests/sweeps/random_sweep_gbuffer_fuzz.py FIRST_SEED LAST_SEED"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as ob
from syzygy_amd import abi, pipelines as pl, scene
from tests import util
from tests.test_gpu_parity import staged


class G:
    pass


g = G()
g.ob, g.abi, g.pl, g.scene = ob, abi, pl, scene
SPECIAL = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, -1.0, 6.0e-8, 65504.0, -65504.0, 0.5, 2.0, 1.0e-3]
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 120)), int(rng.integers(9, 70))
    nsp = int(rng.integers(0, 6))
    inp = util.Inputs(W, H, elevation_degrees=float(rng.uniform(-8.0, 90.0)), spots=nsp)
    cameras, atmospheres, lights = staged(g, inp)
    fr = ob.HostFrame(W, H)
    ob.gbuffer_fill(fr, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
    planes = fr.planes()
    names = list(planes)
    for _ in range(int(rng.integers(1, 30))):
        plane = planes[names[int(rng.integers(0, len(names)))]]
        y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
        h, w = (1, 1) if rng.random() < 0.4 else (int(rng.integers(1, 12)), int(rng.integers(1, 12)))
        region = plane[y0 : y0 + h, x0 : x0 + w]
        channels = slice(0, 4) if rng.random() < 0.3 else slice(int(rng.integers(0, 4)), None, 5)
        mode = rng.random()
        if mode < 0.6:
            region[..., channels] = SPECIAL[int(rng.integers(0, len(SPECIAL)))]
        elif mode < 0.8:
            bits = np.uint16 if plane.dtype == np.float16 else np.uint32
            raw = rng.integers(0, np.iinfo(bits).max, region[..., channels].shape, dtype=np.uint64).astype(bits)
            region[..., channels] = raw.view(plane.dtype)
        else:
            region[..., channels] = (rng.normal(0, 1, region[..., channels].shape) * 10.0 ** rng.uniform(-3, 7)).astype(plane.dtype)
    for _ in range(int(rng.integers(0, 6))):
        y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
        h, w = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        fr.depth[y0 : y0 + h, x0 : x0 + w] = [0.0, 0.5, 1.0e-30, np.nan, -1.0, np.inf][int(rng.integers(0, 6))]
    nslots = 2 + nsp
    maps = {slot: rng.random((int(rng.integers(8, 60)), int(rng.integers(8, 60))), dtype=np.float32) for slot in range(nslots) if rng.random() < 0.3}
    target = pl.SceneTexture(W, H, debug=True)
    deferred = pl.DeferredShadingPipeline((W, H), max_spot_lights=max(nsp, 1), max_shadow_maps=nslots)
    keep, images = [], (abi.Image * nslots)()
    for slot, m in maps.items():
        images[slot] = ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT)
        t = torch.from_numpy(m).cuda()
        keep.append(t)
        deferred.setShadowMap(slot, t)
    host_maps = abi.ShadowMaps(nslots, 0, C.cast(images, C.POINTER(abi.Image)))
    sky = pl.SkyViewComputePipeline.create(transmittance_extent=(64, 16), skyview_extent=(64, 32))
    skip = int(rng.integers(0, 3))
    target.depth.copy_(torch.from_numpy(fr.depth))
    deferred.upload_gbuffer(planes)
    deferred.recordLights(None, inp.rect, target, skip, lights, inp.spots if nsp else None, 0, cameras)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got, got_q = target.debug.cpu().numpy(), target.color_numpy()
    ob.lights(fr, inp.rect, None, host_maps, inp.cam, inp.dirs, 2, skip, inp.spots, nsp, threads=8)
    tlut = ob.transmittance_lut(inp.atm, 64, 16, threads=8)
    slut = ob.skyview_lut(inp.atm, inp.cam, tlut, 64, 32, threads=8)
    ob.composite(fr, inp.rect, None, host_maps, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    same = (got.view(np.uint32) == fr.debug.view(np.uint32)) | (np.isnan(got) & np.isnan(fr.debug))
    if not same.all() or not (got_q == fr.color).all():
        bad += 1
        ys, xs = np.nonzero(~same.all(-1))
        first = (int(ys[0]), int(xs[0])) if len(ys) else None
        print("seed", seed, "MISMATCH", (~same).sum(), "fp32 values; nan gpu/oracle", int(np.isnan(got).sum()), int(np.isnan(fr.debug).sum()),
              "quantised differ", int((got_q != fr.color).sum()), "W,H", W, H, "spots", nsp, "skip", skip, "first", first, flush=True)
    deferred.cleanup()
    sky.destroy()
print("done, mismatching seeds:", bad, "processed up to", seed)
