"""Random parity sweep with HOSTILE PARAMETER BLOCKS on the GPU box: fields of the packed atmosphere, camera, directional- and
spot-light blocks are overwritten with NaN, +-inf, zeros, negative, denormal, huge and tiny values and raw random bit
patterns (planet radius 0, atmosphere inside the planet, zero or NaN sun direction, singular matrices, zero falloff...).
Every pass on the GPU vs the oracle on the same blocks: both LUTs, the lights pass and the final frame bit-identical including
the NaN pattern. Exercises the generic (non-lean) paths and every guard of the exact shortcuts.
usage: python tests/sweeps/random_sweep_params_fuzz.py FIRST_SEED LAST_SEED [extensions]   ("extensions": the opt-in LUTs too)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as ob
from syzygy_amd import abi, pipelines as pl, scene
from tests import util

SPECIAL = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, -1.0, 1.0e-45, 1.0e-38, 3.0e38, -3.0e38, 1.0e-20, 1.0e20, 6.36, 6.42, 0.5]


def as_floats(block):
    return np.frombuffer((C.c_char * C.sizeof(block)).from_buffer(block), np.float32)


def poison(rng, block, count, skip_words=()):
    words = as_floats(block)
    for _ in range(count):
        i = int(rng.integers(0, len(words)))
        if i in skip_words:
            continue
        mode = rng.random()
        if mode < 0.5:
            words[i] = SPECIAL[int(rng.integers(0, len(SPECIAL)))]
        elif mode < 0.7:
            words.view(np.uint32)[i] = int(rng.integers(0, 2 ** 32))
        elif mode < 0.85:
            words[i] = np.float32(words[i] * 10.0 ** rng.uniform(-6, 6) * (1 if rng.random() < 0.8 else -1))
        else:
            words[i] = np.float32(rng.normal(0, 1) * 10.0 ** rng.uniform(-30, 30))


bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 90)), int(rng.integers(9, 60))
    nsp = int(rng.integers(0, 5))
    inp = util.Inputs(W, H, elevation_degrees=float(rng.uniform(-8.0, 90.0)), spots=max(nsp, 1))
    fr = ob.HostFrame(W, H)
    ob.gbuffer_fill(fr, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)  # the G-buffer itself stays sane
    which = rng.random()
    if which < 0.5:
        poison(rng, inp.atm, int(rng.integers(1, 4)))
    elif which < 0.65:
        poison(rng, inp.cam, int(rng.integers(1, 4)))
    elif which < 0.8:
        poison(rng, inp.sun if rng.random() < 0.5 else inp.moon, int(rng.integers(1, 4)))
    elif nsp:
        poison(rng, inp.spots[int(rng.integers(0, nsp))], int(rng.integers(1, 4)))
    else:
        poison(rng, inp.atm, 2)
    dirs = (abi.DirectionalLightPacked * 2)(inp.sun, inp.moon)
    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    cameras.push(inp.cam)
    cameras.recordCopyToDevice()
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1)
    atmospheres.push(inp.atm)
    atmospheres.recordCopyToDevice()
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    lights.push([inp.sun, inp.moon])
    lights.recordCopyToDevice()
    nslots = 2 + nsp
    maps = {slot: rng.random((int(rng.integers(8, 40)), int(rng.integers(8, 40))), dtype=np.float32) for slot in range(nslots) if rng.random() < 0.3}
    target = pl.SceneTexture(W, H, debug=True)
    deferred = pl.DeferredShadingPipeline((W, H), max_spot_lights=max(nsp, 1), max_shadow_maps=nslots)
    keep, images = [], (abi.Image * nslots)()
    for slot, m in maps.items():
        images[slot] = ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT)
        t = torch.from_numpy(m).cuda()
        keep.append(t)
        deferred.setShadowMap(slot, t)
    host_maps = abi.ShadowMaps(nslots, 0, C.cast(images, C.POINTER(abi.Image)))
    tl, sl = (int(rng.integers(8, 64)), int(rng.integers(4, 20))), (int(rng.integers(8, 64)), int(rng.integers(4, 40)))
    sky = pl.SkyViewComputePipeline.create(transmittance_extent=tl, skyview_extent=sl)
    skip = int(rng.integers(0, 3))
    target.depth.copy_(torch.from_numpy(fr.depth))
    deferred.upload_gbuffer(fr.planes())
    deferred.recordLights(None, inp.rect, target, skip, lights, inp.spots if nsp else None, 0, cameras)
    torch.cuda.synchronize()
    got_lights = target.debug.cpu().numpy().copy()
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got, got_q = target.debug.cpu().numpy(), target.color_numpy()
    got_t, got_s = sky.download_lut(sky.transmittanceLUT()), sky.download_lut(sky.skyviewLUT())
    ob.lights(fr, inp.rect, None, host_maps, inp.cam, dirs, 2, skip, inp.spots, nsp, threads=8)
    want_lights = fr.debug.copy()
    tlut = ob.transmittance_lut(inp.atm, tl[0], tl[1], threads=8)
    slut = ob.skyview_lut(inp.atm, inp.cam, tlut, sl[0], sl[1], threads=8)
    ob.composite(fr, inp.rect, None, host_maps, inp.atm, inp.cam, dirs, 0, tlut, slut, threads=8)

    def differ(a, b):
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        return int((~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))).sum())

    report = {"transmittance": differ(got_t, tlut), "skyview": differ(got_s, slut), "lights": differ(got_lights, want_lights),
              "frame": differ(got, fr.debug), "rgba16": int((got_q != fr.color).sum())}
    if len(sys.argv) > 3 and sys.argv[3] == "extensions":
        # the opt-in LUTs without a reference counterpart (multi-scattering, aerial perspective) against their scalar oracles
        max_distance = float(10.0 ** rng.uniform(-3, 0))
        sky.recordMultiScatterLUT(None, 0, atmospheres)
        sky.recordAerialLUT(None, 0, atmospheres, 0, cameras, max_distance)
        torch.cuda.synchronize()
        lum_im, tr_im = sky.aerialLUT()
        want_ms, _ = ob.multiscatter_lut(inp.atm, tlut)
        want_lum, want_tr = ob.aerial_lut(inp.atm, inp.cam, tlut, max_distance, threads=8)
        report.update({"multiscatter": differ(sky.download_lut(sky.multiScatterLUT()), want_ms),
                       "aerial luminance": differ(sky.download_lut(lum_im), want_lum), "aerial transmittance": differ(sky.download_lut(tr_im), want_tr)})
    if any(report.values()):
        bad += 1
        print("seed", seed, "MISMATCH", report, "W,H", W, H, "luts", tl, sl, "spots", nsp, "skip", skip, "poisoned", "atm" if which < 0.5 else
              "cam" if which < 0.65 else "dir" if which < 0.8 else "spot/atm", flush=True)
    deferred.cleanup()
    sky.destroy()
print("done, mismatching seeds:", bad, "processed up to", seed)
