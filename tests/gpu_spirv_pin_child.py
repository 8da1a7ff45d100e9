"""Child process of tests/test_gpu_spirv_pin.py: renders the inputs of tests/golden/spirv_vectors.npz with whichever C-ABI
library SZG_HIP_LIBRARY names (the literal build of the kernels) and prints, as one JSON line, how many recorded values of the
reference's SPIR-V it does NOT reproduce bit for bit. Runs on the GPU box; reads nothing of the reference."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from syzygy_amd import abi, pipelines as pl  # noqa: E402
from syzygy_amd._lib import library_path  # noqa: E402


def staged(ctype, raw):
    raw = bytes(np.ascontiguousarray(raw, np.uint8))
    n = len(raw) // C.sizeof(ctype)
    buf = pl.TStagedBuffer(ctype, n)
    buf.push(list((ctype * n).from_buffer_copy(raw)))
    buf.recordCopyToDevice()
    return buf, n


def mismatches(got, want_bits):
    return int((np.ascontiguousarray(got, np.float32).view(np.uint32) != want_bits).sum())


def distance(stats, name, got, want_bits, got_q=None, want_q=None):
    """Largest / median relative distance of the fp32 values (against max(|want|, 1e-3)) and largest UNORM16 step."""
    want = want_bits.view(np.float32).astype(np.float64)
    rel = np.abs(np.asarray(got, np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    stats.setdefault(name + "_rel", []).extend(rel.ravel().tolist())
    if got_q is not None:
        step = int(np.abs(got_q.astype(np.int64) - want_q.astype(np.int64)).max())
        stats[name + "_unorm_max_step"] = max(stats.get(name + "_unorm_max_step", 0), step)


def main():
    vec = np.load(os.path.join(ROOT, "tests", "golden", "spirv_vectors.npz"))
    out = {"library": os.path.basename(library_path()), "values": 0}
    stats = {}

    # transmittance_LUT.comp: group the texels by atmosphere block
    W, H = (int(v) for v in vec["transmittance_extent"])
    bad = 0
    blocks = vec["transmittance_atm"]
    keys = [bytes(b) for b in blocks]
    for key in sorted(set(keys)):
        idx = [i for i, k in enumerate(keys) if k == key]
        atmospheres, _ = staged(abi.AtmospherePacked, np.frombuffer(key, np.uint8))
        sky = pl.SkyViewComputePipeline.create(transmittance_extent=(W, H), skyview_extent=(64, 32))
        sky.recordTransmittance(None, 0, atmospheres)
        torch.cuda.synchronize()
        lut = sky.download_lut(sky.transmittanceLUT())
        xy = vec["transmittance_xy"][idx]
        bad += mismatches(lut[xy[:, 1], xy[:, 0]], vec["transmittance_texel"][idx])
        distance(stats, "transmittance", lut[xy[:, 1], xy[:, 0]], vec["transmittance_texel"][idx])
        out["values"] += 4 * len(idx)
        sky.destroy()
    out["transmittance_mismatches"] = bad

    # skyview_LUT.comp: group by (atmosphere, camera); the transmittance LUT comes from the GPU as well
    W, H = (int(v) for v in vec["skyview_extent"])
    bad = 0
    keys = [bytes(a) + bytes(c) for a, c in zip(vec["skyview_atm"], vec["skyview_cam"])]
    for key in sorted(set(keys)):
        idx = [i for i, k in enumerate(keys) if k == key]
        atmospheres, _ = staged(abi.AtmospherePacked, vec["skyview_atm"][idx[0]])
        cameras, _ = staged(abi.CameraPacked, vec["skyview_cam"][idx[0]])
        sky = pl.SkyViewComputePipeline.create(transmittance_extent=(512, 128), skyview_extent=(W, H))
        sky.recordTransmittance(None, 0, atmospheres)
        sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
        torch.cuda.synchronize()
        lut = sky.download_lut(sky.skyviewLUT())
        xy = vec["skyview_xy"][idx]
        bad += mismatches(lut[xy[:, 1], xy[:, 0]], vec["skyview_texel"][idx])
        distance(stats, "skyview", lut[xy[:, 1], xy[:, 0]], vec["skyview_texel"][idx])
        out["values"] += 4 * len(idx)
        sky.destroy()
    out["skyview_mismatches"] = bad

    # lights.comp and camera.comp
    out["lights_mismatches"], out["lights_unorm_mismatches"] = 0, 0
    out["camera_mismatches"], out["camera_unorm_mismatches"] = 0, 0
    out["transmittance_lut_sha256_equal"] = True
    for k in range(int(vec["scenarios"])):
        depth = vec[f"depth_{k}"]
        Hf, Wf = depth.shape
        atmospheres, _ = staged(abi.AtmospherePacked, vec[f"atm_{k}"])
        cameras, _ = staged(abi.CameraPacked, vec[f"cam_{k}"])
        lights, ndirs = staged(abi.DirectionalLightPacked, vec[f"dirs_{k}"])
        raw = bytes(np.ascontiguousarray(vec[f"spots_{k}"], np.uint8))
        nspots = len(raw) // C.sizeof(abi.SpotLightPacked)
        spots = (abi.SpotLightPacked * nspots).from_buffer_copy(raw)
        target = pl.SceneTexture(Wf, Hf, debug=True)
        target.depth.copy_(torch.from_numpy(np.ascontiguousarray(depth)))
        deferred = pl.DeferredShadingPipeline((Wf, Hf), max_spot_lights=nspots, max_shadow_maps=ndirs + nspots)
        deferred.upload_gbuffer({n: vec[f"gbuffer_{n}_{k}"] for n in ("diffuse", "specular", "normal", "worldPosition",
                                                                     "occlusionRoughnessMetallic")})
        keep = []
        for s in range(ndirs + nspots):
            t = torch.from_numpy(np.ascontiguousarray(vec[f"shadow_{k}_{s}"])).cuda()
            keep.append(t)
            deferred.setShadowMap(s, t)
        rect = pl.rect(Wf, Hf)
        deferred.recordLights(None, rect, target, 1, lights, spots, 0, cameras)
        torch.cuda.synchronize()
        dbg, col = target.debug.cpu().numpy(), target.color_numpy()
        xy = vec[f"lights_xy_{k}"]
        out["lights_mismatches"] += mismatches(dbg[xy[:, 1], xy[:, 0]], vec[f"lights_value_{k}"])
        out["lights_unorm_mismatches"] += int((col[xy[:, 1], xy[:, 0]] != vec[f"lights_unorm_{k}"]).sum())
        distance(stats, "lights", dbg[xy[:, 1], xy[:, 0]], vec[f"lights_value_{k}"], col[xy[:, 1], xy[:, 0]], vec[f"lights_unorm_{k}"])
        out["values"] += 4 * len(xy)

        sky = pl.SkyViewComputePipeline.create(transmittance_extent=(512, 128), skyview_extent=(96, 48))
        sky.recordTransmittance(None, 0, atmospheres)
        torch.cuda.synchronize()
        tlut = sky.download_lut(sky.transmittanceLUT())
        if hashlib.sha256(np.ascontiguousarray(tlut).tobytes()).digest() != bytes(vec[f"tlut_sha256_{k}"]):
            out["transmittance_lut_sha256_equal"] = False
        sky.upload_lut(sky.skyviewLUT(), vec[f"slut_{k}"])
        sky.recordComposite(None, target, rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
        torch.cuda.synchronize()
        dbg, col = target.debug.cpu().numpy(), target.color_numpy()
        xy = vec[f"camera_xy_{k}"]
        out["camera_mismatches"] += mismatches(dbg[xy[:, 1], xy[:, 0]], vec[f"camera_value_{k}"])
        out["camera_unorm_mismatches"] += int((col[xy[:, 1], xy[:, 0]] != vec[f"camera_unorm_{k}"]).sum())
        distance(stats, "camera", dbg[xy[:, 1], xy[:, 0]], vec[f"camera_value_{k}"], col[xy[:, 1], xy[:, 0]], vec[f"camera_unorm_{k}"])
        out["values"] += 4 * len(xy)
        deferred.cleanup()
        sky.destroy()
    for name, v in stats.items():
        if name.endswith("_rel"):
            out[name + "_median"], out[name + "_max"] = float(np.median(v)), float(np.max(v))
        else:
            out[name] = v
    print(json.dumps(out))


if __name__ == "__main__":
    main()
