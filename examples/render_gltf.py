#!/usr/bin/env python3
"""Load a glTF 2.0 / GLB file the way AssetLibrary::loadGLTFFromPath does (include/szg/assets.h) and render it through the
whole path: shadow raster, G-buffer raster, lights, transmittance + sky-view LUTs, composite, OETF.

    python examples/render_gltf.py [model.glb | model.gltf] --out /tmp/gltf.ppm [--embedded-images]

Without a file a textured sphere is written first (the reference's own assets/sphere.glb is a git-LFS pointer in this
checkout). Meshes are scaled to 8 m and set on the editor's floor plane. Needs an MI355X (no CPU fallback).
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def write_sphere(path):
    from tests import gltf_writer as gw  # the test suite's independent encoder

    pos, nrm, uv, idx = gw.uv_sphere(32, 64)
    b = gw.GltfBuilder()
    y, x = np.mgrid[0:128, 0:256]
    stripes = ((x // 16 + y // 16) % 2).astype(np.uint8)
    color = np.stack([80 + 150 * stripes, 60 + 40 * stripes, 200 - 150 * stripes, np.full_like(stripes, 255)], -1).astype(np.uint8)
    bump = np.stack([127 + 50 * np.sin(x / 2.0), 127 + 50 * np.sin(y / 2.0), np.full(x.shape, 235.0), np.zeros(x.shape)], -1).astype(np.uint8)
    orm = np.zeros((2, 2, 4), np.uint8)
    orm[..., 1], orm[..., 2] = 90, 255  # glossy metal
    textures = [b.texture(b.image_uri(gw.data_uri_png(gw.png_rgba8(t)))) for t in (color, bump, orm)]
    b.doc["materials"] = [{"name": "striped", "pbrMetallicRoughness": {"baseColorTexture": {"index": textures[0]},
                                                                       "metallicRoughnessTexture": {"index": textures[2]}},
                           "normalTexture": {"index": textures[1]}}]
    b.doc["meshes"] = [{"name": "Sphere", "primitives": [{"attributes": {"POSITION": b.accessor(pos), "NORMAL": b.accessor(nrm),
                                                                        "TEXCOORD_0": b.accessor(uv)},
                                                          "indices": b.accessor(idx.astype(np.uint16)), "material": 0}]}]
    with open(path, "wb") as f:
        f.write(b.glb())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?", default="")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--out", default="")
    ap.add_argument("--embedded-images", action="store_true", help="also decode images stored in GLB buffer views (the reference does not)")
    args = ap.parse_args()

    import torch

    from syzygy_amd import abi, assets, lib, meshes, pipelines as pl, scene

    path = args.path
    if not path:
        path = os.path.join(tempfile.mkdtemp(), "sphere.glb")
        write_sphere(path)
    asset = assets.load_gltf(path, abi.SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES if args.embedded_images else 0)
    for line in asset.warnings:
        print("[warning]", line)
    print(f"{path}: {len(asset.meshes)} meshes, {len(asset.materials)} materials")
    if not asset.meshes:
        raise SystemExit("nothing to render")

    # every mesh scaled to 8 m and set on the floor (y = -1, +y down), side by side
    scene_meshes, casters, keep = [], [], []
    for k, m in enumerate(asset.meshes):
        centre, half = m.bounds
        s = 4.0 / max(float(half.max()), 1e-6)
        x = (k - (len(asset.meshes) - 1) / 2.0) * 10.0
        tr = abi.Transform()
        tr.translation[:] = [x - s * float(centre[0]), -1.0 - s * float(centre[1] + half[1]), -s * float(centre[2])]
        tr.eulerAnglesRadians[:] = [0.0, 0.0, 0.0]
        tr.scale[:] = [s, s, s]
        model = meshes.transform_matrix(tuple(tr.translation), (0, 0, 0), (s, s, s))
        scene_meshes.append(asset.instanced(k, [model]))
        bounds = abi.AABB()
        bounds.center[:], bounds.half_extent[:] = [float(v) for v in centre], [float(v) for v in half]
        transforms = (abi.Transform * 1)(tr)
        keep.append(transforms)
        casters.append(abi.ShadowCaster(bounds, transforms, 1, 1, 1, 0))
        print(f"  {m.name}: {len(m.vertices)} vertices, {len(m.indices) // 3} triangles, {len(m.surfaces)} surfaces")
    floor = meshes.reference_default_scene()[2]
    scene_meshes.append(floor)

    W, H = args.width, args.height
    atmosphere = scene.default_atmosphere()
    sun_animation = abi.SunAnimation()
    lib().szg_sun_animation_default(C.byref(sun_animation))
    sun_animation.time = 0.62
    lib().szg_scene_tick_sun(C.byref(sun_animation), C.byref(atmosphere), 0.0)
    camera = scene.default_camera()
    camera.cameraPosition[:] = [-9.0, -9.0, -16.0]
    camera.eulerAngles[:] = [float(v) for v in scene.eulers_from_forward((9.0, 4.0, 16.0))]
    shadow_bounds = abi.AABB()
    lib().szg_calculate_shadow_bounds((abi.ShadowCaster * len(casters))(*casters), len(casters), C.byref(shadow_bounds))
    atm, sun, moon = scene.atmosphere_baked(atmosphere, shadow_bounds)
    spots = (abi.SpotLightPacked * 1)(scene.make_spot((1.0, 0.9, 0.7), (10.0, -14.0, -10.0), scene.eulers_from_forward((-10.0, 10.0, 10.0))))

    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1)
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    for buf, items in ((cameras, [scene.camera_packed(camera, W / H)]), (atmospheres, [atm]), (lights, [sun, moon])):
        buf.push(items)
        buf.recordCopyToDevice()
    target = pl.SceneTexture(W, H)
    deferred = pl.DeferredShadingPipeline((W, H), max_spot_lights=1, max_shadow_maps=3, shadow_map_dim=2048)
    sky = pl.SkyViewComputePipeline.create()
    rect = pl.rect(W, H)
    deferred.recordDrawCommandsMeshes(None, rect, target, 1, lights, spots, 0, cameras, scene_meshes)
    sky.recordDrawCommands(None, target, rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    pl.recordOETF(None, target, W, H)
    torch.cuda.synchronize()
    image = target.color_numpy()
    covered = float((target.depth.cpu().numpy() > 0).mean())
    print(f"rendered {W}x{H}: geometry covers {covered:.1%} of the frame, mean display value {image[..., :3].mean() / 65535.0:.3f}")
    if args.out:
        with open(args.out, "wb") as f:
            f.write(f"P6 {W} {H} 255\n".encode())
            f.write((image[..., :3] >> 8).astype(np.uint8).tobytes())
        print("wrote", args.out)
    deferred.cleanup()
    sky.destroy()
    return image, covered


if __name__ == "__main__":
    main()
