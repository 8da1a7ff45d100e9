#!/usr/bin/env python3
"""An engine-style frame loop over the MI355X path, in the shape of Renderer::recordDraw (renderer.cpp:278-443):

    scene tick (sun animation, mesh-instance animation)            scene.cpp:461-574
    shadow bounds -> baked atmosphere + sun / moon lights           scene.cpp:95-148, :718-737
    staged buffers (cameras, atmospheres, directional lights)       renderer.cpp:302-342
    DeferredShadingPipeline::recordDrawCommands(meshes)             shadow raster, G-buffer raster, lights
    SkyViewComputePipeline::recordDrawCommands                      transmittance LUT, sky-view LUT, composite
    OETF on the presented image                                     editor.cpp:303-340

    python examples/frame_loop.py --frames 60 --width 1920 --height 1080 --out /tmp/frame.ppm

Needs an MI355X (no CPU fallback). Everything on the GPU is enqueued on one stream; the host only ticks the scene.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--shadow-map", type=int, default=2048)
    ap.add_argument("--out", default="", help="write the last frame as a binary PPM (8 bit)")
    args = ap.parse_args()

    import torch

    from syzygy_amd import abi, lib, meshes, pipelines as pl, scene

    W, H = args.width, args.height
    # ---- scene: the editor's start-up scene, its cubes animated (editor.cpp:500-545) -----------------------------
    material = meshes.default_material()
    cv, ci = meshes.cube_mesh()
    pv, pi = meshes.plane_mesh()
    cube_bounds, plane_bounds = abi.AABB(), abi.AABB()
    lib().szg_aabb_create(abi.f3(*cv["position"].min(0)), abi.f3(*cv["position"].max(0)), C.byref(cube_bounds))
    lib().szg_aabb_create(abi.f3(*pv["position"].min(0)), abi.f3(*pv["position"].max(0)), C.byref(plane_bounds))

    def instance(vertices, indices, bounds, animation, items):
        n = len(items)
        originals = (abi.Transform * n)()
        for t, (tr, sc) in zip(originals, items):
            t.translation[:], t.eulerAnglesRadians[:], t.scale[:] = list(tr), [0.0, 0.0, 0.0], list(sc)
        return {"vertices": vertices, "indices": indices, "bounds": bounds, "animation": animation, "originals": originals,
                "transforms": (abi.Transform * n)(*originals), "models": (abi.Mat4 * n)(), "mits": (abi.Mat4 * n)(), "n": n}

    instances = [
        instance(cv, ci, cube_bounds, abi.SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP, [((0, -8, 6), (5, 5, 5))]),
        instance(cv, ci, cube_bounds, abi.SZG_INSTANCE_ANIMATION_DIAGONAL_WAVE, [((0, -8, -6), (5, 5, 5)), ((14, -6, -2), (2, 2, 2))]),
        instance(pv, pi, plane_bounds, abi.SZG_INSTANCE_ANIMATION_NONE, [((0, -1, 0), (20, 1, 20))]),
    ]
    atmosphere = scene.default_atmosphere()
    sun_animation = abi.SunAnimation()
    lib().szg_sun_animation_default(C.byref(sun_animation))
    sun_animation.time = 0.62  # afternoon, the sun behind the camera (scene.cpp:546-574)
    camera = scene.default_camera()
    # the default camera sits 2 m in front of a cube face (scene.cpp:77-83): step back and look at the scene centre
    camera.cameraPosition[:] = [-22.0, -18.0, -42.0]
    camera.eulerAngles[:] = [float(v) for v in scene.eulers_from_forward((22.0, 11.0, 42.0))]
    spots = (abi.SpotLightPacked * 2)(scene.make_spot((1, 0.2, 0.1), (-20.0, -28.0, -20.0), scene.eulers_from_forward((20.0, 20.0, 20.0))),
                                      scene.make_spot((0.1, 0.3, 1), (20.0, -28.0, -20.0), scene.eulers_from_forward((-20.0, 20.0, 20.0))))

    # ---- GPU objects ----------------------------------------------------------------------------------------------
    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1)
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    target = pl.SceneTexture(W, H)
    deferred = pl.DeferredShadingPipeline((W, H), max_spot_lights=len(spots), max_shadow_maps=2 + len(spots), shadow_map_dim=args.shadow_map)
    sky = pl.SkyViewComputePipeline.create()
    rect = pl.rect(W, H)

    t_start = time.perf_counter()
    elapsed, dt = 0.0, 1.0 / 60.0
    for frame in range(args.frames):
        # Scene::tick
        lib().szg_scene_tick_sun(C.byref(sun_animation), C.byref(atmosphere), dt * 50.0)  # 100x speed (scene.cpp:89): a slow sunset over a few hundred frames
        scene_meshes, casters = [], []
        for inst in instances:
            lib().szg_tick_mesh_instance(inst["animation"], inst["originals"], inst["transforms"], inst["n"], elapsed, dt, inst["models"],
                                         inst["mits"])
            scene_meshes.append(meshes.MeshInstanced(inst["vertices"], inst["indices"], [(0, len(inst["indices"]), material)],
                                                     list(inst["models"])))
            casters.append(abi.ShadowCaster(inst["bounds"], inst["transforms"], inst["n"], 1, 1, 0))
        bounds = abi.AABB()
        lib().szg_calculate_shadow_bounds((abi.ShadowCaster * len(casters))(*casters), len(casters), C.byref(bounds))
        atm, sun, moon = scene.atmosphere_baked(atmosphere, bounds)

        # Renderer::recordDraw
        for buf, items in ((cameras, [scene.camera_packed(camera, W / H)]), (atmospheres, [atm]), (lights, [sun, moon])):
            buf.clearStaged()
            buf.push(items)
            buf.recordCopyToDevice()
        deferred.recordDrawCommandsMeshes(None, rect, target, 1, lights, spots, 0, cameras, scene_meshes)
        sky.recordDrawCommands(None, target, rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
        pl.recordOETF(None, target, W, H)
        elapsed += dt
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_start
    image = target.color_numpy()
    print(f"{args.frames} frames of {W}x{H}: {wall / args.frames * 1e3:.2f} ms per frame including host scene prep; "
          f"mean display value {image[..., :3].mean() / 65535.0:.3f}")
    if args.out:
        with open(args.out, "wb") as f:
            f.write(f"P6 {W} {H} 255\n".encode())
            f.write((image[..., :3] >> 8).astype(np.uint8).tobytes())
        print("wrote", args.out)
    deferred.cleanup()
    sky.destroy()
    return image


if __name__ == "__main__":
    main()
