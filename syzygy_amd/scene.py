"""Host input prep: scene parameters -> packed blocks, through the C host
functions of include/szg/host.h (reference: renderer/scene.cpp:52-91, :689-794,
renderer/lights.cpp:9-46, geometry/geometryhelpers.cpp:83-204), plus the seeded
synthetic scene the tests and bench.py use.
"""
import ctypes as C
import math

import numpy as np

from . import abi
from ._lib import lib

SEED = 0x5A2C


def _hash32(*words):
    """Counter-based hash (PCG-style output permutation) of integer words."""
    h = 0x9E3779B9
    for w in words:
        h = (h ^ (int(w) & 0xFFFFFFFF)) * 0x85EBCA6B & 0xFFFFFFFF
        h ^= h >> 13
        h = h * 0xC2B2AE35 & 0xFFFFFFFF
        h ^= h >> 16
    return h


def hash_unit(*words):
    """Deterministic float in [0, 1)."""
    return (_hash32(*words) >> 8) / float(1 << 24)


def default_atmosphere(sun_euler=None):
    """scene.cpp:52-75 Earth defaults; optional (pitch, roll, yaw) for the sun."""
    a = abi.Atmosphere()
    lib().szg_atmosphere_default_earth(C.byref(a))
    if sun_euler is not None:
        a.sunEulerAngles[:] = [float(v) for v in sun_euler]
    return a


def sun_euler_for_elevation(elevation_degrees):
    """sunEulerAngles = (pi + elevation, 0, 0); daytime is pitch in (pi, 2pi)."""
    return (math.pi + math.radians(elevation_degrees), 0.0, 0.0)


def atmosphere_packed(atmosphere):
    out = abi.AtmospherePacked()
    lib().szg_atmosphere_to_device_equivalent(C.byref(atmosphere), C.byref(out))
    return out


def atmosphere_baked(atmosphere, bounds):
    """scene.cpp:718-737 -> (AtmospherePacked, sunlight, moonlight); lights index 0 = sun, 1 = moon."""
    atm = abi.AtmospherePacked()
    sun = abi.DirectionalLightPacked()
    moon = abi.DirectionalLightPacked()
    lib().szg_atmosphere_baked(C.byref(atmosphere), C.byref(bounds), C.byref(atm), C.byref(sun), C.byref(moon))
    return atm, sun, moon


def default_camera():
    c = abi.Camera()
    lib().szg_camera_default(C.byref(c))
    return c


def camera_packed(camera, aspect):
    out = abi.CameraPacked()
    lib().szg_camera_to_device_equivalent(C.byref(camera), C.c_float(aspect), C.byref(out))
    return out


def aabb(center, half_extent):
    b = abi.AABB()
    b.center[:] = [float(v) for v in center]
    b.half_extent[:] = [float(v) for v in half_extent]
    return b


def eulers_from_forward(forward):
    out = (C.c_float * 3)()
    lib().szg_eulers_from_forward(abi.f3(*forward), out)
    return tuple(out)


def forward_from_eulers(eulers):
    out = (C.c_float * 3)()
    lib().szg_forward_from_eulers(abi.f3(*eulers), out)
    return tuple(out)


def make_spot(color_rgb, position, eulers, **overrides):
    """scene.cpp:218-229 addSpotlight defaults -> lights.cpp:29-46 makeSpot."""
    params = abi.SpotlightParams()
    lib().szg_spotlight_params_default(abi.f3(*color_rgb), abi.f3(*position), abi.f3(*eulers), C.byref(params))
    for k, v in overrides.items():
        setattr(params, k, v)
    out = abi.SpotLightPacked()
    lib().szg_make_spot(C.byref(params), C.byref(out))
    return out


def spot_ring(count, seed=SEED, radius=38.0, height=-24.0, target_spread=30.0):
    """`count` spot lights on rings above the scene aiming down at it:
    params from scene.cpp:218-229, hashed colours and aim points. World +y is down,
    so `height` is negative (above the ground plane at y = -1)."""
    lights = (abi.SpotLightPacked * max(count, 1))()
    for i in range(count):
        ring = i // 32
        ang = 2.0 * math.pi * ((i % 32) + 0.37 * ring) / 32.0
        r = radius * (1.0 + 0.45 * ring)
        pos = (r * math.cos(ang), height - 4.0 * ring, r * math.sin(ang) + 20.0)
        tx = (hash_unit(seed, i, 1) - 0.5) * 2.0 * target_spread
        tz = (hash_unit(seed, i, 2) - 0.5) * 2.0 * target_spread + 20.0
        target = (tx, -1.0, tz)
        fwd = tuple(t - p for t, p in zip(target, pos))
        n = math.sqrt(sum(f * f for f in fwd))
        fwd = tuple(f / n for f in fwd)
        eul = eulers_from_forward(fwd)
        col = (0.25 + 0.75 * hash_unit(seed, i, 3), 0.25 + 0.75 * hash_unit(seed, i, 4), 0.25 + 0.75 * hash_unit(seed, i, 5))
        lights[i] = make_spot(col, pos, eul)
    return lights


class SyntheticScene:
    """Ground plane at y = -1 (editor.cpp:541-545) + a grid of cubes of half-size 5
    floating at y = -8 (editor.cpp:510-539), alternating metallic. Keeps the ctypes
    box array alive for as long as the FillScene is used."""

    def __init__(self, grid=(6, 4), spacing=18.0, seed=SEED):
        n = grid[0] * grid[1]
        self.boxes = (abi.FillBox * n)()
        k = 0
        for gz in range(grid[1]):
            for gx in range(grid[0]):
                b = self.boxes[k]
                b.center[:] = [(gx - (grid[0] - 1) / 2.0) * spacing, -8.0, 12.0 + gz * spacing]
                b.half_extent[:] = [5.0, 5.0, 5.0]
                b.metallic = 1.0 if (gx + gz) % 2 == 0 else 0.0
                b.roughness = 60.0 / 255.0
                k += 1
        self.fill = abi.FillScene()
        self.fill.ground_y = -1.0
        self.fill.ground_half_extent = 4000.0
        self.fill.checker_cell = 4.0
        self.fill.ground_roughness = 60.0 / 255.0
        self.fill.box_count = n
        self.fill.boxes = C.cast(self.boxes, C.POINTER(abi.FillBox))
        # bounds of what casts shadows, for the sun's orthographic projection
        self.bounds = aabb((0.0, -7.0, 12.0 + (grid[1] - 1) * spacing / 2.0),
                           (grid[0] * spacing / 2.0 + 10.0, 8.0, grid[1] * spacing / 2.0 + 10.0))


def struct_bytes(obj):
    return np.frombuffer(bytes(obj), dtype=np.uint8).copy()
