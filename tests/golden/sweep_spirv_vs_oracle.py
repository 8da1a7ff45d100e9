#!/usr/bin/env python3
"""Build-container tool: random frames through BOTH the interpreter (the reference's committed SPIR-V, executed literally) and
the literal oracle, every pixel of small frames, lights.comp and camera.comp (+ a few LUT texels per seed). Reports every
value that differs. Nothing is written; tests/golden/make_spirv_vectors.py records the fixed vectors the tests use.

    python tests/golden/sweep_spirv_vs_oracle.py FIRST LAST [W H]
"""
import ctypes as C
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
F32 = np.float32


def main():
    from oracle import binding as ob
    from syzygy_amd import abi, scene
    from tests import util
    from tests.golden import make_spirv_vectors as gen, spirv_interp as si

    first, last = int(sys.argv[1]), int(sys.argv[2])
    Wf, Hf = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (32, 18)
    Store, Unorm16, Nearest, Border, Linear = gen._images()
    mods = {k: si.Module(os.path.join(gen.REFERENCE, p)) for k, p in gen.SHADERS.items() if k in ("transmittance", "skyview", "lights", "camera")}
    builtins = gen.Builtins()

    def bindings(m):
        return {m.names[g]: (m.decor[g][si.DEC_SET][0], m.decor[g][si.DEC_BINDING][0]) for g, (pt, sc) in m.globals.items()
                if sc == si.SC_UNIFORM_CONSTANT}

    bad_seeds = 0
    t0 = time.time()
    for seed in range(first, last):
        rng = np.random.default_rng(0xC0FFEE + seed)
        camera = scene.default_camera()
        camera.cameraPosition[0] += float(rng.uniform(-30, 30))
        camera.cameraPosition[2] += float(rng.uniform(-30, 30))
        camera.cameraPosition[1] = -float(10.0 ** rng.uniform(-0.5, 3.3))  # 0.3 m ... 2 km up (+y is down)
        camera.eulerAngles[0] = float(rng.uniform(-1.3, 1.3))
        camera.eulerAngles[1] = float(rng.uniform(-3.1, 3.1))
        edit = None
        if seed % 3 == 1:
            def edit(a, rng=rng):
                a.absorptionRayleighPerMegameter[:] = [float(x) for x in rng.uniform(0, 3, 3)]
                a.scatteringOzonePerMegameter[:] = [float(x) for x in rng.uniform(0, 1, 3)]
        spots = int(rng.integers(0, 9))
        inp = util.Inputs(Wf, Hf, elevation_degrees=float(rng.uniform(-12, 89)), spots=max(spots, 1), camera=camera, atmosphere_edit=edit)
        nslots = 2 + inp.spot_count
        maps = [rng.random((int(rng.integers(4, 24)), int(rng.integers(4, 24))), dtype=np.float32) for _ in range(nslots)]
        mismatches = []
        with ob.use_literal():
            tlut = ob.transmittance_lut(inp.atm, 512, 128, threads=2)
            slut = ob.skyview_lut(inp.atm, inp.cam, tlut, 64, 32, threads=2)
            frame = ob.HostFrame(Wf, Hf)
            ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=2)
            images = (abi.Image * nslots)()
            for s in range(nslots):
                images[s] = ob.host_image(maps[s], abi.SZG_FORMAT_D32_SFLOAT)
            sm = abi.ShadowMaps(nslots, 0, C.cast(images, C.POINTER(abi.Image)))
            mem = si.Memory()
            a_atm, a_cam, a_dir, a_spot = mem.alloc(bytes(inp.atm)), mem.alloc(bytes(inp.cam)), mem.alloc(bytes(inp.dirs)), mem.alloc(bytes(inp.spots))
            planes = frame.planes()
            gb = {"gbufferDiffuse": Nearest(planes["diffuse"]), "gbufferSpecular": Nearest(planes["specular"]),
                  "gbufferNormal": Nearest(planes["normal"]), "gbufferWorldPosition": Nearest(planes["worldPosition"]),
                  "gbufferOcclusionRoughnessMetallic": Nearest(planes["occlusionRoughnessMetallic"])}
            shadows = [Border(a) for a in maps]
            ml, mc = mods["lights"], mods["camera"]
            bl, bc = bindings(ml), bindings(mc)
            image = Unorm16(np.zeros((Hf, Wf, 4), np.uint16))
            desc = {bl[n]: v for n, v in gb.items()}
            desc.update({bl["image"]: image, bl["shadowMaps"]: shadows, bl["shadowMapSampler"]: "s"})
            it = si.Interpreter(ml, mem, builtins, struct.pack("<QIIQQIIIIffff", a_cam, 0, 0, a_dir, a_spot, 2, inp.spot_count, 1, 0, 0.0, 0.0,
                                                               float(Wf), float(Hf)), desc)
            ob.lights(frame, inp.rect, None, sm, inp.cam, inp.dirs, 2, 1, inp.spots, inp.spot_count, threads=2)
            for y in range(Hf):
                for x in range(Wf):
                    it.run((x, y, 0))
                    want = image.written_f.get((x, y), [F32(0), F32(0), F32(0), F32(1)])
                    if not np.array_equal(np.array(want, np.float32).view(np.uint32), frame.debug[y, x].view(np.uint32)):
                        mismatches.append(("lights", x, y, [float(c) for c in want], frame.debug[y, x].tolist()))
            image = Unorm16(frame.color.copy())
            desc = {bc[n]: v for n, v in gb.items()}
            desc.update({bc["image"]: image, bc["fragmentDepth"]: Nearest(frame.depth), bc["skyview_LUT"]: Linear(slut),
                         bc["transmittance_LUT"]: Linear(tlut), bc["shadowMaps"]: shadows, bc["shadowMapSampler"]: "s"})
            it = si.Interpreter(mc, mem, builtins, struct.pack("<QQIIIIIIIIQII", a_atm, a_cam, 0, 0, Wf, Hf, 0, 0, Wf, Hf, a_dir, 0, 0), desc)
            ob.composite(frame, inp.rect, None, sm, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=2)
            for y in range(Hf):
                for x in range(Wf):
                    try:
                        it.run((x, y, 0))
                    except RuntimeError as e:  # a non-finite texture coordinate: outside what the sampler model defines
                        continue
                    want = image.written_f[(x, y)]
                    if not np.array_equal(np.array(want, np.float32).view(np.uint32), frame.debug[y, x].view(np.uint32)):
                        mismatches.append(("camera", x, y, [float(c) for c in want], frame.debug[y, x].tolist()))
            # a few texels of both LUT shaders
            mt, ms = mods["transmittance"], mods["skyview"]
            st = Store(512, 128)
            it = si.Interpreter(mt, mem, builtins, struct.pack("<QII", a_atm, 0, 0), {bindings(mt)["transmittance_LUT"]: st})
            for _ in range(4):
                x, y = int(rng.integers(512)), int(rng.integers(128))
                it.run((x, y, 0))
                if not np.array_equal(np.array(st.texels[(x, y)], np.float32).view(np.uint32), tlut[y, x].view(np.uint32)):
                    mismatches.append(("transmittance", x, y, [float(c) for c in st.texels[(x, y)]], tlut[y, x].tolist()))
            ss = Store(2048, 1024)
            b = bindings(ms)
            it = si.Interpreter(ms, mem, builtins, struct.pack("<QQIIII", a_atm, a_cam, 0, 0, 0, 0), {b["skyview_LUT"]: ss, b["transmittance_LUT"]: Linear(tlut)})
            for _ in range(6):
                x, y = int(rng.integers(2048)), int(rng.integers(1024))
                it.run((x, y, 0))
                row = ob.skyview_lut(inp.atm, inp.cam, tlut, 2048, 1024, row_begin=y, row_end=y + 1, threads=2)
                if not np.array_equal(np.array(ss.texels[(x, y)], np.float32).view(np.uint32), row[y, x].view(np.uint32)):
                    mismatches.append(("skyview", x, y, [float(c) for c in ss.texels[(x, y)]], row[y, x].tolist()))
        if mismatches:
            bad_seeds += 1
            for m in mismatches[:6]:
                print(f"seed {seed} MISMATCH {m}", flush=True)
            print(f"seed {seed}: {len(mismatches)} mismatching invocations", flush=True)
        if (seed - first) % 5 == 4:
            print(f"... seed {seed} done, {time.time() - t0:.0f} s, mismatching seeds so far {bad_seeds}", flush=True)
    print(f"done, mismatching seeds: {bad_seeds} of {last - first}")


if __name__ == "__main__":
    main()
