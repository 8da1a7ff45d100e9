"""Random sweep of LUT reuse (abi.h szg_skyview_set_lut_reuse) on the GPU box: ONE long-lived pipeline with reuse enabled is
driven through a random walk of frames - each frame keeps the previous parameter blocks, or changes the camera position only,
or one dword of the atmosphere block (sun direction, a coefficient, a NaN), or scribbles over a LUT through a kept pointer and
says so - and after every frame both of its LUTs must equal, bit for bit, the LUTs a FRESH pipeline computes from the same
blocks. The staged buffers are shared and never synchronised between frames (frames in flight).
usage: python tests/sweeps/random_sweep_lut_reuse.py FIRST_SEED LAST_SEED"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from tests import util
from syzygy_amd import abi, pipelines as pl, scene

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    tl = (int(rng.integers(8, 96)), int(rng.integers(4, 40)))
    sl = (int(rng.integers(8, 96)), int(rng.integers(4, 64)))
    sky = pl.SkyViewComputePipeline.create(transmittance_extent=tl, skyview_extent=sl)
    sky.setLUTReuse(True)
    alias = sky.skyviewLUT_tensor()
    # the LUT images are fetched ONCE: every accessor call tells the pipeline that the caller may write the texels, which
    # would force a recompute and defeat what is swept here
    im_t, im_s = sky.transmittanceLUT(), sky.skyviewLUT()
    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1)
    elevation, height = float(rng.uniform(-5, 80)), float(10.0 ** rng.uniform(0, 3.5))
    edits = {}
    history = []
    for step in range(int(rng.integers(4, 10))):
        action = int(rng.integers(0, 6))
        if action == 1:
            height = float(10.0 ** rng.uniform(0, 3.5))
        elif action == 2:
            elevation = float(rng.uniform(-5, 80))
        elif action == 3:
            edits[int(rng.integers(0, 32))] = float(rng.choice([0.0, 1.0, 3.5, float("nan"), 1e-3, 17.0]))
        cam = scene.default_camera()
        cam.cameraPosition[1] = -height
        inp = util.Inputs(32, 32, elevation_degrees=elevation, spots=0, camera=cam)
        words = np.frombuffer(bytes(inp.atm), dtype=np.float32).copy()
        for k, v in edits.items():
            if k not in (19, 23, 27):  # the padding words are not parameters (they are compared too: leave them alone)
                words[k] = v
        atm = abi.AtmospherePacked.from_buffer_copy(words.tobytes())
        cameras.stage([inp.cam])
        atmospheres.stage([atm])
        cameras.recordCopyToDevice()
        atmospheres.recordCopyToDevice()
        if action == 4:
            alias[int(rng.integers(0, sl[1]))].fill_(float(rng.choice([0.0, float("nan"), 5.0])))
            sky.invalidateLUTs(abi.SZG_LUT_SKYVIEW)
        elif action == 5:
            sky.invalidateLUTs(abi.SZG_LUT_TRANSMITTANCE)
        sky.recordTransmittance(None, 0, atmospheres)
        sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
        history.append(action)
        torch.cuda.synchronize()
        got_t, got_s = sky.download_lut(im_t).copy(), sky.download_lut(im_s).copy()
        fresh = pl.SkyViewComputePipeline.create(transmittance_extent=tl, skyview_extent=sl)
        fresh.recordTransmittance(None, 0, atmospheres)
        fresh.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
        torch.cuda.synchronize()
        want_t, want_s = fresh.download_lut(fresh.transmittanceLUT()), fresh.download_lut(fresh.skyviewLUT())
        fresh.destroy()
        if not ((got_t.view(np.uint32) == want_t.view(np.uint32)).all() and (got_s.view(np.uint32) == want_s.view(np.uint32)).all()):
            bad += 1
            print("seed", seed, "step", step, "MISMATCH after actions", history, flush=True)
            break
    sky.destroy()
print("done, mismatching seeds:", bad, "processed up to", seed)
