import sys, numpy as np, torch, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from tests import util
from tests.test_raster import _soup
from syzygy_amd import meshes, abi, pipelines as pl
def run(name, W, H, ms, reps=10):
    inp=util.Inputs(W,H)
    cams=pl.TStagedBuffer(abi.CameraPacked,1); cams.push(inp.cam); cams.recordCopyToDevice()
    target=pl.SceneTexture(W,H)
    d=pl.DeferredShadingPipeline((W,H),max_spot_lights=1,max_shadow_maps=0)
    for _ in range(2): d.recordGBufferRaster(None,inp.rect,target,0,cams,ms)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): d.recordGBufferRaster(None,inp.rect,target,0,cams,ms)
    e1.record(); torch.cuda.synchronize()
    ms_=e0.elapsed_time(e1)/reps
    cov=float((target.depth>0).float().mean())
    prims=sum(len(m.indices)//3*len(m.models) for m in ms)
    print(f"{name}: {W}x{H} prims {prims} coverage {cov:.3f}: {ms_:.3f} ms  -> {52*W*H/ms_/1e6:.0f} GB/s written", flush=True)
    d.cleanup()
inp=util.Inputs(8,8)
run("reference default scene", 3840,2160, meshes.reference_default_scene())
run("fill-scene meshes (24 boxes+ground)", 3840,2160, meshes.meshes_of_fill_scene(inp.synthetic.fill))
run("soup 2k", 3840,2160, _soup(1,500))
run("soup 20k", 3840,2160, _soup(2,5000,spread=120.0))
run("soup 200k small tris", 3840,2160, _soup(3,50000,spread=300.0))
run("reference default scene 8K", 7680,4320, meshes.reference_default_scene())
