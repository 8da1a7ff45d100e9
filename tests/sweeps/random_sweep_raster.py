"""Random parity sweep of the compute rasteriser on the GPU box: random cameras, extents, row tiles and scenes (default
scene, triangle soups up to 24 k primitives, hostile geometry), GPU vs oracle bit for bit.
usage: python tests/sweeps/random_sweep_raster.py FIRST_SEED LAST_SEED"""
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from tests import util
from tests.test_raster import _soup, _hostile_scene, _planes_equal
from oracle import binding as ob
from syzygy_amd import abi, pipelines as pl, scene, meshes, lib
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    W=int(rng.integers(9,200)); H=int(rng.integers(5,120))
    cam=scene.default_camera()
    cam.cameraPosition[:]=[float(rng.uniform(-40,40)), float(rng.uniform(-40,5)), float(rng.uniform(-50,30))]
    cam.eulerAngles[:]=[float(rng.uniform(-1.5,1.5)), float(rng.uniform(-3.1,3.1)), float(rng.uniform(-0.5,0.5))]
    cam.fovDegrees=float(rng.uniform(20.0,130.0)); cam.near_plane=float(10.0**rng.uniform(-2,0.5)); cam.far_plane=float(10.0**rng.uniform(2,4))
    inp=util.Inputs(W,H,camera=cam)
    kind=int(rng.integers(0,4))
    ms={0:lambda: meshes.reference_default_scene(),1:lambda: _soup(seed,int(rng.integers(1,400)),spread=float(rng.uniform(5,200))),
        2:lambda: _hostile_scene(seed, bool(rng.integers(0,2))),3:lambda: meshes.reference_default_scene()+_soup(seed,int(rng.integers(4200,6000)),spread=150.0)}[kind]()
    nranks=int(rng.integers(1,4)); block=int(rng.choice([1,3,8])); rank=int(rng.integers(0,nranks))
    tile=util.rowtile(H,block,rank,nranks) if nranks>1 else None
    rows=H if tile is None else tile.local_rows
    if rows==0: continue
    cameras=pl.TStagedBuffer(abi.CameraPacked,1); cameras.push(inp.cam); cameras.recordCopyToDevice()
    target=pl.SceneTexture(W,rows)
    deferred=pl.DeferredShadingPipeline((W,rows),max_spot_lights=1,max_shadow_maps=0)
    deferred.recordGBufferRaster(None,inp.rect,target,0,cameras,ms,tile=tile)
    torch.cuda.synchronize()
    planes=deferred.download_gbuffer(W,rows); depth=target.depth.cpu().numpy()
    want=ob.HostFrame(W,rows); ob.gbuffer_raster(want,inp.rect,tile,inp.cam,ms,threads=8)
    ok=(depth.view(np.uint32)==want.depth.view(np.uint32)).all()
    try:
        _planes_equal(planes,want.planes())
    except AssertionError as e:
        ok=False; msg=str(e)
    else:
        msg=''
    if not ok:
        bad+=1; print('seed',seed,'MISMATCH kind',kind,'W,H',W,H,'tile',(nranks,block,rank),'depth ne',(depth.view(np.uint32)!=want.depth.view(np.uint32)).sum(),msg,flush=True)
    deferred.cleanup()
print('done, mismatching seeds:',bad,'processed up to',seed)
