"""Random parity sweep on the GPU box: frames with random extents, LUT sizes, cameras, sun elevations, atmosphere edits,
spot-light parameters, shadow maps and row tiles, GPU vs oracle bit for bit (NaN patterns included).
usage: python tests/sweeps/random_sweep_frames.py FIRST_SEED LAST_SEED [tiny]   (400 seeds take ~15 s; "tiny": degenerate LUT extents,
2 ... 7 texels a side, whose marches do produce NaN texels in sane atmospheres)"""
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from tests import util
from tests.test_gpu_parity import staged
from oracle import binding as ob
from syzygy_amd import abi, pipelines as pl, scene
class G: pass
g=G(); g.ob=ob; g.abi=abi; g.pl=pl; g.scene=scene
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    W=int(rng.integers(17,150)); H=int(rng.integers(9,90))
    tl=(int(rng.integers(8,200)), int(rng.integers(4,60))); sl=(int(rng.integers(8,200)), int(rng.integers(4,100)))
    if len(sys.argv)>3 and sys.argv[3]=='tiny':
        tl=(int(rng.choice([2,3,5,7,640])), int(rng.choice([2,3,4,130]))); sl=(int(rng.choice([2,3,5,300])), int(rng.choice([2,3,4,9])))
    nsp=int(rng.integers(0,12))
    cam=scene.default_camera()
    cam.cameraPosition[:]=[float(rng.uniform(-40,40)), float(-10.0**rng.uniform(-1.0,4.0)), float(rng.uniform(-50,30))]
    cam.eulerAngles[:]=[float(rng.uniform(-1.5,1.5)), float(rng.uniform(-3.1,3.1)), float(rng.uniform(-0.3,0.3))]
    cam.fovDegrees=float(rng.uniform(20.0,120.0))
    def edit(a):
        a.sunAngularRadius=float(10.0**rng.uniform(-3.5,-0.5))
        if rng.random()<0.3:
            a.absorptionRayleighPerMegameter[:]=[float(rng.uniform(0,2)) for _ in range(3)]
        if rng.random()<0.3:
            a.scatteringOzonePerMegameter[:]=[float(rng.uniform(0,2)) for _ in range(3)]
    inp=util.Inputs(W,H,elevation_degrees=float(rng.uniform(-8.0,90.0)),spots=nsp,camera=cam,atmosphere_edit=edit)
    # randomise the spot lights
    for i in range(nsp):
        s_=inp.spots[i]
        s_.strength=float(10.0**rng.uniform(0,4)); s_.falloffFactor=float(10.0**rng.uniform(-1,1)); s_.falloffDistance=float(10.0**rng.uniform(-1,1.5))
    nranks=int(rng.integers(1,4)); block=int(rng.choice([1,3,8])); rank=int(rng.integers(0,nranks))
    tile=util.rowtile(H,block,rank,nranks) if nranks>1 else None
    rows=H if tile is None else tile.local_rows
    if rows==0: continue
    # shadow maps for a few slots
    nslots=2+nsp
    maps={}
    for slot in range(nslots):
        if rng.random()<0.3:
            maps[slot]=rng.random((int(rng.integers(8,80)),int(rng.integers(8,80))),dtype=np.float32)
    cameras, atmospheres, lights = staged(g, inp)
    target=pl.SceneTexture(W,rows,debug=True)
    deferred=pl.DeferredShadingPipeline((W,rows),max_spot_lights=max(nsp,1),max_shadow_maps=nslots)
    keep=[]; images=(abi.Image*nslots)()
    for slot,m in maps.items():
        images[slot]=ob.host_image(m,abi.SZG_FORMAT_D32_SFLOAT); t=torch.from_numpy(m).cuda(); keep.append(t); deferred.setShadowMap(slot,t)
    host_maps=abi.ShadowMaps(nslots,0,C.cast(images,C.POINTER(abi.Image)))
    sky=pl.SkyViewComputePipeline.create(transmittance_extent=tl, skyview_extent=sl)
    skip=int(rng.integers(0,3))
    deferred.recordGBufferFill(None,inp.rect,target,0,cameras,inp.synthetic.fill,tile=tile)
    deferred.recordLights(None,inp.rect,target,skip,lights,inp.spots if nsp else None,0,cameras,tile=tile)
    sky.recordDrawCommands(None,target,inp.rect,deferred.gbuffer(),deferred.shadowMaps(),0,atmospheres,0,cameras,0,lights,tile=tile)
    torch.cuda.synchronize()
    got=target.debug.cpu().numpy(); got_q=target.color_numpy()
    fr=ob.HostFrame(W,rows)
    ob.gbuffer_fill(fr,inp.rect,tile,inp.cam,inp.synthetic.fill,threads=8)
    ob.lights(fr,inp.rect,tile,host_maps,inp.cam,inp.dirs,2,skip,inp.spots,nsp,threads=8)
    tlut=ob.transmittance_lut(inp.atm,tl[0],tl[1],threads=8); slut=ob.skyview_lut(inp.atm,inp.cam,tlut,sl[0],sl[1],threads=8)
    ob.composite(fr,inp.rect,tile,host_maps,inp.atm,inp.cam,inp.dirs,0,tlut,slut,threads=8)
    same=((got.view(np.uint32)==fr.debug.view(np.uint32))|(np.isnan(got)&np.isnan(fr.debug))).all() and (got_q==fr.color).all()
    if not same:
        bad+=1; ne=(got.view(np.uint32)!=fr.debug.view(np.uint32))&~(np.isnan(got)&np.isnan(fr.debug))
        print('seed',seed,'MISMATCH',ne.sum(),'of',ne.size,'W,H',W,H,'luts',tl,sl,'spots',nsp,'tile',(nranks,block,rank),'maps',list(maps),'skip',skip,'nan gpu/oracle',np.isnan(got).sum(),np.isnan(fr.debug).sum(),flush=True)
        if __import__('os').environ.get('SZG_SWEEP_VERBOSE'):
            gl=sky.download_lut(sky.skyviewLUT()); print('  sky-view LUT texels differing:',int((gl.view(np.uint32)!=slut.view(np.uint32)).sum()),'transmittance:',int((sky.download_lut(sky.transmittanceLUT()).view(np.uint32)!=tlut.view(np.uint32)).sum()))
            ys,xs=np.nonzero(ne.any(axis=-1))
            for y,x in list(zip(ys,xs))[:8]:
                print('  px',int(x),int(y),'depth',float(fr.depth[y,x]),'metal',float(fr.orm[y,x,2]),'gpu',got[y,x,:3],'oracle',fr.debug[y,x,:3],'pos',fr.position[y,x,:3])
    deferred.cleanup(); sky.destroy()
print('done, mismatching seeds:',bad)
