import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests import util
from oracle import binding as ob
from syzygy_amd import abi, pipelines as pl, scene
import tests.test_gpu_parity as T
class G: pass
g=G(); g.ob=ob; g.abi=abi; g.pl=pl; g.scene=scene
inp = util.Inputs(64,64,elevation_degrees=70.0)
cameras, atmospheres, lights = T.staged(g, inp)
sky = pl.SkyViewComputePipeline.create(transmittance_extent=(512,128), skyview_extent=(512,256))
tlut = ob.transmittance_lut(inp.atm,512,128,threads=8)
sky.upload_lut(sky.transmittanceLUT(), tlut)
back = sky.download_lut(sky.transmittanceLUT())
print('lut roundtrip exact', (back==tlut).all())
sky.recordSkyViewLUT(None,0,atmospheres,0,cameras)
torch.cuda.synchronize()
got = sky.download_lut(sky.skyviewLUT())
want = ob.skyview_lut(inp.atm, inp.cam, tlut, 512, 256, threads=8)
rel = np.abs(got[...,:3]-want[...,:3])/np.maximum(np.abs(want[...,:3]),1e-9)
bad = (rel>1e-4).any(-1)
print('bad rows:', np.nonzero(bad.any(1))[0][[0,-1]], 'bad cols', np.nonzero(bad.any(0))[0][[0,-1]])
print('bad per row (every 16):', bad.sum(1)[::16])
ys,xs=np.nonzero(bad)
for k in range(0,len(ys),len(ys)//8):
    y,x=ys[k],xs[k]; print(y,x,got[y,x,:3],want[y,x,:3])
