import sys, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); os.makedirs('gpurun_out',exist_ok=True)
from conftest import get_scene, GOLDEN
from pathtracing_amd import Context
from PIL import Image
ctx=Context(0); ctx.upload(get_scene('room')); ctx.render_begin(256,256)
img=ctx.render_image(seed=0, iterations=250, spi=256)
np.save('gpurun_out/room256_64k.npy', img)
c=1-np.exp(-img); c=np.power(c,np.float32(1/2.2)); ldr=np.floor(c*255).astype(np.int32)
ref=np.asarray(Image.open(f"{GOLDEN}/reference_results/room-256-i250.png")).astype(np.int32)[...,:3]
d=ldr-ref
print('signed mean diff per channel', d.mean(axis=(0,1)), 'abs mean', np.abs(d).mean())
for name,sl in [('left wall',(slice(100,150),slice(5,30))),('right wall',(slice(100,150),slice(226,250))),('back wall',(slice(60,120),slice(100,160))),('floor',(slice(235,250),slice(60,200))),('ceiling',(slice(5,30),slice(40,90))),('box top',(slice(135,140),slice(50,80)))]:
    print(name, 'gpu', ldr[sl].mean(axis=(0,1)), 'ref', ref[sl].mean(axis=(0,1)))
print('hist of diff', np.bincount((d+10).clip(0,20).ravel()))
