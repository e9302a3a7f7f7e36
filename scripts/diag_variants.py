import sys, os, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); os.makedirs('gpurun_out',exist_ok=True)
from conftest import get_scene
from pathtracing_amd import Context, scene as S
sc=get_scene('room')
ctx=Context(0)
def render(tag):
    ctx.upload(sc); ctx.render_begin(256,256)
    img=ctx.render_image(seed=0, iterations=128, spi=256)
    np.save(f'gpurun_out/room_{tag}.npy', img.reshape(64,4,64,4,3).mean(axis=(1,3)))
for val in (0.9, 0.95):
    r=S.ColorSampler([val,0,0])
    for n in sc.nodes:
        if float(n.material.roughness.color[0])>=0.9: n.material.roughness=r
    render(f'rough{int(val*100)}')
