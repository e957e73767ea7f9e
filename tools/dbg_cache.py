import sys, torch; sys.path.insert(0,'.')
from contextgs_amd.synth import make_scene, orbit_cameras, SynthPipe
from contextgs_amd import context_model as cm, multi_level as ml
from contextgs_amd.renderer import render, prefilter_voxel
calls={'plan':0,'uncached':0,'unique':0}
o1=cm._level_plan_uncached
def w1(*a,**k): calls['uncached']+=1; return o1(*a,**k)
cm._level_plan_uncached=w1
o2=ml.torch_unique_with_indices
def w2(*a,**k): calls['unique']+=1; return o2(*a,**k)
cm.torch_unique_with_indices=w2
pc=make_scene(200000,seed=0); pc.train()
cams=[c.to_torch('cuda') for c in orbit_cameras(4,640,360)]
bg=torch.zeros(3,device='cuda')
for i in range(4):
    vis=prefilter_voxel(cams[i],pc,SynthPipe(),bg)
    pkg=render(cams[i],pc,SynthPipe(),bg,visible_mask=vis,step=20000)
    pkg['render'].sum().backward()
    print(i, calls, pc._level_cache is not None)
