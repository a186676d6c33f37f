import sys; sys.path.insert(0,'.')
import numpy as np, torch
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context
ctx=Context(0)
sim=synthetic_batch(ctx,B=4096,A=8,T=20,seed=1000)
sim.run(8)
for k in range(4):
    b=sim.snapshot(); sim.run(5); sim.step(); a=sim.snapshot()
    b=None
for k in range(3):
    b=sim.snapshot(); sim.step(); a=sim.snapshot()
    same=(b['target_ind']==b['traj_idx'])
    adv=(a['traj_idx']!=b['traj_idx'])
    n=a['cut_len']; hm=a['traj_idx']+1
    remain_w = n - b['target_ind']
    print('step',sim.steps_done,'same start %.3f'%same.mean(),'traj advanced %.3f'%adv.mean(),'hm<n %.3f'%(hm<n).mean(),'window scan len: mean %.1f, >64: %.3f, >2: %.3f'%(remain_w.mean(),(remain_w>64).mean(),(remain_w>2).mean()),
          'all three %.3f'%(same&(hm<n)).mean())
