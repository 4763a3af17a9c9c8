import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from gmr_amd import synth
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle, WORK_ITEM_DTYPE
from tests.util import compiled
cm=compiled('bvh','unitree_g1'); orc=Oracle(cm.blob)
T=9000; n=8
def qdiff(a,b):
    d=np.abs(a-b); d[...,3:7]=np.minimum(d[...,3:7],np.abs(a[...,3:7]+b[...,3:7])); return d.max(axis=-1)
import os
BB=int(os.environ.get('BB','24'))
for hard,seed in ((False,33),(True,34)):
    pos,quat,names,offs,_=synth.synth_clips(cm,n,T,seed=seed,hard=hard,dtype=np.float32)
    sc=cm.slot_columns(names)
    q_true,it_true,_=orc.ik_solve(pos,quat,sc,make_items(offs),n_threads=8)
    C,B=16,BB
    root_task=[i for i,bb in enumerate(cm.task_body[0]) if bb==0][0]; rslot=cm.task_slot[0][root_task]
    for k in range(n):
        starts=np.arange(C,T,C)+k*T
        items=np.zeros(len(starts),dtype=WORK_ITEM_DTYPE)
        b=np.minimum(B,starts-k*T)
        items['frame_begin']=starts-b; items['n_burn']=b; items['n_out']=1
        items['init_row']=np.arange(len(starts)); items['final_row']=-1; items['burn_row']=np.arange(len(starts))
        init=np.tile(cm.robot.qpos0,(len(starts),1))
        for i,f in enumerate(starts-b):
            tp,tq=orc.prepare_targets(pos[f][sc].astype(np.float64),quat[f][sc].astype(np.float64))
            init[i,:3]=tp[rslot]; init[i,3:7]=tq[rslot]/np.linalg.norm(tq[rslot])
        qo,it,qf=orc.ik_solve(pos,quat,sc,items,qpos_init=init,want_final=True,n_threads=8)
        d=qdiff(qf[:len(starts)],q_true[starts-1])
        bad=d>1e-7
        # stretches
        runs=[];c=0
        for x in bad:
            if x:c+=1
            elif c: runs.append(c);c=0
        if c:runs.append(c)
        print('hard',hard,'clip',k,'solves/frame %.2f'%it_true[k*T:(k+1)*T].mean(),'bad frac %.3f'%bad.mean(),'far %.3f'%np.mean(d>0.5),'runs',sorted(runs)[-6:], 'n runs',len(runs))
