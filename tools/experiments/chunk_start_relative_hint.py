import sys, time, os
import numpy as np, torch
sys.path.insert(0,'/root/repo')
from gmr_amd import synth
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle, WORK_ITEM_DTYPE
from tests.util import compiled
cm=compiled('bvh','unitree_g1'); orc=Oracle(cm.blob)
n=12; T=4000
lens=[T]*n
pos,quat,names,offs=synth.synth_clips_torch(cm,lens,seed=33,device='cpu',hard=(np.arange(n)%2==1),yaw0=np.pi)
pos,quat=pos.numpy(),quat.numpy()
sc=cm.slot_columns(names)
q_true=np.load('/tmp/exp/q_true.npy')
def qdiff(a,b):
    d=np.abs(a-b); d[...,3:7]=np.minimum(d[...,3:7],np.abs(a[...,3:7]+b[...,3:7])); return d.max(axis=-1)
def qmul(a,b):
    w1,x1,y1,z1=a; w2,x2,y2,z2=b
    return np.array([w1*w2-x1*x2-y1*y2-z1*z2, w1*x2+x1*w2+y1*z2-z1*y2, w1*y2-x1*z2+y1*w2+z1*x2, w1*z2+x1*y2-y1*x2+z1*w2])
def qconj(a): return np.array([a[0],-a[1],-a[2],-a[3]])
def qrot(q,v):
    return qmul(qmul(q,np.array([0,*v])),qconj(q))[1:]
C=64
root_task=[i for i,bb in enumerate(cm.task_body[0]) if bb==0][0]; rslot=cm.task_slot[0][root_task]
def tgt(f):
    tp,tq=orc.prepare_targets(pos[f][sc].astype(np.float64),quat[f][sc].astype(np.float64))
    return tp[rslot], tq[rslot]/np.linalg.norm(tq[rslot])
for B in (32,):
  for hintframe in (C-1, 8*C-1):
    for k in (3,7,9,10,11):
        starts=np.arange(C,T,C)+k*T
        starts=starts[starts-k*T>hintframe]
        items=np.zeros(len(starts),dtype=WORK_ITEM_DTYPE)
        items['frame_begin']=starts-B; items['n_burn']=B; items['n_out']=1
        items['init_row']=np.arange(len(starts)); items['final_row']=-1; items['burn_row']=np.arange(len(starts))
        qh=q_true[k*T+hintframe]
        init=np.tile(qh,(len(starts),1))
        hp,hq=tgt(k*T+hintframe)
        # root relative to its target at the hint frame
        rel_q=qmul(qconj(hq),qh[3:7]); rel_p=qrot(qconj(hq),qh[:3]-hp)
        for i,f in enumerate(starts-B):
            tp,tq=tgt(f)
            init[i,:3]=tp+qrot(tq,rel_p); init[i,3:7]=qmul(tq,rel_q)
        qo,it,qf=orc.ik_solve(pos,quat,sc,items,qpos_init=init,want_final=True,n_threads=8)
        d=qdiff(qf[:len(starts)],q_true[starts-1])
        bad=d>1e-7
        print('B',B,'hint',hintframe,'clip',k,'bad %d/%d'%(bad.sum(),len(bad)),'first bad',np.nonzero(bad)[0][:8], 'd of first', d[:3])
