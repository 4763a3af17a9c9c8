import sys, time
import numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from gmr_amd import synth
from gmr_amd.schedule import make_items
from gmr_amd._native import INIT_ROOT_TARGET
from oracle.oracle import Oracle, WORK_ITEM_DTYPE
from tests.util import compiled
cm=compiled('bvh','unitree_g1'); orc=Oracle(cm.blob)
n=12; T=4000
lens=[T]*n
pos,quat,names,offs=synth.synth_clips_torch(cm,lens,seed=33,device='cpu',hard=(np.arange(n)%2==1),yaw0=np.pi)
pos,quat=pos.numpy(),quat.numpy()
sc=cm.slot_columns(names)
q_true,it_true,_=orc.ik_solve(pos,quat,sc,make_items(offs),n_threads=8)
def qdiff(a,b):
    d=np.abs(a-b); d[...,3:7]=np.minimum(d[...,3:7],np.abs(a[...,3:7]+b[...,3:7])); return d.max(axis=-1)
C,B=64,32
for k in range(n):
    starts=np.arange(C,T,C)+k*T
    items=np.zeros(len(starts),dtype=WORK_ITEM_DTYPE)
    items['frame_begin']=starts-B; items['n_burn']=B; items['n_out']=1
    items['init_row']=INIT_ROOT_TARGET; items['final_row']=-1; items['burn_row']=np.arange(len(starts))
    qo,it,qf=orc.ik_solve(pos,quat,sc,items,want_final=True,n_threads=8)
    d=qdiff(qf[:len(starts)],q_true[starts-1])
    bad=d>1e-7
    last=np.nonzero(bad)[0].max()+1 if bad.any() else 0
    # what differs at a late mismatch
    yaw0=2*np.arctan2(q_true[k*T,6],q_true[k*T,3])
    j=np.nonzero(bad)[0]
    info=''
    if len(j):
        jj=j[-1]; dd=np.abs(qf[jj]-q_true[starts[jj]-1]); info='worst dof %d diff %.2f'%(dd[7:].argmax(), dd[7:].max())
    print('clip',k,'hard',k%2,'solves/frame %.2f'%it_true[k*T:(k+1)*T].mean(),'bad chunks %d/%d'%(bad.sum(),len(bad)),'last bad chunk',last, info, 'first-frame solves',it_true[k*T:k*T+3])
