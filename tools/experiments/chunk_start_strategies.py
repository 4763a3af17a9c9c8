import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from gmr_amd import synth
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle, WORK_ITEM_DTYPE
from tests.util import compiled
cm=compiled('smplx','unitree_g1'); orc=Oracle(cm.blob)
T=3000
def qdiff(a,b):
    d=np.abs(a-b); d[...,3:7]=np.minimum(d[...,3:7],np.abs(a[...,3:7]+b[...,3:7])); return d.max(axis=-1)
for hard in (False,True):
    pos,quat,names,offs,_=synth.synth_clips(cm,1,T,seed=2000 if hard else 1000,hard=hard,dtype=np.float32)
    sc=cm.slot_columns(names)
    t0=time.time(); q_true,it_true,_=orc.ik_solve(pos,quat,sc,make_items(offs)); print('hard',hard,'seq',time.time()-t0,'s','solves/frame',it_true.mean())
    root_task=[i for i,b in enumerate(cm.task_body[0]) if b==0][0]; rslot=cm.task_slot[0][root_task]
    for C in (16,):
        starts=np.arange(C,T,C)
        for B in (4,8,16,24):
            for strat in ('qpos0','root','coarse16','coarse64'):
                init=np.zeros((len(starts),orc.nq))
                if strat=='root':
                    for i,c in enumerate(starts):
                        f=c-B
                        tp,tq=orc.prepare_targets(pos[f][sc].astype(np.float64),quat[f][sc].astype(np.float64))
                        init[i]=cm.robot.qpos0; init[i,:3]=tp[rslot]; init[i,3:7]=tq[rslot]/np.linalg.norm(tq[rslot])
                elif strat.startswith('coarse'):
                    k=int(strat[6:])
                    # sequential pass over frames 0,k,2k..; anchor for start c = state at the latest coarse frame <= c-B
                    idx=np.arange(0,T,k)
                    qc,itc,_=orc.ik_solve(pos[idx],quat[idx],sc,make_items([0,len(idx)]))
                    for i,c in enumerate(starts):
                        j=max(0,(c-B)//k); init[i]=qc[j]
                    ncoarse=itc.sum()
                items=np.zeros(len(starts),dtype=WORK_ITEM_DTYPE)
                items['frame_begin']=starts-B; items['n_burn']=B; items['n_out']=1
                items['init_row']=-1 if strat=='qpos0' else np.arange(len(starts)); items['final_row']=-1; items['burn_row']=np.arange(len(starts))
                qo,it,qf=orc.ik_solve(pos,quat,sc,items,qpos_init=init,want_final=True,n_threads=8)
                d=qdiff(qf[:len(starts)],q_true[starts-1])
                print(f'C={C} B={B:2d} {strat:9s} match1e-7 {np.mean(d<1e-7):.3f} match1e-3 {np.mean(d<1e-3):.3f} far(>0.5) {np.mean(d>0.5):.3f}', ('coarse solves %d'%ncoarse) if strat.startswith('coarse') else '')
