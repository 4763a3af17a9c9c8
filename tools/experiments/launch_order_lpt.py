import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm=compiled('smplx','unitree_g1'); eng=Engine(cm)
dev=torch.device('cuda',0)
S,T,D=8192,3000,64
pe,qe,names,_,_=synth.synth_clips(cm,D//2,T,seed=1000,hard=False,dtype=np.float32)
ph,qh,_,_,_=synth.synth_clips(cm,D-D//2,T,seed=2000,hard=True,dtype=np.float32)
bp,bq=np.concatenate([pe,ph]),np.concatenate([qe,qh])
pos=torch.from_numpy(bp).to(dev).repeat(S//D,1,1).contiguous(); quat=torch.from_numpy(bq).to(dev).repeat(S//D,1,1).contiguous()
offs=np.arange(S+1,dtype=np.int64)*T
sc=cm.slot_columns(names)
items=make_items(offs)
out=torch.empty((S*T,eng.nq),dtype=torch.float64,device=dev)
def run(items,reps=3):
    ts=[]
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter()
        q,it,_=eng.ik_solve(pos,quat,sc,items,out=out)
        torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return min(ts),it
t0,it=run(items)
work=it.reshape(S,T).to(torch.int64).bitwise_and(0x3FFFFFFF).sum(1).cpu().numpy()
print('default order ms',t0*1e3,'work max/mean',work.max()/work.mean(), 'ideal ms (sum work/2048 slots * 9.7us)', work.sum()/2048*9.7e-3)
order=np.argsort(-work,kind='stable')
t1,_=run(items[order])
print('LPT by true work ms',t1*1e3)
# probe-based: first 64 frames
pw=it.reshape(S,T)[:,:64].to(torch.int64).bitwise_and(0x3FFFFFFF).sum(1).cpu().numpy()
order2=np.argsort(-pw,kind='stable')
t2,_=run(items[order2])
print('LPT by 64-frame probe ms',t2*1e3)
order3=np.argsort(work,kind='stable')
t3,_=run(items[order3]); print('shortest first ms',t3*1e3)
for P in (8, 16, 32):
    pw = it.reshape(S, T)[:, :P].to(torch.int64).bitwise_and(0x3FFFFFFF).sum(1).cpu().numpy()
    tP, _ = run(items[np.argsort(-pw, kind='stable')])
    print('LPT by %d-frame probe ms' % P, tP * 1e3)
# the un-shaped workload of bench.py: distinct clips, any heading, lengths U(1000, 5000)
del pos, quat, out
rng = np.random.default_rng(7)
lens = rng.integers(1000, 5001, size=S)
upos, uquat, unames, uoffs = synth.synth_clips_torch(cm, lens, seed=4242, device=dev, hard=(np.arange(S) % 2 == 1), yaw0=np.pi)
usc = cm.slot_columns(unames)
uitems = make_items(uoffs)
uout = torch.empty((int(uoffs[-1]), eng.nq), dtype=torch.float64, device=dev)
def urun(items, reps=2):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, it, _ = eng.ik_solve(upos, uquat, usc, items, out=uout)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), it
t0, uit = urun(uitems)
uit = (uit.to(torch.int64) & 0x3FFFFFFF)
cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(uit, 0)])
o = torch.from_numpy(uoffs).to(dev)
work = (cs[o[1:]] - cs[o[:-1]]).cpu().numpy()
print('unshaped: default (length-sorted) ms', t0 * 1e3, 'frames/s %.3e' % (uoffs[-1] / t0))
t1, _ = urun(uitems[np.argsort(-work, kind='stable')])
print('unshaped: LPT by true work ms', t1 * 1e3, 'frames/s %.3e' % (uoffs[-1] / t1))
for P in (32, 64):
    pw = (cs[o[:-1] + P] - cs[o[:-1]]).cpu().numpy() / P * lens
    tP, _ = urun(uitems[np.argsort(-pw, kind='stable')])
    print('unshaped: LPT by %d-frame probe x length ms' % P, tP * 1e3, 'frames/s %.3e' % (uoffs[-1] / tP))
