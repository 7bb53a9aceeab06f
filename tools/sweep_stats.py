"""Statistics of the segmented sweeps from the CPU oracle on the 1024x436 synthetic pair, per iteration and direction: fraction of
visited pixels whose candidate is accepted, whose rejection-path candidate equals their own match (skip rule), and of steps that
follow an accepted candidate (the only ones phase B of the speculative form must evaluate itself).  CPU only; test infrastructure."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from eppm_amd import synth
O.set_num_threads(3)
a,b,_,_ = synth.make_pair(436,1024,seed=1234)
u,v,st = O.compute_flow(a,b,dump=True)
L=2
i1,i2,c1,c2 = st["img1_L2"],st["img2_L2"],st["cen1_L2"],st["cen2_L2"]
h,w = i1.shape
nnf, states = O.gen_rand_field(w,h)
cost = O.cost_field(nnf,i1,i2,c1,c2)
SL=10
for it in range(10):
    for d in range(4):
        c2_, n2 = O.seg_propagate_dir(cost,nnf,i1,i2,c1,c2,d)
        acc = (n2["x"]!=nnf["x"])|(n2["y"]!=nnf["y"])
        # orient so that the sweep runs along axis 1, forward
        A = acc if d in (0,2) else acc.T
        NI = nnf if d in (0,2) else nnf.T
        if d>=2: A=A[:,::-1]; NI=NI[:,::-1]
        # candidate on the rejection path vs own: skip if equal
        if d==0: candx=np.minimum(NI["x"][:,:-1]+1,w-1); candy=NI["y"][:,:-1]
        elif d==2: candx=np.maximum(NI["x"][:,:-1]-1,0); candy=NI["y"][:,:-1]
        elif d==1: candy=np.minimum(NI["y"][:,:-1]+1,h-1); candx=NI["x"][:,:-1]
        else: candy=np.maximum(NI["y"][:,:-1]-1,0); candx=NI["x"][:,:-1]
        same=(candx==NI["x"][:,1:])&(candy==NI["y"][:,1:])
        n=A.shape[1]
        # fresh eval needed at position j (j>=1) if A[j-1] accepted and j is not a segment's first step (approx: fwd seg k>=1 starts at 10k; first step pixel 10k ... ignore details)
        prev_acc = A[:,:-1]
        pos = np.arange(1,n)
        first = (pos%SL==0) if d<2 else ((n-1-pos)%SL==SL-1)   # rough
        fresh = prev_acc[:,:] & ~first[None,:]
        # per chain (line, seg): number of fresh steps
        nl = A.shape[0]
        segid = pos//SL
        nseg = segid.max()+1
        per = np.zeros((nl,nseg),int)
        np.add.at(per,(np.repeat(np.arange(nl),len(pos)).reshape(nl,-1), np.tile(segid,(nl,1))), fresh)
        # waves: 4 chains (4 consecutive segs) ; wg: 16 chains
        pw = per[:, :nseg//4*4].reshape(nl,-1,4).max(axis=2)
        print(f"it{it} d{d}: accept {acc.mean():.3f}  skip(rej-path cand==own) {same.mean():.3f}  fresh-steps {fresh.mean():.3f}  chain fresh mean {per.mean():.2f} max {per.max()}  wave(4 chains) mean-of-max {pw.mean():.2f}")
        cost, nnf = c2_, n2
    states,cost,nnf = O.random_search(states,cost,nnf,i1,i2,c1,c2)
