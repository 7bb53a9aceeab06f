import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import oracle as O
from eppm_amd import synth
from conftest import read_ppm
O.set_num_threads(8)
def stats(name, a, b):
    u,v,st=O.compute_flow(a,b,dump=True)
    for lvl, src in ((1,'flow_L2'),(0,'flow_L1')):
        f=st[src]
        h,w = st['img1_L%d'%lvl].shape
        up=O.resize_flow(f, h, w, 2.0); up=O.mul_scalar(up, 2.0)
        fx=up['x']; fy=up['y']
        known=~((fx>1e9)|(fy>1e9))
        ix=np.trunc(np.clip(fx,-32768,32767)).astype(int); iy=np.trunc(np.clip(fy,-32768,32767)).astype(int)
        n=uni=r1=0
        for y0 in range(0,h,16):
            for x0 in range(0,w,16):
                k=known[y0:y0+16,x0:x0+16]
                if not k.any(): continue
                a_=ix[y0:y0+16,x0:x0+16][k]; b_=iy[y0:y0+16,x0:x0+16][k]
                rx=a_.max()-a_.min(); ry=b_.max()-b_.min()
                n+=1; uni+= (rx==0 and ry==0); r1 += (rx<=1 and ry<=1)
        print(name,'level',lvl,'tiles',n,'uniform %.3f'%(uni/n),'range<=1 %.3f'%(r1/n))
a,b,_,_=synth.make_pair_cached(436,1024,seed=1234)
stats('sintel1234',a,b)
a,b,_,_=synth.make_pair_cached(436,1024,seed=1240)
stats('sintel1240',a,b)
from conftest import GOLDEN
import os
stats('middlebury', read_ppm(os.path.join(GOLDEN,'frame10.ppm')), read_ppm(os.path.join(GOLDEN,'frame11.ppm')))
def kstats(name, a, b):
    u,v,st=O.compute_flow(a,b,dump=True)
    for lvl, src in ((1,'flow_L2'),(0,'flow_L1')):
        f=st[src]; h,w = st['img1_L%d'%lvl].shape
        up=O.mul_scalar(O.resize_flow(f, h, w, 2.0), 2.0)
        fx=up['x']; fy=up['y']; known=~((fx>1e9)|(fy>1e9))
        ix=np.trunc(np.clip(fx,-32768,32767)).astype(int); iy=np.trunc(np.clip(fy,-32768,32767)).astype(int)
        ks=[]; ds=[]
        for y0 in range(0,h,16):
            for x0 in range(0,w,16):
                k=known[y0:y0+16,x0:x0+16]
                if not k.any(): continue
                a_=ix[y0:y0+16,x0:x0+16][k]; b_=iy[y0:y0+16,x0:x0+16][k]
                fl=set(zip(a_.tolist(),b_.tolist())); ks.append(len(fl))
                ds.append(len({(p+m,q+n) for p,q in fl for m in (-1,0,1) for n in (-1,0,1)}))
        ks=np.array(ks); ds=np.array(ds)
        print(name,'level',lvl,'distinct flows per tile: median %d mean %.1f p90 %d'%(np.median(ks),ks.mean(),np.percentile(ks,90)),'| distinct candidate displacements: median %d mean %.1f (x1156 texels = %.0f table entries vs 230400 pass-0 terms)'%(np.median(ds),ds.mean(),ds.mean()*1156))
a,b,_,_=synth.make_pair_cached(436,1024,seed=1234)
kstats('sintel1234',a,b)
kstats('middlebury', read_ppm(os.path.join(GOLDEN,'frame10.ppm')), read_ppm(os.path.join(GOLDEN,'frame11.ppm')))
