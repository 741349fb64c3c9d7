#!/usr/bin/env python3
"""Analysis helper (not a test): where and how often the reference's mv_clusters state MOVES on the bench clip, from the oracle's
per-macroblock trace.  python tests/analysis_clusters_trajectory.py [frames]

The state (h264-lab.h:5776-5779, two ratcheting vectors) moves by one quarter sample when a macroblock's vector lies 8+ samples
away from its cluster: single outlier macroblocks, once every few frames on synth_v1.  A move that crosses a rounding boundary
(the candidates are consumed rounded to full samples, h264-lab.h:3498) is what invalidates every frame in flight behind it
(DESIGN.md 5): the printed `flips`.  On the 1080p bench clip the second cluster hovers around x = 2..3 quarter samples, so most
moves are flips -- 23 in 600 frames."""
import sys, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo/tests')
import oracle_lib, clips
w,h,n=1920,1080,int(sys.argv[1]) if len(sys.argv) > 1 else 40
nmbx=120
e = oracle_lib.Encoder(w,h,gop=30,qp=26)
def step(cl, mvx, mvy):
    n_=mvx*mvx+mvy*mvy; n0=cl[0][0]**2+cl[0][1]**2; n1=cl[1][0]**2+cl[1][1]**2
    moved=[]
    if n_<n1:
        new=[(63*cl[0][0]+mvx+32)>>6,(63*cl[0][1]+mvy+32)>>6]
        if new!=cl[0]: moved.append((0,tuple(cl[0]),tuple(new)))
        cl[0][:]=new
    if n_>=n0:
        new=[(63*cl[1][0]+mvx+32)>>6,(63*cl[1][1]+mvy+32)>>6]
        if new!=cl[1]: moved.append((1,tuple(cl[1]),tuple(new)))
        cl[1][:]=new
    return moved
cl=[[0,0],[0,0]]
rnd=lambda v:((v[0]+1)&~3,(v[1]+1)&~3)
buf=np.empty(w*h*3//2,np.uint8)
for t in range(n):
    oracle_lib.lib().synth_v1_frame(buf.ctypes.data,w,h,t,1)
    e.encode(buf)
    tr=e.trace()
    mv=[]; fl=[]
    for k,(typ,cbp,mx,my,bp) in enumerate(tr):
        if typ<5:
            before=(rnd(cl[0]),rnd(cl[1]))
            m=step(cl,mx,my)
            for mm in m: mv.append((k//nmbx,k%nmbx)+mm)
            after=(rnd(cl[0]),rnd(cl[1]))
            if after!=before: fl.append((k//nmbx,k%nmbx,before,after))
    print("frame",t,"state",cl,"moves",len(mv),"flips",fl, "move rows", sorted(set(m[0] for m in mv))[:12])
