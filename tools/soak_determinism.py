import sys, os, hashlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from __graft_entry__ import _pkg
P=_pkg()
ce=P.ClipEncoder(1920,1080,600,gop=30,qp=26); ce.generate_synth()
ref=None; t=time.time()
for i in range(40):
    out,fs,st=ce.encode()
    h=hashlib.md5(out).hexdigest()
    if ref is None: ref=h
    assert h==ref, "run %d differs"%i
print("40 passes identical:", ref, "%.1f s"%(time.time()-t))
ce.close()
for (w,h,n,gop,slices,kbps) in [(3840,2160,60,30,0,0),(352,288,900,30,0,0),(1920,1080,120,1,0,0),(200,120,300,7,0,0),
                              (1920,1080,300,30,8,0),(1920,1080,60,30,0,4000),(1920,1080,60,30,8,4000),(7680,4320,12,30,2,60000)]:
    ce=P.ClipEncoder(w,h,n,gop=gop,qp=26,slices=slices,kbps=kbps); ce.generate_synth()
    hs=set(); launches=set()
    for i in range(6):
        out,fs,st=ce.encode(); hs.add(hashlib.md5(out).hexdigest()); launches.add(st.rounds)
    print(w,h,n,"gop",gop,"slices",slices,"kbps",kbps,"6 passes:", "identical" if len(hs)==1 else "DIFFER", "launches per pass", sorted(launches))
    ce.close()
