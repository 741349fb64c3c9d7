import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from __graft_entry__ import _pkg
import oracle_lib
P=_pkg()
w,h,n=1920,1080,20
c=oracle_lib.synth_c(w,h,n)
e=P.Encoder(w,h,gop=30,qp=26)
e.encode(c[0])
t=time.time()
for i in range(1,n): e.encode(c[i])
dt=time.time()-t
print("H264E_encode 1080p: %.1f ms/frame (%.1f fps)"%(dt/(n-1)*1e3,(n-1)/dt))
e.close()
