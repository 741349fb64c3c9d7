import sys, os, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ['H264E_DEBUG']='1'
from __graft_entry__ import _pkg
P=_pkg()
n=int(sys.argv[1]) if len(sys.argv)>1 else 600
ce=P.ClipEncoder(1920,1080,n,gop=30,qp=26); ce.generate_synth()
t=time.time(); out,fs,st=ce.encode(profile=True); dt=time.time()-t
print("time %.2f s rounds %d reenc %d launches %d mb_ms %.1f enc_ms %.1f read_ms %.1f asm_ms %.1f"%(dt,st.rounds,st.reencoded_gops,st.kernel_launches,st.mb_kernel_ms,st.encode_ms,st.readback_ms,st.assemble_ms))
