"""Offline replay of the frame-level rate controller (a Python port of rc_frame_start / rc_frame_end of h264e_host.c, checked against
the oracle's per-frame QPs in truth()) for experiments with the speculation policy of rate-controlled launches: policy.py (the shipped
chain + hedge leaves, size predictors), tree.py (leaves with children, deeper chains).  Needs the oracle library (make -C oracle)."""
import os, sys, ctypes as C, numpy as np, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import oracle_lib, synth
src=open(os.path.join(ROOT, 'h264-lab_amd', 'csrc', 'h264e_host.c')).read()
m=re.search(r'k_bits_per_mb\[2\]\[41\] = \{(.*?)\};',src,re.S)
rows=re.findall(r'\{([^{}]*)\}',m.group(1))
KB=[[int(x) for x in r.replace('\n',' ').split(',') if x.strip()] for r in rows]
assert len(KB)==2 and len(KB[0])==41
def mul32x32shr16(x,y): return ((x>>16)*(y&0xFFFF) + x*(y>>16) + (((y&0xFFFF)*(x&0xFFFF))>>16)) & 0xFFFFFFFF
def clz(x): return 32-x.bit_length()
def div_q16(numer,denum):
    f=1<<clz(denum)
    while True:
        denum=(denum*f>>16)&0xFFFFFFFF
        numer=mul32x32shr16(numer,f)
        f=((1<<17)-denum)&0xFFFFFFFF
        if denum==0xffff: break
    return numer
class RC:
    def __init__(s): s.qp=s.prev_qp=s.vbv_bits=s.qp_smooth=s.dqp_smooth=s.max_dqp=s.bit_budget=s.vbv_target_level=0
    def copy(s):
        r=RC(); r.__dict__.update(s.__dict__); return r
def frame_start(rc,gop,nmb,vbv,dfb,qmin,qmax,intra):
    npp=min(gop-1,63) if gop-1>=0 else 63
    qp=-1; bit_budget=dfb*8
    while True:
        qp+=1
        gop_bits=KB[0][qp]*npp+KB[1][qp]
        if not (gop_bits*nmb > (npp+1)*dfb*8 and qp<40): break
    peak=div_q16(KB[1][qp]<<16, KB[0][qp]<<16)
    if npp:
        ratio=div_q16((npp+1)<<16,(npp<<16)+peak); nominal_p=mul32x32shr16(dfb*8,ratio)
    else: nominal_p=0
    stationary=min(vbv*8>>4,dfb*8)
    if intra: add=mul32x32shr16(nominal_p,peak)-bit_budget
    else:
        add=nominal_p-bit_budget
        if vbv: add+=(rc.vbv_target_level-rc.vbv_bits)>>4
    if vbv: add=min(add,(vbv*8*7>>3)-rc.vbv_bits)
    bit_budget+=add
    bit_budget=min(bit_budget,dfb*8*16); bit_budget=max(bit_budget,dfb*8>>2)
    if intra: rc.vbv_target_level=rc.vbv_bits+bit_budget-dfb*8
    rc.vbv_target_level-=dfb*8-nominal_p
    rc.vbv_target_level=max(rc.vbv_target_level,stationary)
    rc.bit_budget=bit_budget
    bits=KB[1 if intra else 0]
    qp=0
    while qp<41:
        if bits[qp]*nmb<bit_budget: break
        qp+=1
    qp+=10; qp+=rc.dqp_smooth
    if rc.prev_qp>qp+1: qp=(rc.prev_qp+qp+1)//2
    qp=min(qp,qmax); qp=max(qp,qmin); qp=min(qp,51)
    rc.qp=qp; rc.qp_smooth=qp<<8; rc.prev_qp=qp
    return qp
def frame_end(rc,nmb,vbv,dfb,out_bytes,intra,all_skipped=0):
    if not all_skipped:
        qp=0
        while qp!=41 and KB[intra][qp]*nmb>out_bytes*8-32: qp+=1
        qp+=10
        if (rc.qp_smooth>>8)-rc.dqp_smooth<qp-1: rc.dqp_smooth-=1
        elif (rc.qp_smooth>>8)-rc.dqp_smooth>qp+1: rc.dqp_smooth+=1
        if intra: rc.max_dqp=rc.dqp_smooth
        else: rc.max_dqp=max(rc.max_dqp,(rc.qp_smooth>>8)-qp)
    rc.vbv_bits+=out_bytes*8-dfb*8
    if vbv:
        rc.vbv_bits=max(rc.vbv_bits,0); rc.vbv_bits=min(rc.vbv_bits,vbv*8)
    else: rc.vbv_bits=0
def truth(w,h,n,gop,kbps,slices=0):
    import json
    os.makedirs('/tmp/rc_sim', exist_ok=True)
    fn='/tmp/rc_sim/truth_%d_%d_%d_%d_%d_%d.json'%(w,h,n,gop,kbps,slices)       # cache of the oracle's (bytes, QP) per frame
    if os.path.exists(fn): return [tuple(x) for x in json.load(open(fn))]
    c=synth.clip(w,h,n)
    o=oracle_lib.Encoder(w,h,gop=gop,kbps=kbps,slices=slices)
    lib=oracle_lib.lib(); lib.h264o_get_qp.argtypes=[C.c_void_p]
    res=[]
    for t in range(n):
        b=o.encode(c[t]); res.append((len(b), lib.h264o_get_qp(o.e)))
    json.dump(res,open(fn,'w')); return res
if __name__=='__main__':
    w,h,n,gop,kbps=1920,1080,60,30,4000
    tr=truth(w,h,n,gop,kbps)
    nmb=((w+15)//16)*((h+15)//16); dfb=kbps*1000//8//30; vbv=12500
    # verify my RC port reproduces the true QPs
    rc=RC(); ok=True
    for t,(b,q) in enumerate(tr):
        qq=frame_start(rc,gop,nmb,vbv,dfb,10,50,(t%gop)==0)
        if qq!=q: ok=False; print('mismatch',t,qq,q)
        frame_end(rc,nmb,vbv,dfb,b,1 if (t%gop)==0 else 0)
    print('rc port ok' if ok else 'rc port BAD'); print(tr)
