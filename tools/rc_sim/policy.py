from sim import *
def simulate(tr,w,h,gop,kbps,depth=6,hedge=(-1,1,-2,2),pred_mode='last',verbose=0):
    nmb=((w+15)//16)*((h+15)//16); dfb=kbps*1000//8//30; vbv=12500
    n=len(tr); rc=RC(); last=[0,0]; last_qp=[0,0]
    pos=0; launches=0; rc_frame=-1; qp_exact=None; mainhits=0; leafhits=0
    while pos<n:
        launches+=1
        key=(pos%gop)==0
        if rc_frame!=pos:
            qp_exact=frame_start(rc,gop,nmb,vbv,dfb,10,50,key); rc_frame=pos
        F=min(depth,n-pos)
        ahead=rc.copy(); spec=[qp_exact]
        for i in range(1,F):
            f=pos+i; pk=1 if ((f-1)%gop)==0 else 0
            p=last[pk] if last[pk]>0 else dfb
            if pred_mode=='scaled' and last[pk]>0:
                # scale the last size of this kind by the controller's own bits model for the QP the frame in front runs at
                qprev=spec[i-1]; ql=last_qp[pk]
                p=int(last[pk]*KB[pk][qprev-10]/max(KB[pk][ql-10],1))
            frame_end(ahead,nmb,vbv,dfb,p,pk)
            spec.append(frame_start(ahead,gop,nmb,vbv,dfb,10,50,(f%gop)==0))
        # consume
        i=0
        while True:
            f=pos+i; key=(f%gop)==0; b,q=tr[f]
            assert q==(spec[i] if i<len(spec) else q)
            frame_end(rc,nmb,vbv,dfb,b,1 if key else 0); last[1 if key else 0]=b; last_qp[1 if key else 0]=q
            if f+1>=n: pos=n; break
            qn=frame_start(rc,gop,nmb,vbv,dfb,10,50,((f+1)%gop)==0); rc_frame=f+1; qp_exact=qn
            assert qn==tr[f+1][1],(f,qn,tr[f+1])
            if i+1<F and spec[i+1]==qn:
                mainhits+=1; i+=1; continue
            if i+1<F and any(spec[i+1]+d==qn for d in hedge):
                # leaf: frame f+1 accepted too, launch ends
                leafhits+=1
                b2,q2=tr[f+1]; k2=((f+1)%gop)==0
                frame_end(rc,nmb,vbv,dfb,b2,1 if k2 else 0); last[1 if k2 else 0]=b2; last_qp[1 if k2 else 0]=q2
                pos=f+2
                if pos<n:
                    qp_exact=frame_start(rc,gop,nmb,vbv,dfb,10,50,(pos%gop)==0); rc_frame=pos
                break
            pos=f+1; break
    return launches,mainhits,leafhits
if __name__=='__main__':
    import itertools
    for (w,h,n,gop,kbps) in [(1920,1080,60,30,4000),(1920,1080,240,30,4000),(352,288,300,30,500)]:
        tr=truth(w,h,n,gop,kbps)
        for mode in ('last','scaled'):
            for depth in (6,):
                for hedge in ((-1,1,-2,2),(-1,1)):
                    print(w,h,n,kbps,mode,depth,hedge,simulate(tr,w,h,gop,kbps,depth,hedge,mode))
