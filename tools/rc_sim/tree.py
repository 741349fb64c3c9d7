from sim import *
def run(tr,w,h,gop,kbps,policy,cap=28):
    nmb=((w+15)//16)*((h+15)//16); dfb=kbps*1000//8//30; vbv=12500
    n=len(tr); rc=RC(); last=[0,0]
    pos=0; launches=0; frames_launched=0
    def accept(f):
        b,q=tr[f]; k=1 if (f%gop)==0 else 0
        frame_end(rc,nmb,vbv,dfb,b,k); last[k]=b
    qp_exact=frame_start(rc,gop,nmb,vbv,dfb,10,50,True)
    while pos<n:
        launches+=1
        # build the tree: node = dict(f, qp, rc_after_pred (state after predicted end of this frame), children)
        root=dict(f=pos,qp=qp_exact,st=rc.copy(),ch=[],hl=0,depth=0)
        nodes=[root]; frontier=[root]
        while frontier and len(nodes)<cap:
            nxt=[]
            for nd in frontier:
                f=nd['f']+1
                if f>=n or f>=pos+policy['depth']: continue
                st=nd['st'].copy(); pk=1 if (nd['f']%gop)==0 else 0
                p=last[pk] if last[pk]>0 else dfb
                # the parent's own QP is nd['qp'] (may differ from st.qp when it is a hedge): set the state as if it had run at that QP
                st.qp=nd['qp']; st.qp_smooth=nd['qp']<<8; st.prev_qp=nd['qp']
                frame_end(st,nmb,vbv,dfb,p,pk)
                st2=st.copy(); q=frame_start(st2,gop,nmb,vbv,dfb,10,50,(f%gop)==0)
                kids=[q]
                if nd['hl']<policy['hedge_levels'] and nd['depth']<policy['hedge_pos']:
                    kids+= [q+d for d in policy['hedges'] if 10<=q+d<=50]
                for j,qq in enumerate(kids):
                    if len(nodes)>=cap: break
                    c=dict(f=f,qp=qq,st=st2.copy(),ch=[],hl=nd['hl']+(1 if j else 0),depth=nd['depth']+1)
                    nd['ch'].append(c); nodes.append(c); nxt.append(c)
            frontier=nxt
        frames_launched+=len(nodes)
        # walk the truth
        nd=root
        while True:
            accept(nd['f'])
            f=nd['f']+1
            if f>=n: pos=n; break
            qn=frame_start(rc,gop,nmb,vbv,dfb,10,50,(f%gop)==0); qp_exact=qn
            assert qn==tr[f][1]
            c=[c for c in nd['ch'] if c['qp']==qn]
            if not c: pos=f; break
            nd=c[0]
    return launches,frames_launched
if __name__=='__main__':
    for (w,h,n,gop,kbps) in [(1920,1080,60,30,4000),(1920,1080,240,30,4000),(352,288,300,30,500)]:
        tr=truth(w,h,n,gop,kbps)
        for pol in [dict(depth=6,hedge_levels=1,hedge_pos=99,hedges=(-1,1,-2,2)),   # ~ today, but leaves continue
                    dict(depth=6,hedge_levels=1,hedge_pos=1,hedges=(-1,1,-2,2)),
                    dict(depth=6,hedge_levels=1,hedge_pos=2,hedges=(-1,1,-2,2)),
                    dict(depth=6,hedge_levels=2,hedge_pos=2,hedges=(-1,1)),
                    dict(depth=8,hedge_levels=1,hedge_pos=3,hedges=(-1,1)),
                    dict(depth=6,hedge_levels=1,hedge_pos=3,hedges=(-1,1,-2,2)),
                    dict(depth=4,hedge_levels=2,hedge_pos=3,hedges=(-1,1,-2,2)),
                    ]:
            for cap in (28,48):
                print(w,h,n,kbps,pol,cap,run(tr,w,h,gop,kbps,pol,cap))
