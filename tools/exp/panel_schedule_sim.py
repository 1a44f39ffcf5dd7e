#!/usr/bin/env python3
"""Two-resource simulation of the row-panel schedule of bof_flash_gemm: one disk (reads in fetch order;
writes of finished C panels; separate pure and mixed rates), one GPU (the launch list in order).  Used to choose the ramp
group and to judge a wavefront order of the ramp; reproduces the measured cfg2 / 65536^3 steps within ~5 %."""
import itertools
def sim(n=32768, blk=4096, G=4, order="lmajor", R=20e9, W=16.5e9, Wmix=6e9, Rmix=15e9, tf=148e12, verbose=False):
    Np=n//blk; Nk=Np
    pb=blk*n*4
    # fetch order + launch list
    launches=[]  # (pc, l0, l1)
    fetch=[]
    if order=="lmajor":
        for l in range(Nk):
            for pc in range(G): launches.append((pc,l,l+1))
    else:  # wavefront
        for i in range(max(G,Nk)):
            if i<G:
                for l in range(min(i,Nk)): launches.append((i,l,l+1))   # A_i arrives: B_0..B_{i-1} are in
            if i<Nk:
                for pc in range(min(i+1,G)): launches.append((pc,i,i+1))   # B_i arrives
    for pc in range(G,Np): launches.append((pc,0,Nk))
    seen=set()
    for (pc,l0,l1) in launches:
        if ('A',pc) not in seen: seen.add(('A',pc)); fetch.append(('A',pc))
        for l in range(l0,l1):
            if ('B',l) not in seen: seen.add(('B',l)); fetch.append(('B',l))
    # A ring: 2G slots: A panel pc can be fetched only when panel pc-2G retired (ignore: ring deep enough mostly)
    # disk model: reads in order; writes interleave: while both pending, reads get Rmix, writes Wmix
    t=0.0; arrive={}
    # first pass without writes to get launch/complete times, then iterate with write interference (simple 2-pass)
    def run(read_rate_fn):
        t=0.0; arr={}
        for f in fetch:
            dt=pb/read_rate_fn(t); t+=dt; arr[f]=t
        return arr
    arr=run(lambda t:R)
    def compute(arr):
        tg=0.0; done={}; busy=0.0
        for (pc,l0,l1) in launches:
            ready=max([arr[('A',pc)]]+[arr[('B',l)] for l in range(l0,l1)])
            st=max(tg,ready); d=2.0*blk*n*(l1-l0)*blk/tf
            tg=st+d; busy+=d
            if l1==Nk: done[pc]=tg
        return done,tg,busy
    done,tg,busy=compute(arr)
    # writes with interference: step simulation
    reads_end=max(arr.values())
    # event-driven: disk serves reads (in order) and writes (C panels in completion order)
    # iterate to fixed point: reads slowed while a write is pending
    for it in range(6):
        # build write intervals given done
        order_w=sorted(done.items(), key=lambda x:x[1])
        # simulate disk with small dt
        dt=0.0005; t=0.0; ri=0; rrem=pb; arr2={}; wq=list(order_w); wrem=0; wcur=None; wend={}
        while ri<len(fetch) or wq or wcur is not None:
            if wcur is None and wq and wq[0][1]<=t:
                wcur=wq.pop(0)[0]; wrem=pb
            reading=ri<len(fetch)
            writing=wcur is not None
            rr=(Rmix if writing else R) if reading else 0
            wr=(Wmix if reading else W) if writing else 0
            if reading:
                rrem-=rr*dt
                if rrem<=0:
                    arr2[fetch[ri]]=t+dt; ri+=1; rrem=pb
            if writing:
                wrem-=wr*dt
                if wrem<=0:
                    wend[wcur]=t+dt; wcur=None
            t+=dt
            if t>10: break
        done2,tg2,busy=compute(arr2)
        if max(abs(done2[k]-done[k]) for k in done)<0.002: done=done2; break
        done=done2
    total=max(wend.values())
    if verbose:
        print("reads end %.3f first C %.3f compute end %.3f total %.3f busy %.3f"%(max(arr2.values()),min(done.values()),tg2,total,busy))
    return total, min(done.values()), tg2
for n,blk in ((32768,4096),(65536,4096)):
    for order in ("lmajor","wave"):
        for G in (2,3,4,5,6):
            tot,fc,ce=sim(n,blk,G,order)
            print(n,order,"G",G,"total %.3f firstC %.3f computeEnd %.3f"%(tot,fc,ce))
