"""Model of a force launch's use of the wave slots, calibrated on a per-wave trace (tools/force_trace.py writes
gpurun_out/force_trace_<n>_<theta>.npy; here: the 1M launch with one wave per group, 6 waves per SIMD, as it was at
the start of round 4).  A wave's speed depends on how many waves share its SIMD (cycles per child block with 1 .. 6
waves resident: tools/walk_probe.py and the trace itself); a wave's work = the integral of that speed over its
lifetime; the launch is replayed with the hardware's rule (a workgroup starts when a slot frees, in index order) and
then with other job orders / job lengths.  Output of the run the design was based on: profiles/r04_drain/model.md.
    python tools/drain_model.py gpurun_out/force_trace_1000000_0.5.npy"""
import numpy as np, heapq, sys
rows=np.load(sys.argv[1])
t0=rows[:,0].astype(np.int64); t1=rows[:,1].astype(np.int64)
base=t0.min(); t0=(t0-base)*0.01; t1=(t1-base)*0.01
hw=rows[:,2]; xcc=rows[:,3]&0xF
key=((xcc.astype(np.int64)*16+((hw>>12)&0xF))*16+((hw>>8)&0xF))*4+((hw>>4)&3)
uk,inv=np.unique(key,return_inverse=True)
W=len(t0)
# per-wave rate (blocks per us) with k waves on the SIMD: cycles per block
cpb={1:750,2:870,3:1040,4:1200,5:1360,6:1518}
GHZ=2300.0  # cycles per us
def rate(k): return GHZ/cpb[min(max(k,1),6)]
work=np.zeros(W)
for s in range(len(uk)):
    idx=np.nonzero(inv==s)[0]
    pts=np.unique(np.concatenate([t0[idx],t1[idx]]))
    for u,v in zip(pts[:-1],pts[1:]):
        act=idx[(t0[idx]<=u)&(t1[idx]>=v)]
        if len(act): work[act]+=(v-u)*rate(len(act))
print("blocks per wave from model: mean %.0f (walk_stats says 555)"%work.mean())

def simulate(jobs, nsimd=1024, slots=6, per_xcd=True):
    """jobs: list of (work, nwaves) in dispatch order; a job of nwaves>1 = coop WG: nwaves waves each work/nwaves*ovh on
    distinct SIMDs of the least-loaded CU... simplified: each wave is placed on the SIMD with fewest residents."""
    # state per simd: list of remaining works; last update time
    rem=[[] for _ in range(nsimd)]
    last=np.zeros(nsimd)
    heap=[]  # (finish_time, simd, version)
    ver=[0]*nsimd
    def advance(s,t):
        k=len(rem[s])
        if k:
            d=(t-last[s])*rate(k)
            rem[s]=[x-d for x in rem[s]]
        last[s]=t
    def schedule(s,t):
        ver[s]+=1
        k=len(rem[s])
        if k:
            m=min(rem[s])
            heapq.heappush(heap,(t+max(m,0)/rate(k),s,ver[s]))
    free=[(0,s) for s in range(nsimd)]  # (residents, simd) candidates
    cnt=[0]*nsimd
    t=0.0
    qi=0
    # expand jobs into waves
    waves=[]
    for w,n in jobs:
        if n==1: waves.append(w)
        else: waves.extend([w/n*1.08+3.0]*n)   # +3 blocks of level overhead
    nw=len(waves)
    import bisect
    def place(t):
        nonlocal qi
        # greedy: while there is a simd with free slot, place next wave on the least-loaded simd
        while qi<nw:
            s=min(range(nsimd), key=lambda x: cnt[x]) if False else None
            break
    # simple implementation: maintain buckets by count
    buckets=[set() for _ in range(slots+1)]
    for s in range(nsimd): buckets[0].add(s)
    def pick():
        for c in range(slots):
            if buckets[c]:
                return next(iter(buckets[c]))
        return None
    def add_wave(s,w,t):
        advance(s,t)
        buckets[cnt[s]].discard(s); cnt[s]+=1; buckets[cnt[s]].add(s)
        rem[s].append(w); schedule(s,t)
    while qi<nw:
        s=pick()
        if s is None: break
        add_wave(s,waves[qi],0.0); qi+=1
    tq=0.0
    while heap:
        ft,s,v=heapq.heappop(heap)
        if v!=ver[s]: continue
        t=ft
        advance(s,t)
        # remove finished (<=1e-9)
        keep=[x for x in rem[s] if x>1e-7]
        nd=len(rem[s])-len(keep)
        rem[s]=keep
        buckets[cnt[s]].discard(s); cnt[s]-=nd; buckets[cnt[s]].add(s)
        schedule(s,t)
        while qi<nw:
            s2=pick()
            if s2 is None: break
            add_wave(s2,waves[qi],t); qi+=1
            if qi==nw: tq=t
    return t,tq

order=np.arange(W)
# dispatch order in the real launch: by start time
disp=np.argsort(t0,kind='stable')
jobs=[(work[i],1) for i in disp]
T,tq=simulate(jobs)
print("baseline sim: span %.0f us, T_q %.0f (measured 1150 / 797)"%(T,tq))
for x in (0.1,0.2,0.3,0.4,0.5,1.0):
    nb=int(W*(1-x))
    jobs=[(work[i],1) for i in disp[:nb]]+[(work[i],4) for i in disp[nb:]]
    T,tq=simulate(jobs)
    print("tail %.0f%% coop K=4: span %.0f us T_q %.0f"%(x*100,T,tq))
for x in (0.2,0.4):
    nb=int(W*(1-x))
    jobs=[(work[i],1) for i in disp[:nb]]+[(work[i],2) for i in disp[nb:]]
    T,tq=simulate(jobs)
    print("tail %.0f%% coop K=2: span %.0f us T_q %.0f"%(x*100,T,tq))
# LPT order
lpt=np.argsort(-work)
T,tq=simulate([(work[i],1) for i in lpt]); print("LPT: span %.0f T_q %.0f"%(T,tq))
print("--- more")
for x,K in ((0.1,8),(0.2,8),(0.15,4),(0.25,4)):
    nb=int(W*(1-x))
    jobs=[(work[i],1) for i in disp[:nb]]+[(work[i],K) for i in disp[nb:]]
    T,tq=simulate(jobs); print("tail %.0f%% coop K=%d: span %.0f us T_q %.0f"%(x*100,K,T,tq))
# graded: 64% K=1, then 16% K=2, 12% K=4, 8% K=8
def graded(fr):
    jobs=[];p=0
    for f,K in fr:
        n=int(W*f); jobs+= [(work[i],K) for i in disp[p:p+n]]; p+=n
    jobs+=[(work[i],fr[-1][1]) for i in disp[p:]]
    return simulate(jobs)
print("graded 70/15/15 (1,2,4):",graded([(0.7,1),(0.15,2),(0.15,4)]))
print("graded 75/15/10 (1,4,8):",graded([(0.75,1),(0.15,4),(0.10,8)]))
# LPT + coop tail
for x in (0.1,0.2):
    nb=int(W*(1-x))
    jobs=[(work[i],1) for i in lpt[:nb]]+[(work[i],4) for i in lpt[nb:]]
    print("LPT + tail %.0f%% K=4:"%(x*100), simulate(jobs))
# ideal: total work at full rate
print("ideal (all SIMDs saturated to the end): %.0f us"%(work.sum()/1024/(6*rate(6))))
