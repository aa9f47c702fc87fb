#!/usr/bin/env python3
"""List-scheduling model of the composite kernels' ragged end: 8160 tiles of C3-like work on 1024 SIMDs x 5 wave slots,
a SIMD's throughput by resident waves taken from the forced-occupancy sweep (1.00 / 0.93 / 0.78 / 0.74 / 0.60 for 5 .. 1 waves),
workgroups dealt to the slot that frees first.  Question asked: does splitting tiles into half-tiles (two waves per tile: + 13 %
instructions for the split ones) shorten the kernel?  Answer of the model: longest-first over whole tiles runs at 0.92 of the
ideal, the best split (lightest 30 %) at 0.93 -- the instruction overhead eats the balance gained; not built.  CPU only.
"""
import numpy as np, heapq, sys
rng=np.random.default_rng(1)
RATE={0:0,1:0.60,2:0.74,3:0.78,4:0.93,5:1.0,6:1.0}
def simulate(units, nsimd=1024, slots=5, rate=RATE):
    # units: list of work in dispatch order. returns makespan (work units / full-rate)
    res=[[] for _ in range(nsimd)]  # remaining work per resident wave
    tlast=np.zeros(nsimd)
    t=0.0; nxt=0; n=len(units)
    # initial fill round-robin
    for s in range(slots):
        for i in range(nsimd):
            if nxt<n: res[i].append(units[nxt]); nxt+=1
    heap=[]
    def push(i):
        k=len(res[i])
        if k: 
            per=rate[k]/k
            heapq.heappush(heap,(tlast[i]+min(res[i])/per,i,ver[i]))
    ver=[0]*nsimd
    for i in range(nsimd): push(i)
    busy=np.zeros(7)
    while heap:
        tt,i,v=heapq.heappop(heap)
        if v!=ver[i]: continue
        k=len(res[i]); per=rate[k]/k
        dt=tt-tlast[i]
        busy[k]+=dt
        res[i]=[r-dt*per for r in res[i]]
        # remove finished
        j=int(np.argmin(res[i])); res[i].pop(j)
        tlast[i]=tt; t=tt
        if nxt<n: res[i].append(units[nxt]); nxt+=1
        ver[i]+=1; push(i)
    return t, busy/ (t*nsimd)
# tile work distribution ~ C3: p10 375 p50 453 p90 539
w=rng.normal(455,64,8160).clip(60,900)
w[:200]=rng.uniform(50,300,200)  # some light edge tiles
ideal=w.sum()/1024
def run(name, units):
    t,b=simulate(units)
    print(f"{name:40s} makespan {t:8.1f}  ideal {ideal:8.1f}  eff {ideal/t:.3f}  occ shares 5..1: {b[5]:.2f} {b[4]:.2f} {b[3]:.2f} {b[2]:.2f} {b[1]:.2f}")
ws=np.sort(w)[::-1]
run("LPT full tiles", list(ws))
run("tile order (random)", list(w))
for frac in (0.1,0.2,0.3,0.4,0.5,1.0):
    for ov in (1.13,):
        nsplit=int(frac*len(ws))
        full=ws[:len(ws)-nsplit]; sp=ws[len(ws)-nsplit:]
        halves=np.repeat(sp*ov/2,2)
        units=list(full)+list(np.sort(halves)[::-1])
        t,b=simulate(units)
        print(f"split lightest {frac:.0%} (overhead {ov}) makespan {t:8.1f} eff vs ideal-unsplit {ideal/t:.3f}")
# split heaviest instead -> halves placed by size in LPT order
for frac in (0.2,0.4,0.6):
    nsplit=int(frac*len(ws)); sp=ws[:nsplit]; full=ws[nsplit:]
    units=np.concatenate([full, np.repeat(sp*1.13/2,2)]); units=np.sort(units)[::-1]
    t,b=simulate(list(units)); print(f"split heaviest {frac:.0%}: makespan {t:8.1f} eff {ideal/t:.3f}")
