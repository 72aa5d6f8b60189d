import numpy as np, sys
sys.path.insert(0,'.')
rng=np.random.default_rng(1)
K=1_000_000; n=390_000
w=np.arange(1,K+1,dtype=np.float64)**-1.1; p=w/w.sum(); cdf=np.cumsum(p)
keys=np.searchsorted(cdf, rng.random(n))
hk=(keys.astype(np.uint64)*np.uint64(0x9E3779B97F4A7C15))>>np.uint64(20)
def sim(policy, E=1024, probe=8, doorbits=32768):
    tbl=-np.ones(E,dtype=np.int64); cnt={} ; door=np.zeros(doorbits,dtype=np.uint8); hits=0; fill=0
    for k,h in zip(keys,hk):
        h=int(h); e=(h>>20)&(E-1); found=False; free=-1
        for q in range(probe):
            s=(e+q)&(E-1)
            if tbl[s]==k: found=True; break
            if tbl[s]<0: free=s; break
        if found: hits+=1; continue
        if free<0: continue
        b=(h>>3)&(doorbits-1)
        need=policy(fill/E)
        if door[b]+1>=need:
            tbl[free]=k; fill+=1; hits+=1
        else:
            door[b]+=1
    return hits/n
print("first come", sim(lambda f:1))
print("2nd appearance", sim(lambda f:2))
print("3rd", sim(lambda f:3))
print("adaptive 2/3/4", sim(lambda f: 2 if f<0.5 else (3 if f<0.8 else 4)))
print("adaptive 2/4/6", sim(lambda f: 2 if f<0.5 else (4 if f<0.8 else 6)))
print("2nd, probe 16", sim(lambda f:2, probe=16))
print("2nd, E=2048", sim(lambda f:2, E=2048))
print("ideal top-1024 share", p[:1024].sum(), "top-2048", p[:2048].sum())
