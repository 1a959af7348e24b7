import random, numpy as np
p = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MASK=(1<<29)-1
P=[(p>>(29*i))&MASK for i in range(8)]+[p>>232]
INV=np.float32(256.0/ (p/2**232))
def limbs(v): return [(v>>(29*i))&MASK for i in range(8)]+[v>>232]
def val(l): return sum(x<<(29*i) for i,x in enumerate(l))
def reduce_i64(t):
    vt = t[8] + (t[7]>>29)
    assert abs(vt>>8) < 2**31
    q = int(np.floor(np.float32(np.int32(vt>>8))*INV)) - 1
    t=[t[i]-q*P[i] for i in range(9)]
    for x in t: assert abs(x) < 2**63
    o=[];c=0
    for i in range(8):
        s=t[i]+c; o.append(s&MASK); c=s>>29
    o.append(t[8]+c)
    assert 0<=o[8]<2**32
    return o
def lincomb(vs,cs):
    t=[sum(c*limbs(v)[i] for v,c in zip(vs,cs)) for i in range(9)]
    o=reduce_i64(t); r=val(o)
    assert r%p == sum(c*v for v,c in zip(vs,cs))%p
    assert 0.98*p < r < 2.02*p, r/p
    return r
random.seed(1)
def ev(co,x): return sum(c*x**i for i,c in enumerate(co))%p
for trial in range(2000):
    # degree 2
    co=[random.randrange(p) for _ in range(3)]
    lazy=lambda v,m: v+random.randrange(0,m)*p  # congruent representative
    f=lambda x,m=2: ev(co,x)+random.randrange(0,m)*p
    P1,P2,c=f(1),f(2),co[2]+random.randrange(2)*p
    assert lincomb([P1,P2,c],[-1,2,2])%p==ev(co,3)
    assert lincomb([P1,P2,c],[-2,3,6])%p==ev(co,4)
    co=[random.randrange(p) for _ in range(5)]
    Q=[ev(co,x)+random.randrange(2)*p for x in (1,2,3,4)]+[co[4]+random.randrange(2)*p]
    for x,cs in ((5,[-1,4,-6,4,24]),(6,[-4,15,-20,10,120]),(7,[-10,36,-45,20,360]),(8,[-20,70,-84,35,840])):
        assert lincomb(Q,cs)%p==ev(co,x)
    co=[random.randrange(p) for _ in range(9)]
    w=[ev(co,x)+random.randrange(2)*p for x in range(1,9)]; c=co[8]+random.randrange(2)*p
    for x in range(9,16):
        nw=lincomb(w+[c],[-1,8,-28,56,-70,56,-28,8,40320])
        assert nw%p==ev(co,x)
        w=w[1:]+[nw]
# extreme values: all at 2.01p / 0.99p mixes
for trial in range(2000):
    w=[random.choice([p-1,2*p+p//100,p+1, int(0.99*p)]) for _ in range(9)]
    lincomb(w,[-1,8,-28,56,-70,56,-28,8,40320])
    lincomb(w[:5],[-20,70,-84,35,840])
# final sum: 128 values < 1.06p
for trial in range(200):
    vs=[random.randrange(int(1.06*p)) for _ in range(128)]
    t=[sum(limbs(v)[i] for v in vs) for i in range(9)]
    assert val(reduce_i64(t))%p==sum(vs)%p
print("ok", float(INV))
