import sys, time, threading
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle as O
import sirius_amd as S
cid=0
b=O.make_bases(cid,5,8)
sc=O.ints_to_mont(0,[(i+3)*0x123456789abcdef123456789abcdef % (2**250) for i in range(6)])
exp=None
def ref():
    acc=b[7]
    for i in range(6): acc=O.point_add(cid,acc,O.point_mul(cid,sc[i],b[i]))
    return acc
exp=ref()
t=time.time()
for _ in range(200):
    assert np.array_equal(S.point_lincomb(cid,b[7],b[:6],sc),exp)
print("lincomb 6 pts: %.1f us"%((time.time()-t)/200*1e6))
errs=[]
def worker():
    for _ in range(200):
        if not np.array_equal(S.point_lincomb(cid,b[7],b[:6],sc),exp): errs.append(1)
th=[threading.Thread(target=worker) for _ in range(4)]
[x.start() for x in th]; [x.join() for x in th]
print("concurrent callers ok:", not errs)
for n in (1, 2, 6):
    t=time.time()
    for _ in range(100): S.point_lincomb(cid,b[7],b[:n],sc[:n])
    print(n, "pts: %.1f us"%((time.time()-t)/100*1e6))
