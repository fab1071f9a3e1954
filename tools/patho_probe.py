import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_lib as oracle
from nolzss_amd import _noLZSS as native
def fib(n):
    a,b=b"a",b"ab"
    while len(b)<n: a,b=b,b+a
    return b[:n]
cases={
 'allA_1M': b'A'*(1<<20),
 'allA_8M': b'A'*(1<<23),
 'period3_4M': (b'ACG'*((1<<22)//3+1))[:1<<22],
 'fib_4M': fib(1<<22),
 'abracadabra_x400k': b'abracadabra'*400000,
 'two_long_copies_8M': None,
}
rng=np.random.default_rng(1)
x=np.frombuffer(b'ACGT',dtype=np.uint8)[rng.integers(0,4,1<<22)].tobytes()
cases['two_long_copies_8M']=x+x
native.count_factors(b'ACGT'*1000)
for name,t in cases.items():
    t0=time.time(); z=native.count_factors(t); dt=time.time()-t0
    t1=time.time(); zo=oracle.count_factors(t); do=time.time()-t1
    print(f"{name:24s} n={len(t):9d} gpu {dt*1e3:9.1f} ms z={z}  oracle {do:6.1f}s z={zo} {'OK' if z==zo else 'MISMATCH'}", flush=True)
