"""Host-side cost of one BatchedDMPEnv.step() call at a small batch (GPU box): python tools/step_overhead.py"""
import ctypes as C, os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snac_amd import BatchedDMPEnv
env = BatchedDMPEnv(2, True, 4096, seed=1); env.reset()
N=4096
a = torch.randint(0,5,(N,),dtype=torch.int8,device="cuda"); k = torch.randint(1,4,(N,),dtype=torch.int8,device="cuda")
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print("env.step(a,k)            %.2f us" % t(lambda: env.step(a,k,auto_reset=True)))
print("env.step() rng           %.2f us" % t(lambda: env.step(auto_reset=True)))
bufs = (torch.empty((N,51),dtype=torch.float64,device="cuda"), torch.empty(N,dtype=torch.float32,device="cuda"), torch.empty(N,dtype=torch.uint8,device="cuda"))
print("env.step(a,k,out=bufs)   %.2f us" % t(lambda: env.step(a,k,auto_reset=True,out=bufs)))
obs=torch.empty((N,51),dtype=torch.float64,device="cuda"); r=torch.empty(N,dtype=torch.float32,device="cuda"); d=torch.empty(N,dtype=torch.uint8,device="cuda")
L=env._lib; desc=C.byref(env._desc); st=C.byref(env._state)
po,pr,pd,pa,pk=[C.c_void_p(x.data_ptr()) for x in (obs,r,d,a,k)]
stream=env._stream()
cnt=[0]
def raw():
    cnt[0]+=1
    L.snac_step(desc, st, cnt[0], pa, pk, 1, po, pr, pd, stream)
print("raw ctypes snac_step     %.2f us" % t(raw))
print("torch.empty x3           %.2f us" % t(lambda: (torch.empty((N,51),dtype=torch.float64,device="cuda"), torch.empty(N,dtype=torch.float32,device="cuda"), torch.empty(N,dtype=torch.uint8,device="cuda"))))
print("_stream()                %.2f us" % t(lambda: env._stream()))
def ctx():
    with torch.cuda.device(env.device): pass
print("cuda.device ctx          %.2f us" % t(ctx))
print("_i8 x2                   %.2f us" % t(lambda: (env._i8(a,(N,),"a"), env._i8(k,(N,),"k"))))
x = torch.zeros(8, device="cuda")
print("torch add_ (tiny kernel) %.2f us" % t(lambda: x.add_(1)))
