import json, os, sys, time
sys.path[:0]=['/root/repo','/root/repo/python-paillier_amd']
import numpy as np, torch
from phe import _native as native
g=json.load(open('/root/repo/tests/golden/paillier_2048.json')); H=lambda k:int(g[k],16)
ctx=native.Context(H('n'),H('p'),H('q'),H('hp'),H('hq'),H('p_inverse'),n_limbs=64)
dev=torch.device('cuda',0); B=1<<20
gen=torch.Generator(device=dev); gen.manual_seed(1)
m=torch.randint(-2**31,2**31,(B,64),dtype=torch.int32,device=dev,generator=gen); r=torch.randint(-2**31,2**31,(B,64),dtype=torch.int32,device=dev,generator=gen)
m[:,63]=0; r[:,63]&=0x3fffffff; r[:,0]|=1
c=torch.empty((B,128),dtype=torch.int32,device=dev); out=torch.empty((B,64),dtype=torch.int32,device=dev)
st=torch.cuda.current_stream().cuda_stream
ctx.encrypt_dev(m.data_ptr(),r.data_ptr(),c.data_ptr(),B,st); torch.cuda.synchronize()
def run(chunk):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for lo in range(0,B,chunk):
        k=min(chunk,B-lo)
        ctx.decrypt_dev(c.data_ptr()+lo*512, out.data_ptr()+lo*256, k, st)
    torch.cuda.synchronize(); return time.perf_counter()-t0
run(B)
for chunk in (B, 524288, 262144, 131072, 65536, B, 131072):
    t=min(run(chunk) for _ in range(2)); print(chunk, round(B/t), round(t*1e3,1))
def run_enc(chunk):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for lo in range(0,B,chunk):
        k=min(chunk,B-lo)
        ctx.encrypt_dev(m.data_ptr()+lo*256, r.data_ptr()+lo*256, c.data_ptr()+lo*512, k, st)
    torch.cuda.synchronize(); return time.perf_counter()-t0
for chunk in (B, 131072, 65536):
    t=run_enc(chunk); print('enc', chunk, round(B/t), round(t*1e3,1))
