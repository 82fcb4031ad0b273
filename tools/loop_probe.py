import sys, time, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from cudasift_amd import capi
W,H,B,MAXP=1920,1080,64,32768
dev=torch.device("cuda",0)
ctx=capi.Context(0, torch.cuda.current_stream().cuda_stream); ctx.set_options(quiet=1)
frames=bench.gen_frames_torch(torch,B,0,dev)
S=capi.scratch_floats(W,H,5,False)
scratch=torch.empty((B*S,),dtype=torch.float32,device=dev)
pts=torch.zeros((B*MAXP*576,),dtype=torch.uint8,device=dev)
packed=[torch.empty((B*MAXP*576,),dtype=torch.uint8,device=dev) for _ in range(3)]
cnts=[torch.zeros((2*B+1,),dtype=torch.int32,device=dev) for _ in range(3)]
L=capi.lib()
def run(fn,K=20):
    for k in range(3): fn(k)
    torch.cuda.synchronize(); t=time.perf_counter()
    for k in range(K): fn(k)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/K*1e3
def f_async(k):
    capi.check(L.misift_extract_batch_async(ctx.h, frames.data_ptr(), B, H*W, W,H,W,5,1.0,3.0,0.0, scratch.data_ptr(), pts.data_ptr(), MAXP, cnts[k%3].data_ptr()),"a")
def f_packed(k):
    s=k%3
    capi.check(L.misift_extract_batch_packed_async(ctx.h, frames.data_ptr(), B, H*W, W,H,W,5,1.0,3.0,0.0, scratch.data_ptr(), pts.data_ptr(), MAXP, cnts[s].data_ptr(), cnts[s][B:].data_ptr(), packed[s].data_ptr()),"p")
def f_packed_ev(k):
    f_packed(k); ev=torch.cuda.Event(); ev.record()
cnt_host=(C.c_int*B)()
def f_sync(k):
    capi.check(L.misift_extract_batch(ctx.h, frames.data_ptr(), B, H*W, W,H,W,5,1.0,3.0,0.0, scratch.data_ptr(), pts.data_ptr(), MAXP, cnt_host),"s")
print("sync            %.4f ms/step" % run(f_sync))
print("async           %.4f ms/step" % run(f_async))
print("packed async    %.4f ms/step" % run(f_packed))
print("packed + event  %.4f ms/step" % run(f_packed_ev))
comm=torch.cuda.Stream(device=dev, priority=-1)
pend=[]
def f_pipe(k):
    s=k%3
    f_packed(k); ev=torch.cuda.Event(); ev.record(); pend.append((s,ev))
    if len(pend)>2:
        s0,e0=pend.pop(0)
        with torch.cuda.stream(comm):
            comm.wait_event(e0); c=cnts[s0][:B].cpu()
print("packed + lag-2 count readback on comm stream %.4f ms/step" % run(f_pipe))
free=[None,None,None]
pend2=[]
def f_pipe2(k):
    s=k%3
    if free[s] is not None: torch.cuda.current_stream().wait_event(free[s])
    f_packed(k); ev=torch.cuda.Event(); ev.record(); pend2.append((s,ev))
    if len(pend2)>2:
        s0,e0=pend2.pop(0)
        with torch.cuda.stream(comm):
            comm.wait_event(e0); c=cnts[s0][:B].cpu()
            fe=torch.cuda.Event(); fe.record(comm); free[s0]=fe
print("... + compute stream waits for the slot's free event   %.4f ms/step" % run(f_pipe2))
from cudasift_amd.dist import RecordGather
import torch.distributed as dist
g=RecordGather(dist, torch, 0, 1, dev, dst=0, nslots=3)
def f_bench(k):
    s=k%3
    fe=g.free_event(s)
    if fe is not None: torch.cuda.current_stream().wait_event(fe)
    f_packed(k); ev=torch.cuda.Event(); ev.record(); g.post(s, cnts[s][:B], packed[s], ev)
    if k>=2: g.complete((k-2)%3)
print("bench.py loop body (RecordGather, world 1)              %.4f ms/step" % run(f_bench))
