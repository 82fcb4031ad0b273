import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
from cudasift_amd import capi
from synth import descriptors_to_points, synth_descriptors
ctx=capi.Context(0); ctx.set_options(quiet=1)
for n1,n2 in ((12512,100000),(100000,100000)):
    a=ctx.upload(descriptors_to_points(synth_descriptors(n1,12345),capi.POINT_DTYPE))
    b=ctx.upload(descriptors_to_points(synth_descriptors(n2,12346),capi.POINT_DTYPE))
    ctx.profile_enable(True)
    for _ in range(3): capi.check(capi.lib().misift_match(ctx.h,a.ptr,n1,b.ptr,n2),"m")
    ctx.sync(); ctx.profile_reset()
    reps=10 if n1<50000 else 4
    for _ in range(reps): capi.check(capi.lib().misift_match(ctx.h,a.ptr,n1,b.ptr,n2),"m")
    ctx.sync()
    p=ctx.profile_read()
    ms=p["match_mfma"]["total_ms"]/p["match_mfma"]["calls"]
    print("LIB=%s CHUNKS=%s  %d x %d: match_kernel %.3f ms  frac %.4f"%(os.path.basename(os.environ.get("MISIFT_LIB","intree")),os.environ.get("MISIFT_MATCH_CHUNKS","-"),n1,n2,ms,256.0*n1*n2/(ms*1e-3)/157.3e12),flush=True)
    a.free(); b.free()
