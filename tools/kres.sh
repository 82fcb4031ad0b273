#!/bin/bash
# Developer aid: register / LDS / spill figures of every kernel of one source file.
#   tools/kres.sh kernels_dog.hip [-DFOO=1 ...]
cd "$(dirname "$0")/.."
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Icudasift_amd/csrc \
  -Wno-unused-result -Wno-unused-value "$@" -c cudasift_amd/csrc/$src -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ("VGPRs","AGPRs","SGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","LDS Size \[bytes/block\]","VGPR Spill","SGPR Spill"):
        m=re.search(r"remark: .*?   +"+k+r": (\d+)",l) or re.search(k+r": (\d+)",l)
        if m and cur: rows[cur][k.split(" [")[0].replace("\\","")]=m.group(1)
import subprocess
for k,v in rows.items():
    name=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip()
    name=re.sub(r"\(.*","",name)[:70]
    print("%-70s %s"%(name," ".join("%s=%s"%(a.split(" ")[0],b) for a,b in v.items())))
'
