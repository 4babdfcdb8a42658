#!/bin/bash
# VGPR / LDS / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: scripts/kernel_resources.sh tinygp_amd/csrc/chol.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$(dirname $f) -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /dev/null 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rec={}
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        cur=t.split(':',1)[1].strip(); rec[cur]={}
    elif cur and ':' in t:
        k,v=t.split(':',1); rec[cur][k.strip()]=v.strip()
for f,r in rec.items():
    try: name=subprocess.run(['c++filt',f],capture_output=True,text=True).stdout.strip()
    except Exception: name=f
    name=name.replace('tgp::(anonymous namespace)::','').replace('(anonymous namespace)::','')
    name=re.sub(r'\(.*','',name)[:56]
    print(f\"{name:56s} VGPR {r.get('VGPRs'):>4} AGPR {r.get('AGPRs'):>3} SGPR {r.get('TotalSGPRs'):>3} waves/SIMD {r.get('Occupancy [waves/SIMD]'):>2} LDS {r.get('LDS Size [bytes/block]'):>7} spill {r.get('VGPRs Spill')}\")
"
