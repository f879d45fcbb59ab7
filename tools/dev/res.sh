#!/bin/bash
# usage: res.sh file.hip [extra flags] -> compact table of kernel resource usage
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/root/repo/include "$@" -c $f -o /tmp/w/res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)};rows.append(cur);continue
    for k in ['VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','SGPRs']:
        m=re.search(r'remark:\s+'+k+r': (\d+)',line)
        if m and cur is not None: cur[k.split(' ')[0]]=m.group(1)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    n=re.sub(r'\(.*','',n).replace('spatten::','').replace('void ','')
    print(f\"{n:90s} V{r.get('VGPRs')} A{r.get('AGPRs')} S{r.get('SGPRs')} scr{r.get('ScratchSize')} occ{r.get('Occupancy')} lds{r.get('LDS')}\")
"
