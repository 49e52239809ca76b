#!/usr/bin/env python3
"""Static scan of hipcc's assembly for MFMA loops whose `s_waitcnt vmcnt(N)` drain the load queue (development aid, round 4).

    cd optistate_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s gru_kernels.hip
    python tools/isa_wait_scan.py /tmp/k.s

For every innermost loop with at least six MFMAs: MFMA / vector-load / ds_read counts and the vmcnt waits inside it.  A software
pipeline DEPTH k-pairs deep should show counts near DEPTH x loads-per-k-pair (e.g. vmcnt(23), vmcnt(15)); `vmcnt(2)`, `(1)`, `(0)` at the
top of every iteration means the wait-count pass lost track (conditional requests, a guarded prologue merging into the loop header,
or a prologue the scheduler re-ordered) and the prefetch distance has collapsed to one k-pair: profiles/r04_layer_timestamps.md."""
import re,sys
path=sys.argv[1]
lines=open(path).read().splitlines()
# split kernels
starts=[i for i,l in enumerate(lines) if re.match(r'^_Z\w+:\s*;', l) or re.match(r'^_Z\w+:$', l)]
for si,s in enumerate(starts):
    e=next(i for i in range(s,len(lines)) if lines[i].strip().startswith('s_endpgm'))
    name=lines[s].split(':')[0]
    body=lines[s:e]
    labels={m.group(1):i for i,l in enumerate(body) if (m:=re.match(r'^(\.LBB\d+_\d+):',l))}
    # find innermost loops: backward branches
    loops=[]
    for i,l in enumerate(body):
        m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<i:
            loops.append((labels[m.group(1)],i))
    for lo,hi in loops:
        seg=body[lo:hi+1]
        nm=sum('v_mfma' in x for x in seg)
        if nm<6: continue
        # innermost only
        if any(lo<l2 and h2<hi for l2,h2 in loops if (l2,h2)!=(lo,hi) and sum('v_mfma' in x for x in body[l2:h2+1])>=6): continue
        waits=[x.strip() for x in seg if 's_waitcnt' in x and 'vmcnt' in x]
        nload=sum(('buffer_load' in x or 'global_load' in x) for x in seg)
        print(f"{name[:60]:60s} loop {lo}-{hi}: mfma {nm:3d} vmem-loads {nload:3d} lds {sum('ds_read' in x for x in seg):3d} waits {waits}")
