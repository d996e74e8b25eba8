#!/usr/bin/env python3
"""Steady-state K loop of the 256 x 256 GEMM kernels in a `hipcc -S --cuda-device-only` listing: length, MFMAs, v_readfirstlane / waterfall loops
(a buffer descriptor that ended up in VGPRs: guide T20), v_readlane (spilled SGPRs), SALU / VALU counts.  A healthy loop: 378 lines, 64 MFMAs,
0 waterfalls, 0 readlanes, 0 VALU.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o x.s sprc_amd/csrc/gemm_f16.hip; python tools/kloop_stat.py x.s"""
import sys,re
def stat(path, name):
    txt=open(path).read().split('\n')
    idx=[k for k,l in enumerate(txt) if l.startswith(name+':')]
    if not idx: print('no kernel', name); return
    i=idx[0]; j=i
    while 's_endpgm' not in txt[j]: j+=1
    body=txt[i:j]
    labels={}
    for k,l in enumerate(body):
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=k
    best=None
    for k,l in enumerate(body):
        m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<k:
            a=labels[m.group(1)]; seg=body[a:k]
            nm=sum('v_mfma' in x for x in seg)
            if nm>=32 and (best is None or nm<=best[0]):
                best=(nm,a,k)
    nm,a,k=best; seg=body[a:k]
    print(path.split('/')[-1], name[20:60], 'lines',len(body),'| steady loop: len',k-a,'mfma',nm,'readfirstlane',sum('v_readfirstlane' in x for x in seg),'waterfall',sum('s_cbranch_execnz' in x for x in seg),'readlane',sum('v_readlane' in x for x in seg),'salu',sum(x.strip().startswith('s_') for x in seg),'valu',sum(x.strip().startswith('v_') and 'mfma' not in x for x in seg))
for path in sys.argv[1:]:
    for name in ['_ZN4sprc16gemm_anti_kernelIDF16_DF16_Li0ELb0ELb0ELb0EEEvNS_10GemmParamsE','_ZN4sprc16gemm_anti_kernelIDF16_DF16_Li1ELb0ELb0ELb0EEEvNS_10GemmParamsE','_ZN4sprc16gemm_anti_kernelIDF16_fLi0ELb0ELb0ELb0EEEvNS_10GemmParamsE']:
        stat(path,name)
