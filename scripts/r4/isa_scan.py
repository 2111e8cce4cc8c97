"""largest run of global loads before an s_waitcnt vmcnt, per kernel instantiation, in a device assembly file (round 4):
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iparelagmc_amd/csrc --cuda-device-only -S parelagmc_amd/csrc/kernels.hip -o k.s
  python scripts/r4/isa_scan.py k.s vc_ "<32"
(a wait that only retires OLDER loads also ends a run, so a kernel with interleaved waits reads low: look at the assembly then)"""
import re, subprocess, sys
lines=open(sys.argv[1]).read().split('\n')
starts=[(i,l.split(':')[0]) for i,l in enumerate(lines) if re.match(r'^_ZN3pmc\w+:', l)]
pat=sys.argv[2]; nbtag=sys.argv[3]
for (i,name) in starts:
    if pat not in name: continue
    j=i; best=0; cur=0
    while j < len(lines) and 's_endpgm' not in lines[j]:
        l=lines[j]
        if re.search(r'global_load_dwordx[24]|global_load_dword ', l):
            cur+=1; best=max(best,cur)
        elif 'vmcnt' in l:
            cur=0
        j+=1
    k=j; vg=''
    while k < len(lines) and k < j+200:
        if 'NumVgprs' in lines[k]:
            vg=lines[k].strip(); break
        k+=1
    d=subprocess.run(['c++filt', name],capture_output=True,text=True).stdout.strip()
    m=re.match(r'void pmc::(\w+<[^>]*>)', d)
    if m and nbtag in m.group(1): print(best, m.group(1), vg)
