"""Per-kernel ISA census of a hipcc -S listing: small / dword / dwordx4 global loads, flat loads, stores, `s_waitcnt vmcnt(0)` vs counted waits, branches, VGPRs, scratch.
Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -S --cuda-device-only -o k.s csrc/<file>.hip; python tools/isa_census.py k.s <regex on the demangled name>
(round 6: how the serialized operand loads of the norm kernels and of the GEMM epilogue were found: a load behind a wave-uniform branch gets `s_waitcnt vmcnt(0)` right behind it)."""
import re,sys,subprocess
f=sys.argv[1]; pat=sys.argv[2]
txt=open(f).read().split('\n')
cur=None; stats={}
for l in txt:
    m=re.match(r'^(_Z\S+):',l)
    if m: cur=m.group(1); stats[cur]={'ush':0,'w0':0,'wN':0,'x4':0,'dw':0,'br':0,'flat':0,'st':0,'vgpr':None,'scratch':None}; continue
    if cur is None: continue
    st=stats[cur]
    if re.search(r'global_load_(u|s)(short|byte)',l): st['ush']+=1
    elif 'global_load_dwordx4' in l: st['x4']+=1
    elif re.search(r'global_load_dword\b',l): st['dw']+=1
    if 'flat_load' in l: st['flat']+=1
    if 'global_store' in l: st['st']+=1
    if 's_waitcnt vmcnt(0)' in l: st['w0']+=1
    elif 's_waitcnt vmcnt' in l: st['wN']+=1
    if 's_cbranch' in l: st['br']+=1
    m=re.search(r'\.vgpr_count:\s+(\d+)',l)
    m2=re.match(r'; NumVgprs: (\d+)',l)
    if m2: st['vgpr']=int(m2.group(1))
    m3=re.match(r'; ScratchSize: (\d+)',l)
    if m3: st['scratch']=int(m3.group(1))
for k,v in stats.items():
    d=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    if re.search(pat,d): print(v, d[:140])
