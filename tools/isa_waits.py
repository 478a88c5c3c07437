"""Loads, vmcnt waits and branches per kernel from `hipcc -S` of every csrc/*.hip: a kernel whose wait count approaches its
load count issues its loads one round trip at a time ("conditional loads serialise").
   python tools/isa_waits.py"""
import re,subprocess,sys,glob,os
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'esr_nerf_amd', 'csrc', '*.hip'))):
    out='/tmp/k.s'
    subprocess.run(['/opt/rocm/bin/hipcc','-O3','-std=c++17','-fPIC','--offload-arch=gfx950','-fvisibility=hidden','-fno-fast-math','-I' + os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'include'),'-S','--cuda-device-only','-o',out,f],stderr=subprocess.DEVNULL)
    name=None; stats={}
    for line in open(out):
        m=re.match(r'^(_Z\w+):',line)
        if m: name=m.group(1); stats[name]=dict(ld=0,w0=0,w=0,br=0,v=0); continue
        if line.startswith('.Lfunc_end'): name=None
        if not name: continue
        st=stats[name]
        if 'global_load' in line or 'buffer_load' in line or 'flat_load' in line: st['ld']+=1
        if 's_waitcnt' in line and 'vmcnt(0)' in line: st['w0']+=1
        if 's_waitcnt' in line and 'vmcnt' in line: st['w']+=1
        if 's_cbranch' in line: st['br']+=1
        if re.match(r'\s+v_',line): st['v']+=1
    for n,st in stats.items():
        if st['ld']>=8:
            d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()[:100]
            print(f"{os.path.basename(f):16s} ld {st['ld']:4d} vmcnt0 {st['w0']:4d} vmcnt {st['w']:4d} br {st['br']:4d} valu {st['v']:5d}  {d}")
