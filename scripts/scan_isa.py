"""Per kernel of a `hipcc -S --cuda-device-only` listing: number of global/buffer loads, the longest run of loads issued
without a wait in between, and the number of full `s_waitcnt vmcnt(0)` drains — a tile pass meant to keep 16 rows in flight
that shows "max run 2" has its loads serialised (guarded loads are not hoisted).  python scripts/scan_isa.py file.s"""
import re,subprocess,sys
s=open(sys.argv[1]).read()
parts=re.split(r'\n(?=_Z\S*: *; @)', s)
for p in parts:
    m=re.match(r'(_Z\S*):', p)
    if not m: continue
    body=p.split('s_endpgm')[0]
    if 'global_load' not in body and 'buffer_load' not in body: continue
    ops=re.findall(r'(global_load\w*|global_store\w*|s_waitcnt vmcnt\(\d+\)|buffer_load\w*)', body)
    run=mx=nl=0
    for o in ops:
        if o.startswith('global_load') or o.startswith('buffer_load'): run+=1; nl+=1; mx=max(mx,run)
        elif o.startswith('s_waitcnt'): run=0
    w0=sum(1 for o in ops if o=='s_waitcnt vmcnt(0)')
    short=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
    short=short.replace('(anonymous namespace)::','').replace('void ','')[:70]
    print(f"{nl:4d} loads, max run {mx:3d}, vmcnt(0) x{w0:3d}  {short}")
