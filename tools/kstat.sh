#!/bin/bash
# kstat.sh <file.hip> <kernel name substring>: device asm of one source -> registers of matching kernels,
# total instructions and the basic blocks with many register copies (tuning aid)
src=$1; tag=$2
out=/tmp/$(basename $src .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -I$(dirname $src) -o $out $src 2>&1 | grep error -A5
python3 - "$out" "$tag" <<'PY'
import re,sys
lines=open(sys.argv[1]).read().split('\n'); tag=sys.argv[2]
txt='\n'.join(lines)
for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    if tag in m.group(2): print(m.group(2)[18:84], 'scratch',m.group(3),'sgpr',m.group(4),'vgpr',m.group(5))
start=next(i for i,l in enumerate(lines) if l.startswith('_ZN') and tag in l and ': ' in l)
end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
cur='entry';cnt={};tot=0
for l in lines[start+1:end]:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=m.group(1)
    s=l.strip().split()
    if s and re.match(r'[a-z]',s[0]) and not s[0].startswith(('.',';')): tot+=1
    if s and s[0] in ('v_mov_b32_e32','v_mov_b64_e32'): cnt[cur]=cnt.get(cur,0)+1
print('first match: total instr',tot,'blocks with >8 v_mov:',{k:v for k,v in cnt.items() if v>8})
PY
