"""Opcode histogram of the largest loop of a kernel in a gfx950 code object disassembly.
usage:  llvm-objdump --offloading jolt-atlas_amd/build/atlas_hip.o; llvm-objdump -d <extracted .gfx950 file> > atlas_hip.s
        python tools/isa_loop.py atlas_hip.s k_dot_bind_eval2_f9 Lb0 ChanIo       (substrings of the mangled name)"""
import re,collections,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2:]
parts=re.split(r'\n[0-9a-f]+ <([^>]+)>:\n', s)
def addr(l):
    m=re.search(r'//\s*([0-9A-Fa-f]+):',l); return int(m.group(1),16) if m else None
for i in range(1,len(parts),2):
    name=parts[i]
    if all(p in name for p in pat):
        lines=[l for l in parts[i+1].split('\n') if l.strip() and addr(l) is not None]
        best=None
        for idx,l in enumerate(lines):
            m=re.match(r'\s+s_c?branch\w*\s+(\d+)',l)
            if m and int(m.group(1))>32768:
                span=65536-int(m.group(1))
                if best is None or span>best[0]: best=(span,idx)
        end=addr(lines[best[1]]); start=end+4-best[0]*4
        body=[l for l in lines if start<=addr(l)<=end]
        ops=collections.Counter()
        for line in body:
            m=re.match(r'\s+(\w+)',line)
            if m: ops[m.group(1)]+=1
        print(name, "loop instrs", sum(ops.values()))
        for k,v in ops.most_common(24): print(f"  {k:28s}{v}")
