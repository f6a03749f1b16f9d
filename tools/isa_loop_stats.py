#!/usr/bin/env python3
"""Instruction mix of a kernel's ISA between consecutive s_barrier instructions, and where its scratch (spill) traffic sits.
   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o k.s file.hip ; tools/isa_loop_stats.py k.s <mangled-name-substring> [first last]
prints one line per barrier-to-barrier segment (the k-loop's sub-steps show up as equal-looking lines), then every scratch access
with the loop nest it sits in; `first last`: also dump the instructions of that line range of the kernel (no asm markers)."""
import collections
import sys


def key(o):
    if 'mfma' in o: return 'mfma'
    if o.startswith('ds_read'): return 'ds_read'
    if o.startswith('ds_write'): return 'ds_write'
    if 'v_pk_add' in o: return 'pk_add'
    if 'buffer_load' in o or 'global_load_lds' in o: return 'dma'
    if 'lane' in o: return 'lane'
    if 'scratch' in o: return 'scratch'
    if o.startswith('s_waitcnt'): return 'waitcnt'
    if o.startswith('s_nop'): return 'nop'
    if o.startswith('v_'): return 'valu'
    if o.startswith('s_'): return 'salu'
    return o


def main():
    txt = open(sys.argv[1]).read()
    name = sys.argv[2]
    a = txt.index(name)
    a = txt.index('\n', a)
    b = txt.index('.end_amdhsa_kernel', a)
    lines = txt[a:b].split('\n')
    idx = [i for i, l in enumerate(lines) if l.strip() == 's_barrier']
    print(f"{len(lines)} lines, barriers at {idx}")
    for a_, b_ in zip([0] + idx, idx + [len(lines)]):
        c = collections.Counter()
        for l in lines[a_ + 1:b_]:
            l = l.strip()
            if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
                continue
            c[key(l.split()[0])] += 1
        print(f"  [{a_:5d},{b_:5d})", dict(sorted(c.items())))
    loop = ''
    for i, l in enumerate(lines):
        s = l.strip()
        if s.startswith('.LBB') and 'Loop' in s:
            loop = s
        if 'scratch_' in s:
            print(i, s, '   <-', loop[:60])
    if len(sys.argv) > 4:
        for i in range(int(sys.argv[3]), int(sys.argv[4])):
            if not lines[i].strip().startswith(';;#'):
                print(f"{i}: {lines[i]}")


if __name__ == "__main__":
    main()
