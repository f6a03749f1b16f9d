#!/usr/bin/env python3
"""Static check of a kernel's ISA for registers touched while an LDS read into them may still be in flight.

The hand-ordered k-loops (conv3x3_wino.hip, second generation) request fragments with `ds_read_b128` inside `asm volatile`
statements and wait for them with counted `s_waitcnt lgkmcnt(N)` statements later: the compiler does not know that the
destination registers are not valid yet, so a register copy / spill / reuse it places between the request and the wait
would read (or be overwritten by) a value that arrives later - a timing-dependent wrong result.  This walks the kernel
linearly (LDS operations return in order; `lgkmcnt(N)` leaves the N youngest outstanding; scalar memory reads count in the
same counter, out of order: any of them outstanding makes only lgkmcnt(0) reliable, as the hardware documents) and reports
every instruction that reads or writes a vector register with an outstanding LDS read into it.  Conditional branch targets
reset nothing: the walk is conservative for straight-line loop bodies, which is what these kernels are; behind an unconditional
branch or s_endpgm the walk starts afresh (out-of-line blocks).

    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o k.s file.hip
    tools/isa_inflight_check.py k.s <mangled kernel name>
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def regs(s):
    out = set()
    for m in REG.finditer(s):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernel_lines(path, name):
    txt = open(path).read()
    a = txt.index('\n' + name + ':') + 1
    b = txt.index('.end_amdhsa_kernel', a)
    return txt[a:b].split('\n')


def check(lines):
    pending = []          # in issue order: ('lds', set(dest regs), line no) or ('smem', ...)
    findings = []
    for no, raw in enumerate(lines):
        s = raw.split(';')[0].strip()
        if not s or s.startswith('.') or s.endswith(':'):
            continue
        op = s.split()[0]
        if op in ('s_endpgm', 's_branch', 's_setpc_b64'):
            pending = []        # the next line is only reached by a jump (out-of-line blocks behind the kernel's body): not this path
            continue
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', s)
            if m:
                n = int(m.group(1))
                if any(k == 'smem' for k, _, _ in pending):
                    if n == 0:
                        pending = []
                else:
                    pending = pending[len(pending) - n:] if n else []
            continue
        touched = regs(s)
        for kind, dst, at in pending:
            if kind == 'lds' and dst & touched:
                findings.append((no, s, at, sorted(dst & touched)))
        if op.startswith('ds_read') or op.startswith('ds_load'):
            ops = s[len(op):].split(',')
            pending.append(('lds', regs(ops[0]), no))
        elif op.startswith('ds_'):
            pending.append(('lds', set(), no))
        elif op.startswith('s_load') or op.startswith('s_buffer_load') or op.startswith('s_memtime') or op.startswith('s_memrealtime'):
            pending.append(('smem', set(), no))
    return findings


def main():
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    f = check(lines)
    print(f"{sys.argv[2]}: {len(lines)} lines, {len(f)} accesses to registers with an LDS read in flight")
    for no, s, at, r in f[:200]:
        print(f"  line {no}: {s}    <- ds_read at line {at}, v{r}")
    return 1 if f else 0


if __name__ == "__main__":
    sys.exit(main())
