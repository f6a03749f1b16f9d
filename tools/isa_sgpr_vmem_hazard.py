#!/usr/bin/env python3
"""Static check of a kernel's ISA for the `VALU writes SGPR -> VMEM reads that SGPR` hazard (5 wait states on gfx9 / CDNA; the
hardware does not interlock it and the compiler's hazard recogniser does not look inside `asm volatile` statements, where the
hand-ordered k-loops keep their LDS-DMA `buffer_load ... lds` instructions).  A scalar restored from a spill lane
(`v_readlane_b32 sN, vSPILL, lane`) or produced by `v_readfirstlane_b32` / a VALU compare directly in front of such a
statement is read by the DMA instruction before it has been written: the piece goes to / comes from a stale address.

    tools/isa_sgpr_vmem_hazard.py k.s <mangled kernel name>
walks the kernel linearly (fall-through order; labels do not reset the window - conservative) and lists every VMEM
instruction that reads an SGPR written by a VALU instruction fewer than 5 wait states earlier."""
import re
import sys

SREG = re.compile(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b')
NEED = 5


def sregs(s):
    out = set()
    for m in SREG.finditer(s):
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


def is_vmem(op):
    return op.startswith(('buffer_', 'global_', 'flat_', 'scratch_', 'tbuffer_', 'image_'))


def valu_sgpr_defs(op, s):
    """SGPRs a VALU instruction writes (destination operands only)."""
    if not op.startswith('v_'):
        return set()
    ops = [o.strip() for o in s[len(op):].split(',')]
    if op.startswith(('v_readlane', 'v_readfirstlane')):
        return sregs(ops[0])
    if op.startswith('v_cmp') and not op.endswith('_e32'):
        return sregs(ops[0])                        # VOP3 compare: sdst pair
    if re.match(r'v_(add|sub|subrev)_co_u32|v_(addc|subb|subbrev)_co_u32|v_mad_[iu]64_[iu]32|v_div_scale', op):
        return sregs(ops[1]) if len(ops) > 1 else set()
    return set()


def main():
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    window = []            # (wait states this instruction accounts for, sgprs it wrote as a VALU, line no, text)
    found = 0
    for no, raw in enumerate(lines):
        s = raw.split(';')[0].strip()
        if not s or s.startswith('.') or s.endswith(':'):
            continue
        op = s.split()[0]
        if is_vmem(op):
            reads = sregs(s)
            ws = 0
            for cost, defs, at, txt in reversed(window):
                if ws >= NEED:
                    break
                if defs & reads:
                    found += 1
                    print(f"  line {no}: {s}\n      reads s{sorted(defs & reads)} written {ws} wait states earlier at line {at}: {txt}")
                ws += cost
        cost = 1
        if op == 's_nop':
            cost = int(s.split()[1]) + 1
        window.append((cost, valu_sgpr_defs(op, s), no, s))
        if len(window) > 16:
            window.pop(0)
    print(f"{sys.argv[2]}: {found} VMEM reads of an SGPR inside the VALU-write hazard window")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
