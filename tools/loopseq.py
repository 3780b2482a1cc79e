#!/usr/bin/env python
"""Print a compact opcode string of a kernel's ISA (M=mfma R=ds_read D=lds-dma v=VALU s=SALU) to eyeball scheduling."""
import sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':')
body = s[i:s.index('s_endpgm', i)]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(lines)
seq = []
for l in lines[lo:hi]:
    op = l.split()[0]
    if op.startswith('v_mfma'): seq.append('M')
    elif op.startswith('ds_read'): seq.append('R')
    elif op.startswith('ds_write'): seq.append('W')
    elif op.startswith('buffer_load'): seq.append('D')
    elif op.startswith('global_load') or op.startswith('buffer_load'): seq.append('L')
    elif op.startswith('global_store'): seq.append('S')
    elif op.startswith('s_waitcnt'): seq.append('[' + l.split(None, 1)[1].replace(' ', '') + ']')
    elif op.startswith('s_barrier'): seq.append('|BAR|')
    elif op.startswith('s_nop'): seq.append('n')
    elif op.startswith('s_cbranch') or l.endswith(':'): seq.append('\n{' + l[:16] + '}')
    elif op.startswith('v_'): seq.append('v')
    elif op.startswith('s_'): seq.append('s')
    else: seq.append('?' + op)
print(len(lines), 'instructions')
print(''.join(seq))
