#!/usr/bin/env python3
"""tools/isa_count.py <asm file> <kernel-name substring>: static instruction mix of one kernel,
whole body and per basic block (label), from hipcc --save-temps output (tools/isa_dev.sh)."""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
inside = False
blocks = collections.OrderedDict()
cur = None
for line in open(path):
    m = re.match(r"^(\S+):", line)
    if m:
        lab = m.group(1)
        if lab.startswith("_ZN3pds16stft_wave_kernel"):
            inside = want in lab
            if inside:
                cur = "entry"
                blocks[cur] = collections.Counter()
            continue
        if inside and lab.startswith(".LBB"):
            cur = lab
            blocks[cur] = collections.Counter()
        continue
    if not inside:
        continue
    if line.strip().startswith(".size") or line.strip().startswith("s_endpgm"):
        pass
    t = line.strip().split()
    if not t or t[0].startswith((";", ".")):
        continue
    op = t[0]
    if op.startswith("v_mfma"):
        k = "mfma"
    elif op.startswith("v_"):
        k = "valu"
    elif op.startswith("ds_"):
        k = "lds"
    elif op.startswith(("global_load", "buffer_load")):
        k = "vmem_rd"
    elif op.startswith(("global_store", "buffer_store")):
        k = "vmem_wr"
    elif op.startswith("scratch_"):
        k = "scratch"
    elif op.startswith("s_waitcnt"):
        k = "waitcnt"
    elif op.startswith("s_nop"):
        k = "nop"
    elif op.startswith(("s_load", "s_buffer_load")):
        k = "smem"
    elif op.startswith("s_"):
        k = "salu"
    else:
        k = "other"
    blocks[cur][k] += 1
tot = collections.Counter()
for lab, c in blocks.items():
    tot.update(c)
    n = sum(c.values())
    if n >= 25:
        print(f"{lab:12s} {n:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("TOTAL", sum(tot.values()), dict(tot))
