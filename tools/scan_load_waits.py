#!/usr/bin/env python3
"""Scan every kernel of libpds_amd.so for "load, then wait for it at once" pairs: a global / scratch / LDS read followed
within two instructions by a full s_waitcnt (vmcnt(0) / lgkmcnt(0)).  A long run of them is a kernel that pays one
memory round trip per row (how the float64 + pre-emphasis kernels of the 8-lane geometries and the first two-wave
short-integration kernel were found).  python tools/scan_load_waits.py [top]"""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rt", os.path.join(ROOT, "tools", "resource_table.py"))
rt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rt)
LIB = os.environ.get("PDS_AMD_LIB", os.path.join(ROOT, "pydrobert-speech_amd", "csrc", "libpds_amd.so"))
top = int(sys.argv[1]) if len(sys.argv) > 1 else 25
objdump = os.path.join(rt.LLVM, "llvm-objdump")
res = []
with tempfile.TemporaryDirectory() as tmp:
    for i, elf in enumerate(rt.code_objects(LIB)):
        fn = os.path.join(tmp, f"co{i}.elf")
        open(fn, "wb").write(elf)
        out = subprocess.run([objdump, "-d", "--no-show-raw-insn", fn], capture_output=True, text=True).stdout
        cur, ins = None, []

        def flush():
            if cur and ins:
                pairs = 0
                for j, x in enumerate(ins):
                    if x.startswith(("ds_read", "scratch_load", "global_load", "flat_load")):
                        if any(y.startswith("s_waitcnt") and ("lgkmcnt(0)" in y or "vmcnt(0)" in y) for y in ins[j + 1 : j + 3]):
                            pairs += 1
                res.append((pairs, sum(1 for x in ins if x.startswith("scratch_")), len(ins), cur))

        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                flush()
                cur, ins = m.group(1), []
                continue
            t = line.strip().split("//")[0].strip()
            if t and not t.startswith("Disassembly") and ":" not in t.split()[0]:
                ins.append(t)
        flush()
res.sort(reverse=True)
print("pairs scratch instrs kernel")
for pairs, scr, n, name in res[:top]:
    print(pairs, scr, n, (rt.short(name) if "stft_wave_kernel" in name else name)[:170])
