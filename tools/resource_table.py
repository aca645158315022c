#!/usr/bin/env python3
"""Registers, scratch and spills of every kernel in libpds_amd.so, read from the gfx950 code objects' metadata.

    python tools/resource_table.py [path/to/libpds_amd.so] [--all] > profiles/<tag>_resource_table.txt

The library's .hip_fatbin section holds one clang offload bundle per translation unit; each bundle's gfx950 entry is
an ELF whose AMDGPU metadata note lists, per kernel, .vgpr_count / .agpr_count / .sgpr_count /
.private_segment_fixed_size (scratch bytes per lane) / .vgpr_spill_count / .sgpr_spill_count.  Prints the fused STFT
kernel's instantiations (all kernels with --all), the totals, and the instantiations with scratch.
tests/test_resources.py fails when a default-path instantiation (a kernel the plans pick without opting in) has any.
"""
import os
import re
import signal
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """the gfx950 ELF images inside the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        at = pos + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, at)
            triple = data[at + 24 : at + 24 + tlen].decode()
            at += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off : pos + off + size])
        pos = data.find(MAGIC, pos + 1)
    return out


def kernels_of(elf_bytes):
    """[(name, {field: int})] from the AMDGPU metadata note, and the size of .text"""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
        secs = subprocess.run([f"{LLVM}/llvm-readelf", "-S", f.name], capture_output=True, text=True, check=True).stdout
    text = 0
    for line in secs.splitlines():
        m = re.search(r"\]\s+\.text\s+PROGBITS\s+\S+\s+\S+\s+([0-9a-f]+)", line)
        if m:
            text = int(m.group(1), 16)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:  # first key of a kernel's map (keys are sorted)
            cur = {"agpr_count": int(m.group(1))}
            kernels.append(cur)
            continue
        m = re.match(r"\s+\.(\w+):\s+(.+)", line)
        if m and cur is not None:
            key, val = m.group(1), m.group(2).strip()
            if key == "name":
                cur["name"] = val.strip("'")
            elif val.isdigit():
                cur[key] = int(val)
        if re.match(r"amdhsa\.target", line.strip()):
            cur = None
    return [(k.get("name", "?"), k) for k in kernels], text


def short(name):
    """stft_wave_kernel<...> template arguments from the mangled name"""
    m = re.match(r"_ZN3pds16stft_wave_kernelI(.*)EEvNS_10FastParamsE$", name)
    if not m:
        try:
            return subprocess.run([f"{LLVM}/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()[:100]
        except OSError:
            return name[:100]
    args = m.group(1)
    toks = re.findall(r"Li(\d+)E|Lb([01])E|([fds])", args)
    vals = [a or ("T" if b == "1" else "F" if b else c) for a, b, c in toks]
    names = ["N1", "N2", "ROWS", "MAXW", "MINW", "ELL_LDS", "PRE", "SEG", "MF", "RSG", "TIN", "TOUT", "DLT", "STR", "PF"]
    return "stft_wave<" + " ".join(f"{n}={v}" for n, v in zip(names, vals)) + ">"


def main():
    signal.signal(signal.SIGPIPE, signal.SIG_DFL)  # (| head)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                            "pydrobert-speech_amd", "csrc", "libpds_amd.so")
    show_all = "--all" in sys.argv
    rows, text_total = [], 0
    for elf in code_objects(lib):
        ks, text = kernels_of(elf)
        text_total += text
        rows += ks
    stft = [(n, k) for n, k in rows if "stft_wave_kernel" in n]
    print(f"{lib}: {len(rows)} kernels in the gfx950 code objects ({len(stft)} instantiations of stft_wave_kernel), "
          f".text total {text_total / 1e6:.2f} MB")
    print(f"{'kernel':118s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s} {'lds':>6s}")
    for n, k in sorted(rows if show_all else stft, key=lambda r: short(r[0])):
        print(f"{short(n):118s} {k.get('vgpr_count', 0):4d} {k.get('agpr_count', 0):4d} {k.get('sgpr_count', 0):4d} "
              f"{k.get('private_segment_fixed_size', 0):7d} {k.get('vgpr_spill_count', 0):6d} {k.get('sgpr_spill_count', 0):6d} "
              f"{k.get('group_segment_fixed_size', 0):6d}")
    bad = [(n, k) for n, k in rows if k.get("private_segment_fixed_size", 0) > 0]
    print(f"kernels with scratch: {len(bad)} of {len(rows)}")
    for n, k in sorted(bad, key=lambda r: short(r[0])):
        print(f"  {short(n)}: {k['private_segment_fixed_size']} B, {k.get('vgpr_spill_count', 0)} VGPRs spilled")


if __name__ == "__main__":
    main()
