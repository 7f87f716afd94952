#!/usr/bin/env python3
"""Checks the ISA of gmx_stock.hip: instructions hipcc generated itself (everything outside
;;#ASMSTART .. ;;#ASMEND) must stay out of the registers gmx_stock_asm.inc owns
(v44..v255, a44..a255, s70..s101), and nothing may spill to scratch.  Prints per-kernel
instruction counts.  Usage: check_stock_regs.py [file.s]  (default: compiles gmx_stock.hip)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT = {"v": 44, "a": 44, "s": 70}


def compile_to_asm():
    src = os.path.join(ROOT, "gmix_amd", "csrc")
    out = os.path.join(tempfile.mkdtemp(), "gmx_stock.s")
    subprocess.check_call(
        ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
         "-fno-gpu-flush-denormals-to-zero", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S", os.path.join(src, "gmx_stock.hip"), "-o", out],
        cwd=src, stderr=subprocess.DEVNULL)
    return out


def regs(line):
    line = line.split(";")[0]
    for kind, lo, hi in re.findall(r"\b([vas])\[(\d+):(\d+)\]", line):
        yield kind, int(hi)
    for kind, n in re.findall(r"\b([vas])(\d+)\b", line):
        yield kind, int(n)


def check(path):
    bad, counts, kernel, in_asm = [], {}, None, False
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            counts[kernel] = [0, 0]
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if kernel is None or not line.startswith("\t") or s.startswith((".", ";")) or not s:
            continue
        if s.startswith("s_endpgm"):
            kernel = None
            continue
        counts[kernel][1 if in_asm else 0] += 1
        if "scratch_" in s:
            bad.append((ln, s))
        if not in_asm:
            for kind, n in regs(s):
                if n >= LIMIT[kind]:
                    bad.append((ln, s))
    return bad, counts


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm()
    bad, counts = check(path)
    for k, (c, a) in counts.items():
        print(f"{k[:70]}: {c} compiler instructions, {a} in asm blocks")
    for ln, s in bad[:20]:
        print(f"line {ln}: {s}")
    sys.exit(1 if bad else 0)
