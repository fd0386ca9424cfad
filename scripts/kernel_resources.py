#!/usr/bin/env python3
"""Registers, LDS, scratch and the waves per SIMD they allow, for the step kernels of csrc/slb_kernels.hip as
hipcc builds them for gfx950 (compile only, no GPU needed): reads the .amdhsa_ metadata of the assembly.
usage: scripts/kernel_resources.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "super-lattice-boltzmann-2d_amd", "csrc", "slb_kernels.hip")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "slb.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17",
                        "-S", "--cuda-device-only", "-o", asm, SRC] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        if not any(k in name for k in ("ring_step", "fused2_step", "pipe_step", "multi_step", "fused_step", "sweep_pipe")):
            continue
        get = lambda k: int(re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body).group(1))
        vgpr, lds, scratch = get("next_free_vgpr"), get("group_segment_fixed_size"), get("private_segment_fixed_size")
        sgpr = get("next_free_sgpr")
        by_vgpr = min(8, 512 // max(vgpr, 1))
        wg_by_lds = (160 * 1024) // lds if lds else 99
        rows.append((name, vgpr, sgpr, lds, scratch, by_vgpr, wg_by_lds))
    names = demangle([r[0] for r in rows])
    print("# vgpr sgpr lds_bytes scratch_bytes waves/SIMD(by registers) workgroups/CU(by LDS)  kernel")
    for (name, vgpr, sgpr, lds, scratch, w, g), dn in sorted(zip(rows, names), key=lambda x: x[1]):
        dn = dn.replace("(anonymous namespace)::", "")
        print(f"{vgpr:4d} {sgpr:4d} {lds:6d} {scratch:5d}   {w}   {'-' if g == 99 else g}   {dn}")


if __name__ == "__main__":
    main()
