#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch use of the kernels in an object file built by hipcc (reads the gfx950 code
object out of its fat binary): python tools/kernel_resources.py graphmat_amd/csrc/gm_programs.o [substring ...]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj = sys.argv[1]
    wanted = sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("  - .agpr_count:")[1:]:
        def f(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = f("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if wanted and not all(w in dem for w in wanted):
            continue
        print("vgpr %3s sgpr %3s lds %6s scratch %4s wg %4s  %s" % (f("vgpr_count"), f("sgpr_count"), f("group_segment_fixed_size"),
                                                                    f("private_segment_fixed_size"), f("max_flat_workgroup_size"), dem[:150]))


if __name__ == "__main__":
    main()
