"""Register, scratch and LDS use of every kernel instantiation, from the compiler's own report
(hipcc -Rpass-analysis=kernel-resource-usage on rt_kernels.hip; no GPU needed).
usage: python tools/resource_usage.py > profiles/<tag>_kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
       "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
filt = subprocess.run(["c++filt"], input="\n".join(re.findall(r"Function Name: (\S+)", txt)), capture_output=True, text=True).stdout.split("\n")
print("%-58s %5s %5s %8s %7s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "vspill", "sspill", "occ", "LDS"))
print("%-58s %5s %5s %8s %7s %7s %4s %6s" % ("", "", "", "B/lane", "", "", "w/SIMD", "B/wg"))
for i, b in enumerate(re.split(r"remark: [^\n]*Function Name: ", txt)[1:]):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", filt[i]))
    print("%-58s %5d %5d %8d %7d %7d %4d %6d" % (name[:58], g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"),
                                                g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
