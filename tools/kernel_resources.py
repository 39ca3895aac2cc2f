"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: VGPRs, spills, occupancy, LDS per kernel.
usage: hipcc ... -c file.hip -Rpass-analysis=kernel-resource-usage 2> res.log ; python tools/kernel_resources.py res.log"""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
K_OCC, K_LDS = r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for b in blocks:
    name = b.split("\n")[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    nm = re.sub(r"N3seg\d+_GLOBAL__N_1", "", name)
    nm = re.sub(r"\s*\[-Rpass.*", "", nm)
    print("%-110s vgpr=%d agpr=%d vspill=%d sspill=%d scratch=%d sgpr=%d occ=%d lds=%d" % (nm[-110:], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("SGPRs Spill"),
                                                                                          g(r"ScratchSize \[bytes/lane\]"), g("TotalSGPRs"), g(K_OCC), g(K_LDS)))
