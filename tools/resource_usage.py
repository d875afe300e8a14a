"""Per-kernel register / scratch / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file).
usage: hipcc <build flags> -Rpass-analysis=kernel-resource-usage -c unit.hip -o /tmp/x.o 2> /tmp/ru.txt; python tools/resource_usage.py /tmp/ru.txt"""
import re
import sys
txt = open(sys.argv[1]).read()
K_S, K_O, K_L = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = re.sub(r"^_Z\d+", "", b.split()[0])[:40]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-42s VGPR %4s AGPR %3s scratch %4s occ %2s LDS %s" % (name, g("VGPRs"), g("AGPRs"), g(K_S), g(K_O), g(K_L)))
