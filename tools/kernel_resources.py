"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: VGPRs / AGPRs / scratch / occupancy / LDS per kernel.

usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2> remarks.txt ; python tools/kernel_resources.py remarks.txt
"""
import re
import subprocess
import sys


def main(path):
    txt = open(path).read()
    for b in txt.split("Function Name: ")[1:]:
        name = b.split("\n")[0].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()

        def g(k):
            m = re.search(re.escape(k) + r": (\S+)", b)
            return m.group(1) if m else "?"

        print("%-100s vgpr %4s agpr %3s scratch %4s occ %s lds %s" % (
            dem[:100], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main(sys.argv[1])
