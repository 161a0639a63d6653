#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stderr file) per kernel."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn)[:80]
    if pat and not re.search(pat, dn):
        continue
    print("%-80s S%4s V%4s A%4s scr%5s occ%2s sspill%4s vspill%4s lds%6s" % (dn, g('TotalSGPRs'), g('VGPRs'), g('AGPRs'), g('ScratchSize [bytes/lane]'),
          g('Occupancy [waves/SIMD]'), g('SGPRs Spill'), g('VGPRs Spill'), g('LDS Size [bytes/block]')))
