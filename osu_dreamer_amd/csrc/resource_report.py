"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (VGPR / scratch / LDS per kernel)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: Function Name: ', txt)[1:]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in blocks:
    name = b.split(' ')[0]
    def g(k):
        m = re.search(k + r': (\d+)', b); return int(m.group(1)) if m else -1
    nm = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    nm = re.sub(r'\(anonymous namespace\)::', '', nm)
    nm = re.sub(r'unsigned short', 'bf16', nm).split('(')[0][:64]
    r = (g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g('VGPRs Spill'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]'))
    if flt == 'all' or r[2] > 0 or r[3] > 0 or 'flash' in nm or 'gemm' in nm:
        print(f"{nm:66s} vgpr={r[0]:4d} agpr={r[1]:3d} scratch={r[2]:5d} spill={r[3]:4d} occ={r[4]} lds={r[5]}")
