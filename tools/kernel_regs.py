#!/usr/bin/env python3
"""Register / scratch budget of the kernels in a `hipcc -S --cuda-device-only` listing whose (demangled) name contains a pattern.
usage: kernel_regs.py file.s [pattern]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
parts = re.split(r'; -- Begin function (\S+)\n', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    g = lambda k: (re.search(r'; %s: (\d+)' % k, body) or [0, '?'])[1]
    code = body.split('; -- End function')[0]
    lines = code.split('\n')
    mf = [j for j, l in enumerate(lines) if 'v_mfma' in l]
    sc = [j for j, l in enumerate(lines) if 'scratch_' in l]
    inside = [j for j in sc if mf and mf[0] < j < mf[-1]]
    print('%-90s vgpr %s agpr %s occupancy %s scratch %s B; scratch ops %d (%d between the first and last MFMA); %d MFMA, %d v_accvgpr, %d instructions' % (
        dem.split('(')[0][-90:], g('NumVgprs'), g('NumAgprs'), g('Occupancy'), g('ScratchSize'), len(sc), len(inside), len(mf), len(re.findall(r'v_accvgpr', code)), len([l for l in lines if re.match(r'^\s+[a-z]', l)])))
