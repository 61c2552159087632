#!/usr/bin/env python3
"""Per-kernel instruction census of a `hipcc -S --cuda-device-only` listing: what the contraction kernels' inner loops
are made of (MFMA, LDS reads by width, global loads, scratch) and their register budgets.  usage: isa_census.py file.s"""
import re, subprocess, sys

txt = open(sys.argv[1]).read()
parts = re.split(r'; -- Begin function (\S+)\n', txt)
for i in range(1, len(parts) if len(sys.argv) < 3 else 0, 2):
    name, body = parts[i], parts[i + 1].split('; -- End function')[0]
    if 'mfma_gemm' not in name:
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('nnc::', '').replace('void mfma_gemm_f32_kernel', '')
    dem = dem.split('>(')[0][:95]
    c = lambda p: len(re.findall(r'\b' + p + r'\b', body))
    vg = re.search(r'; NumVgprs: (\d+)', body); ag = re.search(r'; NumAgprs: (\d+)', body); occ = re.search(r'; Occupancy: (\d+)', body)
    print(f"{dem:95s} mfma {c('v_mfma_f32_32x32x2_f32'):4d} ds_r b128 {c('ds_read_b128'):3d} b64 {c('ds_read_b64'):3d} 2xb64 {c('ds_read2_b64') + c('ds_read2st64_b64'):3d} "
          f"b32 {c('ds_read_b32'):3d} 2xb32 {c('ds_read2_b32') + c('ds_read2st64_b32'):3d} | ds_w {len(re.findall(r'ds_write', body)):3d} gl4 {c('global_load_dwordx4'):3d} "
          f"scratch {len(re.findall(r'scratch_', body)):3d} | vgpr {vg.group(1) if vg else '?'} agpr {ag.group(1) if ag else '?'} occ {occ.group(1) if occ else '?'}")


def main_loop(body):
    """the largest backward-branch loop of a function: (label, [instructions])"""
    lines = body.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    best = (None, [])
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = [x.strip() for x in lines[labels[m.group(1)]:i + 1] if re.match(r'^\s+[a-z]', x)]
            if len(seg) > len(best[1]): best = (m.group(1), seg)
    return best


if len(sys.argv) >= 3 and sys.argv[2] == 'loop':
    want = sys.argv[3:] or ['']
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split('; -- End function')[0]
        if 'mfma_gemm' not in name: continue
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('nnc::', '').replace('void mfma_gemm_f32_kernel', '').split('>(')[0]
        if not all(w in dem for w in want): continue
        lab, seg = main_loop(body)
        cls = {}
        for ins in seg:
            op = ins.split()[0]
            k = ('mfma' if 'mfma' in op else 'ds_read' if op.startswith('ds_read') else 'ds_write' if op.startswith('ds_write') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else
                 'waitcnt' if op == 's_waitcnt' else 'barrier' if op == 's_barrier' else 'nop' if op == 's_nop' else 'salu' if op.startswith('s_') else 'accmov' if 'accvgpr' in op else 'valu' if op.startswith('v_') else 'other')
            cls[k] = cls.get(k, 0) + 1
        print(dem[:100], len(seg), dict(sorted(cls.items())))
