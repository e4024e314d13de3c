#!/usr/bin/env python3
"""Attribute the scratch (spill) instructions of one kernel to source lines.

usage: spill_map.py <device .s from hipcc --save-temps -gline-tables-only> [kernel substring] [--by-func]
Walks the .loc directives of the kernel's body and counts scratch_load / scratch_store per (file, line);
prints the histogram bucketed by the enclosing source function (nearest preceding 'UVS_DEV|__global__|template' line
is not known to the assembler, so the bucket is the source line; tools/spill_map.py --ranges maps lines to the
function table given in RANGES below).
"""
import re, sys, collections
path = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 'k_solve'
files = {}; cur = None; inside = False
loads = collections.Counter(); stores = collections.Counter()
for ln in open(path, errors='replace'):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    if re.match(r'^_Z\w*%s\w*:' % kern, ln): inside = True; continue
    if inside and re.match(r'^\s*\.end_amdhsa_kernel|^\.Lfunc_end', ln): inside = False
    if not inside: continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        f_ = files.get(int(m.group(1)), m.group(1))
        if f_.startswith('uvs_solve_kernel') or f_.startswith('uvs_large') or cur is None: cur = (f_, int(m.group(2)))      # context = last location inside the kernel file (inlined helpers keep their caller's bucket)
        continue
    if 'scratch_load' in ln: loads[cur] += 1
    elif 'scratch_store' in ln: stores[cur] += 1
tot_l, tot_s = sum(loads.values()), sum(stores.values())
print(f"{kern}: {tot_l} scratch loads, {tot_s} scratch stores")
# bucket by function using a ctags-like scan of the source
import os
srcdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'uv-slam_amd', 'csrc')
func_tab = {}
for fn in os.listdir(srcdir):
    tab = []
    for i, l in enumerate(open(os.path.join(srcdir, fn), errors='replace'), 1):
        m = re.match(r'^(?:UVS_DEV|__global__|__device__|static __device__)[^(]*?(\w+)\s*\(', l)
        if m and m.group(1) in ('__launch_bounds__', '__attribute__'): m = re.search(r'(\w+)\s*\([^()]*$', l.split(')', 1)[1]) if ')' in l else None
        if m: tab.append((i, m.group(1)))
    func_tab[fn] = tab
def func_of(key):
    if key is None: return '?'
    fn, line = key; name = '?'
    for i, n in func_tab.get(fn, []):
        if i <= line: name = n
        else: break
    return f"{fn}:{name}"
bl = collections.Counter(); bs = collections.Counter()
for k, v in loads.items(): bl[func_of(k)] += v
for k, v in stores.items(): bs[func_of(k)] += v
for f in sorted(set(bl) | set(bs), key=lambda f: -(bl[f] + bs[f])):
    print(f"  {f:50s} loads {bl[f]:4d}  stores {bs[f]:4d}")
if '--lines' in sys.argv:
    for k in sorted(set(loads) | set(stores), key=lambda k: -(loads[k] + stores[k]))[:60]:
        print(f"    {k}: loads {loads[k]} stores {stores[k]}")
