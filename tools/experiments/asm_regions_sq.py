"""Static instruction counts per source region of trace_sq_kernel (experiment).
   hipcc ... -gline-tables-only -S --cuda-device-only drt_sq.hip -o sq.s ; python tools/experiments/asm_regions_sq.py sq.s ILb0ELb0ELb0E [src] [kernel-prefix]"""
import re, sys, collections
asm, inst = sys.argv[1], sys.argv[2]
srcf = sys.argv[3] if len(sys.argv) > 3 else '/root/repo/unbiased-inverse-volume-rendering_amd/csrc/drt_sq.hip'
prefix = sys.argv[4] if len(sys.argv) > 4 else '_ZN3drt15trace_sq_kernel'
src = open(srcf).read().split('\n')
marks = [("prologue", 0)]
for i, l in enumerate(src):
    m = re.search(r'// =================\s*(.*?)\s*=*\s*$', l)
    if m: marks.append((m.group(1)[:40], i + 1))
    m = re.search(r'// ---- (.*?)\s*-*\s*$', l)
    if m: marks.append(("  " + m.group(1)[:38], i + 1))
    m = re.search(r'//@@\s*(.*)$', l)
    if m: marks.append(("  @" + m.group(1)[:37], i + 1))
def region(line):
    r = 0
    for k, (n, l) in enumerate(marks):
        if line >= l: r = k
    return r
lines = open(asm).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(prefix + inst))
# which file number is the main source
fileno = None
cnt = collections.Counter(); cur = 0
other = collections.Counter()
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        # inlined code from headers keeps the region of the last main-file line
        if fileno is None:
            fileno = m.group(1)
        if m.group(1) == fileno: cur = region(int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z_0-9]+)\s', l + ' ')
    if not m or l.strip().startswith(('.', ';')): continue
    op = m.group(1)
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other'
    cnt[(cur, kind)] += 1
print(f"{'region':42s} {'valu':>6s} {'salu':>6s} {'lds':>5s} {'vmem':>5s}")
tot = collections.Counter()
for k, (n, l) in enumerate(marks):
    if any(cnt[(k, x)] for x in ('valu', 'salu', 'lds', 'vmem')):
        print(f"{n:42s} {cnt[(k,'valu')]:6d} {cnt[(k,'salu')]:6d} {cnt[(k,'lds')]:5d} {cnt[(k,'vmem')]:5d}   (line {l})")
    for x in ('valu', 'salu', 'lds', 'vmem'): tot[x] += cnt[(k, x)]
print(f"{'total':42s} {tot['valu']:6d} {tot['salu']:6d} {tot['lds']:5d} {tot['vmem']:5d}")
