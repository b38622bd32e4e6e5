"""Static instruction counts per source region of trace_super_kernel (experiment).
   hipcc ... -gline-tables-only -S --cuda-device-only drt_super.hip -o x.s ; python tools/experiments/asm_regions.py x.s ILb0ELb0ELb0ELb1E"""
import re, sys, collections
asm, inst = sys.argv[1], sys.argv[2]
src = open('/root/repo/unbiased-inverse-volume-rendering_amd/csrc/drt_super.hip').read().split('\n')
marks = [("prologue", 0)]
for name, pat in (("walk:pull", "================= walk:"), ("walk:steps", "const float tmin = fminf(fminf(tnx"), ("walk:finish", "DRT_PROF4(4,"),
                  ("Fe epilogue", "================= (Fe)"), ("A regen", "================= (A)"), ("B transitions", "================= (B)"),
                  ("Fs set-up", "================= (Fs)"), ("end", "if constexpr (ADJ) close_records")):
    for i, l in enumerate(src):
        if pat in l: marks.append((name, i + 1)); break
def region(line):
    r = marks[0][0]
    for n, l in marks:
        if line >= l: r = n
    return r
lines = open(asm).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3drt18trace_super_kernel' + inst))
cnt = collections.Counter(); cur = "prologue"
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        if m.group(1) == '0': cur = region(int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z_0-9]+)\s', l)
    if not m or l.strip().startswith(('.', ';')): continue
    op = m.group(1)
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other'
    cnt[(cur, kind)] += 1
regs = [n for n, _ in marks]
print(f"{'region':16s} {'valu':>6s} {'salu':>6s} {'lds':>5s} {'vmem':>5s}")
for r in regs:
    print(f"{r:16s} {cnt[(r,'valu')]:6d} {cnt[(r,'salu')]:6d} {cnt[(r,'lds')]:5d} {cnt[(r,'vmem')]:5d}")
