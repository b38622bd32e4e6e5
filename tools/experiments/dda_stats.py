"""Experiment: empty-space skipping statistics of the supergrid DDA (headline scene, factor 8), on the CPU oracle.
usage: python tools/experiments/dda_stats.py [film] [spp]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "oracle", "_build", "libdrt_oracle_ddastats.so")
subprocess.run(["gcc", "-O2", "-std=gnu99", "-fPIC", "-fopenmp", "-ffp-contract=off", "-mfma", "-shared", "-o", out,
                os.path.join(ROOT, "tools/experiments/dda_stats.c"), "-lm"], check=True)
import oracle.binding as ob
ob._LIB_PATH = out
ob.build = lambda force=False: out
from uivr_amd import synthetic
film = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene = synthetic.dust_devil_scene(res=256, film=film)
scene.medium.majorant_resolution_factor = 8
osc = ob.OracleScene(scene)
props = dict(max_depth=64, use_nee=True, use_drt=True, use_drt_subsampling=True, use_drt_mis=True)
L = ob.lib()
names = "FLIGHTS CELLS EMPTY DF1_J DF1_S DF2_J DF2_S DF3_J DF3_S M2_J M2_S M4_J M4_S M8_J M8_S ORTH_S ODF_J ODF_S DD3_J DD3_S DD7_J DD7_S DD15_J DD15_S DD15M2_J DD15M2_S DD15M3_J DD15M3_S JLEN MD_E_J MD_E_S MD_K4_J MD_K4_S MD_K8_J MD_K8_S MD_S_J MD_S_S MD_K8R_J MD_K8R_S FD_K8_J FD_K8_S FD_K4_J FD_K4_S LEAD_EMPTY TRAIL_EMPTY ALL_EMPTY_FLIGHTS ALL_EMPTY_CELLS".split()
def get():
    st = (C.c_ulonglong * len(names))(); rh = (C.c_ulonglong * 64)(); lh = (C.c_ulonglong * 64)()
    L.dda_stats_get(st, rh, lh)
    return dict(zip(names, list(st))), np.array(list(rh)), np.array(list(lh))
for label, fn in (("primal", lambda: ob.render_primal(osc, props, spp, 7)),):
    Lr, cnt = fn()
    st, rh, lh = get()
    print(label, cnt)
    print(st)
    n = st["CELLS"]
    print("cells/flight %.2f empty frac %.3f" % (n / st["FLIGHTS"], st["EMPTY"] / n))
    for k in ("DD15", "DD15M2", "MD_E", "MD_K4", "MD_K8", "MD_S", "MD_K8R", "FD_K8", "FD_K4"):
        j, s = st[k + "_J"], st[k + "_S"]
        print(f"{k}: jumps {j/n:.3f} steps {s/n:.3f} per fine cell; cost(2.5/jump) {(2.5*j+s)/n:.3f}  cost(2/jump) {(2*j+s)/n:.3f}")
    print("ORTH only: steps %.3f; DD15M2 cells per jump %.2f" % (st["ORTH_S"]/n, st["JLEN"]/max(1,st["DD15M2_J"])))
    print("empty run hist (len: share of empty cells):", {i: round(float(i * rh[i] / max(1, st['EMPTY'])), 3) for i in range(64) if rh[i]})
    print("flight length hist:", {i: round(float(lh[i] / st['FLIGHTS']), 3) for i in range(64) if lh[i]})
# adjoint
res = ob.h1_step(osc, props, spp, 7)
st, rh, lh = get()
n = st["CELLS"]
print("h1 (primal+adjoint)", st)
print("cells/flight %.2f empty frac %.3f" % (n / st["FLIGHTS"], st["EMPTY"] / n))
for k in ("DD15", "DD15M2", "MD_E", "MD_K4", "MD_K8", "MD_S", "MD_K8R", "FD_K8", "FD_K4"):
    j, s = st[k + "_J"], st[k + "_S"]
    print(f"{k}: jumps {j/n:.3f} steps {s/n:.3f} per fine cell; cost(2.5/jump) {(2.5*j+s)/n:.3f}  cost(2/jump) {(2*j+s)/n:.3f}")
