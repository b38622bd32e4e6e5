"""Quadratic DRT through the queued tracer's QUAD kernels (flags 0) against the round-2 kernels (test hook 4096): counters and gradients."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import uivr_amd as u
from uivr_amd import synthetic
dev = torch.device('cuda', 0)
for res, film, factor, spp in ((64, 64, 4, 4), (128, 96, 8, 8), (160, 64, 4, 4), (256, 128, 4, 4)):
    scene = synthetic.dust_devil_scene(res=res, film=film, device=dev)
    scene.medium.majorant_resolution_factor = factor
    props = dict(u.get_int_config('volpathsimple-drt-quadratic').create(max_depth=64).props(), type="volpathsimple", test_hooks=True)
    integ = u.load_dict(props)
    h = integ.native_handle(scene)
    out = {}
    for flags in (0, 4096):
        h.set_debug_flags(flags)
        h.enable_counters(True); h.reset_counters()
        gi = torch.full((film * film, 3), 1e-3, device=dev)
        g = u.render_backward(scene, integ, gi, sensor=0, spp=spp, seed=5)
        out[flags] = ({k: int(v) for k, v in h.get_counters().items()}, g["_flat"].clone())
    h.set_debug_flags(0); h.enable_counters(False)
    same = out[0][0] == out[4096][0]
    d = float((out[0][1] - out[4096][1]).abs().max()) / float(out[4096][1].abs().max())
    print(res, factor, "counters equal" if same else {k: (out[0][0][k], out[4096][0][k]) for k in out[0][0] if out[0][0][k] != out[4096][0][k]}, "grad rel diff", d)
