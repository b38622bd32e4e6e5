import sys, time, torch
sys.path.insert(0, '.')
import uivr_amd as u
from uivr_amd import synthetic
res = int(sys.argv[1])
dev = torch.device('cuda:0')
scene = synthetic.dust_devil_scene(res=res, film=512, device=dev)
integ = u.get_int_config('volpathsimple-drt').create(max_depth=64)
spp, seed = 32, 7
n = 512 * 512
def ev(): return torch.cuda.Event(enable_timing=True)
acc = {}
for it in range(6):
    marks = [ev()]; marks[0].record(); names = []
    def mark(nm):
        e = ev(); e.record(); marks.append(e); names.append(nm)
    img = u.render_primal(scene, integ, 0, spp, seed); mark('render_primal+develop')
    gi = ((2.0 / (n * 3)) * (img - 0.5)).contiguous(); mark('loss grad')
    grads = u.alloc_grads(scene, integ.param_keys); mark('alloc_grads')
    batch = u.render._sensor_batch(scene, 0, spp, None) if hasattr(u, 'render') and hasattr(u.render, '_sensor_batch') else None
    u.render_backward(scene, integ, gi, 0, spp, seed, grads=grads); mark('render_backward (primal+film+adjoint+untile)')
    torch.cuda.synchronize()
    if it >= 2:
        for i, nm in enumerate(names):
            acc[nm] = acc.get(nm, 0) + marks[i].elapsed_time(marks[i + 1]) / 4
h = integ.native_handle(scene)
print(res, {k: round(v, 3) for k, v in acc.items()})
