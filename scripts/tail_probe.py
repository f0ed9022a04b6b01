# Wave-lifetime timeline of the persistent extend launch.  Needs a library built with -DPT_DEBUG_TAIL
# (hipcc ... -DPT_DEBUG_TAIL -c pt_kernels.hip), which records start/end clocks per wave.
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import pathtracer_amd as pt
from pathtracer_amd import _lib as L
from pathtracer_amd.synthetic import heightfield_box
import torch
data = heightfield_box(2236)
scene = pt.Scene(data, device=0)
cfg = pt.Config(width=3840, height=2160, bounces=1, seed=1, spp_per_batch=2)   # bounces=1: one secondary extend launch
acc = torch.zeros((2160, 3840, 3), dtype=torch.float64, device='cuda')
for i in range(2):
    scene.render_device(cfg, acc.data_ptr(), 2 * i, 2, stream=0)
torch.cuda.synchronize()
buf = np.zeros(16384, np.uint64)
fn = L.lib.pt_debug_tail
fn.argtypes = [C.c_void_p]
print('rc', fn(buf.ctypes.data))
a = buf.reshape(-1, 2).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
total = en.max()
print(f'waves {len(a)} total {total:.0f} us; end percentiles 10/50/90/99/100:', ' '.join(f'{np.percentile(en, p):.0f}' for p in (10, 50, 90, 99, 100)), '| start 50/99/max:', f'{np.percentile(st,50):.0f} {np.percentile(st,99):.0f} {st.max():.0f}')
ts = np.linspace(0, total, 21)
print('active waves over time:', [int(((st <= t) & (en > t)).sum()) for t in ts])
