import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
order = sys.argv[1]
if order == 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available())
import pathtracer_amd as pt
try:
    s = pt.Scene(pt.scene_data('CornellBox'), 0)
    print(order, 'scene ok')
    cfg = pt.Config(dim=64, samples_pr_pixel=2)
    img = s.render(cfg)
    print('render mean', img.mean())
except Exception as e:
    print(order, 'FAILED', e)
if order == 'pt_first':
    import torch
    print('torch avail', torch.cuda.is_available())
os.system("grep -E 'hip|hsa' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
