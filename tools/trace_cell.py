"""Diagnostic: per-workgroup timeline of ttt_cell_kernel (start/end in 10-ns ticks, XCC, CU) for one launch."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd.games import tictactoe
T = int(os.environ.get("MZ_T", "4096"))
cfg = tictactoe.make_config(25)
net = cfg.network_config.network_class(cfg, seed=7)
hid = torch.rand(2, T, 1152, device='cuda')
pb = torch.zeros(T, dtype=torch.int32, device='cuda'); act = torch.randint(0, 9, (T,), device='cuda', dtype=torch.int32)
r = torch.empty(T, device='cuda'); v = torch.empty(T, device='cuda'); lg = torch.empty(T, 9, device='cuda')
rows = int(os.environ.get('MZ_TILE_ROWS', '64'))
nwg = 9 * (T // rows) if os.environ.get('MZ_CELL_KERNEL') else 32 * ((T // 32 + 7) // 8)
trace = torch.zeros(nwg * 8, dtype=torch.int64, device='cuda')
import time
warm = float(os.environ.get('MZ_WARM_SECONDS', '0'))
t_w = time.time()
for _ in range(3):
    net.recurrent_inference_pool(hid, pb, act, 1, r, v, lg)
while time.time() - t_w < warm:            # sustained load first: the chip lowers its clock under load
    for _ in range(200):
        net.recurrent_inference_pool(hid, pb, act, 1, r, v, lg)
    torch.cuda.synchronize()
net._lib.mzdbg_net_set_trace(net._h, ctypes.c_void_p(trace.data_ptr()))
net.recurrent_inference_pool(hid, pb, act, 1, r, v, lg)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nwg, 8)
seg = np.stack([(t[:, 3 + i // 2] >> 32) if i % 2 == 0 else (t[:, 3 + i // 2] & 0xffffffff) for i in range(10)], 1) / 100.0
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0        # microseconds
hw = t[:, 2] & 0xffffffff; xcc = (t[:, 2] >> 32) & 0xf; mma = seg[:, 2]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
print('kernel span us', end.max(), 'n wg', nwg)
if not os.environ.get('MZ_CELL_KERNEL'):
    ghz = t[:, 7] / ((t[:, 1] - t[:, 0]) * 10.0)
    print('in-kernel shader clock GHz: median %.3f min %.3f max %.3f' % (np.median(ghz), ghz.min(), ghz.max()))
if not os.environ.get('MZ_CELL_KERNEL'):
    sg = np.stack([t[:, 4] >> 32, t[:, 4] & 0xffffffff, t[:, 5] >> 32, t[:, 5] & 0xffffffff, t[:, 6]], 1) / 100.0
    for j in range(4):
        print('job', j, '[prologue, setup, mma, epilogue+barrier, after-barrier] us', np.round(sg[t[:, 3] == j].mean(0), 1).tolist(), 'total', round(float((end - start)[t[:, 3] == j].mean()), 1))
order = int(os.environ.get('MZ_CELL_ORDER', '0'))
cells = (np.array([4, 1, 3, 5, 7, 0, 2, 6, 8])[np.arange(nwg) // (T // rows)] if order == 0 else np.arange(nwg) % 9) if os.environ.get('MZ_CELL_KERNEL') else np.array([4, 1, 5, 2])[t[:, 3]]
for name, sel in (('center', cells == 4), ('edge', np.isin(cells, [1, 3, 5, 7])), ('corner', np.isin(cells, [0, 2, 6, 8]))):
    d = (end - start)[sel]
    print(name, 'segments[prolog,ld,mma,commit,barrier,copy,epi1,dense,epi2,pred]', np.round(seg[sel].mean(0), 1).tolist())
    print(name, 'n', sel.sum(), 'dur mean %.1f min %.1f max %.1f' % (d.mean(), d.min(), d.max()), 'start mean %.1f max %.1f' % (start[sel].mean(), start[sel].max()), 'end max %.1f' % end[sel].max())
key = xcc * 10000 + se * 1000 + sh * 100 + cu
uniq, cnt = np.unique(key, return_counts=True)
print('distinct CUs', len(uniq), 'wg per CU min/max', cnt.min(), cnt.max())
busy = np.array([(end - start)[key == k].sum() for k in uniq]); last = np.array([end[key == k].max() for k in uniq])
print('per-CU last end: min %.1f mean %.1f max %.1f' % (last.min(), last.mean(), last.max()))
i = np.argmax(last); k = uniq[i]
print('slowest CU', k, [(int(c), round(float(s), 1), round(float(e), 1)) for c, s, e in zip(cells[key == k], start[key == k], end[key == k])])
i = np.argmin(last); k = uniq[i]
print('fastest CU', k, [(int(c), round(float(s), 1), round(float(e), 1)) for c, s, e in zip(cells[key == k], start[key == k], end[key == k])])
