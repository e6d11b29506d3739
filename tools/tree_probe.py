#!/usr/bin/env python3
"""Per-kernel timing of the tree kernels on injected outputs (profiling aid, not a bench line).

    python tools/tree_probe.py [workload ...]

For every workload: one move in four modes -- {select + expand_backprop, fused step} x {priors, logits} -- timing each
launch with events on the launch stream; prints the mean per kernel and the achieved algorithmic GB/s of the fused step.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from muzero_amd.tree_pool import TreePool  # noqa: E402


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    return out, (e0, e1)


def probe(name, wl, device, moves=int(os.environ.get("PROBE_MOVES", "3")), lanes=0):
    T, A, N = wl['T'], wl['A'], wl['N']
    pool = TreePool(num_trees=T, num_actions=A, max_simulations=N, num_players=wl['players'], discount=wl['discount'],
                    default_value=wl['default'], known_bounds=wl['bounds'], device=device, lanes_per_tree=lanes)
    logits, values, rewards, noise, u = bench.synthetic_outputs(T, A, N, 1234, wl['reward'], device)
    priors = torch.softmax(logits, -1).contiguous()
    res = {'workload': name, 'lanes': lanes}
    for use_logits in (False, True):
        pol = logits if use_logits else priors
        tag = 'logits' if use_logits else 'priors'
        for fused in (False, True):
            ev = {'select': [], 'backprop': [], 'step': []}
            c0 = pool.counters()
            for m in range(moves):
                pool.reset()
                pool.expand_roots(pol[0], logits=use_logits)
                pool.add_root_noise(noise, 0.25)
                pool.select()
                for s in range(N - 1):
                    if fused:
                        _, e = timed(lambda: pool.step(pol[s + 1], rewards[s + 1], values[s + 1], logits=use_logits))
                        ev['step'].append(e)
                    else:
                        _, e = timed(lambda: pool.expand_backprop(pol[s + 1], rewards[s + 1], values[s + 1],
                                                                  logits=use_logits))
                        ev['backprop'].append(e)
                        _, e = timed(lambda: pool.select())
                        ev['select'].append(e)
                pool.expand_backprop(pol[N], rewards[N], values[N], logits=use_logits)
            torch.cuda.synchronize()
            c1 = pool.counters()
            mean_d = (c1[1] - c0[1]) / (c1[0] - c0[0])
            for k, v in ev.items():
                if v:
                    # skip the first move (cold)
                    ms = [a.elapsed_time(b) for a, b in (v[len(v) // moves:] if moves > 1 else v)]
                    res[f'{tag}_{k}_us'] = float(np.mean(ms)) * 1e3
            if fused:
                by = T * bench.bytes_per_simulation(A, mean_d, A)
                res[f'{tag}_step_GBps'] = by / (res[f'{tag}_step_us'] * 1e-6) / 1e9
                res['mean_d'] = mean_d
    pool.close()
    return res


def main():
    names = sys.argv[1:] or ['atari_16384x50', 'go_2048x800']
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    for n in names:
        wl = dict(bench.WORKLOADS[n])
        if n.startswith('go') and os.environ.get('PROBE_GO_N'):
            wl['N'] = int(os.environ['PROBE_GO_N'])
        for lanes in [int(x) for x in os.environ.get('PROBE_LANES', '0').split(',')]:
            print(json.dumps(probe(n, wl, device, lanes=lanes)), flush=True)


if __name__ == '__main__':
    main()
