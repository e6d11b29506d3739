#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/s of the self-play hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2], the config the metric is quoted on -- "4096 concurrent games"):
TicTacToe-shaped search, T=4096 concurrent games PER GPU (weak scaling: games shard across ranks with no
data-path collective), A=9, 25 simulations/move, 2 players, value bounds (-1,1).  One "step" is one move of
all T games: root expansion + Dirichlet noise + 25 x (select -> recurrent_inference -> expand -> backprop) +
root statistics + action sampling, i.e. T*25 simulations; everything is resident in HBM when the clock starts.

The JSON line carries `roofline` for the dominant kernel (HIP-event timed inside the timed region) and
`cpu_baseline` (the CPU oracle -- a port of the reference algorithm -- timed on this box's host cores on a
bounded sample).  Other BASELINE configs are available with --workload for profiling, not as bench lines.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2516.6  # dense f16/bf16 MFMA peak: 16 x the f32 rate (MI355X_MICROARCH.md: "~2.5 PF", "1/16 of BF16")

WORKLOADS = {
    # name: T per GPU, A, N, players, discount, default value, known bounds, network kind
    'tictactoe_4096x25': dict(T=4096, A=9, N=25, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0),
                              net='tictactoe', reward='tanh'),
    'cartpole_1024x50': dict(T=1024, A=2, N=50, players=1, discount=0.99, default=50.0, bounds=(0.0, 100.0),
                             net='cartpole', reward='uniform'),
    # configs[3]: conv-residual dynamics network in the loop (--inject: tree kernels alone, injected outputs)
    'atari_16384x50': dict(T=16384, A=18, N=50, players=1, discount=0.997, default=0.0, bounds=None, net='atari',
                           reward='uniform'),
    'go_2048x800': dict(T=2048, A=362, N=800, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0), net=None,
                        reward='zero'),
}


def bytes_per_simulation(A, d, a_root):
    """algorithmic bytes of one simulation with d expanded nodes on its path."""
    return 20 * A * (d + 1) + 28 * d + 32 + 4 * a_root


def synthetic_outputs(T, A, N, seed, reward, device):
    """Injected network outputs: logits~N(0,1), values~tanh(N(0,1)), rewards by mode; generated up
    front on the device so that only the tree kernels are inside the timed region."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    logits = torch.randn(N + 1, T, A, generator=g, device=device, dtype=torch.float32)
    values = torch.tanh(torch.randn(N + 1, T, generator=g, device=device, dtype=torch.float32))
    if reward == 'uniform':
        rewards = torch.rand(N + 1, T, generator=g, device=device, dtype=torch.float32)
    elif reward == 'tanh':
        rewards = torch.tanh(torch.randn(N + 1, T, generator=g, device=device, dtype=torch.float32))
    else:
        rewards = torch.zeros(N + 1, T, device=device, dtype=torch.float32)
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).to(device)
    u = torch.rand(T, generator=g, device=device, dtype=torch.float64)
    return logits, values, rewards, noise.contiguous(), u


class SyntheticTreeMove:
    """One move of T searches with injected outputs (tree kernels only; workloads without a reference network).  The T
    trees of a GPU are split into `lanes` pools stepped on separate HIP streams (trees are independent): a step of one
    pool ends with its deepest tree, and the other pools' launches fill the chip meanwhile.  Every pool's move is
    enqueued by one native call (mz_search_move_injected)."""
    dominant = 'run_injected_kernel'
    bound = 'hbm'

    def __init__(self, wl, device, seed, lanes=1, event_stride=5, lanes_per_tree=0):
        from muzero_amd.tree_pool import TreePool
        self.event_stride = event_stride
        assert wl['T'] % lanes == 0, 'lanes must divide the trees per GPU'
        self.wl, self.device, self.nl = wl, device, lanes
        Tl = wl['T'] // lanes
        self.logits, self.values, self.rewards, self.noise, self.u = synthetic_outputs(
            wl['T'], wl['A'], wl['N'], seed, wl['reward'], device)
        self.temperature = torch.ones(wl['T'], dtype=torch.float64, device=device)
        self.lanes = []
        for i in range(lanes):
            sl = slice(i * Tl, (i + 1) * Tl)
            stream = torch.cuda.Stream(device=device) if lanes > 1 else torch.cuda.current_stream(device)
            with torch.cuda.stream(stream):
                pool = TreePool(num_trees=Tl, num_actions=wl['A'], max_simulations=wl['N'], num_players=wl['players'],
                                discount=wl['discount'], default_value=wl['default'], known_bounds=wl['bounds'],
                                device=device, lanes_per_tree=lanes_per_tree)
            self.lanes.append(dict(pool=pool, stream=stream, logits=self.logits[:, sl].contiguous(),
                                   values=self.values[:, sl].contiguous(), rewards=self.rewards[:, sl].contiguous(),
                                   noise=self.noise[sl].contiguous(), u=self.u[sl].contiguous(),
                                   temperature=self.temperature[sl].contiguous()))
        if lanes > 1:
            del self.logits, self.values, self.rewards
        torch.cuda.synchronize()
        self.hip_events = HipEvents()
        self.event_sets = []       # per sampled step: events around single fused-step launches
        self.run_sets = []         # per sampled step: one event before the first fused step, one after the last
        self.trees_per_launch = Tl

    def counters(self):
        return sum(np.asarray(l['pool'].counters()) for l in self.lanes)

    def run(self, timed_events=False):
        import ctypes
        N = self.wl['N']
        main = torch.cuda.current_stream(self.device)
        out = []
        if self.nl > 1:
            for l in self.lanes:
                l['stream'].wait_stream(main)
        for i, l in enumerate(self.lanes):
            events = None
            if timed_events and i == 0 and N > 1:     # HIP events around lane 0's fused steps, on its stream
                events = (ctypes.c_void_p * (2 * (N - 1)))()
                if timed_events == 'run':             # the N-1 fused steps as one bracket, no record between them
                    events[0], events[2 * (N - 2) + 1] = self.hip_events.new(), self.hip_events.new()
                    self.run_sets.append(events)
                else:
                    for s in range(self.event_stride // 2, N - 1, self.event_stride):       # a sample of the steps
                        events[2 * s], events[2 * s + 1] = self.hip_events.new(), self.hip_events.new()
                    self.event_sets.append(events)
            with torch.cuda.stream(l['stream']):
                pool = l['pool']
                pool.search_injected(l['logits'], l['rewards'], l['values'], noise=l['noise'], frac=0.25, logits=True,
                                     events=events)
                out.append((pool.root_stats(), pool.select_action(l['u'], l['temperature'])))
        if self.nl > 1:
            for l in self.lanes:
                main.wait_stream(l['stream'])
        return out

    def step_durations(self):
        """mean duration (s) of the fused step launches of lane 0 over the sampled timed steps"""
        import ctypes
        ms = []
        for events in self.event_sets:
            for j in range(0, len(events), 2):
                if events[j]:
                    ms.append(self.hip_events.elapsed_ms(ctypes.c_void_p(events[j]), ctypes.c_void_p(events[j + 1])))
        return float(np.mean(ms)) * 1e-3 if ms else None

    def run_durations(self):
        """mean duration (s) of the ONE launch that holds a move's N simulations (run_injected_kernel: what
        mz_search_move_injected enqueues when no single step is timed), bracketed by two events"""
        import ctypes
        N = self.wl['N']
        ms = [self.hip_events.elapsed_ms(ctypes.c_void_p(ev[0]), ctypes.c_void_p(ev[2 * (N - 2) + 1])) for ev in self.run_sets]
        return float(np.mean(ms)) * 1e-3 if ms else None


class HipEvents:
    """Raw HIP events (hipEventCreate/Record/ElapsedTime through ctypes) recorded on torch's current stream --
    the stream every muzero_amd launch goes to -- so they bracket exactly the launches of the timed region."""
    _hip = None

    def __init__(self):
        import ctypes
        if HipEvents._hip is None:
            HipEvents._hip = ctypes.CDLL('libamdhip64.so')
        self.ct, self.hip = ctypes, HipEvents._hip
        self.free, self.used = [], []

    def new(self):
        if self.free:
            ev = self.free.pop()
        else:
            ev = self.ct.c_void_p()
            rc = self.hip.hipEventCreate(self.ct.byref(ev))
            assert rc == 0, f'hipEventCreate -> {rc}'
        self.used.append(ev)
        return ev

    def reserve(self, n):
        """Create n events now and record each once: the first process on a fresh box has been seen to stall ~40 ms
        somewhere past its 30th event, which must not happen inside a timed region."""
        fresh = [self.new() for _ in range(n)]
        for ev in fresh:
            self.record(ev)
        torch.cuda.synchronize()
        for ev in fresh:
            self.used.remove(ev)
            self.free.append(ev)

    def record(self, ev=None):
        ev = ev if ev is not None else self.new()
        stream = self.ct.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = self.hip.hipEventRecord(ev, stream)
        assert rc == 0, f'hipEventRecord -> {rc}'
        return ev

    def elapsed_ms(self, a, b):
        ms = self.ct.c_float()
        rc = self.hip.hipEventElapsedTime(self.ct.byref(ms), a, b)
        assert rc == 0, f'hipEventElapsedTime -> {rc}'
        return ms.value


class MoveEvents:
    """HIP events for a sample of a move's simulations, recorded by mz_search_move on the launch stream: [0] before the
    inference, [1] after its dominant kernel (TicTacToe: the job kernel; CartPole: the whole simulation, network and
    tree step being one launch for batches <= 8192), [2] after the tree step (TicTacToe: head + tree step are one launch).  A record costs ~13 us of stream time, so only every
    `stride`-th simulation of a sampled step is timed."""

    def __init__(self, num_simulations, stride):
        import ctypes
        self.ev, self.N, self.ct = HipEvents(), num_simulations, ctypes
        self.sampled = list(range(stride // 2, num_simulations, stride)) or [0]
        self.sets = []

    def new_set(self):
        arr = (self.ct.c_void_p * (4 * self.N))()
        for s in self.sampled:
            for j in range(3):
                arr[4 * s + j] = self.ev.new()
        self.sets.append(arr)
        return arr

    def durations(self):
        """-> (dominant kernel s, rest of the simulation s), each averaged over the sampled simulations"""
        first, rest = [], []
        for arr in self.sets:
            for s in self.sampled:
                e0, em, e1 = (self.ct.c_void_p(arr[4 * s + j]) for j in range(3))
                first.append(self.ev.elapsed_ms(e0, em))
                rest.append(self.ev.elapsed_ms(em, e1))
        mean = lambda v: float(np.mean(v)) * 1e-3 if v else None
        return mean(first), mean(rest)


class NetworkMove:
    """One move of T games with the HIP network in the loop (muzero_amd.selfplay.LockstepSearch): the full
    reference path of MCTSAgent.make_move, batched."""
    bound = 'mfma'

    def __init__(self, wl, device, seed, world, lanes=1, event_stride=5, run_mode=0, job_rows=0):
        from muzero_amd.games import atari_shaped, cartpole, tictactoe
        from muzero_amd.selfplay import LockstepSearch, PipelinedSearch, TrajectoryGather
        mod = {'tictactoe': tictactoe, 'cartpole': cartpole, 'atari': atari_shaped}[wl['net']]
        self.wl, self.device = wl, device
        self.config = mod.make_config(num_simulations=wl['N'])
        self.net = self.config.network_config.network_class(self.config, seed=7)       # random-init (glorot) weights
        if run_mode:                                  # A/B measurements only (Network.set_run_mode)
            self.net.set_run_mode(run_mode)
        if job_rows:
            self.net.set_job_rows(job_rows)
        self.lanes = lanes
        if lanes > 1:
            self.search = PipelinedSearch(self.config, self.net, num_trees=wl['T'], lanes=lanes, device=device)
            self.counters = self.search.counters
            self.trees_per_launch = self.search.lanes[0].T       # the events bracket lane 0's launches
        else:
            self.search = LockstepSearch(self.config, self.net, num_trees=wl['T'], device=device)
            self.counters = self.search.pool.counters
            self.trees_per_launch = wl['T']
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        T, A = wl['T'], wl['A']
        if wl['net'] == 'tictactoe':     # random mid-game boards: planes [X, O, to_play]
            cells = torch.randint(0, 3, (T, 9), generator=g, device=device)
            to_play = torch.randint(0, 2, (T, 1), generator=g, device=device).expand(T, 9)
            self.obs = torch.stack([(cells == 0), (cells == 1), to_play.bool()], -1).float().reshape(T, 27).contiguous()
            # the N simulations of a move are ONE launch, ttt_run_kernel (mz_infer_recurrent_run): the jobs
            # (conv 49*137*128, dense 9*128*128, reward 1152, prediction conv 9*128*128), the head (dense 128*128, value
            # 128, policy 9*128) -- SURVEY.md 8(d)'s 2.35 MFLOP per evaluation, multiply-adds x2 -- and the tree steps
            self.dominant = 'ttt_run_kernel'
            self.flops_per_eval = 2 * (49 * 137 * 128 + 9 * 128 * 128 + 1152 + 9 * 128 * 128 + 128 * 128 + 128 + 9 * 128)
            if run_mode == 2 or (job_rows or (64 if wl['T'] > 2560 else 32)) != 64:
                # a launch per simulation (A/B runs; batches of 16-game jobs): the job kernel and its terms alone
                self.dominant = 'ttt_ring_kernel'
                self.flops_per_eval = 2 * (49 * 137 * 128 + 9 * 128 * 128 + 1152 + 9 * 128 * 128)
        elif wl['net'] == 'atari':       # synthetic (2,128) observations
            self.obs = torch.randn(T, 256, generator=g, device=device).contiguous()
            # the N simulations of a move (whole network + tree step of a workgroup's 64 games, looped in the kernel) are ONE
            # launch, atari_run_kernel (mz_infer_recurrent_run): per evaluation input conv 4 taps x 146 x 128,
            # 16 + 2 + 1 128x128 stages, reward 256, heads 128 x 19
            self.dominant = 'atari_run_kernel'
            self.flops_per_eval = 2 * (4 * 146 * 128 + 16 * 128 * 128 + 256 + 2 * 128 * 128 + 128 * 128 + 128 * 19)
        else:
            span = torch.tensor([2.4, 3.0, 0.21, 3.0], device=device)
            self.obs = ((torch.rand(T, 4, generator=g, device=device) * 2 - 1) * span).contiguous()
            # batches <= 8192: the N simulations of a move (network and tree step of 16 games per workgroup) are ONE
            # launch, cartpole16_run_kernel (mz_infer_recurrent_run)
            self.dominant = 'cartpole16_run_kernel' if T <= 8192 else 'cartpole_kernel<false'
            self.flops_per_eval = 223232
        self.noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).to(device).contiguous()
        self.u = torch.rand(T, generator=g, device=device, dtype=torch.float64)
        self.temperature = torch.ones(T, dtype=torch.float64, device=device)
        self.hooks = MoveEvents(wl['N'], event_stride)
        self.evals_per_launch = self.trees_per_launch
        self.one_launch = self.dominant in ('cartpole16_run_kernel', 'atari_run_kernel', 'ttt_run_kernel')
        self.launch_includes = {'cartpole16_run_kernel': 'the tree steps of all its simulations (trees in LDS)',
                                'atari_run_kernel': 'the tree steps (expand, backup, next select) of all its simulations',
                                'ttt_run_kernel': 'the prediction heads and tree steps of all its simulations'
                                }.get(self.dominant)
        if self.one_launch:
            # mz_search_move records the first sampled simulation's events around the whole run: one launch = N evaluations
            # of every game
            self.hooks.sampled = self.hooks.sampled[:1]
            self.evals_per_launch = self.trees_per_launch * wl['N']
        self.gather = TrajectoryGather(T, A, device) if world > 1 else None
        self.job_ticks0 = (0, 0)

    def mark_jobs(self):
        """start of the timed region: the job-phase counters so far (TicTacToe one-launch move)"""
        if self.dominant == 'ttt_run_kernel':
            self.job_ticks0 = self.net.run_info_all()[2:]

    def run(self, timed_events=False):
        events = self.hooks.new_set() if timed_events else None
        visits, root_value, minmax, policy, action = self.search.run_native(self.obs, self.noise, self.u,
                                                                            self.temperature, events=events)
        if self.gather is not None:      # the one exchange step: search statistics of this move to every rank
            self.gather.start(policy, root_value, action)
        return visits, action

    def finish(self):
        if self.gather is not None:
            self.gather.wait()


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_hbm_traffic_<workload>.json:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs, FETCH_SIZE doubled per the gfx950 correction).  Counters cannot be
    read from inside this process, so the figure is the one of the latest committed profile; None when there is none."""
    for path in profile_files('hbm_traffic', workload):
        with open(path) as f:
            data = json.load(f)
        rows = [row for name, row in data.get('kernels', {}).items() if name.startswith(kernel)]
        if rows:
            row = max(rows, key=lambda r: r['launches'])  # e.g. step_kernel<G, P, true> among its instantiations
            return row['traffic_bytes_per_launch'], os.path.basename(path)
    return None, None


def profile_files(what, workload):
    """The committed profiles of a workload, latest round first: profiles/rNN_<what>_<workload>.json and, for the
    workloads that are also profiled with their network in the loop beside the injected tree-only run, ..._net.json."""
    import glob
    files = glob.glob(os.path.join(ROOT, 'profiles', f'*_{what}_{workload}.json'))
    files += glob.glob(os.path.join(ROOT, 'profiles', f'*_{what}_{workload}_net.json'))
    return sorted(files, key=os.path.basename, reverse=True)


def profiled_stats(workload, kernel):
    """Duration statistics of `kernel` from the committed rocprofv3 kernel trace of this workload
    (profiles/*_kernel_medians_<workload>.json, made by tools/profile_workload.sh from `rocprofv3 --kernel-trace`), with the
    ms_per_step the profiled run itself printed: the figures the live HIP-event average should agree with."""
    for path in profile_files('kernel_medians', workload):
        with open(path) as f:
            data = json.load(f)
        rows = [dict(row, name=name) for name, row in data.get('kernels', {}).items() if name.startswith(kernel)]
        if rows:
            row = max(rows, key=lambda r: r['calls'])
            return {'kernel': row['name'], 'median_us': row['median_us'], 'mean_us': row['mean_us'],
                    'min_us': row['min_us'], 'calls': row['calls'], 'ms_per_step_under_rocprof': data.get('ms_per_step'),
                    'source': os.path.basename(path)}
    return None


def host_cores():
    """Cores this process may use: the GPU box exports its per-GPU share as OMP_NUM_THREADS (os.cpu_count() there counts
    the whole machine); elsewhere the visible CPUs."""
    share = int(os.environ.get('OMP_NUM_THREADS', '0') or 0)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return max(1, min(visible, share) if share > 0 else visible)


class CpuBaseline:
    """The CPU oracle -- a port of the reference's Python MCTS (oracle/mcts_oracle.py) calling the numpy network port
    (oracle/net_oracle.py, float32, batch 1 per simulation like agent.py:204-207) -- one single-threaded process per host
    core (oracle/cpu_bench.py), same shapes, each for about `seconds`.  value = the sum of the workers'
    rates.  Workloads without a network time the tree alone.

    The workers are STARTED before this process makes its first HIP call (no child is ever forked from a process that
    holds the GPU) and sit idle on their stdin; `collect()` releases them after the GPU's timed region, so they neither
    share the host cores with the enqueueing thread nor run inside anyone's timing."""

    def __init__(self, wl, seconds, cores=None):
        import subprocess
        self.wl, self.seconds, self.cores = wl, seconds, cores or host_cores()
        cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'cpu_bench.py'), '--actions', str(wl['A']), '--simulations',
               str(wl['N']), '--players', str(wl['players']), '--discount', str(wl['discount']), '--default-value',
               str(wl['default']), '--net', wl['net'] or 'none', '--seconds', str(seconds), '--wait-for-go']
        if wl['bounds'] is not None:
            cmd += ['--bounds', str(wl['bounds'][0]), str(wl['bounds'][1])]
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        self.procs = [subprocess.Popen(cmd + ['--seed', str(99 + i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env,
                                       text=True) for i in range(self.cores)]

    def collect(self):
        for pr in self.procs:                      # imports done: every worker is at its starting line
            assert pr.stdout.readline().strip() == 'ready', 'cpu_baseline worker did not start'
        for pr in self.procs:
            pr.stdin.write('go\n')
            pr.stdin.flush()
        rows = []
        for pr in self.procs:
            out, _ = pr.communicate(timeout=self.seconds * 10 + 120)
            if pr.returncode != 0:
                raise RuntimeError(f'cpu_baseline worker failed with {pr.returncode}')
            rows.append(json.loads(out.strip().splitlines()[-1]))
        wl, cores = self.wl, self.cores
        rate = sum(r['simulations'] / r['seconds'] for r in rows)
        moves = sum(r['moves'] for r in rows)
        what = ('python tree oracle + numpy fp32 network oracle at batch 1' if wl['net'] else 'python tree oracle only')
        return dict(value=rate, unit='simulations/s', cores=cores, kind='port', per_core=rate / cores,
                    sample=f"{moves} single-game moves x {wl['N']} simulations over {cores} single-threaded processes, "
                           f"A={wl['A']}, {what}, {max(r['seconds'] for r in rows):.1f} s each")

    def abandon(self):
        for pr in self.procs:
            pr.stdin.close()                       # readline() returns '' and the worker exits
            pr.wait()


def event_record_cost(hip_events, n=33):
    """Stream time between two hipEventRecords with nothing between them (s, median of n): what every bracketed
    duration below carries on top of the kernel it brackets, and what is subtracted from it."""
    pairs = [(hip_events.record(), hip_events.record()) for _ in range(n)]
    torch.cuda.synchronize()
    return float(np.median([hip_events.elapsed_ms(a, b) for a, b in pairs])) * 1e-3


def timed_region(move, steps, warmup, world, with_net, event_every=10, sync_every=0, prewarm_s=0.0):
    """W untimed warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides; the MAX over
    ranks.  -> (seconds, simulations this rank backed up, sum of their path lengths)
    prewarm_s: untimed moves for that long BEFORE the W warm-up steps.  Five warm-up moves are 6 ms; measured on this part
    the same binary then gives 84 M simulations/s over the next 20 moves and 89 M after 100 or more warm-up moves (the
    GPU's clocks are still on their way up from idle) -- the figure of interest is the second, and it is what a
    timed region of any length reports once the chip has been busy for a few hundred milliseconds."""
    import torch.distributed as dist
    # The pre-warm is bounded by time, so the ranks of an N > 1 run make different numbers of moves in it: it must hold no
    # collective (the ranks' collectives pair up by order), and runs with the per-move all-gather switched off.
    gather, t_pre = getattr(move, 'gather', None), time.perf_counter()
    if gather is not None:
        move.gather = None
    while prewarm_s > 0 and time.perf_counter() - t_pre < prewarm_s:
        for _ in range(8):
            move.run(timed_events=False)
        torch.cuda.synchronize()
    if gather is not None:
        move.gather = gather
    for i in range(warmup):
        move.run(timed_events=(i == 0))              # the first warm-up step also warms the event-recording path
    if with_net:
        move.finish()
        move.hooks.sets.clear()                      # warm-up timings are not reported
    else:
        move.event_sets.clear()
        move.run_sets.clear()
    torch.cuda.synchronize()
    c0 = move.counters()
    if with_net:
        move.mark_jobs()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        # per-launch events on every event_every-th step; the tree-only moves bracket the run of fused steps on the next two
        move.run(timed_events=(i % event_every == 0) or ('run' if not with_net and i % event_every in (1, 2) else False))
        if sync_every and (i + 1) % sync_every == 0:
            torch.cuda.synchronize()
    if with_net:
        move.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    c1 = move.counters()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=move.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed, int(c1[0] - c0[0]), int(c1[1] - c0[1])


def tree_roofline(move, wl, name, mean_d, cost):
    """HBM roofline of the tree kernels of an injected move: the one launch that runs all N simulations
    (run_injected_kernel, the default), with the launch-per-simulation kernel (step_kernel) timed beside it."""
    bytes_sim = bytes_per_simulation(wl['A'], mean_d, wl['A'])
    uncost = lambda t: max(t - cost, 0.5 * t)
    run = uncost(move.run_durations())
    pair = move.step_durations()
    bytes_launch = move.trees_per_launch * wl['N'] * bytes_sim
    traffic, traffic_src = measured_traffic(name, 'run_injected_kernel')
    out = {'bound': 'hbm', 'kernel': 'run_injected_kernel', 'achieved': bytes_launch / run / 1e9, 'peak': HBM_PEAK_GBS,
           'unit': 'GB/s', 'frac': bytes_launch / run / 1e9 / HBM_PEAK_GBS, 'traffic': traffic,
           'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC)', 'traffic_source': traffic_src,
           'avg_launch_us': run * 1e6, 'event_record_cost_us': cost * 1e6, 'bytes_per_launch': bytes_launch,
           'simulations_per_launch': wl['N'], 'us_per_simulation': run * 1e6 / wl['N'], 'mean_path_len': mean_d}
    if pair is not None:
        step = uncost(pair)
        step_bytes = move.trees_per_launch * bytes_sim
        out['launch_per_simulation'] = {'kernel': 'step_kernel', 'avg_launch_us': step * 1e6, 'bytes_per_launch': step_bytes,
                                        'achieved': step_bytes / step / 1e9, 'frac': step_bytes / step / 1e9 / HBM_PEAK_GBS}
    return out


def tree_roofline_leg(name, device, steps, seed):
    """The tree kernels alone (injected network outputs) at one of BASELINE.json's synthetic shapes: the
    north star's HBM-roofline target is about THESE launches.  One native call enqueues a move; the fused step
    (backup of simulation s + select of s+1) is the launch that repeats N-1 times."""
    wl = dict(WORKLOADS[name])
    stride = max(1, wl['N'] // 10)
    move = SyntheticTreeMove(wl, device, seed, 1, stride)
    move.hip_events.reserve(9 * (wl['N'] // stride + 2))
    cost = event_record_cost(move.hip_events)
    elapsed, sims, sum_d = timed_region(move, steps, 2, 1, False, event_every=max(3, steps))   # two 'run' samples a leg
    assert sims == steps * wl['T'] * wl['N'], (sims, steps, wl)
    mean_d = sum_d / sims
    out = tree_roofline(move, wl, name, mean_d, cost)
    out.update({'workload': name, 'trees': wl['T'], 'actions': wl['A'], 'simulations_per_move': wl['N'], 'steps': steps,
                'simulations_per_sec': sims / elapsed, 'ms_per_move': elapsed / steps * 1e3,
                # the first step of a leg is a launch per simulation with events (the launch_per_simulation figures), so the
                # leg's own rate is below that of the one-launch run it reports beside it:
                'one_launch_simulations_per_sec': wl['T'] * wl['N'] / (out['avg_launch_us'] * 1e-6),
                'note': 'simulations_per_sec is over all steps of the leg, one of which is a launch per simulation with '
                        'events; one_launch_simulations_per_sec = trees x simulations / avg_launch_us'})
    for l in move.lanes:
        l['pool'].close()
    del move
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--prewarm-ms', type=float, default=300.0,
                    help='untimed moves for this long before the W warm-up steps of the headline leg (0: none); see '
                         'timed_region')
    ap.add_argument('--workload', default='tictactoe_4096x25', choices=sorted(WORKLOADS))
    ap.add_argument('--trees', type=int, default=None, help='override trees per GPU (profiling only)')
    ap.add_argument('--inject', action='store_true', help='tree kernels only, injected outputs (profiling only)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-legs', action='store_true',
                    help='only the headline leg: no tree-kernel roofline legs, no 512-tree leg, no strong-scaling leg')
    ap.add_argument('--run-mode', type=int, default=0,
                    help='TicTacToe A/B measurements: 1 = the one-launch move with agent-scope fences, 2 = a launch per '
                         'simulation (Network.set_run_mode)')
    ap.add_argument('--lanes-per-tree', type=int, default=0,
                    help='tree-only A/B measurements: lanes of a wavefront per tree (TreePool lanes_per_tree; 0 = its choice)')
    ap.add_argument('--job-rows', type=int, default=0, choices=[0, 16, 32, 64],
                    help='TicTacToe A/B measurements: games per workgroup of the job kernel (Network.set_job_rows)')
    ap.add_argument('--sync-every', type=int, default=0, help='synchronise the host every n timed steps (0 = never)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help="weak: the workload's games PER GPU (default); strong: the workload's games in total, sharded "
                         "over the ranks (BASELINE.json configs[2] read as 4096 games over 1/2/4/8 GPUs)")
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl (= RCCL, one rank per GPU: what the driver runs); gloo rehearses the N > 1 line on a box with '
                         'fewer GPUs than ranks (ranks share GPUs, the collectives go through the host: not a measurement)')
    ap.add_argument('--lanes', type=int, default=1,
                    help='split the games of a GPU into this many lockstep lanes on separate HIP streams')
    ap.add_argument('--event-every', type=int, default=10,
                    help='record the per-kernel HIP events on every K-th timed step (each record costs stream time)')
    ap.add_argument('--event-stride', type=int, default=5,
                    help='... and there on every S-th simulation only')
    args = ap.parse_args()

    if os.environ.get('MZ_BENCH_WATCHDOG'):          # rehearsals: where every thread stands after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['MZ_BENCH_WATCHDOG']), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    wl = dict(WORKLOADS[args.workload])
    if args.trees:
        wl['T'] = args.trees
    with_net = wl['net'] is not None and not args.inject
    default_line = (args.trees is None and args.lanes == 1 and not args.inject and args.run_mode == 0 and
                    args.job_rows == 0)
    extra_legs = default_line and not args.no_extra_legs and args.workload == 'tictactoe_4096x25'

    # BEFORE the first HIP call: the CPU baseline's worker processes (a one-GPU figure: with more ranks their processes
    # would share the host cores).  They idle until the GPU legs are over.
    cpu = None
    if not args.no_cpu_baseline and world == 1 and rank == 0:
        cpu = CpuBaseline(wl if with_net else dict(wl, net=None), args.cpu_seconds)

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.dist_backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group('gloo')
    else:
        torch.cuda.set_device(0)
    device = torch.device('cuda', torch.cuda.current_device())

    full_T = wl['T']
    if args.scaling == 'strong' and world > 1:       # the same games in total, a contiguous shard per rank
        from muzero_amd.selfplay import shard_bounds
        lo, hi = shard_bounds(wl['T'], rank, world)
        wl['T'] = hi - lo
    # per-rank seeds: every rank searches different games
    stride = max(1, min(args.event_stride, wl['N'] // 4))
    move = (NetworkMove(wl, device, 1234 + rank, world, args.lanes, stride, args.run_mode, args.job_rows) if with_net
            else SyntheticTreeMove(wl, device, 1234 + rank, args.lanes, stride, args.lanes_per_tree))
    hip_events = move.hooks.ev if with_net else move.hip_events
    timed_steps = (args.steps + args.event_every - 1) // args.event_every + 1
    hip_events.reserve(timed_steps * 3 * (wl['N'] // stride + 1) + 80)
    record_cost = event_record_cost(hip_events)
    elapsed, sims_rank, sum_d = timed_region(move, args.steps, args.warmup, world, with_net, args.event_every,
                                             args.sync_every, prewarm_s=args.prewarm_ms * 1e-3)
    assert sims_rank == args.steps * wl['T'] * wl['N'], (sims_rank, args.steps, wl['T'], wl['N'])
    total_sims = sims_rank
    ranks_seen = 1
    if world > 1:                                    # ranks hold equal shards under weak scaling, near-equal under strong
        tot = torch.tensor([sims_rank, 1], dtype=torch.int64, device=device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)   # [1]: an all-reduce of ones = the ranks RCCL really connected
        total_sims, ranks_seen = int(tot[0].item()), int(tot[1].item())
    mean_d = sum_d / sims_rank
    tree_bytes_launch = move.trees_per_launch * bytes_per_simulation(wl['A'], mean_d, wl['A'])
    uncost = lambda t: None if t is None else max(t - record_cost, 0.5 * t)     # a bracketed duration without its record
    extra = {}
    # every network kernel but CartPole's large-batch one (batches > 8192) computes in split-f16 arithmetic
    split_f16 = with_net and not (wl['net'] == 'cartpole' and wl['T'] > 8192)
    if with_net:
        k1_raw, rest_raw = move.hooks.durations()
        k1, rest = uncost(k1_raw), uncost(rest_raw)
        flops_launch = move.evals_per_launch * move.flops_per_eval
        achieved = flops_launch / k1 / 1e12
        traffic, traffic_src = (measured_traffic(args.workload, move.dominant) if default_line else (None, None))
        if split_f16:
            # The kernel multiplies on f16 MFMA with every f32 operand split in two halfs: three f16 MFMA products
            # per algorithmic product, held to the f32 tolerance.  Priced against the f16 peak the instructions run at;
            # the f32-equivalent figures say what the same algorithmic work would need of the f32 MFMA peak.
            peak, mfma_dtype, executed = F16_MFMA_PEAK_TFLOPS, 'f16 (split f32: h + l/2048), f32 accumulate', 3 * achieved
        else:
            peak, mfma_dtype, executed = F32_MFMA_PEAK_TFLOPS, 'f32', achieved
        roofline = {'bound': 'mfma', 'kernel': move.dominant, 'achieved': achieved, 'peak': peak,
                    'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic,
                    'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC)', 'traffic_source': traffic_src,
                    'avg_launch_us': k1 * 1e6, 'event_pair_us': k1_raw * 1e6, 'event_record_cost_us': record_cost * 1e6,
                    'flops_per_launch': flops_launch, 'mfma_operands': mfma_dtype,
                    'executed_mfma_tflops': executed, 'executed_frac': executed / peak}
        if move.dominant == 'ttt_run_kernel':
            # the job phase inside the launch (what ttt_ring_kernel was as a launch of its own until round 2): the
            # workgroups' own wall-clock stamps around their jobs, averaged over workgroups and simulations
            _, _, ticks, jobs = move.net.run_info_all()
            t0, j0 = move.job_ticks0
            if jobs > j0:
                job_s = (ticks - t0) / (jobs - j0) * 1e-8
                job_flops = move.trees_per_launch * 2 * (49 * 137 * 128 + 9 * 128 * 128 + 1152 + 9 * 128 * 128)
                roofline['job_phase'] = {
                    'what': 'the jobs (conv taps, Dense, reward dots, prediction-conv terms: 2,310,656 FLOP per evaluation) of '
                            'one simulation inside the launch; mean over workgroups of their own 100 MHz wall-clock stamps',
                    'avg_us': job_s * 1e6, 'achieved': job_flops / job_s / 1e12, 'frac': job_flops / job_s / 1e12 / peak,
                    'executed_frac': 3 * job_flops / job_s / 1e12 / peak}
        one_launch = move.one_launch
        rest_name = ('after_fused_kernel_us' if one_launch else
                     'head_and_tree_step_kernel_us' if wl['net'] == 'tictactoe' else 'tree_step_kernel_us')
        if one_launch:
            roofline['simulations_per_launch'] = wl['N']
            roofline['us_per_simulation'] = k1 * 1e6 / wl['N']
            roofline['launch_includes'] = move.launch_includes
        extra = {rest_name: rest * 1e6, 'tree_algorithmic_bytes_per_launch': tree_bytes_launch, 'mean_path_len': mean_d,
                 'recurrent_inference_evals_per_sec': total_sims / elapsed}
        if rest_name == 'tree_step_kernel_us':       # the tree step is a launch of its own: its HBM roofline beside it
            extra['tree_step_roofline'] = {'bound': 'hbm', 'kernel': 'step_kernel', 'achieved': tree_bytes_launch / rest / 1e9,
                                           'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                           'frac': tree_bytes_launch / rest / 1e9 / HBM_PEAK_GBS, 'avg_launch_us': rest * 1e6}
        network = f"{wl['net']} network, random-init weights, f32 results (1e-5 of the fp64 oracle)"
    else:
        roofline = tree_roofline(move, wl, args.workload, mean_d, record_cost)
        if not (args.trees is None and args.lanes == 1):
            roofline['traffic'] = roofline['traffic_source'] = None
        # all lanes together: algorithmic bytes of every simulation of the timed region over its wall time
        extra = {'aggregate_algorithmic_GBps': sims_rank * bytes_per_simulation(wl['A'], mean_d, wl['A']) / elapsed / 1e9}
        network = 'injected outputs (tree kernels only)'

    # ---- further legs of the default line (outside the headline's timed region)
    if extra_legs:
        del move
        torch.cuda.empty_cache()
        if world == 1:
            # the tree kernels' HBM roofline at BASELINE configs[3] and [4] (tree half), where the driver sees it
            extra['tree_roofline'] = [tree_roofline_leg('atari_16384x50', device, 6, 4321),
                                      tree_roofline_leg('go_2048x800', device, 3, 4322)]
            # one rank's shard of configs[2] read as strong scaling over 8 GPUs: 512 games on this one GPU
            small = dict(wl, T=512)
            mv = NetworkMove(small, device, 77, 1, 1, stride)
            mv.hooks.ev.reserve(8)
            el, sm, _ = timed_region(mv, 40, 5, 1, True, event_every=1 << 30)
            extra['strong_scaling_proxy'] = {'trees': 512, 'simulations_per_sec': sm / el, 'ms_per_move': el / 40 * 1e3,
                                             'note': 'one GPU searching the 512 games an 8-GPU strong-scaling run of '
                                                     '4096 games gives each rank'}
            del mv
        else:
            # the other reading of configs[2]: the SAME 4096 games sharded over the ranks (strong scaling)
            from muzero_amd.selfplay import shard_bounds
            lo, hi = shard_bounds(full_T, rank, world)
            mv = NetworkMove(dict(wl, T=hi - lo), device, 4000 + rank, world, 1, stride)
            mv.hooks.ev.reserve(8)
            el, sm, _ = timed_region(mv, args.steps, args.warmup, world, True, event_every=1 << 30)
            tot = torch.tensor([sm], dtype=torch.int64, device=device)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            extra['strong_scaling'] = {'trees_total': full_T, 'trees_per_gpu': hi - lo,
                                       'simulations_per_sec': int(tot.item()) / el, 'ms_per_step': el / args.steps * 1e3}
            del mv
        torch.cuda.empty_cache()

    if rank == 0:
        out = {
            'metric': 'mcts_simulations_per_sec', 'value': total_sims / elapsed, 'unit': 'simulations/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'prewarm_ms': args.prewarm_ms,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': ('f32 via split-f16 MFMA' if split_f16 else 'f32') if with_net else 'f64',
            'data': 'synthetic',
            'config': {'workload': args.workload, 'trees_per_gpu': wl['T'], 'lanes_per_gpu': args.lanes,
                       'actions': wl['A'],
                       'simulations_per_move': wl['N'], 'network': network,
                       'sharding': f'{world} independent game shards (one rank per GPU)' +
                                   (', per-move all-gather of search statistics' if world > 1 and with_net else ''),
                       'ranks_in_all_reduce': ranks_seen},
            'roofline': roofline,
        }
        if world > 1 and args.dist_backend != 'nccl':
            out['rehearsal'] = f'{args.dist_backend} backend, ranks share GPUs: a test of the N > 1 line, not a measurement'
        if default_line:
            out['roofline']['rocprof'] = profiled_stats(args.workload, roofline['kernel'])
        out.update(extra)
        if cpu is not None:
            out['cpu_baseline'] = cpu.collect()
        print(json.dumps(out))
    elif cpu is not None:
        cpu.abandon()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
