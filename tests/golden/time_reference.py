#!/usr/bin/env python3
"""Calibration of bench.py's cpu_baseline: how fast the build's CPU restatement of the search
(oracle/mcts_oracle.py) runs next to the UNMODIFIED reference agent on the same core, same shapes, same injected network
outputs.  Build container only (needs /root/reference; uses make_golden.py's stub-import harness):

    python tests/golden/time_reference.py [seconds per leg]

Both legs are tree-only: the reference leg is MCTSAgent.expand_node + add_exploration_noise + run_mcts + select_action
(agent.py:154-232) against a fake network handing out pre-generated outputs; the restatement leg is what
oracle/cpu_bench.py --net none times.  Prints simulations/s of each and the ratio restatement / reference per shape."""
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg                       # noqa: E402  (installs the stubs, imports the reference)
from oracle import mcts_oracle                 # noqa: E402

SHAPES = [   # name, A, N, players, discount, default value, known bounds
    ('tictactoe A=9 N=25', 9, 25, 2, 1.0, 0.0, (None, 1.0)),
    ('cartpole A=2 N=50', 2, 50, 1, 0.99, 50.0, (0.0, 100.0)),
    ('atari-shaped A=18 N=50', 18, 50, 1, 0.997, 0.0, None),
    ('go-shaped A=362 N=800', 362, 800, 2, 1.0, 0.0, (-1.0, 1.0)),
]


def outputs(rng, A, N):
    return (rng.standard_normal((N + 1, A)).astype(np.float32), np.tanh(rng.standard_normal(N + 1)).astype(np.float32),
            rng.random(N + 1).astype(np.float32))


def reference_leg(A, N, players, discount, default, bounds, seconds):
    config = mg.make_config(A, players, discount, N, default, bounds)
    rng = np.random.Generator(np.random.PCG64(5))
    legal = list(range(A))
    sims = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        logits, values, rewards = outputs(rng, A, N)
        net = mg.FakeNetwork(config, values, rewards, logits)
        agent = mg.ref_agent.MCTSAgent(config=config, network=net, agent_id='timing')
        root = mg.ref_agent.Node()
        mm = mg.ref_utils.MinMaxStats(known_bounds=config.value_config.known_bounds)
        game = mg.ref_game.Game(environment=mg.FixedEnvironment(A, players, legal))
        observation = np.expand_dims(np.asarray(game.history.make_image()), 0).view(mg.FakeTensor)
        agent.expand_node(node=root, actions=game.legal_actions(),
                          network_output=net.initial_inference(observation).split_batch()[0])
        agent.add_exploration_noise(root)
        agent.run_mcts(root, mm)
        agent.select_action(root, len(game.history))
        sims += N
    return sims / (time.perf_counter() - t0)


def restatement_leg(A, N, players, discount, default, bounds, seconds):
    kb = None
    if bounds is not None:
        kb = mg.ref_utils.KnownBounds(minv=bounds[0], maxv=bounds[1]).endpoints()
    rng = np.random.Generator(np.random.PCG64(5))
    sims = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        logits, values, rewards = outputs(rng, A, N)
        tree = mcts_oracle.TreeOracle(A, N, players, discount, default, kb)
        tree.expand_root(mcts_oracle.masked_policy_f32(logits[0], range(A)), range(A))
        tree.add_root_noise(rng.dirichlet([1.0] * A), 0.25)
        for s in range(N):
            tree.select()
            tree.expand_backprop(mcts_oracle.masked_policy_f32(logits[s + 1], range(A)), rewards[s + 1], values[s + 1])
        tree.select_action(float(rng.random()), 1.0)
        sims += N
    return sims / (time.perf_counter() - t0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    random.seed(1)
    np.random.seed(1)
    print(f'one core, {seconds:.0f} s per leg, Python {sys.version.split()[0]}, numpy {np.__version__}')
    for name, A, N, players, discount, default, bounds in SHAPES:
        ref = reference_leg(A, N, players, discount, default, bounds, seconds)
        port = restatement_leg(A, N, players, discount, default, bounds, seconds)
        print(f'{name:26s} reference {ref:9.0f} sims/s   restatement {port:9.0f} sims/s   ratio {port / ref:5.2f}', flush=True)


if __name__ == '__main__':
    main()
