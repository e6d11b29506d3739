"""CPU oracle for the MCTS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
path (muzero_amd/) never falls back to it: it fails loudly when the HIP library is missing.

This is a from-scratch, single-tree restatement of the reference algorithm on flat per-block arrays
(the same block layout the HIP tree pool uses), in plain Python float (IEEE fp64) arithmetic.
Parity is PINNED: tests/test_oracle_golden.py checks it against every fixture in tests/golden/, which were
produced by the unmodified reference agent (tests/golden/make_golden.py).

Reference lines restated (all /root/reference):
  * Node.update_value                agent.py:119-123
  * MinMaxStats.update / normalize   utils.py:63-76, KnownBounds.endpoints utils.py:45-49
  * exploration_function             config.py:159-160        (2*log(Np+1)/(Nc+1))
  * effective discount               agent.py:150-152
  * expand_node + masked_policy      agent.py:154-160, network.py:142-146   (positional priors, trap T5)
  * add_exploration_noise            agent.py:170-175
  * select_leaf/select_child/ucb     agent.py:177-190         (max over (score, action, child) tuples)
  * backpropagate                    agent.py:192-196
  * run_mcts                         agent.py:198-211
  * softmax_sample / select_action   agent.py:162-168,213-217; config.py:153-157
  * policy target / root value       agent.py:229-231

Semantics are the reference's era semantics (SURVEY.md trap T3): scores are fp64, non-root priors are
fp32 values widened to fp64, root priors become fp64 after the noise mix.
"""
import math
from bisect import bisect_right
from itertools import accumulate

import numpy as np

NAN = float('nan')
INF = float('inf')


def masked_policy_f32(logits, actions):
    """network.py:142-146 in numpy float32: exp(logits)*mask / sum, no max-subtraction.

    np.exp on float32 is SIMD-dependent (trap T4, <=2 ulp), so bit-exact tests inject the fixture priors
    instead of calling this; it is used where priors are derived from logits (tolerance tests, baselines).
    """
    logits = np.asarray(logits, np.float32)
    mask = np.zeros_like(logits)
    mask[list(actions)] = 1
    policy = np.exp(logits) * mask
    return policy / policy.sum()


class TreeOracle:
    """One search tree.  Block e holds the children of the node expanded by network call e (root = 0)."""

    def __init__(self, num_actions, num_simulations, num_players=1, discount=1.0, default_value=0.0,
                 known_bounds=None):
        self.A = int(num_actions)
        self.N = int(num_simulations)
        self.discount = float(discount) * (-1 if num_players == 2 else 1)     # agent.py:150-152
        self.default_value = float(default_value)
        self.known_bounds = None if known_bounds is None else (float(known_bounds[0]), float(known_bounds[1]))
        # config.py:159-160; parent visit count <= N. Tabulated exactly like the product does (trap T6).
        self.log_table = [2 * math.log(n + 1) for n in range(self.N + 2)]
        self.reset()

    # ------------------------------------------------------------------ state
    def reset(self):
        """agent.py:220-221: a fresh root and MinMaxStats per move (no tree reuse)."""
        self.actions = []      # per block: list of actions, in insertion order
        self.count = []
        self.vsum = []
        self.prior = []
        self.reward = []
        self.child = []
        self.root_count = 0
        self.root_vsum = 0.0
        self.mm_min = INF      # utils.py:60-61
        self.mm_max = -INF
        self.path = []         # [(block, k)] of the last select()
        self.sum_path_len = 0  # measurement: sum over simulations of expanded nodes on the path (d)

    def _new_block(self, actions, priors):
        n = len(actions)
        self.actions.append(list(actions))
        self.count.append([0] * n)
        self.vsum.append([0.0] * n)
        self.prior.append([float(p) for p in priors])
        self.reward.append([NAN] * n)
        self.child.append([-1] * n)
        return len(self.actions) - 1

    # ------------------------------------------------------------------ MinMaxStats (utils.py:63-76)
    def _mm_update(self, value):
        if self.known_bounds is None:
            if value > self.mm_max:
                self.mm_max = value
            if value < self.mm_min:
                self.mm_min = value

    def _normalize(self, value):
        if self.known_bounds is None:
            if self.mm_max == self.mm_min:
                return value
            a, b = self.mm_min, self.mm_max
        else:
            a, b = self.known_bounds
        return (value - a) / (b - a)

    # ------------------------------------------------------------------ expansion
    def expand_root(self, priors_full, legal_actions):
        """Root expansion: child k (action legal[k]) gets priors_full[k] -- positional (trap T5)."""
        assert not self.actions, 'reset() first'
        legal = [int(a) for a in legal_actions]
        assert 1 <= len(legal) <= self.A
        self._new_block(legal, [float(np.float32(priors_full[k])) for k in range(len(legal))])

    def add_root_noise(self, noise, frac):
        """agent.py:170-175 (fp64; two roundings per term, no contraction)."""
        frac = float(frac)
        for k in range(len(self.actions[0])):
            self.prior[0][k] = self.prior[0][k] * (1 - frac) + float(noise[k]) * frac

    # ------------------------------------------------------------------ selection
    def _score(self, e, k, parent_count):
        c = self.count[e][k]
        value = self.vsum[e][k] / c if c > 0 else NAN
        if value != value:
            exploit = self.default_value
        else:
            exploit = self.reward[e][k] + self.discount * value
        explore = self.prior[e][k] * (self.log_table[parent_count] / (c + 1))
        return self._normalize(exploit) + explore

    def select(self):
        """select_leaf: returns (parent block, action) of the leaf and records the path."""
        e, parent_count = 0, self.root_count
        self.path = []
        while True:
            acts = self.actions[e]
            best_k, best_s = 0, self._score(e, 0, parent_count)
            for k in range(1, len(acts)):
                s = self._score(e, k, parent_count)
                # Python max over (score, action, child) tuples: replace only when strictly greater
                if s > best_s or (s == best_s and acts[k] > acts[best_k]):
                    best_k, best_s = k, s
            self.path.append((e, best_k))
            if self.count[e][best_k] > 0:          # visited <=> expanded (every simulated leaf is expanded)
                parent_count = self.count[e][best_k]
                e = self.child[e][best_k]
            else:
                break
        self.sum_path_len += len(self.path)
        e, k = self.path[-1]
        return e, self.actions[e][k]

    # ------------------------------------------------------------------ expand + backprop
    def expand_backprop(self, priors, reward, value):
        """expand_node(leaf, all actions) + backpropagate(leaf, value) for the path of the last select()."""
        e, k = self.path[-1]
        new_block = self._new_block(range(self.A), [float(np.float32(p)) for p in priors])
        self.child[e][k] = new_block
        self.reward[e][k] = float(np.float32(reward))
        v = float(np.float32(value))
        for (pe, pk) in reversed(self.path):
            self.vsum[pe][pk] += v
            self.count[pe][pk] += 1
            self._mm_update(self.vsum[pe][pk] / self.count[pe][pk])
            v = self.reward[pe][pk] + self.discount * v
        self.root_vsum += v
        self.root_count += 1
        self._mm_update(self.root_vsum / self.root_count)
        return new_block

    def run(self, values, rewards, priors, first_call=1):
        """run_mcts with injected outputs: call i supplies values[i], rewards[i], priors[i]."""
        trace_parent, trace_action = [], []
        for s in range(self.N):
            e, a = self.select()
            trace_parent.append(e)
            trace_action.append(a)
            i = first_call + s
            self.expand_backprop(priors[i], rewards[i], values[i])
        return trace_parent, trace_action

    # ------------------------------------------------------------------ outputs
    def root_visits(self):
        visits = np.zeros(self.A, np.int32)
        for k, a in enumerate(self.actions[0]):
            visits[a] = self.count[0][k]
        return visits

    def root_value(self):
        return self.root_vsum / self.root_count if self.root_count > 0 else NAN

    def policy_target(self):
        """agent.py:229-231: python floats -> tf.constant -> float32."""
        pol = [0.0] * self.A
        for k, a in enumerate(self.actions[0]):
            pol[a] = self.count[0][k] / self.root_count
        return np.asarray(pol, np.float64).astype(np.float32)

    def select_action(self, u, temperature):
        """softmax_sample (agent.py:162-168) with the uniform draw u = random.random() injected.

        random.choices(population, weights) == population[bisect_right(cum, u*total, 0, n-1)].
        """
        dist = [(self.count[0][k], a) for k, a in enumerate(self.actions[0])]
        if temperature == 0.0:
            return max(dist)[1]
        weights = [c ** (1 / temperature) for c, _ in dist]
        cum = list(accumulate(weights))
        total = cum[-1] + 0.0
        return dist[bisect_right(cum, u * total, 0, len(dist) - 1)][1]


def temperature_for(config_temperature, num_moves, freezing_moves):
    """config.py:153-157."""
    return config_temperature if num_moves < freezing_moves else 0.0


def run_fixture(fx):
    """Replay one tests/golden fixture through the oracle; returns (oracle, trace_parent, trace_action, action)."""
    kb = (float(fx['known_min']), float(fx['known_max'])) if int(fx['has_bounds']) else None
    tree = TreeOracle(int(fx['A']), int(fx['N']), int(fx['num_players']), float(fx['discount']),
                      float(fx['default_value']), kb)
    legal = [int(a) for a in fx['legal']]
    tree.expand_root(fx['priors'][0], legal)
    tree.add_root_noise(fx['noise'], float(fx['frac']))
    tp, ta = tree.run(fx['values'], fx['rewards'], fx['priors'])
    t = temperature_for(float(fx['temperature']), int(fx['num_moves']), int(fx['freezing_moves']))
    action = tree.select_action(float(fx['u']), t)
    return tree, tp, ta, action
