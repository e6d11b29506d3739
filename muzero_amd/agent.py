"""Agent API of the reference (agent.py:22-235) on top of the HIP tree pool.

`MCTSAgent(config, network, agent_id=None, debug=False)` keeps the reference's constructor and its eleven
methods.  The tree is not a graph of Python objects any more: it lives in a `TreePool` (one tree for this
single-game agent; `selfplay.LockstepSearch` is the same machinery for thousands of games) and `Node` is a
read-only view of one node of that pool.  `network` is duck-typed exactly like in the reference
(agent.py:141): anything with `initial_inference`, `recurrent_inference` and `training_steps`; a HIP-backed
`muzero_amd.network.Network` keeps every tensor on the device.

Host randomness is consumed like the reference consumes it -- `np.random.dirichlet` once per move
(agent.py:172) and one `random.random()` per sampled action (`random.choices`, agent.py:168) -- so a seeded run
reproduces the reference's moves when the network outputs agree.
"""
import random
import time
from math import isnan
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .exceptions import MuZeroError, MuZeroImplementationError
from .game import Game, GameHistory
from .network import NetworkOutput
from .tree_pool import TreePool
from .utils import MinMaxStats, random_id


class Agent:
    """An agent playing a game in an Environment (agent.py:22-56)."""

    def __init__(self, config, agent_id: Optional[str] = None) -> None:
        self.config = config
        self.agent_id = agent_id if agent_id else 'agent_' + random_id()

    def play_game(self, environment) -> GameHistory:
        start = time.time()
        game = Game(environment=environment)
        while not game.terminal():
            game.apply(self.make_move(game))
        history = game.history
        history.metadata['agent_id'] = self.agent_id
        history.metadata['timing'] = time.time() - start
        history.metadata.update(self.fill_metadata())
        return history

    def make_move(self, game: Game) -> int:
        raise MuZeroImplementationError('make_move', 'Agent')

    def fill_metadata(self) -> Dict[str, str]:
        return {}


class _Search:
    """One single-tree search in progress: the pool, its hidden states, and the pending expansion."""

    def __init__(self, agent: 'MCTSAgent'):
        self.agent = agent
        self.pool: TreePool = agent._pool
        self.hidden: List = [None] * (self.pool.N + 1)     # hidden state per block (whatever the network returned)
        self.pending = None                                # (priors f32[A], reward, hidden) of the selected leaf
        self.leaf_selected = False
        self.origin: Dict[int, Tuple[int, int]] = {}       # block -> (parent block, action that led to it)
        self.root_actions: List[int] = []

    def slot_of(self, parent_block: int, action: int) -> int:
        return self.root_actions.index(action) if parent_block == 0 else action

    def node_of_block(self, root: 'Node', block: int) -> 'Node':
        """The view of the node that owns `block`, with its chain of parents up to `root`."""
        if block == 0:
            return root
        parent_block, action = self.origin[block]
        parent = self.node_of_block(root, parent_block)
        node = Node(parent=parent, _search=self, _slot=self.slot_of(parent_block, action), _action=action)
        node._block = block
        return node

    def refresh(self, mm: Optional[MinMaxStats]):
        if mm is None:
            return
        _, _, minmax, _ = self.pool.root_stats()
        lo, hi = (float(x) for x in minmax[0].tolist())
        mm.minimum, mm.maximum = lo, hi


class Node:
    """View of one tree node (reference Node, agent.py:103-132).

    A node is identified by the child slot it occupies in its parent's block (`parent`, `slot`) and, once
    expanded, by the block holding its own children."""

    def __init__(self, parent: Optional['Node'] = None, prior: float = 1.0, _search: Optional[_Search] = None,
                 _slot: int = -1, _action: int = -1) -> None:
        self.parent = parent
        self._prior = prior
        self._search = _search
        self._slot = _slot
        self._action = _action
        self._block = 0 if (parent is None and _search is not None) else -1
        self.to_play = None

    # -- raw record of this node inside the parent's block
    def _record(self):
        if self.parent is None or self._search is None:
            return None
        rec = self._search.pool.read_block(0, self.parent._block)
        return {k: v[self._slot] for k, v in rec.items()}

    def _own_block(self) -> int:
        if self._block >= 0 or self.parent is None or self._search is None:
            return self._block
        rec = self._record()
        if rec['count'] > 0:
            self._block = int(rec['child'])
        return self._block

    def expanded(self) -> bool:
        if self._search is None:
            return False
        if self.parent is None:
            return self._search.pool.num_blocks >= 1
        return self._own_block() >= 0

    @property
    def children(self) -> Dict[int, 'Node']:
        if not self.expanded():
            return {}
        rec = self._search.pool.read_block(0, self._own_block())
        return {int(a): Node(parent=self, prior=float(p), _search=self._search, _slot=k, _action=int(a))
                for k, (a, p) in enumerate(zip(rec['actions'], rec['prior']))}

    @property
    def prior(self) -> float:
        rec = self._record()
        return float(rec['prior']) if rec is not None else self._prior

    @property
    def visit_count(self) -> int:
        if self._search is None:
            return 0
        if self.parent is None:
            return int(self._search.pool.root_stats()[0].sum().item()) if self.expanded() else 0
        return int(self._record()['count'])

    @property
    def value_sum(self) -> float:
        if self._search is None:
            return 0.0
        if self.parent is None:
            n = self.visit_count
            return float(self._search.pool.root_stats()[1][0].item()) * n if n else 0.0
        return float(self._record()['value_sum'])

    @property
    def value(self) -> float:
        if self._search is None:
            return float('nan')
        if self.parent is None:
            return float(self._search.pool.root_stats()[1][0].item()) if self.expanded() else float('nan')
        rec = self._record()
        return float(rec['value_sum'] / rec['count']) if rec['count'] > 0 else float('nan')

    @property
    def reward(self) -> Optional[float]:
        if self.parent is None or self._search is None:
            return None
        rec = self._record()
        return float(rec['reward']) if rec['count'] > 0 else None

    @property
    def hidden_state(self):
        block = self._own_block() if self.parent is not None else self._block
        return self._search.hidden[block] if (self._search is not None and block >= 0) else None

    def print(self, _prefix: str = '', name: str = 'Root', _last: bool = True) -> None:
        print(_prefix, '`- ' if _last else '|- ',
              f'{name}-{self.visit_count}: prior={self.prior:.2f}; value={self.value:.4f}', sep='')
        _prefix += '   ' if _last else '|  '
        kids = self.children
        for i, (action, child) in enumerate(kids.items()):
            child.print(_prefix, action, i == len(kids) - 1)


class MCTSAgent(Agent):
    """Monte-Carlo tree search on the GPU tree pool (reference MCTSAgent, agent.py:135-235)."""

    def __init__(self, config, network, agent_id: Optional[str] = None, debug: bool = False, device=None) -> None:
        super().__init__(config=config, agent_id=agent_id)
        self.network = network
        self.debug = debug
        self.effective_discount = config.game_config.discount
        if config.game_config.num_players == 2:
            self.effective_discount *= -1
        self.device = torch.device(device) if device is not None else getattr(network, 'device', torch.device('cuda'))
        kb = config.value_config.known_bounds
        self._pool = TreePool(num_trees=1, num_actions=config.game_config.action_space_size,
                              max_simulations=config.mcts_config.num_simulations,
                              num_players=config.game_config.num_players, discount=config.game_config.discount,
                              default_value=config.mcts_config.default_value,
                              known_bounds=None if kb is None else kb.endpoints(), device=self.device)
        self.A = config.game_config.action_space_size

    # ------------------------------------------------------------------ helpers
    def new_root(self) -> Node:
        """`Node()` of agent.py:220, bound to this agent's pool."""
        self._pool.reset()
        return Node(_search=_Search(self))

    def _dev(self, array, dtype):
        return torch.as_tensor(np.ascontiguousarray(array), dtype=dtype, device=self._pool.device)

    @staticmethod
    def _search_of(node: Node) -> _Search:
        if node._search is None:
            raise MuZeroError('this Node is not bound to a tree pool: create roots with MCTSAgent.new_root()')
        return node._search

    # ------------------------------------------------------------------ reference methods
    @staticmethod
    def expand_node(node: Node, actions: List[int], network_output: NetworkOutput) -> None:
        """agent.py:154-160.  Root: one launch.  Leaf: recorded and fused with the following backpropagate()."""
        search = MCTSAgent._search_of(node)
        agent, pool = search.agent, search.pool
        policy = np.asarray(network_output.masked_policy(actions), np.float32)      # full length A, positional
        if node.parent is None:
            A = agent.A
            legal = np.zeros((1, A), np.int32)
            legal[0, :len(actions)] = actions
            pool.expand_roots(agent._dev(policy.reshape(1, A), torch.float32), agent._dev(legal, torch.int32),
                              agent._dev([len(actions)], torch.int32))
            search.hidden[0] = network_output.hidden_state
            search.root_actions = [int(a) for a in actions]
            node._block = 0
        else:
            if list(actions) != list(range(agent.A)):
                raise MuZeroError('non-root nodes are expanded over the whole action space (agent.py:208-210)')
            search.pending = (policy, float(network_output.reward), network_output.hidden_state,
                              node.parent._block, node._action)

    @staticmethod
    def softmax_sample(distribution: List[Tuple[int, int]], temperature: float) -> Tuple[int, int]:
        """agent.py:162-168 on a host list (kept for callers; select_action samples on the device)."""
        if temperature == 0.0:
            return max(distribution)
        weights = [count ** (1 / temperature) for count, _ in distribution]
        return random.choices(distribution, weights=weights, k=1)[0]

    def add_exploration_noise(self, node: Node) -> None:
        """agent.py:170-175: host Dirichlet draw, fp64 mix on the device."""
        search = self._search_of(node)
        n = len(search.pool.read_block(0, 0)['count'])
        noise = np.zeros((1, self.A), np.float64)
        noise[0, :n] = np.random.dirichlet([self.config.mcts_config.root_dirichlet_alpha] * n)
        search.pool.add_root_noise(self._dev(noise, torch.float64), self.config.mcts_config.root_exploration_fraction)

    def _select_leaf_below_root(self, node: Node, min_max_stats: MinMaxStats) -> Tuple[int, Node]:
        """agent.py:177-181 from a node below the root: the descent is made on the host with the reference's own
        arithmetic (select_child / ucb_score: Python floats, fp32 priors widened), then the whole chain root -> node -> leaf
        is handed to the pool as its selection (mz_pool_set_selection), so that the following backpropagate walks up to
        the root like the reference's does."""
        search = self._search_of(node)
        if search.leaf_selected:
            raise MuZeroError('a selected leaf awaits expand_node + backpropagate')
        if not node.expanded():
            raise MuZeroError('select_leaf needs an expanded node (agent.py:179 would return no action)')
        chain, n = [], node
        while n.parent is not None:                      # root -> node
            chain.append((n.parent._block if n.parent.parent is None else n.parent._own_block(), n._action))
            n = n.parent
        chain.reverse()
        action, cur = None, node
        while cur.expanded():
            block = cur._own_block() if cur.parent is not None else cur._block
            action, cur = self.select_child(cur, min_max_stats)
            chain.append((block, action))
        search.pool.set_selection(0, [b for b, _ in chain], [a for _, a in chain])
        search.leaf_selected = True
        return action, cur

    def select_leaf(self, node: Node, min_max_stats: MinMaxStats) -> Tuple[int, Node]:
        """agent.py:177-181.  From the root: one select launch; from any other expanded node: a host descent whose path
        the pool adopts.  Returns (last action, leaf view)."""
        search = self._search_of(node)
        if node.parent is not None:
            return self._select_leaf_below_root(node, min_max_stats)
        parent_block, action = search.pool.select()
        parent_block, action = int(parent_block[0].item()), int(action[0].item())
        search.leaf_selected = True
        parent = search.node_of_block(node, parent_block)
        slot = search.slot_of(parent_block, action)
        prior = float(search.pool.read_block(0, parent_block)['prior'][slot])
        leaf = Node(parent=parent, prior=prior, _search=search, _slot=slot, _action=action)
        return action, leaf

    def ucb_score(self, node: Node, min_max_stats: MinMaxStats) -> float:
        """agent.py:187-190 evaluated on the host from the node's record (inspection; the kernel scores on device)."""
        value = node.value
        exploitation = self.config.mcts_config.default_value if isnan(value) else node.reward + self.effective_discount * value
        exploration = node.prior * self.config.exploration_function(node.parent.visit_count, node.visit_count)
        return min_max_stats.normalize(exploitation) + exploration

    def select_child(self, node: Node, min_max_stats: MinMaxStats) -> Tuple[int, Node]:
        """agent.py:183-185 on the host (inspection)."""
        self._search_of(node).refresh(min_max_stats)
        best = None
        for action, child in node.children.items():
            score = self.ucb_score(child, min_max_stats)
            if best is None or (score, action) > best[:2]:
                best = (score, action, child)
        return best[1], best[2]

    def backpropagate(self, node: Node, value: float, min_max_stats: MinMaxStats) -> None:
        """agent.py:192-196 for the leaf returned by select_leaf, fused with its pending expansion."""
        search = self._search_of(node)
        if search.pending is None or not search.leaf_selected:
            raise MuZeroError('backpropagate expects select_leaf + expand_node on the leaf first (agent.py:202-211)')
        policy, reward, hidden, parent_block, action = search.pending
        pool = search.pool
        pool.expand_backprop(self._dev(policy.reshape(1, self.A), torch.float32), self._dev([reward], torch.float32),
                             self._dev([value], torch.float32))
        search.hidden[pool.num_blocks - 1] = hidden
        search.origin[pool.num_blocks - 1] = (parent_block, action)
        search.pending, search.leaf_selected = None, False
        search.refresh(min_max_stats)

    def run_mcts(self, root: Node, min_max_stats: MinMaxStats) -> None:
        """agent.py:198-211: num_simulations x (select, recurrent_inference, expand, backprop)."""
        search = self._search_of(root)
        pool = search.pool
        N = self.config.mcts_config.num_simulations
        parent_block, action = pool.select()
        for s in range(N):
            if self.debug:
                root.print()
            pb, a = int(parent_block[0].item()), int(action[0].item())
            hidden = search.hidden[pb]
            batch_hidden = hidden.unsqueeze(0) if isinstance(hidden, torch.Tensor) else np.expand_dims(np.asarray(hidden), 0)
            output = self.network.recurrent_inference(batch_hidden, np.asarray([a], np.int32)).split_batch()[0]
            policy = np.asarray(output.masked_policy(self.config.action_space()), np.float32)
            args = (self._dev(policy.reshape(1, self.A), torch.float32), self._dev([output.reward], torch.float32),
                    self._dev([output.value], torch.float32))
            search.hidden[s + 1] = output.hidden_state
            search.origin[s + 1] = (pb, a)
            if s + 1 < N:
                parent_block, action = pool.step(*args)
            else:
                pool.expand_backprop(*args)
        search.refresh(min_max_stats)

    def select_action(self, node: Node, num_moves: int) -> int:
        """agent.py:213-217: temperature schedule, then the device softmax_sample with one host uniform."""
        search = self._search_of(node)
        t = self.config.visit_softmax_temperature_fn(num_moves=num_moves, training_steps=self.network.training_steps())
        u = random.random() if t != 0.0 else 0.0          # random.choices consumes exactly one random()
        search.pool.set_temperature(t)                    # exact count ** (1/t) weights, tabulated by this host
        chosen = search.pool.select_action(self._dev([u], torch.float64), self._dev([t], torch.float64))
        return int(chosen[0].item())

    def make_move(self, game: Game) -> int:
        """agent.py:219-232."""
        root = self.new_root()
        min_max_stats = MinMaxStats(known_bounds=self.config.value_config.known_bounds)
        image = game.history.make_image()
        observation = image.unsqueeze(0) if isinstance(image, torch.Tensor) else np.expand_dims(np.asarray(image), 0)
        self.expand_node(node=root, actions=game.legal_actions(),
                         network_output=self.network.initial_inference(observation).split_batch()[0])
        self.add_exploration_noise(root)
        self.run_mcts(root, min_max_stats)
        _visits, root_value, _minmax, policy = self._pool.root_stats()
        game.store_search_statistics(float(root_value[0].item()), policy[0].cpu())
        return self.select_action(root, len(game.history))

    def fill_metadata(self) -> Dict[str, str]:
        return {'network_id': str(self.network.training_steps())}
