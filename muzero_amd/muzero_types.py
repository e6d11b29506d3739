"""Type aliases of the reference (muzero_types.py:1-19); tensors are torch tensors / numpy arrays here."""
from typing import Any, List, NamedTuple, NewType

Value = NewType('Value', float)
Player = NewType('Player', int)
Action = NewType('Action', int)
Observation = NewType('Observation', Any)
Policy = NewType('Policy', Any)
ObservationBatch = NewType('ObservationBatch', Any)
ActionBatch = NewType('ActionBatch', Any)
ValueBatch = NewType('ValueBatch', Any)
PolicyBatch = NewType('PolicyBatch', Any)


class State(NamedTuple):
    observation: Observation
    to_play: Player
    legal_actions: List[Action]
