import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


def golden_names(prefix=''):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz')))
    return names


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False) as data:
        return {k: data[k] for k in data.files}


TREE_FIXTURES = [n for n in golden_names() if not n.startswith(('ttt_game', 'replay_', 'ttt_env_'))]
REPLAY_FIXTURES = golden_names('replay_')
GAME_FIXTURES = golden_names('ttt_game')


@pytest.fixture(scope='session')
def has_gpu():
    import torch
    return torch.cuda.is_available()
