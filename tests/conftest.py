import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ant_model():
    from robosumo_selfplay_amd import mjcf
    return mjcf.load_model("RoboSumo-Ant-vs-Ant-v0")


@pytest.fixture(scope="session")
def spider_model():
    from robosumo_selfplay_amd import mjcf
    return mjcf.load_model("RoboSumo-Spider-vs-Spider-v0")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
