import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's greedy steps are GEMV-bound: on the GPU boxes' 128-thread hosts PyTorch runs them faster on 16 threads
    # than on all of them (bench.py's cpu_baseline probe: 98 ms/token on 16), and the GPU suite spends most of its time there.
    try:
        import torch
        if (os.cpu_count() or 1) > 32:
            torch.set_num_threads(16)
    except Exception:
        pass


@pytest.fixture(scope="session")
def lib():
    """The shared library must exist (built in-tree by __graft_entry__.build())."""
    from qwen3_asr_rs_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.load()


@pytest.fixture(scope="session")
def tiny_dir():
    from qwen3_asr_rs_amd import synthetic
    return synthetic.write_checkpoint("/tmp/q3a_ckpt_tiny", "tiny", seed=1)


@pytest.fixture(scope="session")
def tiny_untied_dir():
    from qwen3_asr_rs_amd import synthetic
    return synthetic.write_checkpoint("/tmp/q3a_ckpt_tiny_untied", "tiny_untied", seed=2, shards=3)


@pytest.fixture(scope="session")
def tiny_oracle(tiny_dir):
    from oracle import q3asr_oracle as O
    return O.AsrOracle(tiny_dir)
