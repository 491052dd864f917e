"""pytest configuration: markers + import paths.

`-m "not gpu"` runs on CPU (oracle vs golden vectors, host logic, C-ABI load/symbol check);
`-m gpu` are the parity tests proper and call the HIP path through the C ABI.
"""
import os
import sys

# in-process tensor-parallel tests drive up to 8 ranks of one process on one GPU, each on its own stream, and their one-shot
# collectives wait for each other INSIDE kernels: every stream needs a hardware queue of its own (HIP's default is 4 per process)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import gc

import pytest


@pytest.fixture(autouse=True)
def _no_collector_inside_a_test():
    """Engines of earlier tests that are still waiting for the cycle collector are destroyed HERE, between tests, and the collector stays
    off while a test runs.  Destroying an engine frees device memory, and hipFree waits for every stream of the device; in the
    in-process tensor-parallel tests (several ranks of ONE process on ONE GPU, one host thread each) a collection that fires on rank A's
    thread would then wait for rank B's kernel, which waits inside its one-shot collective for the launch rank A's thread has not made
    yet — a deadlock of the test arrangement (ranks of the product are one process per GPU: a rank's hipFree only ever waits for itself)."""
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        gc.enable()


# (r02-r05 forced nvr_config.async_decode = 0 on every Config of the suite here, because the logits accessors used to refer to the step launched
# ahead.  Since r06 they refer to the step just returned (model_runner.cpp: present_step), so the suite runs the engine nvr_config_default builds —
# launch-ahead on — against the oracle, per-step logits included; the tests that need the synchronous engine spell async_decode=0 out.)
