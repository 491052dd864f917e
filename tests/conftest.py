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
