import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# VH_JIT=off (tools/verify_gpu.sh's second pass): the library never compiles a per-query kernel, whatever a plan's flags ask for;
# tests that assert the compiled kernel ran skip those assertions (the pre-built kernels then answer the same queries)
JIT_OFF = os.environ.get("VH_JIT", "") in ("0", "off")
