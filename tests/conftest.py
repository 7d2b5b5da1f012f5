import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the library's VH_TEST_* / VH_POISON / VH_PART_TABLE_KB hooks are behind ONE gate it reads once (viya_hip.hip test_env): open it for
# every test process and the workers they spawn (inherited environment)
os.environ.setdefault("VH_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# VH_JIT=off (tools/verify_gpu.sh's second pass): the library never compiles a per-query kernel, whatever a plan's flags ask for;
# tests that assert the compiled kernel ran skip those assertions (the pre-built kernels then answer the same queries)
JIT_OFF = os.environ.get("VH_JIT", "") in ("0", "off")
