import os
import sys

import pytest

# The tensor-parallel loopback tests run several shards' streams side by side on ONE device, kernels of one stream waiting for
# kernels of another: each stream needs its own hardware queue.  The one-process group makes sure of that itself (CU-masked
# streams, jh_tp_group_create); the raised queue count keeps the other multi-stream tests (pipeline stages, concurrent sessions)
# off shared queues.  Read when HIP initialises, so it is set before anything imports torch.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

try:   # torch bundles its own HIP runtime: when a test mixes torch.cuda with libjlamahip.so, torch must be loaded first
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from jlama_amd import _native as N
        N.init(0)
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run on the HIP path: no device => hard failure, never a silent skip-to-CPU."""
    from jlama_amd import _native as N
    info = N.init(0)
    return info


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(autouse=True)
def _clear_library_options(request):
    """Options set through jh_set_option (the library reads no tuning knob from the environment) live for one test only."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from jlama_amd import _native as N
        N.clear_options()
