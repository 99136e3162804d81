import os as _os

# ROCm 7.2 hipGraph "packet capture" corrupts earlier graphs once a process holds ~2900 kernel nodes (see
# cpcsv/graphs.many_graphs_safe); the switch is read when the HIP runtime initialises, i.e. before torch touches the GPU
if "torch" not in __import__("sys").modules:       # provably before the HIP runtime reads its flags: cpcsv.runtime trusts this marker
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    _os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]

import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_gpu_state(request):
    """Trainers of a finished GPU test hold ~15 captured HIP graphs each (private memory pools, thousands of kernel nodes).
    Left to the cyclic collector they pile up across tests of one process - and the ROCm 7.2 graph runtime has crashed in
    hipGraphLaunch with several dead trainers' graphs still instantiated. Drop them at the end of every GPU test."""
    yield
    if "gpu" in request.keywords and "torch" in sys.modules:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.empty_cache()
