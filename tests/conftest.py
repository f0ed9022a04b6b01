import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.skip("no GPU")
    return 0


def record_parity(name, values):
    """Achieved parity numbers of a GPU test, merged into a JSON file (copied to profiles/<round>/parity.json
    after the run): PT_PARITY_JSON, else gpurun_out/parity.json under the repository."""
    import json
    path = os.environ.get("PT_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            with open(path) as f:
                allv = json.load(f)
        except (OSError, ValueError):
            allv = {}
        allv[name] = values
        with open(path, "w") as f:
            json.dump(allv, f, indent=1, sort_keys=True)
    except OSError:
        pass
