import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a second whole-model repeat of a geometry another test already covers in the same "
                                       "precision class; skipped unless CTRLORA_RUN_SLOW=1 (keeps the driver's GPU run short)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("CTRLORA_RUN_SLOW") == "1":
        return
    import pytest as _pytest
    skip = _pytest.mark.skip(reason="slow repeat: set CTRLORA_RUN_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


import pytest  # noqa: E402


@pytest.fixture(autouse=True, scope="module")
def _release_gpu_memory_between_modules():
    """GPU runs only: models, hipGraph pools and the caching allocator's blocks of one test module are released before the
    next one starts (the suite builds ~40 SD1.5-width models; without this the late modules run inside a fragmented pool)."""
    yield
    if "torch" in sys.modules:
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
