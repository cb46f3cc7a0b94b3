import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tests run on seeded random weights and synthetic text: the two explicit opt-ins of the product path
# (clip_api.load / tokenizer.get_tokenizer refuse to fall back silently; tests/test_host_logic.py checks that they do)
os.environ.setdefault("KEMR_ALLOW_RANDOM_WEIGHTS", "1")
os.environ.setdefault("KEMR_ALLOW_HASH_TOKENIZER", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU (the HIP path has no CPU fallback)")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test; build it in-tree if it is missing."""
    from knowledge_enhanced_multimodal_retrieval_amd import build
    build.build()
