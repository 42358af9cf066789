import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests load the in-tree C-ABI library and the C oracle.  On a checkout where they have not been built yet (they are
    git-ignored build products) compile them once -- the same step as __graft_entry__.build(); hipcc cross-compiles without a GPU.
    The product itself never builds on demand: zopt_amd._lib.lib() fails loudly when the library is missing."""
    from zopt_amd import _lib
    if (not os.path.exists(_lib.LIB_PATH) or not os.path.exists(_lib.LAB_LIB_PATH)) and not os.environ.get("ZOPT_AMD_LIB"):
        import __graft_entry__
        __graft_entry__.build()
