import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def ftk():
    """The product package with its native library loaded (fails loudly if it was not built)."""
    import feature_tracker_amd as F
    from feature_tracker_amd import _native
    _native.lib()
    return F


@pytest.fixture(scope="session")
def gpu_ctx(ftk):
    return ftk.default_context()


@pytest.fixture
def switch():
    """Sets one of the library's FTK_* experiment switches for the duration of a test.  The library reads them ONCE per context
    (ftk_context_create), so every live context is told to read them again (ftk_context_refresh_env) after each change and once
    more when the test's environment has been restored."""
    import feature_tracker_amd as F
    saved = {}

    def set_switch(name, value):
        saved.setdefault(name, os.environ.get(name))
        if value is None:
            os.environ.pop(name, None)  # unset
        else:
            os.environ[name] = str(value)
        F.refresh_env_switches()

    yield set_switch
    for name, old in saved.items():
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    if saved:
        F.refresh_env_switches()
