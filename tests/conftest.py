import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """Tests import the ctypes wrappers at collection time: build what is missing first.  The oracle needs gcc (seconds); the HIP
    library needs hipcc (cross-compiles for gfx950 without a GPU, about a minute) — the same thing __graft_entry__.build() does."""
    so = os.path.join(ROOT, "oracle", "libpiscesoracle.so")
    src = os.path.join(ROOT, "oracle", "pisces_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "pisces_amd", "libpisceship.so")):
        from pisces_amd import build
        build.build_native()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build it on demand (gcc only, seconds)."""
    so = os.path.join(ROOT, "oracle", "libpiscesoracle.so")
    src = os.path.join(ROOT, "oracle", "pisces_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    yield
