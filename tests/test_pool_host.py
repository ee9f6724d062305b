"""The host thread pool behind dh_parallel_for (dentist_amd/csrc/dh_parallel.h: workers asleep on a futex word) on its
own: regions of every shape visit every index exactly once, several callers at once are served one region at a time,
and the optional spinning mode (DH_POOL_SPIN_US) behaves the same."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "native", "libdh_pool_host.so")

DRIVER = """
import ctypes, sys
L = ctypes.CDLL(sys.argv[1])
L.dh_pool_host_regions.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]
n = ctypes.c_int64()
bad = L.dh_pool_host_regions(2000, 37, ctypes.byref(n))
bad2 = L.dh_pool_host_concurrent(4, 300, 1000)
# regions of 3 and of 200 chunks in turn: the claim race of the first compare-and-swap version (a worker still looking at
# the last region's ticket took the next region's chunk count for the old one's) hung or double-ran a chunk within seconds
bad3 = L.dh_pool_host_alternate(60000, 3, 200)
print(bad, n.value, bad2 + bad3)
"""


@pytest.fixture(scope="module")
def built():
    src = os.path.join(ROOT, "tests", "native", "pool_host.cpp")
    hdr = os.path.join(ROOT, "dentist_amd", "csrc", "dh_parallel.h")
    if not os.path.exists(PATH) or os.path.getmtime(PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["make", "-C", ROOT, "-B", "tests/native/libdh_pool_host.so"], check=True)
    return PATH


# (a process per setting: the pool reads its environment once, when the first region runs)
@pytest.mark.parametrize("env", [{}, {"DH_HOST_THREADS": "1"}, {"DH_HOST_THREADS": "3"}, {"DH_HOST_THREADS": "16"},
                                 {"DH_HOST_THREADS": "4", "DH_POOL_SPIN_US": "20"}])
def test_every_index_once_and_concurrent_callers(built, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", DRIVER, built], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    bad, regions, bad2 = (int(x) for x in out.stdout.split())
    assert regions == 4 * len(range(0, 2001, 37))
    assert bad == 0 and bad2 == 0
