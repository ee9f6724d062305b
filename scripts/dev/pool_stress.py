"""dev: tests/native/libdh_pool_host.so's stress entries at the pool sizes given (one process per size)."""
import ctypes, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "native", "libdh_pool_host.so"))
    t = time.time()
    a = L.dh_pool_host_alternate(300000, 3, 200)
    b = L.dh_pool_host_alternate(100000, 1, 4000)
    c = L.dh_pool_host_concurrent(4, 30000, 1000)
    print("threads", os.environ.get("DH_HOST_THREADS"), "bad", a, b, c, "%.1f s" % (time.time() - t), flush=True)
else:
    for th in sys.argv[1:] or ["64", "32", "8", "3", "128"]:
        e = dict(os.environ, DH_HOST_THREADS=th)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, timeout=900)
        if r.returncode:
            print("threads", th, "rc", r.returncode)
