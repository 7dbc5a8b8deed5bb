"""tools/fuzzmore.py -- the GPU fuzz tests again with other seeds (tests/test_gpu_parity.py seeds its generators with constants)."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib
import conftest  # noqa
t = importlib.import_module("test_gpu_parity")
from oracle import binding as ob
oracle = ob.Oracle()
orig = np.random.default_rng
for k in range(1, 7):
    np.random.default_rng = lambda seed=None, k=k: orig((seed or 0) + 1000003 * k)
    for name in [n for n in dir(t) if n.startswith("test_fuzz")]:
        fn = getattr(t, name)
        if "descriptors" in name:
            c = t.h.Context(0)
            try:
                fn(c, oracle)
            finally:
                c.close()
        else:
            fn(oracle)
        print("seed offset", k, name, "ok", flush=True)
np.random.default_rng = orig
