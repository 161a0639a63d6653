#!/usr/bin/env python3
"""One process, the engine fixture's life cycle of tests/test_gpu_parity.py::test_c2_full_size_properties repeated; a watchdog dumps every thread's Python stack
if an iteration takes more than 60 s (faulthandler), so that a hang shows WHERE it sits."""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_gpu_parity as T
from ldweaver_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with Engine(0) as e:
    for it in range(n):
        faulthandler.dump_traceback_later(60, exit=True)
        t0 = time.time()
        T.test_c2_full_size_properties(e)
        faulthandler.cancel_dump_traceback_later()
        print(f"iteration {it}: {time.time() - t0:.1f} s", flush=True)
