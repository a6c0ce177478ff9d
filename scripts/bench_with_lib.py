"""bench.py against another build of the library (same-box A/B of a kernel change over the whole iteration):
bench_with_lib.py <path/to/lib.so> [bench.py arguments]."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disentangle_mlp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
