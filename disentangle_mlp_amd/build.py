"""Build libvaegan_hip.so (the C-ABI HIP library) in-tree for gfx950, and its tuning twin.

    python -m disentangle_mlp_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under csrc/build/ and
rebuilt when a source or header is newer.  The .so files are git-ignored but travel
to the GPU box with the working tree.

Two libraries come out of the same sources:
  libvaegan_hip.so         the product: no global mutable state, no vg_debug_* symbols;
  libvaegan_hip_tuning.so  compiled with -DVG_TUNING: the tile-forcing knobs (process-globals) that tests/
                           and scripts/ use to reach every kernel variant (``_lib.load_tuning()``).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libvaegan_hip.so")
OUT_TUNING = os.path.join(HERE, "libvaegan_hip_tuning.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{CSRC}",
         "-Wno-unused-result"]
# conv_ring.hip unrolls the 25 K steps of a channel chunk (compile-time tap offsets, static fragment registers):
# past LLVM's default pragma-unroll budget the loop stays rolled and the fragment arrays go to scratch
FILE_FLAGS = {"conv_ring.hip": ["-mllvm", "-pragma-unroll-threshold=131072", "-Wno-inline-asm"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _build_one(out, bdir_name, extra, force, verbose):
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    bdir = os.path.join(CSRC, "build", bdir_name) if bdir_name else os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(bdir, s[:-4] + ".o")
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(bdir, s[:-4] + ".o") for s in srcs]
    if jobs or any(_newer(o, out) for o in objs):        # also after an object was compiled by hand
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


def build(force=False, verbose=True, tuning=True):
    """Compiles every HIP source; returns the product library's path.  The two libraries are built side by side:
    conv_ring.hip (its unrolled K loop, 100+ kernel instantiations) is minutes of one compiler thread in each."""
    if not tuning:
        return _build_one(OUT, "", [], force, verbose)
    with ThreadPoolExecutor(max_workers=2) as ex:
        a = ex.submit(_build_one, OUT, "", [], force, verbose)
        b = ex.submit(_build_one, OUT_TUNING, "tuning", ["-DVG_TUNING"], force, verbose)
        b.result()
        return a.result()


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
