"""Builds the native library in-tree (struspattern_amd/_build/libstruspattern_amd.so).

hipcc cross-compiles gfx950 code objects without a GPU; the built .so travels to the GPU box
with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# SPA_LIB: a variant library built out of tree (tests/micro/ab.sh); the product library otherwise
LIB = os.environ.get("SPA_LIB") or os.path.join(HERE, "_build", "libstruspattern_amd.so")


HOST_LIB = os.path.join(HERE, "_build", "libstrus_pattern.so")
HOST_MODULE = os.path.join(HERE, "_build", "modstrus_analyzer_pattern.so")
HOST_TEST = os.path.join(HERE, "_build", "testStrusInterface")


def _sources():
    out = []
    for d in (os.path.join(HERE, "csrc"), os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith((".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                out.append(os.path.join(d, f))
    return out


def build_host(quiet=True):
    """C++ host side of the drop-in (strus plugin interfaces over the C-ABI) + module entry point."""
    build(quiet=quiet)
    cmd = ["make", "-C", os.path.join(HERE, "host")]
    if quiet:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return HOST_LIB, HOST_MODULE, HOST_TEST


def needs_build():
    if os.environ.get("SPA_LIB"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force=False, quiet=True):
    if force or needs_build():
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            raise RuntimeError("hipcc not found at %s: cannot build the HIP extension" % hipcc)
        cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j4", "HIPCC=" + hipcc]
        if quiet:
            cmd.insert(1, "-s")
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=False, quiet=False))
