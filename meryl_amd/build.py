"""Build recipe for the native library (gfx950 only, in-tree).

    python -m meryl_amd.build            # builds meryl_amd/libmeryl_gpu_count.so

hipcc cross-compiles for gfx950 without a GPU.  The built .so stays in-tree
(git-ignored) so that it travels with the source snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmeryl_gpu_count.so")
SOURCES = ["mgc_kmer.hip", "mgc_sort.hip", "mgc_scan.hip", "mgc_finish.hip", "mgc_misc.hip", "mgc_parse.hip",
           "mgc_encode.hip", "mgc_decode.hip", "mgc_merge.hip", "mgc_lookup.hip",
           "mgc_api.cpp", "mgc_stream.cpp", "mgc_runs.cpp", "mgc_node.cpp", "meryl_db.cpp", "meryl_seq.cpp"]
HEADERS = ["mgc_device.h", "mgc_common.hpp", "mdb_layout.h", "mgc_session.hpp", "mgc_runs.hpp",
           os.path.join("..", "..", "include", "meryl_gpu_count.h"),
           os.path.join("..", "..", "include", "meryl_db.h"), os.path.join("..", "..", "include", "meryl_seq.h"),
           os.path.join("..", "..", "include", "meryl_lookup.h")]
OBJDIR = os.path.join(HERE, "build")
# -no-hip-rt: the library carries no DT_NEEDED on a particular libamdhip64; it binds to the
# HIP runtime already in the process (torch's bundled one under Python -- two HIP/HSA runtimes
# in one process cannot share streams or ordering -- or /opt/rocm's for the standalone CLI).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function"]
LFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-no-hip-rt"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the native library cannot be built")


def _sources():
    return [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]


def _header_mtime():
    deps = [os.path.join(CSRC, f) for f in HEADERS if os.path.exists(os.path.join(CSRC, f))]
    deps.append(os.path.abspath(__file__))
    return max(os.path.getmtime(d) for d in deps)


def _obj(f):
    return os.path.join(OBJDIR, f + ".o")


def _stale_objects(force=False):
    """sources whose object is missing or older than the source / any header"""
    ht = _header_mtime()
    out = []
    for f in _sources():
        o = _obj(f)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(os.path.join(CSRC, f)), ht):
            out.append(f)
    return out


def _stale():
    if not os.path.exists(LIB) or _stale_objects():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(_obj(f)) > t for f in _sources())


CLI = os.path.join(HERE, "bin", "meryl")


def build_cli(force=False, verbose=False):
    """The `meryl` front end (meryl_amd/bin/meryl): links the library and the system HIP runtime."""
    src = os.path.join(CSRC, "meryl_main.cpp")
    if (not force and os.path.exists(CLI) and os.path.getmtime(CLI) >= os.path.getmtime(src)
            and os.path.getmtime(CLI) >= os.path.getmtime(LIB)):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    tmp = "%s.tmp%d" % (CLI, os.getpid())             # several ranks may build at once: private file, atomic rename
    cmd = [hipcc(), "-O2", "-std=c++17", "-pthread", src, "-o", tmp, "-L" + HERE, "-lmeryl_gpu_count",
           "-Wl,-rpath,$ORIGIN/..", "-L" + rocm_lib, "-lamdhip64", "-Wl,-rpath," + rocm_lib, "-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, CLI)
    return CLI


LOOKUP_CLI = os.path.join(HERE, "bin", "meryl-lookup")


def build_lookup_cli(force=False, verbose=False):
    """`meryl-lookup -existence` (meryl_amd/bin/meryl-lookup): links the library and the system HIP runtime."""
    src = os.path.join(CSRC, "meryl_lookup_main.cpp")
    if (not force and os.path.exists(LOOKUP_CLI) and os.path.getmtime(LOOKUP_CLI) >= os.path.getmtime(src)
            and os.path.getmtime(LOOKUP_CLI) >= os.path.getmtime(LIB)):
        return LOOKUP_CLI
    os.makedirs(os.path.dirname(LOOKUP_CLI), exist_ok=True)
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    tmp = "%s.tmp%d" % (LOOKUP_CLI, os.getpid())
    cmd = [hipcc(), "-O2", "-std=c++17", "-pthread", src, "-o", tmp, "-L" + HERE, "-lmeryl_gpu_count",
           "-Wl,-rpath,$ORIGIN/..", "-L" + rocm_lib, "-lamdhip64", "-Wl,-rpath," + rocm_lib, "-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, LOOKUP_CLI)
    return LOOKUP_CLI


def build(force=False, verbose=False):
    """Compile every HIP/C++ source for gfx950 (one object per source, rebuilt only when the source or a
    header changed, in parallel) and link libmeryl_gpu_count.so (and the CLI).  Returns the library path."""
    if not force and not _stale():
        build_cli(False, verbose)
        build_lookup_cli(False, verbose)
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    todo = _stale_objects(force)
    procs = []
    for f in todo:
        tmp = "%s.tmp%d" % (_obj(f), os.getpid())
        cmd = [hipcc()] + CFLAGS + ["-c", os.path.join(CSRC, f), "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((f, tmp, cmd, subprocess.Popen(cmd)))
    failed = []
    for f, tmp, cmd, p in procs:
        if p.wait() != 0:
            failed.append(f)
        else:
            os.replace(tmp, _obj(f))
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    tmp = "%s.tmp%d" % (LIB, os.getpid())
    cmd = [hipcc()] + LFLAGS + ["-o", tmp] + [_obj(f) for f in _sources()] + ["-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    build_cli(True, verbose)
    build_lookup_cli(True, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
